"""CPU: the C-ABI library loads, exports every symbol include/kivi_hip.h declares, rejects bad
arguments without touching a device, and the Python layer refuses CPU tensors (no fallback)."""
import os
import re

import pytest
import torch

from helpers import ROOT


@pytest.fixture(scope="module")
def lib():
    from kivi_amd import _lib, build
    build.build()          # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "kivi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kivi_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from kivi_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kivi_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in kivi_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.kivi_abi_version() == _lib.ABI_VERSION == 3


def test_variant_tables(lib):
    nk, nv = lib.kivi_gemv_k_num_variants(), lib.kivi_gemv_v_num_variants()
    assert nk > 0 and nv > 0
    names = [lib.kivi_gemv_k_variant_name(i).decode() for i in range(nk)]
    names += [lib.kivi_gemv_v_variant_name(i).decode() for i in range(nv)]
    assert len(set(names)) == len(names) and all(names)
    assert lib.kivi_gemv_k_variant_name(10 ** 6) == b""


def test_argument_validation_without_device(lib):
    """Rejected arguments return a negative code before any HIP call (pointers are never dereferenced)."""
    from kivi_amd._lib import KiviHipError, check
    # bits = 3
    assert lib.kivi_quant_pack_lastdim(None, None, None, None, 4, 64, 32, 3, None) == -1
    assert b"bits" in lib.kivi_last_error()
    # T % group_size != 0 (new_pack.py:222)
    assert lib.kivi_quant_pack_lastdim(None, None, None, None, 4, 48, 32, 2, None) == -1
    # nh % nh_kv != 0 (matmul.py:216)
    args = (None, 0, 0, None, 0, 0, 0, None, None, 0, 0, 0, None, 0, 0)
    assert lib.kivi_gemv_k(*args, 1, 3, 2, 128, 64, 32, 2, None) == -1
    assert b"nh_kv" in lib.kivi_last_error()
    # GEMV supports 2 and 4 bits only (matmul.py:215)
    assert lib.kivi_gemv_v(*args, 1, 2, 2, 64, 128, 32, 8, None) == -1
    assert lib.kivi_gemv_outer_dim(None, None, None, None, None, 2, 128, 64, 2, 32, 4, 0, None) == -1
    with pytest.raises(KiviHipError):
        check(lib.kivi_unpack_dequant_lastdim(None, None, None, None, 1, 48, 32, 2, None), "unpack")
    # empty problems are no-ops
    assert lib.kivi_quant_pack_lastdim(None, None, None, None, 0, 64, 32, 2, None) == 0


def test_python_api_has_no_cpu_fallback():
    from kivi_amd._lib import KiviHipError
    from kivi_amd.quant import kivi_gemv, matmul, new_pack
    x = torch.randn(1, 1, 4, 64).half()
    with pytest.raises(KiviHipError):
        new_pack.triton_quantize_and_pack_along_last_dim(x, 32, 2)
    with pytest.raises(KiviHipError):
        new_pack.quant_and_pack_kcache(x, 32, 2)
    with pytest.raises(KiviHipError):
        new_pack.unpack_tensor(torch.zeros(1, 1, 4, 4, dtype=torch.int32), 2, 3)
    fA = torch.zeros(1, 2, 1, 64).half()
    qB = torch.zeros(1, 2, 64, 8, dtype=torch.int32)
    sz = torch.zeros(1, 2, 64, 4).half()
    with pytest.raises(KiviHipError):
        matmul.cuda_bmm_fA_qB_outer(32, fA, qB, sz, sz, 2)
    with pytest.raises(KiviHipError):
        kivi_gemv.gemv_forward_cuda_outer_dim(fA.view(2, 1, 64), qB.view(2, 64, 8), sz.view(2, 64, 4), sz.view(2, 64, 4),
                                              2, 32, 2, 2)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under kivi_amd/ (or the drop-in shims) may reference it."""
    bad = []
    for base in ("kivi_amd", "quant", "models"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|kivi_oracle", txt, flags=re.M):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_decoder_wrapper_uses_hf_parameter_names():
    """kivi_amd.llama keeps the Hugging Face checkpoint names, so a Llama / Mistral state dict loads unmodified."""
    import torch
    from kivi_amd.llama import LlamaForCausalLM_KIVI, make_config
    cfg = make_config(dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, num_hidden_layers=2,
                           intermediate_size=512, vocab_size=100))
    m = LlamaForCausalLM_KIVI(cfg)
    keys = set(m.state_dict().keys())
    want = {"model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"}
    for i in range(2):
        for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj",
                  "mlp.up_proj", "mlp.down_proj", "input_layernorm", "post_attention_layernorm"):
            want.add(f"model.layers.{i}.{n}.weight")
    assert want <= keys, sorted(want - keys)
    assert m.model.layers[0].self_attn.k_proj.weight.shape == (128, 256)
    with __import__("pytest").raises(Exception):   # no CPU fallback: the forward pass needs the HIP path
        m(torch.zeros((1, 4), dtype=torch.long))


def test_cache_reserve_keeps_contents_and_page_layout():
    """KiviLayerCache.reserve (host logic, no kernels): bigger buffers, live contents bit-identical, K pages keep the
    page-outside-head memory order."""
    import torch
    from kivi_amd.cache import KiviConfig, KiviLayerCache
    cfg = KiviConfig(2, 2, 32, 32)
    lc = KiviLayerCache(cfg, 2, 3, 128, 100, "cpu", page_tokens=64)
    assert lc.cap == 128 and lc.n_pages == 2
    g = torch.Generator().manual_seed(0)
    for name in ("k_code", "v_code"):
        getattr(lc, name).copy_(torch.randint(-2**31, 2**31 - 1, getattr(lc, name).shape, generator=g, dtype=torch.int64).int())
    for name in ("k_scale", "k_mn", "v_scale", "v_mn", "k_res", "v_res"):
        getattr(lc, name).copy_(torch.randn(getattr(lc, name).shape, generator=g).half())
    before = {n: getattr(lc, n).clone() for n in ("k_code", "k_scale", "k_mn", "v_code", "v_scale", "v_mn", "k_res", "v_res")}
    lc.kv_seq_len = 128
    lc.ensure_room(1)
    assert lc.cap == 256 and lc.n_pages == 4
    assert torch.equal(lc.k_code[:, :, :2], before["k_code"]) and torch.equal(lc.k_scale[:, :, :2], before["k_scale"])
    assert torch.equal(lc.k_mn[:, :, :2], before["k_mn"]) and torch.equal(lc.v_code[:, :, :128], before["v_code"])
    assert torch.equal(lc.v_scale[:, :, :128], before["v_scale"]) and torch.equal(lc.v_mn[:, :, :128], before["v_mn"])
    assert torch.equal(lc.k_res, before["k_res"]) and torch.equal(lc.v_res, before["v_res"])
    assert lc.k_code.stride(2) == 3 * lc.k_code.stride(1)      # page stride = nh_kv head slabs: pages outside heads
    lc.reserve(10)                                             # never shrinks
    assert lc.cap == 256


def test_decoder_wrapper_accepts_a_patched_hf_config():
    """The reference's usage (README.md:72-75): an HF LlamaConfig with k_bits / v_bits / group_size / residual_length
    patched on goes straight into the model class."""
    pytest = __import__("pytest")
    transformers = pytest.importorskip("transformers")
    from kivi_amd.llama import LlamaForCausalLM_KIVI
    cfg = transformers.LlamaConfig(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, num_hidden_layers=1,
                                   intermediate_size=512, vocab_size=64)
    cfg.k_bits, cfg.v_bits, cfg.group_size, cfg.residual_length = 2, 2, 32, 32
    m = LlamaForCausalLM_KIVI(cfg)
    att = m.model.layers[0].self_attn
    assert (att.k_bits, att.v_bits, att.group_size, att.residual_length) == (2, 2, 32, 32)
    assert att.num_key_value_heads == 1 and att.head_dim == 128


def _layer_desc(lib, **over):
    """A kivi_layer_desc over fake (never dereferenced) device pointers, for argument-validation tests."""
    from kivi_amd import _lib
    f = dict(B=2, nh_kv=2, D=128, k_bits=2, v_bits=2, group_size=32, residual_length=32, inv_scale=0.088,
             cap=256, page_tokens=2048, v_window_rows=65, s_pitch=264,
             k_code=0x1000, kc_sb=1, kc_sh=1, kc_sp=1, kc_sr=128, k_scale=0x1000, k_mn=0x1000, ks_sb=1, ks_sh=1, ks_sp=1,
             ks_sr=64, k_res=0x1000, kr_sb=2 * 32 * 128, kr_sh=32 * 128, kr_st=128,
             v_code=0x1000, vc_sb=1, vc_sh=1, vc_sr=8, v_scale=0x1000, v_mn=0x1000, vs_sb=1, vs_sh=1, vs_sr=4,
             v_res=0x1000, vr_sb=2 * 65 * 128, vr_sh=65 * 128, vr_st=128, scores=0x1000, s_sb=1, s_sh=1,
             workspace=None, workspace_bytes=0)
    f.update(over)
    return _lib.LayerDesc(**f)


@pytest.mark.parametrize("over,msg", [(dict(k_bits=3), b"k_bits"), (dict(group_size=24), b"group_size"),
                                      (dict(residual_length=48), b"residual_length"), (dict(k_res=None), b"null")])
def test_decode_layer_refuses_before_anything_is_committed(lib, over, msg):
    """kivi_decode_layer (llama_kivi.py:314-399 in one call): everything its K flush could reject is validated before
    the attend launch, so a refused step leaves the caller's six lengths untouched (no half-advanced cache)."""
    import ctypes
    d = _layer_desc(lib, **over)
    state = (ctypes.c_int64 * 6)(64, 31, 63, 0, 32, 95)      # the NEXT step would flush K (residual 31 + 1 == R)
    before = list(state)
    rc = lib.kivi_decode_layer(ctypes.byref(d), state, 0x1000, 1, 1, 4, 0x1000, 1, 1, 0x1000, 1, 1, None, 0, 0x1000, 1, 1, None)
    assert rc < 0 and msg in lib.kivi_last_error(), lib.kivi_last_error()
    assert list(state) == before
    # inconsistent lengths are refused too
    bad = (ctypes.c_int64 * 6)(64, 31, 63, 0, 32, 96)
    d = _layer_desc(lib)
    assert lib.kivi_decode_layer(ctypes.byref(d), bad, 0x1000, 1, 1, 4, 0x1000, 1, 1, 0x1000, 1, 1, None, 0, 0x1000, 1, 1, None) < 0
    assert list(bad) == [64, 31, 63, 0, 32, 96]


def test_cache_tuples_are_single_use_handles():
    """The reference's past_key_value is an immutable snapshot (llama_kivi.py:454-455); the in-place cache's tuple is a
    handle.  Indexing or replaying one after the cache has advanced raises instead of silently using the later state."""
    from types import SimpleNamespace
    from kivi_amd.attention import LlamaAttention_KIVI
    from kivi_amd.cache import KiviConfig, KiviLayerCache
    lc = KiviLayerCache(KiviConfig(2, 2, 32, 32), 1, 2, 128, 64, "cpu")
    lc.kv_seq_len = lc.k_res_len = lc.v_res_len = 5
    t = lc.as_tuple()
    assert t[-1] == 5 and t[1].shape == (1, 2, 5, 128) and t[0] is None      # materialised while fresh: fine
    stale = lc.as_tuple()
    lc.kv_seq_len = lc.k_res_len = lc.v_res_len = 6                           # the cache moves on
    with pytest.raises(RuntimeError, match="stale"):
        stale[1]
    assert stale[-1] == 5                                                     # the length member never touches the cache
    cfg = SimpleNamespace(hidden_size=256, num_attention_heads=2, num_key_value_heads=2, max_position_embeddings=131072,
                          rope_theta=10000.0, k_bits=2, v_bits=2, group_size=32, residual_length=32)
    attn = LlamaAttention_KIVI(cfg)
    with pytest.raises(RuntimeError, match="single-use"):
        attn(torch.zeros(1, 1, 256), past_key_value=stale, use_cache=True)
    # clone() = an independent cache for continuing one prefix twice
    c2 = lc.clone()
    c2.k_res[...] = 1.0
    assert c2.kv_seq_len == lc.kv_seq_len and c2.k_res.data_ptr() != lc.k_res.data_ptr()
    assert c2.k_code.stride() == lc.k_code.stride()


def test_initial_capacity_follows_the_prompt_not_the_context_window():
    """A 128k-context config must not pre-allocate 128k tokens of cache per sequence (the reference's tuple grows with
    the sequence); kivi_max_cache_len is an opt-in reservation."""
    from types import SimpleNamespace
    from kivi_amd.attention import LlamaAttention_KIVI
    base = dict(hidden_size=256, num_attention_heads=2, num_key_value_heads=1, max_position_embeddings=131072,
                rope_theta=500000.0, k_bits=2, v_bits=2, group_size=32, residual_length=128)
    a = LlamaAttention_KIVI(SimpleNamespace(**base))
    assert a._capacity(1000) == 1000 + 128
    b = LlamaAttention_KIVI(SimpleNamespace(**base, kivi_max_cache_len=9000))
    assert b._capacity(1000) == 9000 and b._capacity(20000) == 20000 + 128


def test_rope_scaling_matches_hf():
    """config.json rope_scaling (Llama-3.1 'llama3', 'linear') gives HF's frequencies; unknown types raise."""
    from types import SimpleNamespace
    from kivi_amd.attention import _rope_inv_freq
    base = 1.0 / (500000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))
    assert torch.equal(_rope_inv_freq(SimpleNamespace(), 128, 500000.0), base)
    assert torch.equal(_rope_inv_freq(SimpleNamespace(rope_scaling={"type": "linear", "factor": 4.0}), 128, 500000.0), base / 4)
    rs = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
          "original_max_position_embeddings": 8192}
    got = _rope_inv_freq(SimpleNamespace(rope_scaling=rs), 128, 500000.0)
    wavelen = 2 * torch.pi / base
    assert torch.equal(got[wavelen < 8192 / 4.0], base[wavelen < 8192 / 4.0])          # high frequencies untouched
    assert torch.allclose(got[wavelen > 8192], base[wavelen > 8192] / 8.0, rtol=1e-6)   # low frequencies / factor
    assert bool((got <= base).all() and (got >= base / 8.0 * (1 - 1e-6)).all())
    try:
        from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
        import transformers
        hf = transformers.LlamaConfig(hidden_size=4096, num_attention_heads=32, rope_theta=500000.0, rope_scaling=dict(rs),
                                      max_position_embeddings=131072)
        ref, _ = ROPE_INIT_FUNCTIONS["llama3"](hf, "cpu")
        assert torch.allclose(got, ref, rtol=1e-6, atol=0)
    except Exception as e:  # transformers API drift: the closed-form checks above still hold
        print("HF comparison skipped:", e)
    with pytest.raises(NotImplementedError):
        _rope_inv_freq(SimpleNamespace(rope_scaling={"rope_type": "yarn", "factor": 2.0}), 128, 10000.0)


def test_mistral_module_reads_a_mistral_config():
    """models/mistral_kivi.py:69-109: bias-free projections, GQA ratio 4, sliding_window carried (never applied)."""
    pytest.importorskip("transformers")
    import transformers
    import models.mistral_kivi as M
    cfg = transformers.MistralConfig(hidden_size=512, num_attention_heads=8, num_key_value_heads=2, num_hidden_layers=1,
                                     intermediate_size=256, vocab_size=64, sliding_window=4096)
    cfg.k_bits, cfg.v_bits, cfg.group_size, cfg.residual_length = 2, 2, 32, 128
    m = M.MistralForCausalLM_KIVI(cfg)
    att = m.model.layers[0].self_attn
    assert isinstance(att, M.MistralAttention_KIVI) and issubclass(M.MistralFlashAttention_KIVI, M.MistralAttention_KIVI)
    assert M.MistralAttention_KIVI._prefill_uses_mask and not M.MistralFlashAttention_KIVI._prefill_uses_mask   # eager vs flash prompt pass
    assert att.sliding_window == 4096 and att.num_key_value_groups == 4 and att.q_proj.bias is None
    assert att.residual_length == 128 and att.k_proj.weight.shape == (2 * 64, 512)


def _gqa_args(**over):
    """kivi_gqa_decode_args over fake (never dereferenced) device pointers."""
    from kivi_amd import _lib
    f = dict(B=2, nh=8, nh_kv=2, D=128, group_size=32, bits=2, inv_scale=0.088,
             q=0x1000, q_sb=8 * 128, q_sh=128, mask=None, mask_sb=0,
             kt=0x1000, kt_sb=2 * 6144, kt_sh=6144, kt_ss=2 * 2 * 6144, Tq=512,
             kres=0x1000, kres_sb=2 * 32 * 128, kres_sh=32 * 128, kres_st=128, knew=0x1000, knew_sb=2 * 128, knew_sh=128, k_res_len=7,
             vt=0x1000, vt_sb=2 * 6144, vt_sh=6144, vt_ss=2 * 2 * 6144, Tv=487,
             vres=0x1000, vres_sb=2 * 65 * 128, vres_sh=65 * 128, vres_st=128, v_win_start=0, v_res_len=32,
             vnew=0x1000, vnew_sb=2 * 128, vnew_sh=128, v_flush=1,
             scores=0x1000, s_sb=8 * 528, s_sh=528, stats=0x1000, stats_bytes=2 * 8 * 5 * 2 * 4,
             workspace=0x1000, workspace_bytes=65536 + 4 * (1 + 1) * 2 * 4 * 128 * 4, out=0x1000, out_sb=8 * 128, out_sh=128,
             residual_length=32, v_window_rows=65, kt_superblocks=2, vt_superblocks=2, flags=0, kt_range=0x1000, vt_range=0x1000)
    f.update(over)
    return _lib.GqaDecodeArgs(**f)


@pytest.mark.parametrize("over,rc_expected,msg", [
    (dict(nh=6), -3, b"nh / nh_kv"),                        # ratio 3: not on the matrix pipe (KIVI_EUNSUPPORTED)
    (dict(bits=8), -3, b"2- and 4-bit"),
    (dict(bits=4, nh=16), -3, b"4-bit codes on the matrix pipe need nh / nh_kv in {1, 4}"),  # 4-bit: ratio 4 (round 4) or multi-head (round 6)
    (dict(bits=4, kt_ss=6144), None, b"16-byte aligned super-blocks"),                      # ... in 10240-word super-blocks
    (dict(Tq=500), None, b"inconsistent lengths"),          # packed keys come in whole 32-token blocks
    (dict(Tv=480), None, b"inconsistent lengths"),          # Tq + k_res != Tv + v_res
    (dict(s_sh=516), None, b"score rows"),                  # rows must hold the step and be 16-byte aligned
    (dict(stats_bytes=16), None, b"statistics buffer"),
    (dict(workspace_bytes=65536), None, b"workspace"),
    (dict(workspace_bytes=65536 + 4 * 1 * 2 * 4 * 128 * 4), None, b"workspace"),   # one slot per slice is not enough: + the window block's
    (dict(vnew=0x1004), None, b"value rows"),
    (dict(kt_range=None), None, b"range flags"),              # every store comes with its range flags (include/kivi_hip.h)
    (dict(vt_range=0x1002), None, b"range flags"),
    # what the step writes must lie inside the caller's buffers: K append row, V append row, the VT slot of the
    # token leaving the window, the packed prefix
    (dict(k_res_len=32, Tq=480, Tv=480), None, b"residual_length"),
    (dict(residual_length=16), None, b"residual_length"),
    (dict(v_win_start=33), None, b"window rows"),
    (dict(v_window_rows=32), None, b"window rows"),
    (dict(vt_superblocks=0), None, b"exceed the stores"),
    (dict(Tq=1024, Tv=999, kt_superblocks=1), None, b"exceed the stores"),
    (dict(v_res_len=31, Tv=488), None, b"v_flush"),
    # ring window (flags 4): residual_length + 1 rows are enough, the start may sit anywhere inside the buffer
    (dict(flags=4, v_window_rows=32), None, b"window rows"),
    (dict(flags=4, v_window_rows=33, v_win_start=33), None, b"window rows"),
])
def test_gqa_decode_validates_before_launching(lib, over, rc_expected, msg):
    """kivi_gqa_decode (the grouped-query layer step, llama_kivi.py:314-399 / mistral_kivi.py:381-445) refuses bad shapes,
    lengths and scratch sizes with a message before anything is enqueued -- callable without a GPU for that reason."""
    import ctypes
    a = _gqa_args(**over)
    rc = lib.kivi_gqa_decode(ctypes.byref(a), None)
    assert rc < 0 and (rc_expected is None or rc == rc_expected), rc
    assert msg in lib.kivi_last_error(), lib.kivi_last_error()
    assert lib.kivi_gqa_decode(None, None) < 0


def _mf_layer_desc(**over):
    """A kivi_mf_layer_desc over fake (never dereferenced) device pointers, for argument-validation tests."""
    from kivi_amd import _lib
    f = dict(B=2, nh_kv=2, D=128, bits=2, group_size=32, residual_length=32, inv_scale=0.088,
             cap=512, v_window_rows=33, s_pitch=520,
             kt=0x1000, kt_sb=2 * 6144, kt_sh=6144, kt_ss=2 * 2 * 6144, vt=0x1000, vt_sb=2 * 6144, vt_sh=6144, vt_ss=2 * 2 * 6144,
             k_res=0x1000, kr_sb=2 * 32 * 128, kr_sh=32 * 128, kr_st=128, v_res=0x1000, vr_sb=2 * 33 * 128, vr_sh=33 * 128, vr_st=128,
             scores=0x1000, s_sb=8 * 520, s_sh=520, stats=0x1000, stats_bytes=1 << 16, workspace=0x1000, workspace_bytes=1 << 20,
             flags=4, kt_range=0x1000, vt_range=0x1000)
    f.update(over)
    return _lib.MfLayerDesc(**f)


@pytest.mark.parametrize("over,msg", [(dict(bits=8), b"2- and 4-bit"), (dict(bits=4, nh_kv=4), b"need nh / nh_kv in {1, 4}"), (dict(residual_length=48), b"inconsistent lengths"),
                                      (dict(kt=None), b"null"), (dict(vt_range=None), b"null"), (dict(cap=64), b"capacity"), (dict(kt_ss=100), b"alignment"),
                                      (dict(v_window_rows=32), b"ring window"), (dict(s_pitch=90), b"score rows")])
def test_mf_decode_layer_refuses_before_anything_is_committed(lib, over, msg):
    """kivi_mf_decode_layer (the layer step on the matrix-pipe layout: kivi_gqa_decode + bookkeeping + kivi_kt_pack flush):
    what its K flush or its launches could reject is validated first, a refused step leaves the six lengths untouched."""
    import ctypes
    d = _mf_layer_desc(**over)
    state = (ctypes.c_int64 * 6)(64, 31, 63, 5, 32, 95)      # the NEXT step would flush K (residual 31 + 1 == R)
    before = list(state)
    rc = lib.kivi_mf_decode_layer(ctypes.byref(d), state, 0x1000, 8 * 128, 128, 8, 0x1000, 2 * 128, 128, 0x1000, 2 * 128, 128, None, 0,
                                  0x1000, 8 * 128, 128, None)
    assert rc < 0 and msg in lib.kivi_last_error(), lib.kivi_last_error()
    assert list(state) == before
    bad = (ctypes.c_int64 * 6)(64, 31, 63, 0, 32, 96)          # inconsistent lengths
    d = _mf_layer_desc()
    assert lib.kivi_mf_decode_layer(ctypes.byref(d), bad, 0x1000, 8 * 128, 128, 8, 0x1000, 2 * 128, 128, 0x1000, 2 * 128, 128, None, 0,
                                    0x1000, 8 * 128, 128, None) < 0
    assert list(bad) == [64, 31, 63, 0, 32, 96]
    assert lib.kivi_mf_decode_layer(None, None, None, 0, 0, 0, None, 0, 0, None, 0, 0, None, 0, None, 0, 0, None) < 0


def test_mf_cache_ring_window_view():
    """The fp16 value window of the matrix-pipe cache is a ring of R + 1 rows (round 3): the 9-tuple member V_full is read
    through it in token order, also when the live rows wrap."""
    from kivi_amd.attention import KiviConfig, make_layer_cache
    lc = make_layer_cache(KiviConfig(2, 2, 32, 32), 1, 2, 128, 100, "cpu", num_heads=2)
    assert lc.ring and lc.v_res.shape[2] == 33
    lc.v_res.copy_(torch.arange(33, dtype=torch.float16)[None, None, :, None].expand(1, 2, 33, 128))
    lc.v_res_start, lc.v_res_len = 30, 5
    assert lc.v_res_view()[0, 0, :, 0].tolist() == [30.0, 31.0, 32.0, 0.0, 1.0]
    lc.v_res_start, lc.v_res_len = 3, 4
    assert lc.v_res_view()[0, 0, :, 0].tolist() == [3.0, 4.0, 5.0, 6.0]
    r8 = make_layer_cache(KiviConfig(2, 2, 32, 32), 1, 1, 128, 100, "cpu", num_heads=8)       # nh / nh_kv = 8: the same kernels since round 4
    assert r8.ring and r8.v_res.shape[2] == 33


def test_layer_cache_factory_picks_the_layout():
    """make_layer_cache: the matrix-pipe layout only for the shapes its kernels cover, the hook-state layout otherwise
    (constructing either on the CPU allocates plain tensors; no kernel is involved)."""
    from kivi_amd.attention import KiviConfig, KiviLayerCache, KiviLayerCacheMF, make_layer_cache
    mf = make_layer_cache(KiviConfig(2, 2, 32, 128), 1, 8, 128, 1000, "cpu", num_heads=32)
    assert isinstance(mf, KiviLayerCacheMF) and mf.n_sb == 2 and mf.kt.shape == (1, 8, 2, 6144) and not mf.kt.any()
    assert mf.kt.stride(2) == 8 * 6144 and mf.kt.stride(1) == 6144      # super-block index outside the head index in memory
    mha = make_layer_cache(KiviConfig(2, 2, 32, 32), 1, 8, 128, 1000, "cpu", num_heads=8)       # round 3: multi-head models too
    assert isinstance(mha, KiviLayerCacheMF) and mha.n_sb == 2
    for kw in (dict(num_heads=24), dict(num_heads=16), dict(num_heads=None)):
        assert isinstance(make_layer_cache(KiviConfig(2, 2, 32, 128), 1, 8, 128, 1000, "cpu", **kw), KiviLayerCache)
    mf4 = make_layer_cache(KiviConfig(4, 4, 32, 128), 1, 8, 128, 1000, "cpu", num_heads=32)     # round 4: 4-bit K / V, nh / nh_kv = 4
    assert isinstance(mf4, KiviLayerCacheMF) and mf4.kt.shape == (1, 8, 2, 10240) and mf4.vt.shape == (1, 8, 2, 10240)
    mf41 = make_layer_cache(KiviConfig(4, 4, 32, 128), 1, 8, 128, 1000, "cpu", num_heads=8)     # round 6: 4-bit multi-head (LongChat-7B / Llama-2-7B + KIVI-4)
    assert isinstance(mf41, KiviLayerCacheMF) and mf41.kt.shape == (1, 8, 2, 10240)
    for kw in (dict(num_heads=16), dict(num_heads=64)):                                         # ... other ratios: the hook-state layout
        assert isinstance(make_layer_cache(KiviConfig(4, 4, 32, 128), 1, 8, 128, 1000, "cpu", **kw), KiviLayerCache)
    assert isinstance(make_layer_cache(KiviConfig(4, 2, 32, 128), 1, 8, 128, 1000, "cpu", num_heads=32), KiviLayerCache)
    assert isinstance(make_layer_cache(KiviConfig(2, 2, 64, 128), 1, 8, 128, 1000, "cpu", num_heads=32), KiviLayerCache)
    mf.reserve(3000)
    assert mf.n_sb == 6 and mf.cap >= 3000


def test_store_wrappers_refuse_without_range_flags(lib):
    """kivi_kt_pack / kivi_vt_pack / kivi_gqa_scores / kivi_gqa_output / the relayouts towards the layout refuse a store without
    its range flags before touching a device (include/kivi_hip.h, RANGE FLAGS); reading a store (to_ref) does not need them."""
    st = (0x1000, 2 * 6144, 6144, 2 * 2 * 6144)
    assert lib.kivi_kt_pack(0x1000, 8192, 4096, 128, *st, None, 0, 1, 2, 32, 128, 32, 2, None) < 0 and b"range flags" in lib.kivi_last_error()
    assert lib.kivi_vt_pack(0x1000, 8192, 4096, 128, *st, None, 1, 2, 32, 128, 32, 2, None) < 0 and b"range flags" in lib.kivi_last_error()
    assert lib.kivi_gqa_scores(0x1000, 1024, 128, *st, None, 0x1000, 1024, 128, 1, 8, 2, 128, 32, 32, 2, None) < 0
    assert b"range flags" in lib.kivi_last_error()
    assert lib.kivi_gqa_output(0x1000, 1024, 128, *st, None, 0x1000, 1024, 128, 1, 8, 2, 128, 32, 32, 2, 0x1000, 1 << 20, None) < 0
    assert b"range flags" in lib.kivi_last_error()
    rel = (0x1000, 4096, 2048, 16, 0x1000, 0x1000, 1024, 512, 4)
    assert lib.kivi_kt_relayout(0, *st, None, *rel, 1, 2, 32, 128, 32, 2, None) < 0 and b"range flags" in lib.kivi_last_error()
    assert lib.kivi_vt_relayout(0, *st, None, *rel, 1, 2, 32, 128, 32, 2, None) < 0 and b"range flags" in lib.kivi_last_error()
    assert lib.kivi_kt_relayout(1, *st, None, *rel, 1, 2, 0, 128, 32, 2, None) == 0        # (T = 0: nothing to launch)


def test_store_range_flags_travel_with_the_store():
    """kivi_amd.quant.mfma keeps a store's range flags in the same allocation (behind the super-blocks): growth and clone carry
    them along, prefill of a reused cache clears them, foreign tensors are refused."""
    from kivi_amd.attention import KiviConfig, make_layer_cache
    from kivi_amd.quant import mfma
    st = mfma.alloc_store(2, 3, 2, "cpu")
    fl = mfma.range_flags(st)
    assert fl.shape == (2, 3) and not fl.any() and fl.data_ptr() == st.data_ptr() + 2 * 2 * 3 * 6144 * 4
    fl[1, 2] = 1
    st.fill_(-1)                                         # the view covers the super-blocks only
    assert mfma.range_flags(st).tolist() == [[0, 0, 0], [0, 0, 1]]
    with pytest.raises(ValueError):
        mfma.range_flags(torch.zeros((2, 2, 3, 6144), dtype=torch.int32).permute(0, 2, 1, 3))
    lc = make_layer_cache(KiviConfig(2, 2, 32, 32), 2, 3, 128, 600, "cpu", num_heads=3)
    mfma.range_flags(lc.vt)[0, 1] = 1
    lc.reserve(3000)
    assert lc.n_sb == 6 and mfma.range_flags(lc.vt).tolist() == [[0, 1, 0], [0, 0, 0]] and not mfma.range_flags(lc.kt).any()
    c = lc.clone()
    assert mfma.range_flags(c.vt).tolist() == [[0, 1, 0], [0, 0, 0]] and c.vt.data_ptr() != lc.vt.data_ptr()


def test_product_python_reads_the_environment_only_through_the_tuning_module():
    """the product reads no tuning knob unless the process was started with KIVI_TUNING=1.  The only module
    of the package that touches os.environ is kivi_amd/_tuning.py (build.py: the HIPCC path of the build script)."""
    import glob
    offenders = []
    for path in sorted(glob.glob(os.path.join(ROOT, "kivi_amd", "**", "*.py"), recursive=True)):
        rel = os.path.relpath(path, ROOT)
        if rel in ("kivi_amd/_tuning.py", "kivi_amd/build.py"):
            continue
        if re.search(r"\bos\.environ\b|\bgetenv\b", open(path).read()):
            offenders.append(rel)
    assert not offenders, offenders
    import importlib
    from kivi_amd import _tuning
    saved = {k: os.environ.get(k) for k in ("KIVI_TUNING", "KIVI_NO_MFMA_LAYOUT")}
    try:
        os.environ["KIVI_NO_MFMA_LAYOUT"] = "1"
        os.environ.pop("KIVI_TUNING", None)
        importlib.reload(_tuning)
        assert not _tuning.ENABLED and not _tuning.flag("KIVI_NO_MFMA_LAYOUT") and _tuning.knob("KIVI_NO_MFMA_LAYOUT", "d") == "d"
        os.environ["KIVI_TUNING"] = "1"
        importlib.reload(_tuning)
        assert _tuning.ENABLED and _tuning.flag("KIVI_NO_MFMA_LAYOUT")
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        importlib.reload(_tuning)


@pytest.mark.parametrize("T0,R", [(0, 32), (5, 32), (70, 32), (300, 128), (511, 64), (4080, 32)])
def test_step_advance_follows_the_reference_cache_policy(lib, T0, R):
    """kivi_mf_step_advance + the caller's K flush = the length bookkeeping of the hook (llama_kivi.py:343-356, :386-399,
    :425-452): after every step K holds floor(kv / R) * R packed + kv mod R residual keys, V max(kv - R, 0) packed + min(kv, R)
    window values; the ring start moves with every quantised value; v_flush announces what the NEXT step does.  kivi_mf_step_key
    changes exactly when a store gains a super-block (or the one- / two-launch decision flips)."""
    import ctypes
    from kivi_amd import _lib
    rows = R + 1
    hs = _lib.MfStep(Tq=(T0 // R) * R, Tv=max(T0 - R, 0), k_res_len=T0 % R, v_res_len=min(T0, R), v_win_start=0, v_flush=int(min(T0, R) + 1 > R))
    kv, start = T0, 0
    keys = []
    for s in range(3 * R + 600):
        keys.append(lib.kivi_mf_step_key(ctypes.byref(hs), 2, 8, 2, R, 4))
        assert keys[-1] >= 0
        rc = lib.kivi_mf_step_advance(ctypes.byref(hs), R, rows)
        assert rc in (0, 1)
        if kv >= R:
            start = (start + 1) % rows
        kv += 1
        if rc == 1:
            assert hs.k_res_len == R
            hs.Tq += R
            hs.k_res_len = 0
        assert (hs.Tq, hs.k_res_len) == ((kv // R) * R, kv % R)
        assert (hs.Tv, hs.v_res_len, hs.v_win_start) == (max(kv - R, 0), min(kv, R), start)
        assert hs.v_flush == int(hs.v_res_len + 1 > R)
    changes = sum(1 for a, b in zip(keys, keys[1:]) if a != b)
    kv0, kv1 = T0, T0 + len(keys) - 1
    sb = lambda t: (t + 511) // 512                      # noqa: E731
    expect = (sb((kv1 // R) * R) - sb((kv0 // R) * R)) + (sb(max(kv1 - R, 0)) - sb(max(kv0 - R, 0)))
    assert expect <= changes <= expect + 3, (changes, expect)      # + the first V flush / a one- vs two-launch flip
    bad = _lib.MfStep(Tq=64, Tv=10, k_res_len=3, v_res_len=32, v_win_start=0, v_flush=1)
    assert lib.kivi_mf_step_advance(ctypes.byref(bad), 32, 33) < 0
    assert lib.kivi_mf_step_key(None, 1, 1, 1, 32, 0) == -1


def test_launch_plan_is_a_function_of_the_geometry_class(lib):
    """kivi_mf_launch_plan (include/kivi_hip.h): 0 = two launches, S >= 1 = one launch with S slices per row.  The plan of a step's
    whole geometry class (dyn = 1: what a captured graph replays) never changes inside the class, forced forms are honoured when they
    are valid and refused (0 = two launches) when they are not, slices always fit the LDS, and the shapes the measurements name get
    the forms DESIGN section 3.6 lists."""
    from kivi_amd import _lib
    P = lambda B, nh, nkv, Tq, kres, R, flags=0, bits=2, dyn=0: lib.kivi_mf_launch_plan(B, nh, nkv, Tq, kres, R, flags, bits, dyn)   # noqa: E731
    assert P(32, 32, 32, 4064, 16, 32) == 1                      # headline: a block per row (mf_row_kernel)
    assert P(64, 32, 8, 8064, 0, 128) == 1                       # BASELINE config 4: a block per unit
    assert P(16, 32, 8, 32640, 0, 128) == 4                      # config-5 per-GPU slice: 32k rows do not fit the LDS -> 4 slices of 16 super-blocks
    assert P(16, 64, 8, 8064, 0, 128) == 4                       # 70B-like slice (nh / nh_kv = 8: 4608 keys per block)
    assert P(64, 64, 8, 8064, 0, 128) == 2                       # ... at B = 64: the fewest slices that fit (1024 blocks: ticket ids)
    assert P(4, 32, 8, 8064, 0, 128) == 4 and P(32, 32, 8, 8064, 0, 128) == 1 and P(16, 32, 8, 8064, 0, 128) == 2
    assert P(8, 32, 8, 2048, 0, 128) == 0                        # few short rows: two launches
    assert P(4, 32, 32, 4064, 16, 32) == 1
    # multi-head rows beyond 16 super-blocks (LongChat-7B-32K, docs/long_bench.md:5-26): two launches (the sliced one-launch form was
    # measured in round 6 and loses: profiles/r06_long_rows.log); KIVI_GQA_SLICES(n) still reaches it
    assert P(1, 32, 32, 32736, 16, 32) == 0 and P(8, 32, 32, 32768, 0, 128) == 0 and P(16, 32, 32, 16384, 0, 128) == 0
    assert P(8, 32, 32, 32768, 0, 128, _lib.gqa_slices(5)) == 5 and P(8, 32, 32, 32768, 0, 128, _lib.gqa_slices(4)) == 0
    assert P(8, 32, 32, 32768, 0, 128, _lib.GQA_FORCE_SPLIT) == 0 and P(8, 32, 32, 32768, 0, 128, _lib.GQA_FORCE_ROW) == 0
    assert P(64, 32, 8, 8064, 0, 128, _lib.GQA_FORCE_SPLIT) == 0 and P(4, 32, 8, 8064, 0, 128, _lib.GQA_FORCE_ROW) == 1
    assert P(2, 8, 2, 1024, 76, 128, _lib.gqa_slices(2)) == 2 and P(2, 8, 2, 1024, 76, 128, _lib.gqa_slices(3)) == 0   # 2 super-blocks: no 3 slices
    assert P(2, 4, 4, 1088, 12, 32, _lib.gqa_slices(2)) == 2 and P(2, 4, 4, 1088, 12, 32, _lib.gqa_slices(1)) == 1     # nh == nh_kv through the slice kernel
    assert P(0, 32, 8, 64, 0, 32) == -1 and P(2, 32, 5, 64, 0, 32) == -1
    # Tq = 0 (before the first K flush): one, empty, slice -- the one-launch form when the units fill the chip (advisor r5)
    assert P(64, 32, 8, 0, 5, 128) == 1 and P(32, 64, 8, 0, 5, 128) == 1 and P(2, 32, 8, 0, 5, 128) == 0 and P(32, 32, 32, 0, 5, 32) == 1
    # the bands where only the class bound exceeded a block (advisor r4 / r5): the caps are whole super-blocks + a full residual now,
    # and eager steps are planned for the class bound too
    assert P(32, 32, 32, 8192, 31, 32) == 1 and P(32, 32, 32, 8160, 0, 32) == 1 and P(32, 32, 32, 8192 + 32, 0, 32) == 0
    assert P(64, 32, 8, 9216, 127, 128) == 1 and P(64, 32, 8, 9088, 0, 128) == 1 and P(64, 32, 8, 9216 + 128, 0, 128) == 2
    for nh, nkv, cap in ((32, 8, 9216 + 128), (64, 8, 4608), (32, 32, 8192 + 128)):
        for B in (1, 4, 16, 64):
            for Tq in range(0, 40000, 1664):
                for R in (32, 128):
                    Tq_ = Tq // R * R
                    nsb = (Tq_ + 511) // 512
                    cls = P(B, nh, nkv, Tq_, 0, R, 0, 2, 1)
                    for kres in (0, R // 2, R - 1):              # one class, one plan: eager (dyn = 0) and device-resident lengths alike
                        assert P(B, nh, nkv, Tq_, kres, R, 0, 2, 1) == cls == P(B, nh, nkv, Tq_, kres, R, 0, 2, 0)
                    S = cls
                    if S > 1:                                    # the longest row of a block: max(ceil(nsb / S), 2) super-blocks + the residual
                        assert S <= nsb and max((nsb + S - 1) // S, 2) * 512 + R + 1 <= (8192 if nh == nkv else cap)
                    elif S == 1:
                        assert nsb * 512 + R <= cap


def test_device_error_is_clear_without_a_device(lib):
    """kivi_device_error (include/kivi_hip.h): the sticky error of sliced launches; nothing was launched -> 0, and asking does not
    allocate or touch a device."""
    assert lib.kivi_device_error() == 0 and lib.kivi_device_error() == 0


def test_decode_layer_dyn_refuses_before_launching(lib):
    """kivi_mf_decode_layer_dyn validates the step being captured and the room its geometry class needs (score pitch, capacity)
    without a device."""
    import ctypes
    from kivi_amd import _lib
    d = _mf_layer_desc(s_pitch=600)
    ok = _lib.MfStep(Tq=64, Tv=63, k_res_len=31, v_res_len=32, v_win_start=5, v_flush=1)
    args = (0x1000, 8 * 128, 128, 8, 0x1000, 2 * 128, 128, 0x1000, 2 * 128, 128, None, 0, 0x1000, 8 * 128, 128, None)
    for hs, dev, desc, msg in ((ok, None, d, b"null"), (_lib.MfStep(Tq=64, Tv=63, k_res_len=31, v_res_len=32, v_win_start=5, v_flush=0), 0x1000, d, b"inconsistent"),
                               (ok, 0x1000, _mf_layer_desc(s_pitch=520), b"score pitch"), (ok, 0x1000, _mf_layer_desc(flags=0, s_pitch=600), b"ring window"),
                               (ok, 0x1004, d, b"8-byte aligned")):
        rc = lib.kivi_mf_decode_layer_dyn(ctypes.byref(desc), ctypes.byref(hs), dev, *args)
        assert rc < 0 and msg in lib.kivi_last_error(), (rc, lib.kivi_last_error())
    assert lib.kivi_mf_step_upload(None, None, None) < 0


def test_tuning_build_still_compiles():
    """The product library carries no environment knob and no losing / diagnostic instantiation; they live behind
    -DKIVI_TUNING (python -m kivi_amd.build --tuning; __graft_entry__.build() builds it too since round 6, for the fault-injection test).
    Its front-end pass is checked here independently of any built file (syntax + template instantiation of every source, no code generation)."""
    import glob
    import os
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    srcs = sorted(glob.glob(os.path.join(root, "kivi_amd", "csrc", "*.hip")))
    procs = [subprocess.Popen([hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-DKIVI_TUNING", "-Wno-unused-value", s],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for s in srcs]
    for s, p_ in zip(srcs, procs):
        out, _ = p_.communicate()
        assert p_.returncode == 0, (s, out[-2000:])
    lib = os.path.join(root, "kivi_amd", "libkivi_hip.so")
    assert b"getenv" not in subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True).stdout, "the product library reads the environment"
