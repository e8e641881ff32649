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
    assert lib.kivi_abi_version() == 1


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
