"""GPU: the MFMA-friendly cache layout (kivi_amd/quant/mfma.py, kivi_amd/csrc/kivi_gqa.hip, kivi_mf.hip; nh / nh_kv in
{1, 4, 8}): relayout kernels are bit-exact inverses and reproduce the hook-state tensors of the reference pack; the
matrix-pipe qK^T AND sV each agree with the oracle's restatement of gemv_cuda.cu:348-427 within the north_star GEMV bar
(1e-3), in isolation and stage by stage inside the decode step."""
import pytest
import torch

from helpers import gemv_close, make_kv, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from kivi_amd.quant import matmul, mfma, new_pack
    return mfma, new_pack, matmul


@pytest.mark.parametrize("B,nh_kv,T,off,kind", [(1, 1, 32, 0, "outlier"), (2, 3, 544, 0, "outlier"), (1, 2, 128, 480, "outlier"),
                                                  (2, 2, 1024, 64, "outlier"), (1, 2, 96, 32, "tiny")])
def test_kt_pack_equals_reference_pack(mods, oracle, B, nh_kv, T, off, kind):
    """kivi_kt_pack (direct per-channel quantise into the layout) == the reference-layout pack (bit-exact vs the
    reference's new_pack.py through the golden fixtures) after kivi_kt_relayout, at a token offset too."""
    mfma, new_pack, _ = mods
    k = make_kv(7, B, nh_kv, off + T, 128, kind).cuda()
    store = mfma.alloc_store(B, nh_kv, (off + T + 511) // 512, "cuda")
    if off:
        mfma.kt_pack(k[:, :, :off], store, 0)
    mfma.kt_pack(k[:, :, off:], store, off)
    code, scale, mn = mfma.kt_to_ref(store, off + T)
    rc, rs, rm = new_pack.quantize_and_pack_k_tmajor(k, 32, 2)
    assert same_bits(code, rc) and same_bits(scale, rs) and same_bits(mn, rm)
    oc, os_, om = oracle.quantize_and_pack_along_last_dim(k.cpu().transpose(2, 3).contiguous(), 32, 2)
    assert same_bits(code, oc) and same_bits(scale, os_) and same_bits(mn, om)
    # reverse direction reproduces the storage word for word (written region)
    store2 = mfma.alloc_store(B, nh_kv, store.shape[2], "cuda")
    mfma.kt_from_ref(store2, code, scale, mn)
    assert torch.equal(store2, store)


def test_kt_vt_pack_nan_inputs_follow_torch_min_max(mods):
    """As tests/test_pack_gpu.py::test_nan_inputs_follow_torch_min_max for the packers of the matrix-pipe layout (seen through
    the relayout kernels): a NaN makes its group's scale and zero point NaN and its codes 0, nothing else moves."""
    mfma, _, _ = mods
    k = make_kv(31, 1, 2, 64, 128, "randn")
    v = make_kv(32, 1, 2, 64, 128, "randn")
    k2, v2 = k.clone(), v.clone()
    k2[0, 1, 40, 9] = float("nan")                      # K: channel 9, token group 1 (per-channel groups along the tokens)
    v2[0, 0, 5, 100] = -float("nan")                    # V: token 5, channel group 3 (per-token groups along the channels)
    outs = []
    for kk, vv in ((k, v), (k2, v2)):
        kt, vt = mfma.alloc_store(1, 2, 1, "cuda"), mfma.alloc_store(1, 2, 1, "cuda")
        mfma.kt_pack(kk.cuda(), kt)
        mfma.vt_pack(vv.cuda(), vt)
        outs.append([t.cpu() for t in mfma.kt_to_ref(kt, 64)] + [t.cpu() for t in mfma.vt_to_ref(vt, 64)])
    (kc0, ks0, km0, vc0, vs0, vm0), (kc1, ks1, km1, vc1, vs1, vm1) = outs
    khit = torch.zeros_like(ks0, dtype=torch.bool)
    khit[0, 1, 9, 1] = True                              # K_scale_T (B, nh_kv, D, T / 32)
    vhit = torch.zeros_like(vs0, dtype=torch.bool)
    vhit[0, 0, 5, 3] = True                              # V_scale (B, nh_kv, T, D / 32)
    for s0, s1, m0, m1, hit in ((ks0, ks1, km0, km1, khit), (vs0, vs1, vm0, vm1, vhit)):
        assert torch.isnan(s1[hit]).all() and torch.isnan(m1[hit]).all()
        assert same_bits(s1[~hit], s0[~hit]) and same_bits(m1[~hit], m0[~hit])
    kw0, kw1 = kc0.reshape(1, 2, 128, 2, 2), kc1.reshape(1, 2, 128, 2, 2)      # 2 words per (channel, token group)
    vw0, vw1 = vc0.reshape(1, 2, 64, 4, 2), vc1.reshape(1, 2, 64, 4, 2)        # 2 words per (token, channel group)
    assert (kw1[khit] == 0).all() and torch.equal(kw1[~khit], kw0[~khit])
    assert (vw1[vhit] == 0).all() and torch.equal(vw1[~vhit], vw0[~vhit])


@pytest.mark.parametrize("B,nh_kv,T,kind", [(1, 1, 1, "randn"), (2, 2, 33, "outlier"), (1, 3, 512, "randn"), (2, 1, 1000, "outlier"),
                                             (1, 2, 75, "tiny_rows"), (3, 8, 2049, "randn")])
def test_vt_pack_equals_reference_pack(mods, oracle, B, nh_kv, T, kind):
    """kivi_vt_pack (per-token V quantise straight into the VT layout, llama_kivi.py:441-448) == the last-dim pack
    (bit-exact vs the reference through the golden fixtures) followed by kivi_vt_relayout, word for word, for whole and
    ragged 32-token blocks; and == the oracle after the reverse relayout.  Also through a strided view (the prompt's
    value_states[:, :, :-R])."""
    mfma, new_pack, _ = mods
    if kind == "tiny_rows":      # one-ulp ranges along the channel axis (V groups): scale 0 with the max code
        v = make_kv(9, B, nh_kv, 128, T, "tiny").transpose(2, 3).contiguous().cuda()
    else:
        v = make_kv(9, B, nh_kv, T, 128, kind).cuda()
    nsb = (T + 511) // 512
    a, b_ = mfma.alloc_store(B, nh_kv, nsb, "cuda"), mfma.alloc_store(B, nh_kv, nsb, "cuda")
    mfma.vt_pack(v, a)
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, 32, 2)
    mfma.vt_from_ref(b_, code, scale, mn)
    assert torch.equal(a, b_)
    oc, os_, om = oracle.quantize_and_pack_along_last_dim(v.cpu(), 32, 2)
    c2, s2, m2 = mfma.vt_to_ref(a, T)
    assert same_bits(c2, oc) and same_bits(s2, os_) and same_bits(m2, om)
    big = torch.zeros((B, nh_kv, T + 40, 128), dtype=torch.float16, device="cuda")
    big[:, :, :T] = v
    c = mfma.alloc_store(B, nh_kv, nsb, "cuda")
    mfma.vt_pack(big[:, :, :T], c)                       # strided view: rows of the longer tensor
    assert torch.equal(c, a)


@pytest.mark.parametrize("B,nh_kv,T", [(1, 1, 1), (2, 2, 33), (1, 3, 512), (2, 1, 1000)])
def test_vt_relayout_round_trip(mods, B, nh_kv, T):
    mfma, new_pack, _ = mods
    v = make_kv(9, B, nh_kv, T, 128).cuda()
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, 32, 2)
    store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda")
    store.fill_(-1)                                            # the relayout must zero the unwritten slots of the last block
    mfma.vt_from_ref(store, code, scale, mn)
    c2, s2, m2 = mfma.vt_to_ref(store, T)
    assert same_bits(c2, code) and same_bits(s2, scale) and same_bits(m2, mn)
    nb = (T + 31) // 32
    if T % 32:                                                 # slots of tokens >= T inside the last block read as zero
        cz, sz, mz = mfma.vt_to_ref(store, nb * 32)
        assert not cz[:, :, T:].any() and not sz[:, :, T:].view(torch.int16).any() and not mz[:, :, T:].view(torch.int16).any()


@pytest.mark.parametrize("B,nh,nh_kv,T", [(1, 4, 1, 32), (2, 8, 2, 544), (1, 8, 1, 1024), (3, 16, 2, 96), (16, 32, 8, 8192),
                                          (1, 32, 8, 4096), (1, 1, 1, 32), (2, 3, 3, 544), (1, 2, 2, 480), (32, 32, 32, 4096),
                                          (1, 32, 32, 32768)])
def test_gqa_scores_vs_oracle(mods, oracle, B, nh, nh_kv, T):
    """Matrix-pipe qK^T (ratio 1, 4 and 8; partial super-blocks; both block shapes; the BASELINE configs[1] size) against
    the oracle on sampled heads and against the VALU kernel of the hook-state layout everywhere."""
    mfma, new_pack, matmul = mods
    k = make_kv(3, B, nh_kv, T, 128, "outlier").cuda()
    q = (make_kv(4, B, nh, 1, 128) * 1.5).half().cuda()
    store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda")
    mfma.kt_pack(k, store, 0)
    out = torch.full((B, nh, 1, T + 8), 7.0, dtype=torch.float16, device="cuda")
    mfma.gqa_scores(q, store, T, out)
    assert bool((out[..., T:] == 7.0).all()) and torch.isfinite(out).all()
    code, scale, mn = new_pack.quantize_and_pack_k_tmajor(k, 32, 2)
    ref_gpu = matmul.cuda_bmm_fA_qB_outer(32, q, code, scale, mn, 2)
    ok, ratio = gemv_close(out[..., :T], ref_gpu.cpu(), rtol=1.5e-3)     # two roundings of the same exact sum apart
    assert ok, ratio
    ratio_h = nh // nh_kv
    for (b, hk) in {(0, 0), (B - 1, nh_kv - 1)}:
        hs = slice(hk * ratio_h, (hk + 1) * ratio_h)
        ref = oracle.bmm_fA_qB_outer(32, q[b:b + 1, hs].cpu(), code[b:b + 1, hk:hk + 1].cpu(), scale[b:b + 1, hk:hk + 1].cpu(),
                                     mn[b:b + 1, hk:hk + 1].cpu(), 2)
        ok, ratio = gemv_close(out[b:b + 1, hs, :, :T], ref)
        assert ok, (b, hk, ratio)


@pytest.mark.parametrize("nh,nh_kv", [(8, 2), (2, 2), (16, 2)])
def test_gqa_scores_exact_arithmetic(mods, nh, nh_kv):
    """Integer-valued inputs (quant/test.py:182-183 style): every product and sum is exact in fp32, so head mapping, the
    layout's bit positions and the hi / lo operand split must reproduce the dequantised matmul exactly."""
    mfma, new_pack, _ = mods
    B, T = 2, 1056
    g = torch.Generator().manual_seed(0)
    k = torch.randint(0, 4, (B, nh_kv, T, 128), generator=g).half().cuda()     # groups span 0..3 -> scale 1 or less, exact
    k[:, :, ::32] = 0
    k[:, :, 1::32] = 3                                                          # every group: min 0, max 3 -> scale 1, codes = values
    q = torch.randint(-5, 6, (B, nh, 1, 128), generator=g).half().cuda()
    store = mfma.alloc_store(B, nh_kv, 3, "cuda")
    mfma.kt_pack(k, store, 0)
    out = torch.empty((B, nh, 1, T), dtype=torch.float16, device="cuda")
    mfma.gqa_scores(q, store, T, out)
    ref = torch.matmul(q.float(), k.float().repeat_interleave(nh // nh_kv, dim=1).transpose(2, 3))
    assert torch.equal(out.float(), ref.half().float())


# ---------------------------------------------------------------------------------------------------------------------
# the matrix-pipe sV in isolation (VERDICT r2 #1a): kivi_gqa_output vs oracle.bmm_fA_qB_outer at the north_star GEMV bar

def _probs(kind, B, nh, T, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "softmax":        # peaked rows: a few dominant keys
        return torch.softmax(torch.randn((B, nh, 1, T), generator=g) * 3, -1).half()
    if kind == "uniform":        # the worst cancellation between the code sum and the zero-point sum
        return torch.full((B, nh, 1, T), 1.0 / T).half()
    if kind == "sparse":         # exact zeros and probabilities down to the fp16 subnormals
        p = torch.softmax(torch.randn((B, nh, 1, T), generator=g) * 12, -1)
        return p.half()
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["softmax", "uniform", "sparse"])
@pytest.mark.parametrize("B,nh,nh_kv,T", [(2, 4, 1, 33), (1, 8, 2, 544), (2, 32, 8, 8192), (1, 32, 8, 32768),
                                          (2, 2, 2, 33), (1, 3, 3, 544), (4, 32, 32, 4064), (1, 4, 4, 32768),
                                          (2, 8, 1, 33), (1, 16, 2, 544), (2, 64, 8, 8192), (1, 8, 1, 32768)])
def test_gqa_output_vs_oracle(mods, oracle, B, nh, nh_kv, T, kind):
    """out = probs @ dequant(V) on the VT layout (ratio 1, 4 and 8; ragged last block; one to 64 super-blocks; rows cut into
    slices) against the oracle's restatement of the reference kernel (gemv_cuda.cu:348-427 at llama_kivi.py:382) on
    sampled kv heads with gemv_close(rtol=1e-3), and against the VALU kernel of the hook-state layout everywhere."""
    mfma, new_pack, matmul = mods
    v = make_kv(21, B, nh_kv, T, 128, "outlier" if kind == "softmax" else "randn").cuda()
    store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda")
    mfma.vt_pack(v, store)
    pitch = (T + 7) // 8 * 8 + 8
    probs = torch.zeros((B, nh, 1, pitch), dtype=torch.float16, device="cuda")
    probs[..., :T] = _probs(kind, B, nh, T, 5).cuda()
    probs[..., T:] = 1.0                                             # beyond the row: must not be read as a probability
    out = mfma.gqa_output(probs, store, T)
    assert torch.isfinite(out).all()
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, 32, 2)
    ref_gpu = matmul.cuda_bmm_fA_qB_outer(32, probs[..., :T], code, scale, mn, 2)
    ok, ratio = gemv_close(out, ref_gpu.cpu(), rtol=1.5e-3)          # two roundings of the same exact sum apart
    assert ok, ratio
    ratio_h = nh // nh_kv
    for (b, hk) in {(0, 0), (B - 1, nh_kv - 1)}:
        hs = slice(hk * ratio_h, (hk + 1) * ratio_h)
        ref = oracle.bmm_fA_qB_outer(32, probs[b:b + 1, hs, :, :T].cpu().contiguous(), code[b:b + 1, hk:hk + 1].cpu(),
                                     scale[b:b + 1, hk:hk + 1].cpu(), mn[b:b + 1, hk:hk + 1].cpu(), 2)
        ok, ratio = gemv_close(out[b:b + 1, hs], ref, rtol=1e-3)
        assert ok, (b, hk, ratio)


def test_gqa_output_exact_arithmetic(mods):
    """Integer-valued V and power-of-two probabilities: every product and sum is exact, so the layout's bit positions, the
    centring of the codes (A x (code - 1.5) + 1.5 sum A) and the hi / lo split must reproduce the dequantised matmul
    exactly."""
    mfma, new_pack, _ = mods
    for nh, nh_kv in ((4, 1), (2, 2), (8, 1)):
        B, T = 2, 700
        g = torch.Generator().manual_seed(0)
        v = torch.randint(0, 4, (B, nh_kv, T, 128), generator=g).half()
        v[..., ::32] = 0
        v[..., 1::32] = 3                                               # every group: min 0, max 3 -> scale 1, codes = values
        v = v.cuda()
        p = (torch.randint(0, 3, (B, nh, 1, 704), generator=g).float() * 2.0 ** -10).half().cuda()
        store = mfma.alloc_store(B, nh_kv, 2, "cuda")
        mfma.vt_pack(v, store)
        out = mfma.gqa_output(p, store, T)
        ref = torch.matmul(p[..., :T].float(), v.float().repeat_interleave(nh // nh_kv, dim=1))
        assert torch.equal(out.float(), ref.half().float())


# ---------------------------------------------------------------------------------------------------------------------
# the whole decode step on the matrix-pipe layout (kivi_gqa_decode through KiviLayerCacheMF) vs the CPU restatement of
# models/llama_kivi.py:314-455 / models/mistral_kivi.py:381-445 (oracle/hook_ref.py, pinned against the reference classes)

def _cmp_cache(t_gpu, t_ref):
    names = ["K_code_T", "K_full", "K_scale_T", "K_mn_T", "V_code", "V_full", "V_scale", "V_mn"]
    for n, a, b in zip(names, t_gpu[:8], t_ref[:8]):
        if b is None:
            assert a is None or a.numel() == 0, n
            continue
        assert a is not None and tuple(a.shape) == tuple(b.shape), (n, None if a is None else a.shape, b.shape)
        assert same_bits(a, b), n
    assert t_gpu[8] == t_ref[8]


@pytest.mark.parametrize("form", ["split", "row"])
@pytest.mark.parametrize("nh,nh_kv,T0,R,masked,kind", [(4, 1, 5, 32, False, "randn"), (8, 2, 70, 32, True, "outlier"), (8, 1, 33, 32, False, "outlier"),
                                                         (16, 2, 600, 64, False, "randn"), (8, 2, 1100, 128, True, "outlier"),
                                                         (32, 8, 300, 128, False, "outlier"), (2, 2, 5, 32, False, "randn"),
                                                         (3, 3, 70, 32, True, "outlier"), (4, 4, 600, 64, False, "outlier"),
                                                         (2, 2, 1100, 128, True, "randn"), (16, 2, 1100, 32, True, "outlier")])
def test_mf_decode_steps_match_reference_logic(oracle, nh, nh_kv, T0, R, masked, kind, form):
    """Every step of R + 9 (one K flush, V flushes, the window ring wrapping, cache growth, a partial last super-block), stage by
    stage against the reference logic (oracle/hook_ref.py), no step-level escape -- for keys with large-magnitude channels
    (kind "outlier": what per-channel K quantisation exists for; scores reach |s| ~ 100) as well as plain normal ones, and for
    BOTH forms of the step: "split" (mf_k_kernel -> score rows in memory -> mf_v_kernel) and "row" (mf_row_kernel /
    mf_row4_kernel: the scores never leave the LDS; KIVI_GQA_DUMP_SCORES makes the test instantiations of those kernels also
    write the rows their softmax consumes):
      A. the fp16 row fed to the softmax (packed qK^T | residual scores, / sqrt(D), + mask; llama_kivi.py:324-372) within
         the north_star GEMV bar (1e-3 of max(|ref|, rms) + 1 ulp) of the reference's row;
      B. the output within 2e-3 of the reference's attend half (:375-399) run ON THE ROW THE GPU PRODUCED: the GEMV bar
         for the packed sV + one fp16 ulp (4.9e-4) of the probability of a dominant key (the row sum of exp is added in a
         different order) + the roundings of the two partial sums.
    A and B bound the end-to-end difference by the triangle inequality; what they leave out -- how a one-ulp difference in
    a dominant score moves its probability -- is the reference softmax itself, run on the CPU in stage B.
    After R + 9 steps the 9-tuple is bit-identical to the reference logic's."""
    mk = lambda seed, h, T: make_kv(seed, 2, h, T, 128, kind)        # noqa: E731
    mo = lambda seed, h, T: make_kv(seed, 2, h, T, 128)              # noqa: E731
    _stage_ab_steps(nh, nh_kv, T0, R, masked, form, R + 9, k_prompt=mk, k_step=mk, v_prompt=mo, v_step=mo, q_step=mo)


@pytest.mark.parametrize("nh,nh_kv,T0,R,masked,kind,S,bits", [
    (8, 2, 1100, 128, True, "outlier", 2, 2),        # two slices of one super-block each; a K flush adds a third super-block
    (8, 2, 1100, 96, False, "randn", 3, 2),          # R = 96: token Tv sits a super-block before the last -> an EMPTY middle slice, a last slice of two
    (16, 2, 2100, 32, True, "outlier", 4, 2),        # nh / nh_kv = 8, five super-blocks in four slices of two: an empty third slice
    (8, 2, 1100, 128, False, "outlier", 2, 4),       # 4-bit codes
    (4, 4, 1100, 32, True, "outlier", 2, 2),         # nh == nh_kv through the slice kernel (mf_row4_kernel<R = 1>): two slices
    (2, 2, 1100, 128, False, "randn", 1, 2),         # ... and unsliced (KIVI_GQA_SLICES(1): the A/B form against mf_row_kernel)
])
def test_mf_sliced_rows_match_reference_logic(oracle, nh, nh_kv, T0, R, masked, kind, S, bits):
    """The one-launch form with every row cut into S slices (mf_row4_kernel, S > 1: a block per slice, the slices of a unit exchange
    their softmax statistics inside the launch and meet in the workspace): stage A / B at every step exactly as for the other two
    forms, 9-tuples bit-identical -- incl. slices that are empty, a last slice that starts at the super-block of token Tv, masks
    (added per segment inside the K walk), V flushes by the last slice only."""
    mk = lambda seed, h, T: make_kv(seed, 2, h, T, 128, kind)        # noqa: E731
    mo = lambda seed, h, T: make_kv(seed, 2, h, T, 128)              # noqa: E731
    _stage_ab_steps(nh, nh_kv, T0, R, masked, f"slices{S}", min(R + 9, 72), k_prompt=mk, k_step=mk, v_prompt=mo, v_step=mo, q_step=mo, bits=bits)


def _stage_ab_steps(nh, nh_kv, T0, R, masked, form, steps, k_prompt, k_step, v_prompt, v_step, q_step, check_at=None, bits=2):
    """The stage A / B loop of test_mf_decode_steps_match_reference_logic over caller-made inputs (f(seed, heads, T))."""
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, KiviLayerCacheMF, kivi_attention_decode, make_layer_cache
    from oracle import hook_ref as H
    if form == "row" and nh // nh_kv == 8 and T0 + steps + 1 > 4608:
        pytest.skip("nh / nh_kv = 8: the eight score rows of a unit fit the LDS up to 4608 keys")
    B, D, g = 2, 128, 32
    cfg = KiviConfig(bits, bits, g, R)
    k0, v0 = k_prompt(1, nh_kv, T0), v_prompt(2, nh_kv, T0)
    layer = make_layer_cache(cfg, B, nh_kv, D, T0 + 8, "cuda", num_heads=nh)    # small capacity: the cache must grow
    assert isinstance(layer, KiviLayerCacheMF)
    if form.startswith("slices"):                                                # the one-launch form with the rows cut into S slices
        layer.flags = _lib.gqa_slices(int(form[6:])) | _lib.GQA_DUMP_SCORES
    else:
        layer.flags = _lib.GQA_FORCE_SPLIT if form == "split" else (_lib.GQA_FORCE_ROW | _lib.GQA_DUMP_SCORES)
    layer.prefill(k0.cuda(), v0.cuda())
    if form.startswith("slices"):
        plan = _lib.load().kivi_mf_launch_plan(B, nh, nh_kv, layer.k_quant_len, layer.k_res_len, R, layer.flags, bits, 0)
        assert plan == int(form[6:]), ("the shape must take the sliced form from the first step on", plan)
    past = H.prefill_cache(k0, v0, bits, bits, g, R)
    _cmp_cache(layer.as_tuple(), past)
    gen = torch.Generator().manual_seed(5)
    worst_a = worst_b = 0.0
    for s in range(steps):
        q = q_step(100 + s, nh, 1)
        kn, vn = k_step(200 + s, nh_kv, 1), v_step(300 + s, nh_kv, 1)
        mask = None
        n = T0 + s + 1
        if masked:
            mask = torch.zeros((B, 1, 1, n), dtype=torch.float16)
            mask[0, ..., : min(7, n - 1)] = torch.finfo(torch.float16).min          # left padding of batch row 0
            mask[1, ..., torch.randint(0, n - 1, (3,), generator=gen)] = -3.0
        if layer._native:
            layer._native[4][0].fill_(float("nan"))                                  # (a stale row must not pass for this step's)
        out = kivi_attention_decode(q.cuda(), kn.cuda(), vn.cuda(), layer, attention_mask=None if mask is None else mask.cuda())
        assert torch.isfinite(out).all(), s
        x_gpu = layer._native[4][0][:B, :nh, :, :n].cpu()                            # the row the softmax / the sV launch consumed
        ref, past_ref, x_ref = H.decode_step(q, kn, vn, past, bits, bits, g, R, attention_mask=mask, return_scores=True)
        assert torch.isfinite(x_ref.float()).all() and torch.isfinite(ref.float()).all(), "test inputs must keep the reference finite"
        live = x_ref.float() > -60000                                                # fully masked keys: both sides at the fp16 minimum
        ok, ra = gemv_close(torch.where(live, x_gpu.float(), 0.0), torch.where(live, x_ref.float(), 0.0), rtol=1e-3, ulps=1)
        assert ok, ("scores", s, ra)
        assert torch.equal(x_gpu[~live], x_ref[~live])
        ref_b, past_b = H.decode_step(q, kn, vn, past, bits, bits, g, R, attention_mask=mask, scores_override=x_gpu)
        ok, rb = gemv_close(out, ref_b, rtol=2e-3, ulps=1)
        assert ok, ("attend", s, rb)
        worst_a, worst_b = max(worst_a, ra), max(worst_b, rb)
        past = past_ref
        if s in (check_at or (0, R - 1, R, steps - 1)):
            _cmp_cache(layer.as_tuple(), past)
    print(f"worst ratio: scores {worst_a:.3f} of the 1e-3 bar, attend {worst_b:.3f} of the 2e-3 bar")
    return layer


# ---------------------------------------------------------------------------------------------------------------------
# dynamic range of the matrix-pipe operands: the fp16 A operands q * scale / p * scale must stay finite and
# accurate from subnormal scales to the largest finite ones, as the reference's fp32 scale * code + zero does
# (quant/csrc/gemv_cuda.cu:407-413).  Values are uniform in +-mag with a per-channel spread, so a unit mixes small and large
# scales; q (or nothing, for sV) is scaled so that the REFERENCE's fp16 result stays finite.

def _ranged(seed, B, h, T, mag, spread_dim):
    """Uniform in +-mag, times a factor per index of `spread_dim` (3: channel, 2: token) drawn from three decades below 1."""
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand((B, h, T, 128), generator=g) * 2 - 1) * mag
    n = x.shape[spread_dim]
    shape = [1, 1, 1, 1]
    shape[spread_dim] = n
    return (x * torch.logspace(0, -3, n)[torch.randperm(n, generator=g)].reshape(shape)).half()


MAGS = [1e-4, 1.0, 1e3, 3e4]


@pytest.mark.parametrize("mag", MAGS)
@pytest.mark.parametrize("B,nh,nh_kv,T", [(2, 2, 2, 1056), (2, 8, 2, 1056), (1, 8, 1, 1056), (32, 32, 32, 4096)])
def test_gqa_scores_dynamic_range(mods, oracle, B, nh, nh_kv, T, mag):
    """Matrix-pipe qK^T over keys of magnitude 1e-4 .. 3e4 (K scales from the fp16 subnormals to ~2e4; channels spread over three
    decades): finite, within the GEMV bar of the oracle on sampled units and of the VALU kernel (fp32 scale * code + zero)
    everywhere; the store's range flags are set exactly for the units that hold a scale >= 256."""
    mfma, new_pack, matmul = mods
    k = _ranged(3, B, nh_kv, T, mag, 3).cuda()                      # per-channel magnitudes: K groups run along the tokens
    q = (make_kv(4, B, nh, 1, 128) * min(1.0, 300.0 / mag)).half().cuda()
    store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda")
    mfma.kt_pack(k, store, 0)
    code, scale, mn = new_pack.quantize_and_pack_k_tmajor(k, 32, 2)
    big = (scale.float() >= 256).flatten(2).any(-1)
    assert torch.equal(mfma.range_big(store), big) and bool(big.any()) == (mag >= 1e3)
    assert torch.equal(mfma.range_small(store), (scale.float() < 2.0 ** -8).flatten(2).all(-1))
    out = torch.full((B, nh, 1, T + 8), 7.0, dtype=torch.float16, device="cuda")
    mfma.gqa_scores(q, store, T, out)
    assert torch.isfinite(out).all()
    ref_gpu = matmul.cuda_bmm_fA_qB_outer(32, q, code, scale, mn, 2)
    assert torch.isfinite(ref_gpu).all(), "test inputs must keep the reference finite"
    ok, ratio = gemv_close(out[..., :T], ref_gpu.cpu(), rtol=1.5e-3)
    assert ok, ratio
    rh = nh // nh_kv
    for (b, hk) in {(0, 0), (B - 1, nh_kv - 1)}:
        hs = slice(hk * rh, (hk + 1) * rh)
        ref = oracle.bmm_fA_qB_outer(32, q[b:b + 1, hs].cpu(), code[b:b + 1, hk:hk + 1].cpu(), scale[b:b + 1, hk:hk + 1].cpu(),
                                     mn[b:b + 1, hk:hk + 1].cpu(), 2)
        ok, ratio = gemv_close(out[b:b + 1, hs, :, :T], ref)
        assert ok, (b, hk, ratio)


@pytest.mark.parametrize("mag", MAGS)
@pytest.mark.parametrize("B,nh,nh_kv,T", [(2, 2, 2, 1000), (2, 8, 2, 1000), (1, 8, 1, 1000), (4, 32, 32, 4064)])
def test_gqa_output_dynamic_range(mods, oracle, B, nh, nh_kv, T, mag):
    """Matrix-pipe sV over values of magnitude 1e-4 .. 3e4 (token magnitudes spread over three decades): as above."""
    mfma, new_pack, matmul = mods
    v = _ranged(21, B, nh_kv, T, mag, 2).cuda()                     # per-token magnitudes: V groups run along the channels
    store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda")
    mfma.vt_pack(v, store)
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, 32, 2)
    big = (scale.float() >= 256).flatten(2).any(-1)
    assert torch.equal(mfma.range_big(store), big) and bool(big.any()) == (mag >= 1e3)
    assert torch.equal(mfma.range_small(store), (scale.float() < 2.0 ** -8).flatten(2).all(-1))
    pitch = (T + 7) // 8 * 8 + 8
    probs = torch.zeros((B, nh, 1, pitch), dtype=torch.float16, device="cuda")
    probs[..., :T] = _probs("softmax", B, nh, T, 5).cuda()
    out = mfma.gqa_output(probs, store, T)
    assert torch.isfinite(out).all()
    ref_gpu = matmul.cuda_bmm_fA_qB_outer(32, probs[..., :T], code, scale, mn, 2)
    assert torch.isfinite(ref_gpu).all(), "test inputs must keep the reference finite"
    ok, ratio = gemv_close(out, ref_gpu.cpu(), rtol=1.5e-3)
    assert ok, ratio
    rh = nh // nh_kv
    for (b, hk) in {(0, 0), (B - 1, nh_kv - 1)}:
        hs = slice(hk * rh, (hk + 1) * rh)
        ref = oracle.bmm_fA_qB_outer(32, probs[b:b + 1, hs, :, :T].cpu().contiguous(), code[b:b + 1, hk:hk + 1].cpu(),
                                     scale[b:b + 1, hk:hk + 1].cpu(), mn[b:b + 1, hk:hk + 1].cpu(), 2)
        ok, ratio = gemv_close(out[b:b + 1, hs], ref, rtol=1e-3)
        assert ok, (b, hk, ratio)


@pytest.mark.parametrize("form", ["split", "row"])
@pytest.mark.parametrize("m0,m1", [(1e-4, 1e-4), (1e-4, 1.0), (1.0, 1e3), (1e3, 1.0), (1e3, 1e3), (3e4, 3e4)])
@pytest.mark.parametrize("nh,nh_kv", [(2, 2), (8, 2), (8, 1)])
def test_mf_decode_steps_dynamic_range(oracle, nh, nh_kv, m0, m1, form):
    """The whole step, stage by stage as above, with a prompt of magnitude m0 and new tokens of magnitude m1 (K and V both):
    (1, 1e3) sets the range flags of both stores in the middle of the run -- the K flag by the K flush (kivi_kt_pack), the V
    flag by the V flush inside the step's own launch -- (1e3, 1) keeps them set while the new groups are small, (3e4, 3e4)
    runs scales ~2e4 through every path.  R + 3 steps: one K flush, V flushes from the first step on."""
    R, T0 = 32, 600
    qmag = min(1.0, 300.0 / max(m0, m1))
    layer = _stage_ab_steps(
        nh, nh_kv, T0, R, False, form, R + 3,
        k_prompt=lambda seed, h, T: _ranged(seed, 2, h, T, m0, 3), k_step=lambda seed, h, T: _ranged(seed, 2, h, T, m1, 3),
        v_prompt=lambda seed, h, T: _ranged(seed, 2, h, T, m0, 2), v_step=lambda seed, h, T: _ranged(seed, 2, h, T, m1, 3),
        q_step=lambda seed, h, T: (make_kv(seed, 2, h, T, 128) * qmag).half(), check_at=(0, 5, R - 1, R, R + 2))
    from kivi_amd.quant import mfma
    expect = max(m0, m1) >= 1e3
    assert bool(mfma.range_big(layer.kt).any()) == expect and bool(mfma.range_big(layer.vt).any()) == expect
    if max(m0, m1) <= 1e-4:
        assert bool(mfma.range_small(layer.kt).all()) and bool(mfma.range_small(layer.vt).all())


@pytest.mark.parametrize("B,nh,nh_kv,T0,R,masked", [(2, 4, 4, 5, 32, False), (8, 32, 32, 1500, 32, True), (2, 2, 2, 8100, 32, False),
                                                     (4, 8, 8, 4080, 128, False), (2, 8, 2, 5, 32, False), (3, 16, 4, 1500, 64, True),
                                                     (2, 8, 2, 9000, 128, False), (8, 32, 8, 8000, 128, False),
                                                     (2, 16, 2, 1500, 64, True), (2, 64, 8, 4000, 32, False)])
def test_mf_row_kernel_matches_two_launch_form(oracle, B, nh, nh_kv, T0, R, masked):
    """The one-launch row kernels (mf_row_kernel for nh == nh_kv, mf_row4_kernel for nh / nh_kv == 4 and -- rows up to 4608 keys --
    8: scores never leave the LDS) against the two-launch form of the same step (stage-checked above) on cloned caches: same packed qK^T arithmetic
    -> same scores; the softmax sum and the sV partial sums are added in a different order -> outputs within 1.5e-3 (GEMV
    bar + one fp16 ulp of a dominant probability); every cache write identical: 9-tuples bit-identical after a K flush and
    V flushes.  Also vs the reference logic end to end at the hook bar."""
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from oracle import hook_ref as H
    D, g = 128, 32
    ratio = nh // nh_kv
    cfg = KiviConfig(2, 2, g, R)
    k0, v0 = make_kv(1, B, nh_kv, T0, D), make_kv(2, B, nh_kv, T0, D)
    a = make_layer_cache(cfg, B, nh_kv, D, T0 + 2 * R + 8, "cuda", num_heads=nh)
    a.prefill(k0.cuda(), v0.cuda())
    b_ = a.clone()
    a.flags, b_.flags = _lib.GQA_FORCE_ROW, _lib.GQA_FORCE_SPLIT
    samples = sorted({(0, 0), (B - 1, nh_kv - 1)})
    pasts = {(b, h): H.prefill_cache(k0[b:b + 1, h:h + 1], v0[b:b + 1, h:h + 1], 2, 2, g, R) for b, h in samples}
    probe_lib = _lib.load()
    for s in range(min(R + 3, 40)):
        q, kn, vn = make_kv(100 + s, B, nh, 1, D).cuda(), make_kv(200 + s, B, nh_kv, 1, D).cuda(), make_kv(300 + s, B, nh_kv, 1, D).cuda()
        mask = None
        if masked:
            mask = torch.zeros((B, 1, 1, T0 + s + 1), dtype=torch.float16, device="cuda")
            mask[0, ..., :9] = torch.finfo(torch.float16).min
        e0, e1 = probe_lib.kivi_event_create(), probe_lib.kivi_event_create()
        probe_lib.kivi_set_launch_events(e0, e1)
        oa = kivi_attention_decode(q, kn, vn, a, attention_mask=mask)
        torch.cuda.synchronize()
        assert (b"mf_row_kernel" if ratio == 1 else b"mf_row4_kernel") in (probe_lib.kivi_last_timed_kernel() or b"")
        ob = kivi_attention_decode(q, kn, vn, b_, attention_mask=mask)
        ok, r = gemv_close(oa, ob, rtol=1.5e-3, ulps=1)
        assert ok, (s, r)
        for (bb, h) in samples:
            hs = slice(h * ratio, (h + 1) * ratio)
            ref, pasts[(bb, h)] = H.decode_step(q[bb:bb + 1, hs].cpu(), kn[bb:bb + 1, h:h + 1].cpu(), vn[bb:bb + 1, h:h + 1].cpu(),
                                                pasts[(bb, h)], 2, 2, g, R, attention_mask=None if mask is None else mask[bb:bb + 1].cpu())
            ok, r = gemv_close(oa[bb:bb + 1, hs], ref, rtol=3e-3, ulps=1)
            assert ok, (s, bb, h, r)
    ta, tb = a.as_tuple(), b_.as_tuple()
    for x, y in zip(ta[:8], tb[:8]):
        assert (x is None and y is None) or same_bits(x, y)
    for (bb, h) in samples:
        sl = tuple(None if x is None else x[bb:bb + 1, h:h + 1] for x in ta[:8]) + (ta[8],)
        _cmp_cache(sl, pasts[(bb, h)])


@pytest.mark.parametrize("B,nh,nh_kv,T0,R", [(64, 32, 8, 8192 - 128, 128), (2, 32, 8, 8192 - 128, 128), (2, 32, 8, 32768 - 128, 128),
                                             (1, 64, 8, 4096, 32)])
def test_mf_decode_full_size_rows_vs_oracle(oracle, B, nh, nh_kv, T0, R):
    """BASELINE configs 4 / 5 row lengths (T = 8k and 32k, residual 128, nh / nh_kv = 4) and a ratio-8 shape: sampled
    (batch row, kv head) slices of the inputs go through the CPU restatement (heads are independent)."""
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from oracle import hook_ref as H
    D, g = 128, 32
    cfg = KiviConfig(2, 2, g, R)
    gen = torch.Generator(device="cuda").manual_seed(11)
    k0 = torch.randn((B, nh_kv, T0, D), generator=gen, device="cuda", dtype=torch.float16)
    v0 = torch.randn((B, nh_kv, T0, D), generator=gen, device="cuda", dtype=torch.float16)
    layer = make_layer_cache(cfg, B, nh_kv, D, T0 + 64, "cuda", num_heads=nh)
    layer.prefill(k0, v0)
    ratio = nh // nh_kv
    samples = sorted({(0, 0), (B - 1, nh_kv - 1)})
    pasts = {(b, hk): H.prefill_cache(k0[b:b + 1, hk:hk + 1].cpu(), v0[b:b + 1, hk:hk + 1].cpu(), 2, 2, g, R) for b, hk in samples}
    del k0, v0
    for s in range(3):
        q = torch.randn((B, nh, 1, D), generator=gen, device="cuda", dtype=torch.float16)
        kn = torch.randn((B, nh_kv, 1, D), generator=gen, device="cuda", dtype=torch.float16)
        vn = torch.randn((B, nh_kv, 1, D), generator=gen, device="cuda", dtype=torch.float16)
        out = kivi_attention_decode(q, kn, vn, layer)
        for (b, hk) in samples:
            hs = slice(hk * ratio, (hk + 1) * ratio)
            ref, pasts[(b, hk)] = H.decode_step(q[b:b + 1, hs].cpu(), kn[b:b + 1, hk:hk + 1].cpu(), vn[b:b + 1, hk:hk + 1].cpu(),
                                                pasts[(b, hk)], 2, 2, g, R)
            ok, r_ = gemv_close(out[b:b + 1, hs], ref, rtol=3e-3, ulps=1)
            assert ok, (s, b, hk, r_)
    t = layer.as_tuple()
    for (b, hk) in samples:
        sl = tuple(None if x is None else x[b:b + 1, hk:hk + 1] for x in t[:8]) + (t[8],)
        _cmp_cache(sl, pasts[(b, hk)])


def test_mf_cache_from_tuple_and_clone(oracle):
    """A plain reference 9-tuple adopted into the layout (from_tuple) continues exactly like the cache that produced it;
    clone() is independent."""
    from kivi_amd.attention import KiviConfig, KiviLayerCacheMF, kivi_attention_decode, make_layer_cache
    B, nh, nh_kv, D, T0, R = 2, 8, 2, 128, 200, 32
    cfg = KiviConfig(2, 2, 32, R)
    k0, v0 = make_kv(1, B, nh_kv, T0, D).cuda(), make_kv(2, B, nh_kv, T0, D).cuda()
    a = make_layer_cache(cfg, B, nh_kv, D, T0 + 40, "cuda", num_heads=nh)
    a.prefill(k0, v0)
    b = KiviLayerCacheMF.from_tuple(cfg, tuple(a.as_tuple()), T0 + 40, nh)
    c = a.clone()
    for s in range(4):
        q, kn, vn = make_kv(100 + s, B, nh, 1, D).cuda(), make_kv(200 + s, B, nh_kv, 1, D).cuda(), make_kv(300 + s, B, nh_kv, 1, D).cuda()
        oa = kivi_attention_decode(q, kn, vn, a)
        ob = kivi_attention_decode(q, kn, vn, b)
        assert torch.equal(oa, ob)
    assert c.kv_seq_len == T0 and a.kv_seq_len == T0 + 4
    for x, y in zip(a.as_tuple()[:8], b.as_tuple()[:8]):
        assert (x is None and y is None) or same_bits(x, y)


@pytest.mark.parametrize("nh,nh_kv", [(4, 4), (8, 2)])
def test_decode_flushes_of_subnormal_groups_are_bit_exact(oracle, nh, nh_kv):
    """K / V values in the fp16-subnormal neighbourhood: groups whose range is 0 or ONE subnormal ulp have scale 0, and
    the reference then gives code 0 to the minimum (0 / 0 = NaN) and the maximum code to everything above it (d / 0 = inf).
    Every quantiser on the decode path (prefill packs, K flush, V leaving the window; matrix-pipe layout, one-launch row
    kernel for nh == nh_kv) must reproduce that: the cache tuples stay bit-identical through a flush."""
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from oracle import hook_ref as H
    B, D, g, R, T0 = 2, 128, 32, 32, 75
    cfg = KiviConfig(2, 2, g, R)

    def tiny_k(seed, T):      # one-ulp ranges along tokens (K groups)
        return make_kv(seed, B, nh_kv, T, D, "tiny")

    def tiny_v(seed, T):      # one-ulp ranges along channels (V groups)
        return make_kv(seed, B, nh_kv, D, T, "tiny").transpose(2, 3).contiguous()

    k0, v0 = tiny_k(1, T0), tiny_v(2, T0)
    layer = make_layer_cache(cfg, B, nh_kv, D, T0 + R + 16, "cuda", num_heads=nh)
    layer.prefill(k0.cuda(), v0.cuda())
    past = H.prefill_cache(k0, v0, 2, 2, g, R)
    _cmp_cache(layer.as_tuple(), past)
    kseq, vseq = tiny_k(3, R + 9), tiny_v(4, R + 9)
    for s in range(R + 9):
        q = make_kv(100 + s, B, nh, 1, D)
        kn, vn = kseq[:, :, s:s + 1].contiguous(), vseq[:, :, s:s + 1].contiguous()
        out = kivi_attention_decode(q.cuda(), kn.cuda(), vn.cuda(), layer)
        ref, past = H.decode_step(q, kn, vn, past, 2, 2, g, R)
        assert torch.isfinite(out).all()
        assert (out.cpu().float() - ref.float()).abs().max() <= 1e-6          # |V| < 4e-6
        if s % 8 == 0 or s >= R - 2:
            _cmp_cache(layer.as_tuple(), past)


@pytest.mark.parametrize("form", ["row", "split"])
@pytest.mark.parametrize("nh,nh_kv,bits", [(4, 4, 2), (8, 2, 2), (16, 2, 2), (8, 2, 4)])
def test_big_value_units_keep_their_small_probabilities(nh, nh_kv, bits, form):
    """A unit whose V store holds scales >= 256 (range word byte 0) AND a row whose probability mass sits on the newest token: the packed
    tokens' probabilities are ~2e-6 (fp16 subnormals), their values 2e4 times the new token's, so they still make half of the output.
    The sV product of such a unit needs its operand 2^7 lower and takes that where nothing is rounded (mf_sp / mf_ksh: here, a row sum of ~1,
    4 bits in the placement of p'' -- still p times a power of two -- and 3 in the scales); while it was p'' that went 2^10 lower (through
    round 6's first sessions) those probabilities were rounded to 0-2 subnormal ulps and this test's outputs were up to 37 % off (ratio 184
    of the bar on that library, profiles/r06_big_value_units.log; found by tools/fuzz_decode.py).  The attend half on the GPU's own rows
    against the fp64 reference at the stage-B bar (2e-3 + 1 ulp): 0.47-0.67 of it in the row and the two-launch form, R = 1, 4, 8, 2 and 4 bits."""
    import math

    import torch_ref64 as T64
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from kivi_amd.quant import mfma
    B, D, g, R, T0 = 3, 128, 32, 32, 200
    ratio = nh // nh_kv
    cfg = KiviConfig(bits, bits, g, R)
    gen = torch.Generator(device="cuda").manual_seed(91)
    k0 = (torch.randn((B, nh_kv, T0, D), device="cuda", generator=gen) * 0.05).half()
    v0 = (torch.randn((B, nh_kv, T0, D), device="cuda", generator=gen) * (200.0 if bits == 2 else 1000.0)).half()   # group scales ~270 (range / 3 | 15)
    layer = make_layer_cache(cfg, B, nh_kv, D, T0 + 8, "cuda", num_heads=nh)
    assert getattr(layer, "layout", "") == "mfma"
    layer.flags |= _lib.GQA_DUMP_SCORES | (_lib.GQA_FORCE_SPLIT if form == "split" else _lib.GQA_FORCE_ROW)
    layer.prefill(k0, v0)
    past = T64.prefill_cache(k0, v0, bits, bits, g, R)
    assert bool(mfma.range_big(layer.vt).all()) and not bool(mfma.range_big(layer.kt).any())
    worst = 0.0
    for s in range(3):
        base = torch.randn((B, nh_kv, 1, D), device="cuda", generator=gen)
        q = (base.repeat_interleave(ratio, dim=1) + 0.01 * torch.randn((B, nh, 1, D), device="cuda", generator=gen)).half()
        kn = (base * (13.0 * math.sqrt(D) / base.pow(2).sum(-1, keepdim=True))).half()      # the new key scores 13: p(packed token) ~ 2e-6
        vn = (torch.randn((B, nh_kv, 1, D), device="cuda", generator=gen) * 0.01).half()
        n = T0 + s + 1
        out = kivi_attention_decode(q, kn, vn, layer)
        x_gpu = layer._native[4][0][:B, :nh, :, :n].contiguous()
        ref_b, new_past, pre = T64.decode_step(q, kn, vn, past, bits, bits, g, R, scores_override=x_gpu)
        ok, ra = gemv_close(x_gpu, pre, rtol=1e-3, ulps=1)
        assert ok, ("scores", s, ra)
        p = torch.softmax(x_gpu.float(), -1)
        assert 1e-7 < p[..., : n - R - 1].max().item() < 1e-5 and p[..., -1].min().item() > 0.99
        ok, rb = gemv_close(out, ref_b, rtol=2e-3, ulps=1)
        worst = max(worst, rb)
        assert ok, ("attend half", s, rb)
        past = new_past
    print(f"worst attend-half ratio {worst:.3f} of 2e-3 (+1 ulp)")
