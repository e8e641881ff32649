"""GPU: 4-bit K / V on the matrix pipe (kivi_mfma_layout.h "KT4 / VT4": round 4, nh / nh_kv = 4 -- the reference's published
Mistral-7B + KIVI-4 shape, docs/long_bench.md:35-53; round 6, nh == nh_kv -- its multi-head KIVI-4 models, LongChat-7B-32K /
Llama-2-7B, docs/long_bench.md:5-26): the packers and relayouts are bit-exact against the reference-layout
4-bit pack (itself bit-exact vs the reference through the golden fixtures), qK^T and sV agree with the oracle's restatement of
gemv_cuda.cu:265-427 within the north_star GEMV bar, the decode step stage by stage in both forms.  The reference-class hook
fixtures of this shape (tests/golden/hook_*_b4_*.npz) are replayed by tests/test_hook_gpu.py on every layout."""
import pytest
import torch

from helpers import gemv_close, make_kv, same_bits
from test_mfma_gpu import MAGS, _probs, _ranged, _stage_ab_steps

pytestmark = pytest.mark.gpu
BITS = 4


@pytest.fixture(scope="module")
def mods():
    from kivi_amd.quant import matmul, mfma, new_pack
    return mfma, new_pack, matmul


@pytest.mark.parametrize("B,nh_kv,T,off,kind", [(1, 1, 32, 0, "outlier"), (2, 3, 544, 0, "outlier"), (1, 2, 128, 480, "randn"),
                                                  (2, 2, 1024, 64, "outlier"), (1, 2, 96, 32, "tiny")])
def test_kt4_pack_equals_reference_pack(mods, oracle, B, nh_kv, T, off, kind):
    mfma, new_pack, _ = mods
    k = make_kv(7, B, nh_kv, off + T, 128, kind).cuda()
    store = mfma.alloc_store(B, nh_kv, (off + T + 511) // 512, "cuda", BITS)
    if off:
        mfma.kt_pack(k[:, :, :off], store, 0, 32, BITS)
    mfma.kt_pack(k[:, :, off:], store, off, 32, BITS)
    code, scale, mn = mfma.kt_to_ref(store, off + T, bits=BITS)
    rc, rs, rm = new_pack.quantize_and_pack_k_tmajor(k, 32, BITS)
    assert same_bits(code, rc) and same_bits(scale, rs) and same_bits(mn, rm)
    oc, os_, om = oracle.quantize_and_pack_along_last_dim(k.cpu().transpose(2, 3).contiguous(), 32, BITS)
    assert same_bits(code, oc) and same_bits(scale, os_) and same_bits(mn, om)
    store2 = mfma.alloc_store(B, nh_kv, store.shape[2], "cuda", BITS)
    mfma.kt_from_ref(store2, code, scale, mn, 32, BITS)
    assert torch.equal(store2, store)


@pytest.mark.parametrize("B,nh_kv,T,kind", [(1, 1, 1, "randn"), (2, 2, 33, "outlier"), (1, 3, 512, "randn"), (2, 1, 1000, "outlier"),
                                             (1, 2, 75, "tiny_rows"), (3, 8, 2049, "randn")])
def test_vt4_pack_equals_reference_pack(mods, oracle, B, nh_kv, T, kind):
    mfma, new_pack, _ = mods
    if kind == "tiny_rows":
        v = make_kv(9, B, nh_kv, 128, T, "tiny").transpose(2, 3).contiguous().cuda()
    else:
        v = make_kv(9, B, nh_kv, T, 128, kind).cuda()
    nsb = (T + 511) // 512
    a, b_ = mfma.alloc_store(B, nh_kv, nsb, "cuda", BITS), mfma.alloc_store(B, nh_kv, nsb, "cuda", BITS)
    b_.fill_(-1)                                                # the relayout must zero the unwritten slots of the last block
    mfma.vt_pack(v, a, 32, BITS)
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, 32, BITS)
    mfma.vt_from_ref(b_, code, scale, mn, 32, BITS)
    nb = (T + 31) // 32
    # whole blocks in use compare word for word; blocks never written stay as they were
    assert torch.equal(_blocks(a, nb), _blocks(b_, nb))
    oc, os_, om = oracle.quantize_and_pack_along_last_dim(v.cpu(), 32, BITS)
    c2, s2, m2 = mfma.vt_to_ref(a, T, bits=BITS)
    assert same_bits(c2, oc) and same_bits(s2, os_) and same_bits(m2, om)
    if T % 32:
        cz, sz, mz = mfma.vt_to_ref(b_, nb * 32, bits=BITS)
        assert not cz[:, :, T:].any() and not sz[:, :, T:].view(torch.int16).any() and not mz[:, :, T:].view(torch.int16).any()


def _blocks(store, nb):
    """The words of the first nb 32-token blocks of every unit: codes, scale, zero points (4-bit super-block: 16 x 512 | 1024 | 1024)."""
    out = []
    for blk in range(nb):
        sb, j = blk // 16, blk % 16
        out += [store[:, :, sb, j * 512:(j + 1) * 512], store[:, :, sb, 8192 + j * 64: 8192 + (j + 1) * 64],
                store[:, :, sb, 9216 + j * 64: 9216 + (j + 1) * 64]]
    return torch.cat(out, dim=-1)


@pytest.mark.parametrize("B,nh,nh_kv,T", [(1, 4, 1, 32), (2, 8, 2, 544), (1, 4, 1, 1024), (3, 8, 2, 96), (16, 32, 8, 8192), (1, 32, 8, 4096),
                                          (1, 1, 1, 32), (2, 2, 2, 544), (3, 3, 3, 96), (8, 32, 32, 4096), (1, 4, 4, 32768)])
def test_gqa4_scores_vs_oracle(mods, oracle, B, nh, nh_kv, T):
    mfma, new_pack, matmul = mods
    k = make_kv(3, B, nh_kv, T, 128, "outlier").cuda()
    q = (make_kv(4, B, nh, 1, 128) * 1.5).half().cuda()
    store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda", BITS)
    mfma.kt_pack(k, store, 0, 32, BITS)
    out = torch.full((B, nh, 1, T + 8), 7.0, dtype=torch.float16, device="cuda")
    mfma.gqa_scores(q, store, T, out, 32, BITS)
    assert bool((out[..., T:] == 7.0).all()) and torch.isfinite(out).all()
    code, scale, mn = new_pack.quantize_and_pack_k_tmajor(k, 32, BITS)
    ref_gpu = matmul.cuda_bmm_fA_qB_outer(32, q, code, scale, mn, BITS)
    ok, ratio = gemv_close(out[..., :T], ref_gpu.cpu(), rtol=1.5e-3)
    assert ok, ratio
    for (b, hk) in {(0, 0), (B - 1, nh_kv - 1)}:
        hs = slice(hk * (nh // nh_kv), (hk + 1) * (nh // nh_kv))
        ref = oracle.bmm_fA_qB_outer(32, q[b:b + 1, hs].cpu(), code[b:b + 1, hk:hk + 1].cpu(), scale[b:b + 1, hk:hk + 1].cpu(),
                                     mn[b:b + 1, hk:hk + 1].cpu(), BITS)
        ok, ratio = gemv_close(out[b:b + 1, hs, :, :T], ref)
        assert ok, (b, hk, ratio)


@pytest.mark.parametrize("ratio", [4, 1])
def test_gqa4_exact_arithmetic(mods, ratio):
    """Integer-valued K / V (codes = values 0..15) and small-integer q / power-of-two probabilities: every product and sum is
    exact, so the field positions, the four views, the head mapping, the hi / lo split and the centring (-7.5) must reproduce
    the dequantised matmuls exactly."""
    mfma, _, _ = mods
    B, nh_kv, T = 2, 2, 1056
    nh = nh_kv * ratio
    g = torch.Generator().manual_seed(0)
    k = torch.randint(0, 16, (B, nh_kv, T, 128), generator=g).half().cuda()
    k[:, :, ::32] = 0
    k[:, :, 1::32] = 15                                                         # every group: min 0, max 15 -> scale 1
    q = torch.randint(-5, 6, (B, nh, 1, 128), generator=g).half().cuda()
    store = mfma.alloc_store(B, nh_kv, 3, "cuda", BITS)
    mfma.kt_pack(k, store, 0, 32, BITS)
    out = torch.empty((B, nh, 1, T), dtype=torch.float16, device="cuda")
    mfma.gqa_scores(q, store, T, out, 32, BITS)
    ref = torch.matmul(q.float(), k.float().repeat_interleave(ratio, dim=1).transpose(2, 3))
    assert torch.equal(out.float(), ref.half().float())
    T = 700
    v = torch.randint(0, 16, (B, nh_kv, T, 128), generator=g).half()
    v[..., ::32] = 0
    v[..., 1::32] = 15
    v = v.cuda()
    p = (torch.randint(0, 3, (B, nh, 1, 704), generator=g).float() * 2.0 ** -10).half().cuda()
    vst = mfma.alloc_store(B, nh_kv, 2, "cuda", BITS)
    mfma.vt_pack(v, vst, 32, BITS)
    o = mfma.gqa_output(p, vst, T, None, 32, BITS)
    ref = torch.matmul(p[..., :T].float(), v.float().repeat_interleave(ratio, dim=1))
    assert torch.equal(o.float(), ref.half().float())


@pytest.mark.parametrize("kind", ["softmax", "uniform", "sparse"])
@pytest.mark.parametrize("B,nh,nh_kv,T", [(2, 4, 1, 33), (1, 8, 2, 544), (2, 32, 8, 8192), (1, 32, 8, 32768),
                                          (2, 1, 1, 33), (1, 2, 2, 544), (4, 32, 32, 4064), (1, 4, 4, 32768)])
def test_gqa4_output_vs_oracle(mods, oracle, B, nh, nh_kv, T, kind):
    mfma, new_pack, matmul = mods
    v = make_kv(21, B, nh_kv, T, 128, "outlier" if kind == "softmax" else "randn").cuda()
    store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda", BITS)
    mfma.vt_pack(v, store, 32, BITS)
    pitch = (T + 7) // 8 * 8 + 8
    probs = torch.zeros((B, nh, 1, pitch), dtype=torch.float16, device="cuda")
    probs[..., :T] = _probs(kind, B, nh, T, 5).cuda()
    probs[..., T:] = 1.0
    out = mfma.gqa_output(probs, store, T, None, 32, BITS)
    assert torch.isfinite(out).all()
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, 32, BITS)
    ref_gpu = matmul.cuda_bmm_fA_qB_outer(32, probs[..., :T], code, scale, mn, BITS)
    ok, ratio = gemv_close(out, ref_gpu.cpu(), rtol=1.5e-3)
    assert ok, ratio
    for (b, hk) in {(0, 0), (B - 1, nh_kv - 1)}:
        hs = slice(hk * (nh // nh_kv), (hk + 1) * (nh // nh_kv))
        ref = oracle.bmm_fA_qB_outer(32, probs[b:b + 1, hs, :, :T].cpu().contiguous(), code[b:b + 1, hk:hk + 1].cpu(),
                                     scale[b:b + 1, hk:hk + 1].cpu(), mn[b:b + 1, hk:hk + 1].cpu(), BITS)
        ok, ratio = gemv_close(out[b:b + 1, hs], ref, rtol=1e-3)
        assert ok, (b, hk, ratio)


@pytest.mark.parametrize("ratio", [4, 1])
@pytest.mark.parametrize("mag", MAGS)
def test_gqa4_dynamic_range(mods, oracle, mag, ratio):
    """As tests/test_mfma_gpu.py::test_gqa_scores_dynamic_range / _output_: scales from the fp16 subnormals to ~4e3 (a 4-bit scale is
    a fifteenth of the range); the range words carry the mark for a scale >= 256 exactly for the units that hold one, and the mark
    for a scale >= 2^-8 likewise.  Every case is held to the same bars as the 2-bit twins: the GEMV bar (1e-3) against the ORACLE
    (gemv_cuda.cu:265-345 restated: fp32 scale * code + zero) on sampled units, 1.5e-3 against the VALU kernel everywhere.
    mag 1e-4 with the token magnitudes spread three decades below puts V values at 1e-7, the V scales at 1-100 fp16-subnormal ulps
    (2^-24) and the OUTPUTS into the fp16 subnormals (~100 ulps): round 4 left p'' * scale a subnormal fp16 there (ratio 1.9, bar
    widened to 3); since round 5 a unit whose scales are all < 2^-8 places p'' 2^8 higher (mf_range_shift) and the case meets the
    plain bar."""
    mfma, new_pack, matmul = mods
    B, nh_kv, T = 2, 2, 1056
    nh = nh_kv * ratio
    k = _ranged(3, B, nh_kv, T, mag, 3).cuda()
    q = (make_kv(4, B, nh, 1, 128) * min(1.0, 300.0 / mag)).half().cuda()
    kst = mfma.alloc_store(B, nh_kv, 3, "cuda", BITS)
    mfma.kt_pack(k, kst, 0, 32, BITS)
    code, scale, mn = new_pack.quantize_and_pack_k_tmajor(k, 32, BITS)
    assert torch.equal(mfma.range_big(kst), (scale.float() >= 256).flatten(2).any(-1))
    assert torch.equal(mfma.range_small(kst), (scale.float() < 2.0 ** -8).flatten(2).all(-1))
    out = torch.full((B, nh, 1, T + 8), 7.0, dtype=torch.float16, device="cuda")
    mfma.gqa_scores(q, kst, T, out, 32, BITS)
    ref = matmul.cuda_bmm_fA_qB_outer(32, q, code, scale, mn, BITS)
    assert torch.isfinite(out).all() and torch.isfinite(ref).all()
    ok, ratio = gemv_close(out[..., :T], ref.cpu(), rtol=1.5e-3)
    assert ok, ("scores", ratio)
    for (b, hk) in {(0, 0), (B - 1, nh_kv - 1)}:
        hs = slice(hk * (nh // nh_kv), (hk + 1) * (nh // nh_kv))
        oref = oracle.bmm_fA_qB_outer(32, q[b:b + 1, hs].cpu(), code[b:b + 1, hk:hk + 1].cpu(), scale[b:b + 1, hk:hk + 1].cpu(),
                                      mn[b:b + 1, hk:hk + 1].cpu(), BITS)
        ok, ratio = gemv_close(out[b:b + 1, hs, :, :T], oref)
        assert ok, ("scores vs oracle", b, hk, ratio)
    T = 1000
    v = _ranged(21, B, nh_kv, T, mag, 2).cuda()
    vst = mfma.alloc_store(B, nh_kv, 2, "cuda", BITS)
    mfma.vt_pack(v, vst, 32, BITS)
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, 32, BITS)
    assert torch.equal(mfma.range_big(vst), (scale.float() >= 256).flatten(2).any(-1))
    assert torch.equal(mfma.range_small(vst), (scale.float() < 2.0 ** -8).flatten(2).all(-1))
    probs = torch.zeros((B, nh, 1, 1008), dtype=torch.float16, device="cuda")
    probs[..., :T] = _probs("softmax", B, nh, T, 5).cuda()
    o = mfma.gqa_output(probs, vst, T, None, 32, BITS)
    ref = matmul.cuda_bmm_fA_qB_outer(32, probs[..., :T], code, scale, mn, BITS)
    assert torch.isfinite(o).all() and torch.isfinite(ref).all()
    ok, ratio = gemv_close(o, ref.cpu(), rtol=1.5e-3)
    assert ok, ("output", ratio)
    for (b, hk) in {(0, 0), (B - 1, nh_kv - 1)}:
        hs = slice(hk * (nh // nh_kv), (hk + 1) * (nh // nh_kv))
        oref = oracle.bmm_fA_qB_outer(32, probs[b:b + 1, hs, :, :T].cpu().contiguous(), code[b:b + 1, hk:hk + 1].cpu(),
                                      scale[b:b + 1, hk:hk + 1].cpu(), mn[b:b + 1, hk:hk + 1].cpu(), BITS)
        ok, ratio = gemv_close(o[b:b + 1, hs], oref, rtol=1e-3)
        assert ok, ("output vs oracle", b, hk, ratio)


@pytest.mark.parametrize("form", ["split", "row"])
@pytest.mark.parametrize("nh,nh_kv,T0,R,masked,kind", [(4, 1, 5, 32, False, "randn"), (8, 2, 70, 32, True, "outlier"),
                                                         (16, 4, 600, 64, False, "randn"),
                                                         (2, 2, 5, 32, False, "randn"), (3, 3, 70, 32, True, "outlier"), (4, 4, 600, 64, False, "outlier"),
                                                         (2, 2, 1100, 128, True, "randn")])
def test_mf4_decode_steps_match_reference_logic(oracle, nh, nh_kv, T0, R, masked, kind, form):
    """tests/test_mfma_gpu.py::test_mf_decode_steps_match_reference_logic at 4 bits: R + 9 steps (a K flush through kt_pack4,
    V flushes into the 4-bit words, the window ring wrapping, cache growth), stage A (the softmax's input row, 1e-3) and stage B
    (the attend half on the GPU's row, 2e-3) in both forms, 9-tuples bit-identical to the reference logic's.  (R = 128 at 4 bits:
    the reference-class fixture hook_flash_gqa4_b4_r128_mask and tests/test_fullsize_gpu.py, across a K flush of 128 tokens.)"""
    mk = lambda seed, h, T: make_kv(seed, 2, h, T, 128, kind)        # noqa: E731
    mo = lambda seed, h, T: make_kv(seed, 2, h, T, 128)              # noqa: E731
    _stage_ab_steps(nh, nh_kv, T0, R, masked, form, R + 9, k_prompt=mk, k_step=mk, v_prompt=mo, v_step=mo, q_step=mo, bits=BITS)


@pytest.mark.parametrize("nh,nh_kv,T0,R,masked,kind,S", [(4, 4, 1100, 32, True, "outlier", 2), (2, 2, 1100, 128, False, "randn", 1),
                                                           (2, 2, 2100, 32, True, "outlier", 4)])
def test_mf4_sliced_multi_head_rows_match_reference_logic(oracle, nh, nh_kv, T0, R, masked, kind, S):
    """4-bit multi-head rows through the slice kernel (mf_row4_kernel<R = 1, BITS = 4>: what serves KIVI-4 rows beyond 8192 keys in one
    launch): stage A / B at every step, 9-tuples bit-identical (cf. tests/test_mfma_gpu.py::test_mf_sliced_rows_match_reference_logic)."""
    mk = lambda seed, h, T: make_kv(seed, 2, h, T, 128, kind)        # noqa: E731
    mo = lambda seed, h, T: make_kv(seed, 2, h, T, 128)              # noqa: E731
    _stage_ab_steps(nh, nh_kv, T0, R, masked, f"slices{S}", min(R + 9, 72), k_prompt=mk, k_step=mk, v_prompt=mo, v_step=mo, q_step=mo, bits=BITS)


@pytest.mark.parametrize("form", ["split", "row"])
@pytest.mark.parametrize("nh,nh_kv", [(8, 2), (2, 2)])
@pytest.mark.parametrize("m0,m1", [(1e-4, 1e-4), (1e-4, 1.0), (1.0, 3e3), (3e4, 3e4)])
def test_mf4_decode_steps_dynamic_range(oracle, m0, m1, form, nh, nh_kv):
    R, T0 = 32, 600
    qmag = min(1.0, 300.0 / max(m0, m1))
    layer = _stage_ab_steps(
        nh, nh_kv, T0, R, False, form, R + 3,
        k_prompt=lambda seed, h, T: _ranged(seed, 2, h, T, m0, 3), k_step=lambda seed, h, T: _ranged(seed, 2, h, T, m1, 3),
        v_prompt=lambda seed, h, T: _ranged(seed, 2, h, T, m0, 2), v_step=lambda seed, h, T: _ranged(seed, 2, h, T, m1, 3),
        q_step=lambda seed, h, T: (make_kv(seed, 2, h, T, 128) * qmag).half(), check_at=(0, 5, R - 1, R, R + 2), bits=BITS)
    from kivi_amd.quant import mfma
    expect = max(m0, m1) >= 3e3                          # (a 4-bit scale is a fifteenth of the range)
    assert bool(mfma.range_big(layer.kt).any()) == expect and bool(mfma.range_big(layer.vt).any()) == expect
    # (1e-4, 1e-4): every unit stays "small" (q'' / p'' 2^8 higher); (1e-4, 1): the K flush / the V flushes of the new tokens end that mid-run
    assert bool(mfma.range_small(layer.kt).all()) == (max(m0, m1) <= 1e-4) and bool(mfma.range_small(layer.vt).all()) == (max(m0, m1) <= 1e-4)
