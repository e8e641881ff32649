"""GPU: the MFMA-friendly cache layout for grouped queries (kivi_amd/quant/mfma.py, kivi_amd/csrc/kivi_gqa.hip):
relayout kernels are bit-exact inverses and reproduce the hook-state tensors of the reference pack; the matrix-pipe
qK^T / sV agree with the oracle's restatement of gemv_cuda.cu:348-427 within the GEMV bar."""
import pytest
import torch

from helpers import gemv_close, make_kv, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    from kivi_amd.quant import matmul, mfma, new_pack
    return mfma, new_pack, matmul


@pytest.mark.parametrize("B,nh_kv,T,off", [(1, 1, 32, 0), (2, 3, 544, 0), (1, 2, 128, 480), (2, 2, 1024, 64)])
def test_kt_pack_equals_reference_pack(mods, oracle, B, nh_kv, T, off):
    """kivi_kt_pack (direct per-channel quantise into the layout) == the reference-layout pack (bit-exact vs the
    reference's new_pack.py through the golden fixtures) after kivi_kt_relayout, at a token offset too."""
    mfma, new_pack, _ = mods
    k = make_kv(7, B, nh_kv, off + T, 128, "outlier").cuda()
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
                                          (1, 32, 8, 4096)])
def test_gqa_scores_vs_oracle(mods, oracle, B, nh, nh_kv, T):
    """Matrix-pipe qK^T (ratio 4 and 8; partial super-blocks; both block shapes) against the oracle on sampled heads and
    against the VALU kernel of the paged layout everywhere."""
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


def test_gqa_scores_exact_arithmetic(mods):
    """Integer-valued inputs (quant/test.py:182-183 style): every product and sum is exact in fp32, so head mapping, the
    layout's bit positions and the hi / lo operand split must reproduce the dequantised matmul exactly."""
    mfma, new_pack, _ = mods
    B, nh, nh_kv, T = 2, 8, 2, 1056
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
