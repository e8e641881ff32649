"""FULL-COVERAGE parity at BASELINE.json's sizes: every (batch row, kv head) unit of the full-size launches, not 2-4 sampled ones.

tests/test_fullsize_gpu.py holds a few units of these launches to the CPU oracle (seconds per unit); the other ~99.7 % were covered
only by HIP-vs-HIP properties.  Here a second reference that shares no code with the HIP kernels or the oracle -- plain torch on the
GPU in fp64, tests/torch_ref64.py, itself checked against the pinned oracle on small inputs by tests/test_torch_ref64_cpu.py --
follows ALL units of the same inputs:

  * quantise + pack (quant/new_pack.py:30-48, :217-252) and the 9-tuple: bit-exact over every unit;
  * qK^T and sV (quant/matmul.py:178-219 -> quant/csrc/gemv_cuda.cu:348-427): the bare north_star bar, 1e-3 of max(|ref|, rms), every output;
  * decode steps of the hook (models/llama_kivi.py:314-399) through K and V flushes: stage A (rows the softmax consumes), stage B
    (attend half on those rows, 2e-3), end to end (3e-3), and the 9-tuple of every unit bit-identical at the end.
A unit-index-dependent bug (64-bit stride, ticket order, range-word index, slice exchange) cannot hide behind a sample here.
"""
import pytest
import torch

import torch_ref64 as T64
from helpers import gemv_close

pytestmark = pytest.mark.gpu

NAMES = ["K_code_T", "K_full", "K_scale_T", "K_mn_T", "V_code", "V_full", "V_scale", "V_mn"]


def eq_bits(a, b):
    if a is None or b is None:
        return (a is None or a.numel() == 0) and (b is None or b.numel() == 0)
    if a.shape != b.shape:
        return False
    a, b = a.contiguous(), b.contiguous()
    if a.dtype == torch.float16:
        a, b = a.view(torch.int16), b.view(torch.int16)
    return bool(torch.equal(a, b))


def assert_tuple(layer, past, what):
    t = layer.as_tuple()
    for n, a, r in zip(NAMES, t[:8], past[:8]):
        assert eq_bits(a, r), (what, n)
    assert t[8] == past[8], what


# (name, B, nh, nh_kv, T0, R, bits, steps, forced flags, expected kernel)
SHAPES = [
    ("C2", 32, 32, 32, 4080, 32, 2, 20, 0, "mf_row_kernel"),                 # bench.py's workload: K flush at step 16, V flush every step
    ("config4", 64, 32, 8, 8192 - 4, 128, 2, 6, 0, "mf_row4_kernel"),        # BASELINE configs[3]: K flush at step 4
    ("config4_4bit", 64, 32, 8, 8192 - 4, 128, 4, 6, 0, "mf_row4_kernel"),
    ("config5_slice", 16, 32, 8, 32768 + 125, 128, 2, 5, 0, "mf_row4_kernel"),   # BASELINE configs[4] per GPU: 4 slices per row, K flush at step 3
    ("config5_slice_split", 16, 32, 8, 32768 + 125, 128, 2, 4, 1, "mf_k_kernel"),
    ("longchat_32k", 8, 32, 32, 32768 + 100, 128, 2, 30, 0, "mf_k_kernel"),   # LongChat-7B-32K (docs/long_bench.md:5-26): multi-head rows > 8192 keys, K flush at step 28
    ("longchat_32k_sliced", 8, 32, 32, 32768 + 100, 128, 2, 3, 5 << 8, "mf_row4_kernel"),     # ... forced into ONE launch, 5 slices per row
    ("longchat_16k_4bit", 8, 32, 32, 16384 + 100, 128, 4, 4, 0, "mf_k_kernel"),               # KIVI-4, multi-head (round 6: 4 bits on the matrix pipe for nh == nh_kv)
    ("C2_4bit", 32, 32, 32, 4080, 32, 4, 20, 0, "mf_row_kernel"),
    ("gqa_6k_three_blocks", 96, 32, 8, 6016 - 4, 128, 2, 6, 0, "mf_row4_kernel"),             # 768 units of <= 6.3k keys: the three-blocks-per-CU instantiation (round 6), K flush at step 4
]


@pytest.mark.parametrize("name,B,nh,nh_kv,T0,R,bits,steps,flags,kernel", SHAPES, ids=[s[0] for s in SHAPES])
def test_decode_steps_every_unit(name, B, nh, nh_kv, T0, R, bits, steps, flags, kernel):
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from test_fullsize_gpu import LaunchProbe
    D, g = 128, 32
    cfg = KiviConfig(bits, bits, g, R)
    gen = torch.Generator(device="cuda").manual_seed(61)
    k0 = torch.randn((B, nh_kv, T0, D), device="cuda", dtype=torch.float16, generator=gen)
    v0 = torch.randn((B, nh_kv, T0, D), device="cuda", dtype=torch.float16, generator=gen)
    layer = make_layer_cache(cfg, B, nh_kv, D, T0 + steps + 1, "cuda", num_heads=nh)
    mf = getattr(layer, "layout", "") == "mfma"
    if mf:
        layer.flags |= _lib.GQA_DUMP_SCORES | flags
    layer.prefill(k0, v0)
    past = T64.prefill_cache(k0, v0, bits, bits, g, R)
    del k0, v0
    assert_tuple(layer, past, "after the prompt pass")
    probe = LaunchProbe()
    seen = set()
    worst = {"A": 0.0, "B": 0.0, "E": 0.0}
    for s in range(steps):
        q = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        kn = torch.randn((B, nh_kv, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        vn = torch.randn((B, nh_kv, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        n = T0 + s + 1
        mask = None
        if s % 2:                                                  # left padding on every other sequence, every other step
            mask = torch.zeros((B, 1, 1, n), dtype=torch.float16, device="cuda")
            for b in range(0, B, 2):
                mask[b, :, :, : 37 + 11 * b + s] = torch.finfo(torch.float16).min
        probe.arm()
        out = kivi_attention_decode(q, kn, vn, layer, attention_mask=mask)
        seen.add(probe.kernel().split("<")[0].strip("( "))
        ref, new_past, pre = T64.decode_step(q, kn, vn, past, bits, bits, g, R, attention_mask=mask)
        if mf:
            x_gpu = layer._native[4][0][:B, :nh, :, :n]                 # (the scratch rows are shared per stream and only ever grow)
            live = pre.float() > -60000
            ok, ra = gemv_close(torch.where(live, x_gpu.float(), 0.0), torch.where(live, pre.float(), 0.0), rtol=1e-3, ulps=1)
            assert ok, (name, "scores", s, ra)
            assert torch.equal(x_gpu[~live], pre[~live]), (name, "masked scores", s)
            ref_b, _, _ = T64.decode_step(q, kn, vn, past, bits, bits, g, R, attention_mask=mask, scores_override=x_gpu.contiguous())
            ok, rb = gemv_close(out, ref_b, rtol=2e-3, ulps=1)
            assert ok, (name, "attend half", s, rb)
            worst["A"], worst["B"] = max(worst["A"], ra), max(worst["B"], rb)
        # end to end: REPORTED against the hook bar (3e-3), asserted at twice that.  Stage A + B above are the rigorous bars; what they
        # leave out is the reference softmax's own sensitivity to a one-ulp difference of a score (an fp16 score of 2-4 has an ulp of
        # 2e-3: its probability moves by 0.2 %), which over every output of every unit -- millions of samples -- reaches 4e-3 of the
        # output's rms where the three sampled units of tests/test_fullsize_gpu.py stay under 3e-3.
        _, re_ = gemv_close(out, ref, rtol=3e-3)
        assert re_ <= 2.0, (name, "output", s, re_)
        worst["E"] = max(worst["E"], re_)
        past = new_past
    assert_tuple(layer, past, "after the last step")
    if kernel is not None:
        assert seen == {kernel}, seen
    print(f"{name}: kernels {sorted(seen)}; worst ratio: scores {worst['A']:.3f} of 1e-3 (+1 ulp), attend {worst['B']:.3f} of 2e-3, "
          f"output {worst['E']:.3f} of 3e-3 over {B * nh_kv} units x {steps} steps")


@pytest.mark.parametrize("B,nh,nh_kv,T,bits", [(32, 32, 32, 4096, 2), (64, 32, 8, 8192, 2), (32, 32, 32, 4096, 4), (8, 32, 32, 32768, 2)],
                         ids=["C2", "config4", "C2_4bit", "longchat_32k"])
def test_kgemv_every_unit(B, nh, nh_kv, T, bits):
    """BASELINE configs[1] (and the grouped-query / long-row shapes) through the reference's operator: pack bit-exact and every one of the
    B x nh x T scores within the bare north_star bar -- on the hook-state layout (gemv_k_kernel) and on the matrix-pipe layout."""
    from kivi_amd.quant import matmul, mfma, new_pack
    D, g = 128, 32
    gen = torch.Generator(device="cuda").manual_seed(62)
    k = torch.randn((B, nh_kv, T, D), device="cuda", dtype=torch.float16, generator=gen)
    q = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16, generator=gen)
    code, scale, mn = new_pack.quantize_and_pack_k_tmajor(k, g, bits)
    parts = [T64.quant_pack_lastdim(k[b0:b0 + 2].transpose(2, 3).contiguous(), g, bits) for b0 in range(0, B, 2)]
    rc, rs, rm = (torch.cat([p[i] for p in parts], 0) for i in range(3))
    assert eq_bits(code, rc) and eq_bits(scale, rs) and eq_bits(mn, rm), "per-channel K pack"
    ref = T64.scores64(q, rc, rs, rm, g, bits)
    out = matmul.cuda_bmm_fA_qB_outer(g, q, code, scale, mn, bits)
    ok, r1 = gemv_close(out, ref)
    assert ok, ("hook-state layout", r1)
    r2 = None
    if mfma.supported(bits, bits, g, D, 32, nh // nh_kv):
        store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda", bits)
        mfma.kt_pack(k, store, 0, g, bits)
        c2, s2, m2 = mfma.kt_to_ref(store, T, D, g, bits)
        assert eq_bits(c2, rc) and eq_bits(s2, rs) and eq_bits(m2, rm), "kivi_kt_pack"
        o2 = torch.empty((B, nh, 1, T), dtype=torch.float16, device="cuda")
        mfma.gqa_scores(q, store, T, o2, g, bits)
        ok, r2 = gemv_close(o2, ref)
        assert ok, ("matrix-pipe layout", r2)
    print(f"worst ratio of the bare 1e-3 bar over {B * nh * T} scores: hook-state layout {r1:.3f}, matrix-pipe layout {r2}")


@pytest.mark.parametrize("B,nh,nh_kv,Tv,bits", [(32, 32, 32, 4064, 2), (64, 32, 8, 8061, 2), (32, 32, 32, 4064, 4), (8, 32, 32, 32701, 2)],
                         ids=["C2", "config4", "C2_4bit", "longchat_32k"])
def test_svgemv_every_unit(B, nh, nh_kv, Tv, bits):
    """The sV product (llama_kivi.py:382: a non-contiguous slice of the probabilities) over every unit, both layouts."""
    from kivi_amd.quant import matmul, mfma, new_pack
    D, g, L = 128, 32, 33
    gen = torch.Generator(device="cuda").manual_seed(63)
    v = torch.randn((B, nh_kv, Tv, D), device="cuda", dtype=torch.float16, generator=gen)
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, g, bits)
    parts = [T64.quant_pack_lastdim(v[b0:b0 + 2], g, bits) for b0 in range(0, B, 2)]
    rc, rs, rm = (torch.cat([p[i] for p in parts], 0) for i in range(3))
    assert eq_bits(code, rc) and eq_bits(scale, rs) and eq_bits(mn, rm), "per-token V pack"
    w = torch.softmax(torch.randn((B, nh, 1, Tv + L), device="cuda", generator=gen) * 3, -1).half()
    a = w[..., :-L]
    ref = T64.output64(a, rc, rs, rm, g, bits)
    out = matmul.cuda_bmm_fA_qB_outer(g, a, code, scale, mn, bits)
    ok, r1 = gemv_close(out, ref)
    assert ok, ("hook-state layout", r1)
    r2 = None
    if mfma.supported(bits, bits, g, D, 32, nh // nh_kv):
        store = mfma.alloc_store(B, nh_kv, (Tv + 511) // 512, "cuda", bits)
        mfma.vt_pack(v, store, g, bits)
        c2, s2, m2 = mfma.vt_to_ref(store, Tv, D, g, bits)
        assert eq_bits(c2, rc) and eq_bits(s2, rs) and eq_bits(m2, rm), "kivi_vt_pack"
        pitch = (Tv + 7) // 8 * 8
        ap = torch.zeros((B, nh, 1, pitch), dtype=torch.float16, device="cuda")
        ap[..., :Tv] = a
        o2 = mfma.gqa_output(ap, store, Tv, None, g, bits)
        ok, r2 = gemv_close(o2, ref)
        assert ok, ("matrix-pipe layout", r2)
    print(f"worst ratio of the bare 1e-3 bar over {B * nh * D} outputs: hook-state layout {r1:.3f}, matrix-pipe layout {r2}")
