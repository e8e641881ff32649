"""CPU: the oracle (oracle/kivi_oracle.c) against the fixtures minted from the REAL reference
(oracle/pin_reference.py, run where /root/reference exists).  Nothing here touches the GPU."""
import numpy as np
import pytest
import torch

from helpers import f16, load_golden, same_bits


def test_half_conversion_exhaustive(oracle):
    L = oracle.lib()
    allh = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16)
    f = allh.float()
    # h2f over every half
    ours = np.array([L.kivi_oracle_h2f(int(i)) for i in range(0, 65536, 1)], dtype=np.float32)
    ref = f.numpy()
    fin = np.isfinite(ref)
    assert (ours[fin] == ref[fin]).all() and (np.isnan(ours) == np.isnan(ref)).all()
    # f2h against torch's RNE conversion on a float sweep that includes every rounding boundary
    g = torch.Generator().manual_seed(1)
    xs = torch.cat([f[torch.isfinite(f)], f[torch.isfinite(f)] * (1 + 2 ** -12), f[torch.isfinite(f)] * (1 - 2 ** -12),
                    torch.randn(20000, generator=g) * 70000, torch.randn(20000, generator=g) * 1e-6,
                    torch.tensor([65504.0, 65519.9, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 0.0, -0.0])])
    want = xs.half().view(torch.int16).numpy().view(np.uint16)
    got = np.array([L.kivi_oracle_f2h(float(v)) for v in xs.numpy()], dtype=np.uint16)
    assert (got == want).all()


@pytest.mark.parametrize("name,fx", sorted(load_golden("lastdim_").items()))
def test_lastdim_pack_matches_reference(oracle, name, fx):
    x = f16(fx["x"])
    g, bits = int(fx["g"]), int(fx["bits"])
    # the fixtures hold the reference's CPU outputs: NaN->int is INT_MIN there
    code, scale, mn = oracle.quantize_and_pack_along_last_dim(x, g, bits, nan_mode=oracle.NAN_CPU)
    assert same_bits(code, torch.from_numpy(fx["code"]))
    assert same_bits(scale, f16(fx["scale"])) and same_bits(mn, f16(fx["mn"]))
    deq = oracle.unpack_and_dequant_vcache(torch.from_numpy(fx["code"]), f16(fx["scale"]), f16(fx["mn"]), g, bits)
    assert same_bits(deq, f16(fx["deq"]))
    if not int(fx["has_nan"]):
        # without constant groups the CUDA flavour (NaN -> 0) is identical
        code2, _, _ = oracle.quantize_and_pack_along_last_dim(x, g, bits, nan_mode=oracle.NAN_CUDA)
        assert same_bits(code2, code)


def test_constant_group_modes(oracle):
    """scale == 0 -> 0/0: reference on x86 packs INT_MIN (only element 0 of a word survives the shift),
    the reference's CUDA conversion gives 0; the oracle restates both."""
    x = torch.full((1, 1, 1, 64), 1.5, dtype=torch.float16)
    x[..., 32:] = torch.arange(32).half()
    c_cpu, s, m = oracle.quantize_and_pack_along_last_dim(x, 32, 2, nan_mode=oracle.NAN_CPU)
    c_gpu, s2, m2 = oracle.quantize_and_pack_along_last_dim(x, 32, 2, nan_mode=oracle.NAN_CUDA)
    assert c_cpu[0, 0, 0, 0].item() == -2147483648 and c_cpu[0, 0, 0, 1].item() == -2147483648
    assert c_gpu[0, 0, 0, 0].item() == 0 and c_gpu[0, 0, 0, 1].item() == 0
    assert same_bits(c_cpu[..., 2:], c_gpu[..., 2:]) and same_bits(s, s2) and same_bits(m, m2)
    assert s[0, 0, 0, 0].item() == 0.0 and m[0, 0, 0, 0].item() == 1.5


@pytest.mark.parametrize("name,fx", sorted(load_golden("kcache_").items()))
def test_kcache_pack_matches_reference(oracle, name, fx):
    k = f16(fx["k"])
    g, bits = int(fx["g"]), int(fx["bits"])
    code, scale, mn = oracle.quant_and_pack_kcache(k, g, bits, nan_mode=oracle.NAN_CPU)
    assert same_bits(code, torch.from_numpy(fx["code"]))
    assert same_bits(scale, f16(fx["scale"])) and same_bits(mn, f16(fx["mn"]))
    deq = oracle.unpack_and_dequant_kcache(code, scale, mn, g, bits)
    assert same_bits(deq, f16(fx["deq"]))
    # hook identity (llama_kivi.py:345): last-dim packing of K^T == transpose of the T-major codes
    c2, s2, m2 = oracle.quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), g, bits, nan_mode=oracle.NAN_CPU)
    assert same_bits(c2, code.transpose(2, 3).contiguous())
    assert same_bits(s2, scale.squeeze(3).transpose(2, 3).contiguous())
    assert same_bits(m2, mn.squeeze(3).transpose(2, 3).contiguous())


@pytest.mark.parametrize("name,fx", sorted(load_golden("packtensor_").items()))
def test_pack_tensor_matches_reference(oracle, name, fx):
    data = torch.from_numpy(fx["data"])
    bits, pack_dim = int(fx["bits"]), int(fx["pack_dim"])
    code = oracle.pack_tensor(data, bits, pack_dim)
    assert same_bits(code, torch.from_numpy(fx["code"]))
    assert bool((oracle.unpack_tensor(code, bits, pack_dim).int() == data).all())


@pytest.mark.parametrize("name,fx", sorted(load_golden("gemvexact_").items()))
def test_fused_gemv_indexing_matches_reference(oracle, name, fx):
    """Exact-arithmetic cases: the reference's unpack_and_dequant_vcache + fp64 matmul is exact, so the
    fused oracle (gemv_cuda.cu lane order) must agree bit for bit in both layouts and both fma modes."""
    fA, qB = f16(fx["fA"]), torch.from_numpy(fx["qB"])
    scales, zeros = f16(fx["scales"]), f16(fx["zeros"])
    g, bits = int(fx["g"]), int(fx["bits"])
    want = f16(fx["out"])
    assert same_bits(oracle.bmm_fA_qB_outer(g, fA, qB, scales, zeros, bits), want)
    assert same_bits(oracle.bmm_fA_qB_outer(g, fA, qB, scales, zeros, bits, use_fma=False), want)
    assert same_bits(oracle.bmm_fA_qB_outer(g, fA, qB, scales, zeros, bits, fakequant=True), want)
    B, nh, _, K = fA.shape
    nh_kv = qB.shape[1]
    fpi = 32 // bits
    N = qB.shape[-1] * fpi
    w_t = qB.reshape(-1, K, N // fpi).transpose(1, 2).contiguous()      # matmul.py:205
    s_t = scales.reshape(-1, K, N // g).transpose(1, 2).contiguous()    # matmul.py:213
    z_t = zeros.reshape(-1, K, N // g).transpose(1, 2).contiguous()     # matmul.py:214
    got = oracle.gemv_forward_outer_dim(fA.reshape(B * nh, 1, K), w_t, s_t, z_t, bits, g, nh, nh_kv)
    assert same_bits(got.view(B, nh, 1, N), want)


def test_cfg1_fakequant_path(oracle):
    """BASELINE config 1 (B=1,H=1,T=128,D=128,g=32, 2-bit): the reference's fake-quant qK procedure
    (quant/test.py:187-195) reproduced by the oracle's fake-quant GEMV; fused arithmetic stays close."""
    fx = load_golden("cfg1_fakequant_qk")["cfg1_fakequant_qk"]
    q = f16(fx["q"])
    code_T, scale_T, mn_T = torch.from_numpy(fx["code_T"]), f16(fx["scale_T"]), f16(fx["mn_T"])
    want = f16(fx["out_fakequant"]).float()
    rms = want.pow(2).mean().sqrt()
    fq = oracle.bmm_fA_qB_outer(32, q, code_T, scale_T, mn_T, 2, fakequant=True).float()
    assert (fq - want).abs().max() <= 2e-3 * rms
    fused = oracle.bmm_fA_qB_outer(32, q, code_T, scale_T, mn_T, 2).float()
    assert (fused - want).abs().max() <= 1e-2 * rms
    # and the packed inputs themselves come out of the oracle's pack
    k = f16(fx["k"])
    c2, s2, m2 = oracle.quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), 32, 2)
    assert same_bits(c2, code_T) and same_bits(s2, scale_T) and same_bits(m2, mn_T)


def test_oracle_rejects_bad_arguments(oracle):
    x = torch.zeros((1, 1, 2, 48), dtype=torch.float16)
    with pytest.raises(ValueError):
        oracle.quantize_and_pack_along_last_dim(x, 32, 2)        # T % group_size != 0 (new_pack.py:222)
    with pytest.raises(ValueError):
        oracle.quantize_and_pack_along_last_dim(torch.zeros((1, 1, 2, 64), dtype=torch.float16), 32, 3)
    fA = torch.zeros((1, 3, 1, 32), dtype=torch.float16)
    qB = torch.zeros((1, 2, 32, 2), dtype=torch.int32)
    sz = torch.zeros((1, 2, 32, 1), dtype=torch.float16)
    with pytest.raises(ValueError):
        oracle.bmm_fA_qB_outer(32, fA, qB, sz, sz, 2)             # nh % nh_kv != 0 (matmul.py:216)


def test_threshold_quantiser_equals_division():
    """The 2-bit kernels count thresholds instead of dividing (kivi_quant.h): code = [d > t0*s] + [d >= t1*s] + [d > t2*s].
    For EVERY positive fp16 scale and every d within 3 ulps of a decision boundary (plus random pairs) this equals the
    reference's rint(clamp(fp16(d / s))) with the division done in fp32 and rounded to fp16 (new_pack.py:240-241)."""
    allh = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16)
    s_all = allh[1:]
    taus = np.array([0.5 + 2 ** -12, 1.5 - 2 ** -11, 2.5 + 2 ** -10], dtype=np.float32)
    strict = [True, False, True]

    def ref_code(d, s):
        with np.errstate(over="ignore"):
            q = (d.astype(np.float32) / s.astype(np.float32)).astype(np.float16).astype(np.float32)
        return np.rint(np.clip(q, 0, 3)).astype(np.int32)

    def thr_code(d, s):
        df, sf = d.astype(np.float32), s.astype(np.float32)
        # scale inf -> code 0; scale 0 -> thresholds under the smallest fp16: 0/0 -> 0, d/0 = inf -> 3 (kivi_quant.h)
        sf = np.where((sf > 0) & np.isfinite(sf), sf, np.where(sf == 0, np.float32(2.0 ** -30), np.float32(np.nan)))
        c = np.zeros(d.shape, np.int32)
        for t, st in zip(taus, strict):
            with np.errstate(invalid="ignore"):
                th = t * sf
            ok = np.isfinite(sf)
            assert (th[ok].astype(np.float64) == np.float64(t) * sf[ok].astype(np.float64)).all()   # the product is exact
            c += (df > th) if st else (df >= th)
        return c

    for k in range(3):
        with np.errstate(over="ignore"):
            bits = (taus[k] * s_all.astype(np.float32)).astype(np.float16).view(np.uint16).astype(np.int32)
        for off in range(-3, 4):
            d = np.clip(bits + off, 0, 0x7BFF).astype(np.uint16).view(np.float16)
            assert (ref_code(d, s_all) == thr_code(d, s_all)).all(), (k, off)
    rng = np.random.default_rng(0)
    d = allh[rng.integers(0, allh.size, 2_000_000)]
    s = s_all[rng.integers(0, s_all.size, 2_000_000)]
    assert (ref_code(d, s) == thr_code(d, s)).all()
    # degenerate scales: 0/0 and x/inf, inf/inf -> NaN or 0 -> code 0 (CUDA float->int of NaN)
    with np.errstate(invalid="ignore", divide="ignore"):
        for sv, dv in ((0.0, 0.0), (np.inf, 1.0), (np.inf, np.inf), (np.inf, 0.0)):
            d1, s1 = np.array([dv], np.float16), np.array([sv], np.float16)
            q = (d1.astype(np.float32) / s1.astype(np.float32)).astype(np.float16).astype(np.float32)
            ref = 0 if np.isnan(q[0]) else int(np.rint(np.clip(q, 0, 3))[0])
            assert thr_code(d1, s1)[0] == ref == 0
        # scale 0 with d > 0: a group whose range is one subnormal ulp (2^-24 / 3 rounds to 0) -> d / 0 = inf -> code 3
        for dv in (2.0 ** -24, 2.0 ** -23, 1.0):
            d1, s1 = np.array([dv], np.float16), np.array([0.0], np.float16)
            q = (d1.astype(np.float32) / s1.astype(np.float32)).astype(np.float16).astype(np.float32)
            assert thr_code(d1, s1)[0] == int(np.rint(np.clip(q, 0, 3))[0]) == 3


def test_packed_pack_kernel_shortcuts_are_exact():
    """quant_pack_lastdim2_kernel (kivi_pack.hip, make_group2) takes two shortcuts; both hold for EVERY fp16 input:
    (1) scale = fp16(range / 3) computed as fp16(range * fp32(1/3));
    (2) tau_k * scale is never an fp16 value, so the three decisions  d > th0, d >= th1, d > th2  on a non-negative fp16 d
        all equal  bits(d) > bits(RTZ_fp16(th))  (v_cvt_pkrtz_f16_f32)."""
    b = np.arange(0, 0x7C01, dtype=np.uint16)                      # +0 .. +inf
    r = b.view(np.float16).astype(np.float32)
    with np.errstate(over="ignore"):
        div = (r / np.float32(3.0)).astype(np.float16).view(np.uint16)
        mul = (r * np.float32(0.3333333432674408)).astype(np.float16).view(np.uint16)
        mul_once = (r.astype(np.float64) * np.float64(np.float32(0.3333333432674408))).astype(np.float16).view(np.uint16)
    assert (div == mul).all()
    assert (div == mul_once).all()      # the compiler may fuse the product and the rounding (v_fma_mixlo_f16: one rounding)
    s = b[1:0x7C00].view(np.float16).astype(np.float32)            # every positive finite scale
    dbits = np.arange(0, 0x7C01, dtype=np.int32)                   # every d >= +0 (incl. +inf)
    dval = dbits.astype(np.uint16).view(np.float16).astype(np.float32)
    for tau, strict in ((0.500244140625, True), (1.49951171875, False), (2.5009765625, True)):
        th = np.float32(tau) * s
        assert (th.astype(np.float64) == np.float64(tau) * s.astype(np.float64)).all()       # exact product
        with np.errstate(over="ignore"):
            h = th.astype(np.float16)
        hb = h.view(np.uint16).astype(np.int32)
        rtz = np.where(h.astype(np.float32) > th, hb - 1, hb)      # round toward zero (never inf for a finite th)
        assert (rtz < 0x7C00).all()
        assert (rtz.astype(np.uint16).view(np.float16).astype(np.float32) < th).all()        # th is never representable
        # the decision itself, on a sample of scales against every d
        for i in range(0, s.size, 257):
            ref = (dval > th[i]) if strict else (dval >= th[i])
            assert ((dbits > rtz[i]) == ref).all(), (tau, i)


def test_reciprocal_quantiser_codes_are_exact():
    """quant_pack_lastdimN_kernel (kivi_pack.hip) quantises through ONE reciprocal per group: code = low bits of
    fp16(min(max(fp16(d * fp32(1 / s)), 0), maxq) + 1024) instead of rint(clamp(fp16(d / s))) (new_pack.py:240-241).
    Exhaustive over every fp16 d >= +0 (incl. +inf) and every positive finite fp16 scale -- 10^9 pairs -- for maxq in
    {3, 15, 255}, with the product rounded twice (fp32, then fp16: v_mul_f32 + v_cvt) and once (v_fma_mixlo_f16): the
    quotient differs by an ulp in a few thousand pairs, the CODE never.  Also: scale = fp16(range * fp32(1 / maxq)) equals
    fp16(range / maxq) for every fp16 range, both roundings."""
    allb = np.arange(0, 0x7C01, dtype=np.uint16)
    d32 = allb.view(np.float16).astype(np.float32)[None, :]
    d64 = d32.astype(np.float64)
    sb = np.arange(1, 0x7C00, dtype=np.uint16)
    s_all = sb.view(np.float16).astype(np.float32)

    def magic_code(q16, mq):
        q = q16.astype(np.float32)
        q = np.clip(np.where(np.isnan(q), np.float32(0), q), 0, mq)
        return ((q + np.float32(1024.0)).astype(np.float16).view(np.uint16) & 0x3FF).astype(np.int32)   # fp16 add = exact sum, one RNE

    differ = [0, 0]
    with np.errstate(over="ignore", invalid="ignore"):
        for i in range(0, s_all.size, 128):
            s = s_all[i:i + 128, None]
            qr = (d32 / s).astype(np.float16)
            r = np.float32(1.0) / s
            q_twice = (d32 * r).astype(np.float16)
            q_once = (d64 * r.astype(np.float64)).astype(np.float16)        # 11 x 24-bit product is exact in a double
            for j, q in enumerate((q_twice, q_once)):
                m = q.view(np.uint16) != qr.view(np.uint16)
                differ[j] += int(m.sum())
                if m.any():
                    a = np.nan_to_num(qr[m].astype(np.float32), nan=0.0)
                    for mq in (3, 15, 255):
                        assert (np.rint(np.clip(a, 0, mq)).astype(np.int32) == magic_code(q[m], mq)).all(), (hex(sb[i]), j, mq)
            if i % 4096 == 0:   # where the quotients agree the magic-number rint must agree with rint too: a sample of scales
                a = np.nan_to_num(qr.astype(np.float32), nan=0.0)
                for mq in (3, 15, 255):
                    assert (np.rint(np.clip(a, 0, mq)).astype(np.int32) == magic_code(qr, mq)).all(), (hex(sb[i]), mq)
    assert differ[0] > 0 and differ[1] > 0          # the shortcut is NOT the division; only the codes agree
    # the magic-number rint on every clamped fp16 value (ties to even at .5)
    q_all = allb.view(np.float16)
    for mq in (3, 15, 255):
        a = q_all.astype(np.float32)
        assert (np.rint(np.clip(a, 0, mq)).astype(np.int32) == magic_code(q_all, mq)).all()
    # degenerate scales, as the kernel meets them: r = inf (scale 0), r = 0 (scale inf)
    with np.errstate(over="ignore", invalid="ignore"):
        for r_, dv, want in ((np.inf, 0.0, 0), (np.inf, 6e-8, 15), (0.0, 1.0, 0), (0.0, np.inf, 0)):
            q = (np.array([dv], np.float16).astype(np.float32) * np.float32(r_)).astype(np.float16)
            assert magic_code(q, 15)[0] == want
        rng = allb.view(np.float16).astype(np.float32)
        for mq in (15, 255):
            ref = (rng / np.float32(mq)).astype(np.float16).view(np.uint16)
            rq = np.float32(1.0) / np.float32(mq)
            assert ((rng * rq).astype(np.float16).view(np.uint16) == ref).all()
            assert ((rng.astype(np.float64) * np.float64(rq)).astype(np.float16).view(np.uint16) == ref).all()
