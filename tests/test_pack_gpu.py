"""GPU parity: fused quantise+pack / unpack+dequant HIP kernels (through the C ABI) vs the CPU oracle and
the fixtures minted from the real reference.  Bar: bit-exact (code, scale, mn, dequantised fp16)."""
import pytest
import torch

from helpers import f16, load_golden, make_kv, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def np_mod():
    from kivi_amd.quant import new_pack
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return new_pack


@pytest.mark.parametrize("name,fx", sorted(load_golden("lastdim_").items()))
def test_lastdim_golden(np_mod, oracle, name, fx):
    x = f16(fx["x"])
    g, bits = int(fx["g"]), int(fx["bits"])
    code, scale, mn = np_mod.triton_quantize_and_pack_along_last_dim(x.cuda(), g, bits)
    assert same_bits(scale, f16(fx["scale"])) and same_bits(mn, f16(fx["mn"]))
    if int(fx["has_nan"]):
        # constant groups: reference-on-CPU packs INT_MIN, the CUDA path (and we) pack code 0
        want, _, _ = oracle.quantize_and_pack_along_last_dim(x, g, bits, nan_mode=oracle.NAN_CUDA)
    else:
        want = torch.from_numpy(fx["code"])
    assert same_bits(code, want)
    deq = np_mod.unpack_and_dequant_vcache(torch.from_numpy(fx["code"]).cuda(), f16(fx["scale"]).cuda().unsqueeze(-1),
                                           f16(fx["mn"]).cuda().unsqueeze(-1), g, bits)
    assert same_bits(deq, f16(fx["deq"]))
    # reference-named wrappers
    c2, s2, m2 = np_mod.quant_and_pack_vcache(x.cuda(), g, bits)
    assert same_bits(c2, want) and s2.shape == scale.shape + (1,) and same_bits(s2.squeeze(-1), scale)


@pytest.mark.parametrize("name,fx", sorted(load_golden("kcache_").items()))
def test_kcache_golden(np_mod, name, fx):
    k = f16(fx["k"])
    g, bits = int(fx["g"]), int(fx["bits"])
    code, scale, mn = np_mod.quant_and_pack_kcache(k.cuda(), g, bits)
    assert same_bits(code, torch.from_numpy(fx["code"]))
    assert same_bits(scale, f16(fx["scale"])) and same_bits(mn, f16(fx["mn"]))
    assert code.is_contiguous() and scale.shape == (k.shape[0], k.shape[1], k.shape[2] // g, 1, k.shape[3])
    deq = np_mod.unpack_and_dequant_kcache(code, scale, mn, g, bits)
    assert same_bits(deq, f16(fx["deq"]))
    # the hook's call: pack K^T along the last dim, with and without the reference's .contiguous() copy
    ct, st, mt = np_mod.triton_quantize_and_pack_along_last_dim(k.cuda().transpose(2, 3).contiguous(), g, bits)
    cv, sv, mv = np_mod.triton_quantize_and_pack_along_last_dim(k.cuda().transpose(2, 3), g, bits)
    want_c = torch.from_numpy(fx["code"]).transpose(2, 3)
    assert same_bits(ct, want_c) and same_bits(cv, want_c)
    assert same_bits(st, sv) and same_bits(mt, mv)
    assert same_bits(st, f16(fx["scale"]).squeeze(3).transpose(2, 3))


@pytest.mark.parametrize("name,fx", sorted(load_golden("packtensor_").items()))
def test_pack_unpack_tensor_golden(np_mod, name, fx):
    data = torch.from_numpy(fx["data"]).cuda()
    bits, pack_dim = int(fx["bits"]), int(fx["pack_dim"])
    code = np_mod.pack_tensor(data, bits, pack_dim)
    assert same_bits(code, torch.from_numpy(fx["code"]))
    back = np_mod.unpack_tensor(code, bits, pack_dim)
    assert back.dtype == torch.int16 and bool((back.int() == data).all())


CASES = [
    # B, nh, rows, T, g, bits, kind
    (2, 3, 7, 128, 32, 2, "randn"),
    (1, 2, 5, 256, 64, 2, "outlier"),
    (1, 2, 3, 256, 128, 4, "randn"),
    (2, 1, 9, 64, 32, 4, "int"),
    (1, 1, 4, 128, 32, 8, "randn"),
    (1, 2, 33, 96, 32, 2, "randn"),       # T not a power of two
    (1, 1, 3, 48, 16, 2, "randn"),        # group 16 -> generic kernel
    (1, 1, 2, 96, 48, 4, "randn"),        # group not a power of two -> generic kernel
    (3, 2, 1, 128, 32, 2, "randn"),       # the per-step V shape (B, nh, 1, D)
    (1, 2, 8, 128, 32, 2, "tiny"),        # subnormal neighbourhood: one-ulp ranges (scale 0, d / 0 = inf -> max code)
    (1, 2, 8, 128, 32, 4, "tiny"),
]


@pytest.mark.parametrize("B,nh,rows,T,g,bits,kind", CASES)
def test_lastdim_vs_oracle(np_mod, oracle, B, nh, rows, T, g, bits, kind):
    x = make_kv(11, B, nh, rows, T, kind)
    code, scale, mn = np_mod.triton_quantize_and_pack_along_last_dim(x.cuda(), g, bits)
    oc, os_, om = oracle.quantize_and_pack_along_last_dim(x, g, bits)
    assert same_bits(code, oc) and same_bits(scale, os_) and same_bits(mn, om)
    deq = np_mod.unpack_and_dequant_vcache(code, scale.unsqueeze(-1), mn.unsqueeze(-1), g, bits)
    assert same_bits(deq, oracle.unpack_and_dequant_vcache(oc, os_, om, g, bits))


def test_lastdim_hard_values(np_mod, oracle):
    """Rounding traps: wide dynamic range, subnormals, +-0, ties at x.5, constant and all-zero groups, +-65504."""
    g = torch.Generator().manual_seed(5)
    x = (torch.randn((1, 4, 64, 128), generator=g) * torch.exp(4 * torch.randn((1, 4, 64, 128), generator=g))).half()
    x[0, 0, 0, :32] = 0.0
    x[0, 0, 1, :32] = 3.25
    x[0, 0, 2, :32] = torch.tensor([0.0, 1.0, 2.0, 3.0] * 8)                 # scale exactly 1: codes are ties-free
    x[0, 0, 3, :32] = torch.tensor([0.0, 0.5, 1.5, 2.5, 3.0, 1.0, 2.0, 0.25] * 4)  # exact .5 quotients (ties to even)
    x[0, 0, 4, :32] = torch.tensor([6e-8, 1.2e-7, 0.0, 5.9e-8] * 8)           # fp16 subnormals
    x[0, 0, 5, :32] = torch.tensor([65504.0, -65504.0] * 16)                  # range overflows to inf in fp16
    x[0, 0, 6, :32] = torch.tensor([-0.0, 0.0] * 16)
    for bits in (2, 4, 8):
        code, scale, mn = np_mod.triton_quantize_and_pack_along_last_dim(x.cuda(), 32, bits)
        oc, os_, om = oracle.quantize_and_pack_along_last_dim(x, 32, bits)
        assert same_bits(scale, os_) and same_bits(mn, om)
        assert same_bits(code, oc)


@pytest.mark.parametrize("g,bits", [(32, 2), (64, 2), (128, 2), (32, 4), (64, 4), (128, 4), (32, 8), (16, 4)])
def test_lastdim_every_exponent(np_mod, oracle, g, bits):
    """The packed-math kernels (2 bits: quant_pack_lastdim2_kernel, thresholds rounded toward zero to fp16, scale through a
    multiply; 4 / 8 bits: quant_pack_lastdimN_kernel, one reciprocal per group + magic-number rint) on groups whose range
    sits at every fp16 exponent, subnormal scales included, values on a few-ulp grid so that many d fall on or next to a
    decision boundary; full blocks (1024 chunks per block) and a ragged tail."""
    gen = torch.Generator().manual_seed(17)
    for rows in (4096 * 32 // g * 2, 37):
        ngrp = rows * 128 // g
        e = torch.randint(-24, 16, (ngrp, 1), generator=gen).float()
        base = torch.randint(0, 2048, (ngrp, 1), generator=gen).float()
        x = ((torch.randint(-24, 25, (ngrp, g), generator=gen).float() + base) * torch.exp2(e - 5)).half()
        x = x.reshape(1, 1, rows, 128)
        code, scale, mn = np_mod.triton_quantize_and_pack_along_last_dim(x.cuda(), g, bits)
        oc, os_, om = oracle.quantize_and_pack_along_last_dim(x, g, bits)
        for name, a, b in (("scale", scale, os_), ("mn", mn, om), ("code", code, oc)):
            a, b = a.cpu().contiguous().view(torch.int16 if a.dtype == torch.float16 else torch.int32), b.contiguous().view(
                torch.int16 if b.dtype == torch.float16 else torch.int32)
            bad = (a.reshape(ngrp, -1) != b.reshape(ngrp, -1)).any(dim=1).nonzero().flatten()
            if bad.numel():
                gi = int(bad[0])
                raise AssertionError(f"{name}: {bad.numel()} of {ngrp} groups differ; group {gi}: x bits "
                                     f"{[hex(int(v) & 0xFFFF) for v in x.reshape(ngrp, g)[gi].view(torch.int16)]} "
                                     f"kernel {a.reshape(ngrp, -1)[gi].tolist()} oracle {b.reshape(ngrp, -1)[gi].tolist()}")


@pytest.mark.parametrize("bits", [2, 4])
def test_nan_inputs_follow_torch_min_max(np_mod, bits):
    """torch.min / max propagate NaN (new_pack.py:236-237): a group that contains a NaN -- of either sign -- gets scale = mn =
    NaN and (the reference's CUDA float->int of NaN) codes 0; every other group is untouched.  The packed-math kernels get
    this from v_pk_minimum3_f16 / v_pk_maximum3_f16; checked for the last-dim pack and the per-channel K pack."""
    fpi = 32 // bits
    x = make_kv(23, 1, 2, 8, 128, "randn")
    c0, s0, m0 = [t.cpu() for t in np_mod.triton_quantize_and_pack_along_last_dim(x.cuda(), 32, bits)]
    y = x.clone()
    y[0, 0, 0, 40] = float("nan")                       # row 0, group 1
    y[0, 1, 3, 70] = -float("nan")                      # row 3 of head 1, group 2
    c1, s1, m1 = [t.cpu() for t in np_mod.triton_quantize_and_pack_along_last_dim(y.cuda(), 32, bits)]
    hit = torch.zeros_like(s0, dtype=torch.bool)
    hit[0, 0, 0, 1] = hit[0, 1, 3, 2] = True
    assert torch.isnan(s1[hit]).all() and torch.isnan(m1[hit]).all()
    assert same_bits(s1[~hit], s0[~hit]) and same_bits(m1[~hit], m0[~hit])
    wpg = 32 // fpi                                      # code words per group
    cw0, cw1 = c0.reshape(1, 2, 8, 4, wpg), c1.reshape(1, 2, 8, 4, wpg)
    assert (cw1[hit] == 0).all() and torch.equal(cw1[~hit], cw0[~hit])
    # per-channel K (groups along the token axis; the tiled kernel needs >= 8 groups)
    k = make_kv(24, 1, 1, 256, 128, "randn")
    kc0, ks0, km0 = [t.cpu() for t in np_mod.quantize_and_pack_k_tmajor(k.cuda(), 32, bits)]
    k2 = k.clone()
    k2[0, 0, 70, 5] = float("nan")                      # channel 5, token group 2
    k2[0, 0, 255, 127] = -float("nan")                  # channel 127, token group 7
    kc1, ks1, km1 = [t.cpu() for t in np_mod.quantize_and_pack_k_tmajor(k2.cuda(), 32, bits)]
    khit = torch.zeros_like(ks0, dtype=torch.bool)
    khit[0, 0, 5, 2] = khit[0, 0, 127, 7] = True
    assert torch.isnan(ks1[khit]).all() and torch.isnan(km1[khit]).all()
    assert same_bits(ks1[~khit], ks0[~khit]) and same_bits(km1[~khit], km0[~khit])
    kw0, kw1 = kc0.reshape(1, 1, 128, 8, wpg), kc1.reshape(1, 1, 128, 8, wpg)
    assert (kw1[khit] == 0).all() and torch.equal(kw1[~khit], kw0[~khit])


@pytest.mark.parametrize("B,nh,T,D,g,bits,kind", [
    (2, 2, 64, 128, 32, 2, "randn"), (1, 3, 128, 64, 64, 2, "outlier"), (1, 2, 256, 128, 128, 4, "randn"),
    (1, 2, 64, 128, 32, 2, "tiny"), (1, 2, 64, 128, 32, 4, "tiny"),
    (1, 1, 32, 80, 32, 2, "randn"), (1, 2, 96, 128, 32, 4, "int"), (1, 1, 64, 128, 32, 8, "randn"),
    (1, 1, 48, 33, 16, 2, "randn"),
])
def test_k_tmajor_vs_oracle(np_mod, oracle, B, nh, T, D, g, bits, kind):
    k = make_kv(21, B, nh, T, D, kind)
    code_T, scale_T, mn_T = np_mod.quantize_and_pack_k_tmajor(k.cuda(), g, bits)
    oc, os_, om = oracle.quant_and_pack_kcache(k, g, bits)
    assert same_bits(code_T, oc.transpose(2, 3)) and same_bits(scale_T, os_.squeeze(3).transpose(2, 3))
    assert same_bits(mn_T, om.squeeze(3).transpose(2, 3))
    # strided input: K as the hook holds it, a (B, T, nh, D) projection output viewed as (B, nh, T, D)
    kv = k.transpose(1, 2).contiguous().cuda().transpose(1, 2)
    assert not kv.is_contiguous() or nh == 1
    c2, s2, m2 = np_mod.quantize_and_pack_k_tmajor(kv, g, bits)
    assert same_bits(c2, code_T) and same_bits(s2, scale_T) and same_bits(m2, mn_T)


def test_k_tmajor_in_place_append(np_mod, oracle):
    """Appending R tokens into capacity-strided cache buffers == packing the whole prefix at once."""
    B, nh, D, g, bits, cap = 2, 2, 128, 32, 2, 256
    k = make_kv(31, B, nh, 192, D)
    code = torch.zeros((B, nh, D, cap // 16), dtype=torch.int32, device="cuda")
    scale = torch.zeros((B, nh, D, cap // g), dtype=torch.float16, device="cuda")
    mn = torch.zeros_like(scale)
    for t0 in range(0, 192, 64):
        np_mod.quantize_and_pack_k_tmajor(k[:, :, t0:t0 + 64].cuda(), g, bits, out=(code, scale, mn), token_offset=t0)
    oc, os_, om = oracle.quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), g, bits)
    assert same_bits(code[..., :192 // 16], oc) and same_bits(scale[..., :192 // g], os_) and same_bits(mn[..., :6], om)
    assert bool((code[..., 12:] == 0).all())


def test_full_size_properties(np_mod):
    """BASELINE config 2 sized K slab (B=32 would be 1 GiB; 4 batches = 128 MiB keeps the box light):
    size-independent properties instead of the (slow) oracle."""
    B, nh, T, D, g, bits = 4, 32, 4096, 128, 32, 2
    torch.manual_seed(0)
    k = torch.randn((B, nh, T, D), device="cuda", dtype=torch.float16)
    code_T, scale_T, mn_T = np_mod.quantize_and_pack_k_tmajor(k, g, bits)
    # 1. same answer through the reference's transpose-copy route
    c2, s2, m2 = np_mod.triton_quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), g, bits)
    assert same_bits(code_T, c2) and same_bits(scale_T, s2) and same_bits(mn_T, m2)
    # 2. round trip: |x - dequant| <= scale/2 (+ fp16 rounding slack), min and max reproduce exactly
    deq = np_mod.unpack_and_dequant_vcache(code_T, scale_T.unsqueeze(-1), mn_T.unsqueeze(-1), g, bits)  # (B,nh,D,T)
    kt = k.transpose(2, 3)
    err = (deq.float() - kt.float()).abs().view(B, nh, D, T // g, g)
    bound = scale_T.float().unsqueeze(-1) * 0.5 * (1 + 2 ** -8) + 2 ** -10 * kt.float().abs().view(B, nh, D, T // g, g) + 1e-3
    assert bool((err <= bound).all())
    # 3. idempotence: re-quantising the dequantised tensor reproduces the same codes wherever the
    #    group's min/max survived (always, since code 0 and code 3 dequantise to mn and ~mx)
    c3, s3, m3 = np_mod.triton_quantize_and_pack_along_last_dim(deq, g, bits)
    same = (c3 == code_T).float().mean().item()
    assert same > 0.97, same
    # 4. codes use the full range and scale/mn are finite
    assert torch.isfinite(scale_T).all() and torch.isfinite(mn_T).all()
    hist = torch.bincount((np_mod.unpack_tensor(code_T[:1, :2], bits, 3).flatten()).int(), minlength=4)
    assert (hist > 0).all()
