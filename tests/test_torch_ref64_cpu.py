"""The second reference (tests/torch_ref64.py: plain torch, fp64, used by the full-coverage GPU tests) against the pinned CPU oracle
on small inputs: pack bit-exact, GEMVs within the north_star bar, one hook step incl. the K and V flushes with bit-identical 9-tuples."""
import pytest
import torch

import torch_ref64 as T64
from helpers import gemv_close, make_kv, same_bits


@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("kind", ["randn", "outlier", "int"])
def test_pack_matches_oracle(oracle, bits, kind):
    x = make_kv(3, 2, 3, 64, 128, kind)
    if kind == "int":
        x = x + torch.arange(128).half() * 0.125          # (no constant groups: 0 / 0 is outside both references' contract)
    for t in (x, x.transpose(2, 3).contiguous()):
        c, s, m = T64.quant_pack_lastdim(t, 32, bits)
        oc, os_, om = oracle.quantize_and_pack_along_last_dim(t, 32, bits)
        assert torch.equal(c, oc) and same_bits(s, os_) and same_bits(m, om)
        assert torch.equal(T64.unpack_lastdim(c, bits).to(torch.int16), oracle.unpack_tensor(oc, bits, 3)) if hasattr(oracle, "unpack_tensor") else True


@pytest.mark.parametrize("bits,ratio", [(2, 1), (2, 4), (4, 4), (2, 8)])
def test_gemvs_match_oracle(oracle, bits, ratio):
    B, nh_kv, T, D, g = 2, 2, 256, 128, 32
    nh = nh_kv * ratio
    k, v = make_kv(5, B, nh_kv, T, D), make_kv(6, B, nh_kv, T, D)
    gen = torch.Generator().manual_seed(7)
    q = torch.randn((B, nh, 1, D), generator=gen).half()
    a = torch.softmax(torch.randn((B, nh, 1, T), generator=gen) * 2, -1).half()
    kc, ks, km = oracle.quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), g, bits)
    vc, vs, vm = oracle.quantize_and_pack_along_last_dim(v, g, bits)
    ok, r = gemv_close(T64.scores64(q, kc, ks, km, g, bits), oracle.bmm_fA_qB_outer(g, q, kc, ks, km, bits))
    assert ok, r
    ok, r = gemv_close(T64.output64(a, vc, vs, vm, g, bits), oracle.bmm_fA_qB_outer(g, a, vc, vs, vm, bits))
    assert ok, r


@pytest.mark.parametrize("bits,ratio,R,T0,masked", [(2, 1, 32, 70, False), (2, 4, 32, 95, True), (4, 4, 64, 200, False)])
def test_hook_step_matches_oracle_hook(oracle, bits, ratio, R, T0, masked):
    from oracle import hook_ref as H
    B, nh_kv, D, g = 2, 2, 128, 32
    nh = nh_kv * ratio
    k0, v0 = make_kv(8, B, nh_kv, T0, D), make_kv(9, B, nh_kv, T0, D)
    past = T64.prefill_cache(k0, v0, bits, bits, g, R)
    ref_past = H.prefill_cache(k0, v0, bits, bits, g, R)
    gen = torch.Generator().manual_seed(10)
    for s in range(R + 3):                                 # through a K flush and V flushes
        q = torch.randn((B, nh, 1, D), generator=gen).half()
        kn = torch.randn((B, nh_kv, 1, D), generator=gen).half()
        vn = torch.randn((B, nh_kv, 1, D), generator=gen).half()
        mask = None
        if masked:
            mask = torch.zeros((B, 1, 1, T0 + s + 1), dtype=torch.float16)
            mask[0, :, :, :9 + s] = torch.finfo(torch.float16).min
        out, past, pre = T64.decode_step(q, kn, vn, past, bits, bits, g, R, attention_mask=mask)
        ref, ref_past, ref_pre = H.decode_step(q, kn, vn, ref_past, bits, bits, g, R, attention_mask=mask, return_scores=True)
        live = ref_pre.float() > -60000
        ok, r = gemv_close(torch.where(live, pre.float(), 0.0), torch.where(live, ref_pre.float(), 0.0), ulps=1)
        assert ok, ("scores", s, r)
        ok, r = gemv_close(out, ref, rtol=3e-3, ulps=1)
        assert ok, ("out", s, r)
        for a, b in zip(past[:8], ref_past[:8]):
            assert (a is None) == (b is None) and (a is None or same_bits(a.contiguous(), b.contiguous())), s
        assert past[8] == ref_past[8]
