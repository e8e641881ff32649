"""CPU: the torch port used as bench.py's cpu_baseline agrees with the pinned C oracle."""
import torch

from helpers import make_kv, same_bits


def test_torch_fakequant_matches_oracle(oracle):
    from oracle import torch_fakequant as TF
    for bits, g in ((2, 32), (4, 64)):
        x = make_kv(3, 1, 2, 64, 128)
        code, scale, mn = TF.quant_pack_lastdim(x, g, bits)
        oc, os_, om = oracle.quantize_and_pack_along_last_dim(x, g, bits)
        assert same_bits(code, oc) and same_bits(scale, os_) and same_bits(mn, om)
        deq = TF.dequant_lastdim(code, scale, mn, g, bits)
        assert same_bits(deq, oracle.unpack_and_dequant_vcache(oc, os_, om, g, bits))


def test_fakequant_layer_close_to_fused_oracle(oracle):
    from oracle import torch_fakequant as TF
    B, nh, T, D = 1, 2, 128, 128
    k, v = make_kv(5, B, nh, T, D), make_kv(6, B, nh, T, D)
    q = make_kv(7, B, nh, 1, D)
    a = torch.softmax(make_kv(8, B, nh, 1, T).float(), -1).half()
    scores, out, stages = TF.fakequant_decode_layer(q, a, k, v, 32, 2)
    kc, ks, km = oracle.quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), 32, 2)
    ref = oracle.bmm_fA_qB_outer(32, q, kc, ks, km, 2, fakequant=True)
    assert (scores.float() - ref.float()).abs().max() <= 2e-3 * ref.float().pow(2).mean().sqrt()
    fused = oracle.bmm_fA_qB_outer(32, q, kc, ks, km, 2)
    assert (scores.float() - fused.float()).abs().max() <= 1e-2 * fused.float().pow(2).mean().sqrt()
    assert set(stages) == {"pack_s", "dequant_s", "gemv_s"} and out.shape == (B, nh, 1, D)


def test_hook_reference_cache_policy(oracle):
    """Appendix A of SURVEY.md: lengths of the quantised prefixes / residuals after prefill + decode steps."""
    from oracle import hook_ref as H
    R, g = 32, 32
    B, nh, nh_kv, D = 1, 4, 2, 128
    T0 = 70
    k, v = make_kv(1, B, nh_kv, T0, D), make_kv(2, B, nh_kv, T0, D)
    past = H.prefill_cache(k, v, 2, 2, g, R)
    assert past[0].shape[-1] * 16 == 64 and past[1].shape[2] == 6        # floor(70/32)*32 quantised, 70 % 32 fp16
    assert past[4].shape[2] == 38 and past[5].shape[2] == 32 and past[8] == 70
    for step in range(30):
        q = make_kv(100 + step, B, nh, 1, D)
        kn, vn = make_kv(200 + step, B, nh_kv, 1, D), make_kv(300 + step, B, nh_kv, 1, D)
        out, past = H.decode_step(q, kn, vn, past, 2, 2, g, R)
        L = T0 + step + 1
        kq = (L // R) * R
        assert past[8] == L and out.shape == (B, nh, 1, D) and torch.isfinite(out).all()
        assert (past[0].shape[-1] * 16 if past[0] is not None else 0) == kq
        assert (past[1].shape[2] if past[1] is not None else 0) == L - kq
        assert past[4].shape[2] == L - R and past[5].shape[2] == R


def test_fakequant_workspace_path_is_the_same_arithmetic():
    """bench.py times the port with its intermediates kept between repetitions (ws): same ops, same roundings -> same bits."""
    from oracle import torch_fakequant as TF
    B, nh, T, D = 1, 2, 128, 128
    k, v = make_kv(5, B, nh, T, D), make_kv(6, B, nh, T, D)
    q = make_kv(7, B, nh, 1, D)
    a = torch.softmax(make_kv(8, B, nh, 1, T).float(), -1).half()
    s0, o0, _ = TF.fakequant_decode_layer(q, a, k, v, 32, 2)
    ws = {}
    for _ in range(2):
        s1, o1, st = TF.fakequant_decode_layer(q, a, k, v, 32, 2, ws)
        assert same_bits(s0, s1) and same_bits(o0, o1)
    assert st["pack_s"] < 1e-3          # the packed cache of the first call is reused
