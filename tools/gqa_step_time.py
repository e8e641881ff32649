#!/usr/bin/env python3
"""Per-dispatch medians of the two launches of the matrix-pipe decode step (kivi_gqa_decode) at a given shape, rotating
over several layer caches.  KIVI_GQA_TIME_V=1 (set by this script for the second pass) moves the event pair to the sV launch."""
import argparse, os, subprocess, sys
os.environ.setdefault("KIVI_TUNING", "1")   # the knobs below are honoured in tuning sessions only (kivi_amd/_tuning.py)
_TUNING = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kivi_amd", "_variants", "libkivi_tuning.so")
if os.path.exists(_TUNING):      # environment knobs exist in the -DKIVI_TUNING build only (tools/build_variant.sh)
    os.environ.setdefault("KIVI_HIP_LIB", _TUNING)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(args):
    import torch
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    lib = _lib.load()
    B, nh, kv, T, R = args.batch, args.heads, args.kv_heads, args.tokens, args.residual
    cfg = KiviConfig(2, 2, 32, R)
    layers = []
    for _ in range(args.layers):
        lc = make_layer_cache(cfg, B, kv, 128, T + 64, "cuda", num_heads=nh)
        lc.prefill(torch.randn((B, kv, T, 128), device="cuda", dtype=torch.float16), torch.randn((B, kv, T, 128), device="cuda", dtype=torch.float16))
        layers.append(lc)
    q = torch.randn((B, nh, 1, 128), device="cuda", dtype=torch.float16)
    k = torch.randn((B, kv, 1, 128), device="cuda", dtype=torch.float16)
    v = torch.randn((B, kv, 1, 128), device="cuda", dtype=torch.float16)
    ev = []
    for it in range(args.iters + 2):
        for lc in layers:
            e0, e1 = lib.kivi_event_create(), lib.kivi_event_create()
            lib.kivi_set_launch_events(e0, e1)
            kivi_attention_decode(q, k, v, lc)
            if it >= 2:
                ev.append((e0, e1))
    torch.cuda.synchronize()
    us = sorted(lib.kivi_event_elapsed_us(a, b) for a, b in ev)
    nbytes = B * kv * (128 * T // 4 + 2 * 128 * (T // 32) * 2)
    print(f"{(lib.kivi_last_timed_kernel() or b'').decode()[:44]:44s} median {us[len(us) // 2]:7.2f} us  min {us[0]:7.2f}  {nbytes / us[len(us) // 2] / 1e6:5.2f} TB/s = {nbytes / us[len(us) // 2] / 8e6:.3f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=8064)
    ap.add_argument("--residual", type=int, default=128)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        run(args)
    else:
        for tv in ("", "1"):
            env = dict(os.environ)
            if tv:
                env["KIVI_GQA_TIME_V"] = "1"
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:], env=env)
