#!/usr/bin/env python3
"""Prefill of the matrix-pipe cache layout at the config-4 shape (B=64, 8 kv heads, 8192 tokens, D=128 = 1 GiB of K and of V):
kivi_kt_pack / kivi_vt_pack (quantise straight into the KT / VT layouts), and the two-pass V route they replace (last-dim
pack + relayout)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kivi_amd.quant import mfma, new_pack

B, H, T, D = int(os.environ.get("B", "64")), 8, int(os.environ.get("T", "8192")), 128
BITS = int(os.environ.get("BITS", "2"))                       # 4: the KT4 / VT4 packers (nh / nh_kv = 4 models)
k = torch.randn((B, H, T, D), device="cuda", dtype=torch.float16)
store = mfma.alloc_store(B, H, (T + 511) // 512, "cuda", BITS)


def timed(fn, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], out


n = k.numel()
alg = n * 2 + n * BITS // 8 + n // 32 * 4
t, _ = timed(lambda: mfma.kt_pack(k, store, 0, 32, BITS))
print(f"kt_pack            {t:8.1f} us  {alg / t / 1e6:.2f} TB/s algorithmic = {alg / t / 8e6:.3f}")
t, _ = timed(lambda: mfma.vt_pack(k, store, 32, BITS))
print(f"vt_pack            {t:8.1f} us  {alg / t / 1e6:.2f} TB/s algorithmic = {alg / t / 8e6:.3f}")
t, out = timed(lambda: new_pack.triton_quantize_and_pack_along_last_dim(k, 32, BITS))
print(f"V last-dim pack    {t:8.1f} us  {alg / t / 1e6:.2f} TB/s algorithmic = {alg / t / 8e6:.3f}")
vc, vs, vm = out
rel = (n * BITS // 8 + n // 32 * 4) * 2
t, _ = timed(lambda: mfma.vt_from_ref(store, vc, vs, vm, 32, BITS))
print(f"vt_from_ref        {t:8.1f} us  {rel / t / 1e6:.2f} TB/s (read + write of the packed bytes)")
