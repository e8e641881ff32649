#!/usr/bin/env python3
"""Per-step error ratios (|out - ref| / hook bar) of the matrix-pipe decode step and of the hook-layout kernels on the same
inputs, against the CPU restatement of the reference logic."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import gemv_close, make_kv
from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode, make_layer_cache
from oracle import hook_ref as H

nh, nh_kv, T0, R, masked = [int(x) for x in (sys.argv[1:6] or [8, 2, 70, 32, 1])]
B, D, g = 2, 128, 32
cfg = KiviConfig(2, 2, g, R)
k0, v0 = make_kv(1, B, nh_kv, T0, D), make_kv(2, B, nh_kv, T0, D)
mf = make_layer_cache(cfg, B, nh_kv, D, T0 + 8, "cuda", num_heads=nh)
hk = KiviLayerCache(cfg, B, nh_kv, D, T0 + 8, "cuda")
mf.prefill(k0.cuda(), v0.cuda()); hk.prefill(k0.cuda(), v0.cuda())
past = H.prefill_cache(k0, v0, 2, 2, g, R)
gen = torch.Generator().manual_seed(5)
rm, rh = [], []
for s in range(R + 9):
    q = make_kv(100 + s, B, nh, 1, D)
    kn, vn = make_kv(200 + s, B, nh_kv, 1, D), make_kv(300 + s, B, nh_kv, 1, D)
    mask = None
    if masked:
        n = T0 + s + 1
        mask = torch.zeros((B, 1, 1, n), dtype=torch.float16)
        mask[0, ..., : min(7, n - 1)] = torch.finfo(torch.float16).min
        mask[1, ..., torch.randint(0, n - 1, (3,), generator=gen)] = -3.0
    mc = None if mask is None else mask.cuda()
    o1 = kivi_attention_decode(q.cuda(), kn.cuda(), vn.cuda(), mf, attention_mask=mc)
    o2 = kivi_attention_decode(q.cuda(), kn.cuda(), vn.cuda(), hk, attention_mask=mc)
    ref, past = H.decode_step(q, kn, vn, past, 2, 2, g, R, attention_mask=mask)
    rm.append(gemv_close(o1, ref, rtol=3e-3)[1]); rh.append(gemv_close(o2, ref, rtol=3e-3)[1])
print("matrix pipe:", " ".join(f"{x:.2f}" for x in rm))
print("hook layout:", " ".join(f"{x:.2f}" for x in rh))
print(f"max {max(rm):.3f} / {max(rh):.3f}   mean {sum(rm)/len(rm):.3f} / {sum(rh)/len(rh):.3f}")
