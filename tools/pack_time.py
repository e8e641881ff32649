#!/usr/bin/env python3
"""Time the last-dim quantise + pack kernel (the V prefill, new_pack.py:217-252) on a C2-sized tensor (1 GiB fp16 in)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kivi_amd.quant import new_pack
x = [torch.randn((32, 32, 4096, 128), device="cuda", dtype=torch.float16) for _ in range(3)]
for bits in (2, 4, 8):
    ts = []
    for it in range(4):
        for xi in x:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = new_pack.triton_quantize_and_pack_along_last_dim(xi, 32, bits); e1.record()
            torch.cuda.synchronize()
            if it:
                ts.append(e0.elapsed_time(e1) * 1e3)
            del out
    ts.sort()
    n = x[0].numel()
    alg = n * 2 + n * bits // 8 + n // 32 * 4
    print(f"bits {bits}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us   {alg / ts[len(ts) // 2] / 1e6:.2f} TB/s algorithmic = {alg / ts[len(ts) // 2] / 8e6:.3f} of 8 TB/s  (KIVI_PACK_UNROLL={os.environ.get('KIVI_PACK_UNROLL', 'default')})")

# per-channel K pack straight from the un-transposed tensor into the hook layout (quant_and_pack_kcache, new_pack.py:8-27)
for bits in (2, 4):
    ts = []
    for it in range(4):
        for xi in x:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = new_pack.quantize_and_pack_k_tmajor(xi, 32, bits); e1.record()
            torch.cuda.synchronize()
            if it:
                ts.append(e0.elapsed_time(e1) * 1e3)
            del out
    ts.sort()
    n = x[0].numel()
    alg = n * 2 + n * bits // 8 + n // 32 * 4
    print(f"K per-channel pack, bits {bits}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us   {alg / ts[len(ts) // 2] / 1e6:.2f} TB/s algorithmic = {alg / ts[len(ts) // 2] / 8e6:.3f} of 8 TB/s")
