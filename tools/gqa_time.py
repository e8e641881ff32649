#!/usr/bin/env python3
"""Per-dispatch times of the grouped-query qK^T / sV kernels at a given shape: matrix-pipe layout vs the VALU kernels of
the paged layout (rotating over several caches so nothing lives in the Infinity Cache)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=8)
    ap.add_argument("--tokens", type=int, default=8192)
    ap.add_argument("--nbuf", type=int, default=6)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    from kivi_amd import _lib
    from kivi_amd.quant import matmul, mfma, new_pack
    lib = _lib.load()
    B, nh, kv, T = args.batch, args.heads, args.kv_heads, args.tokens
    dev = "cuda"
    stores, refs = [], []
    for i in range(args.nbuf):
        k = torch.randn((B, kv, T, 128), device=dev, dtype=torch.float16)
        st = mfma.alloc_store(B, kv, (T + 511) // 512, dev)
        mfma.kt_pack(k, st, 0)
        stores.append(st)
        refs.append(new_pack.quantize_and_pack_k_tmajor(k, 32, 2))
        del k
    q = torch.randn((B, nh, 1, 128), device=dev, dtype=torch.float16)
    out = torch.empty((B, nh, 1, T), device=dev, dtype=torch.float16)
    nbytes = B * kv * (128 * T // 4 + 2 * 128 * (T // 32) * 2) + B * nh * (128 * 2 + T * 2)

    def timed(fn):
        ev = []
        for it in range(args.iters + 1):
            for i in range(args.nbuf):
                e0, e1 = lib.kivi_event_create(), lib.kivi_event_create()
                lib.kivi_set_launch_events(e0, e1)
                fn(i)
                if it:
                    ev.append((e0, e1))
        torch.cuda.synchronize()
        us = sorted(lib.kivi_event_elapsed_us(a, b) for a, b in ev)
        return us[len(us) // 2], us[0], (lib.kivi_last_timed_kernel() or b"").decode()[:60]

    for name, fn in (("mfma qK^T", lambda i: mfma.gqa_scores(q, stores[i], T, out)),
                     ("valu qK^T", lambda i: matmul.cuda_bmm_fA_qB_outer(32, q, *refs[i], 2))):
        med, mn, kern = timed(fn)
        print(f"{name}: median {med:7.2f} us  min {mn:7.2f} us  {nbytes / med / 1e6:6.2f} TB/s = {nbytes / med / 8e6:.3f} of 8 TB/s   [{kern}]")


if __name__ == "__main__":
    main()
