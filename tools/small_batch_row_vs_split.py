#!/usr/bin/env python3
"""Few rows (B x nh < 192): the one-launch row kernel (GQA_FORCE_ROW) against the default two-launch form, ms per 32-layer step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kivi_amd import _lib
from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
cfg = KiviConfig(2, 2, 32, 32)
nh, D, L = 32, 128, 32
for B, T0 in ((1, 4080), (2, 4080), (4, 4080), (5, 4080), (1, 8000)):
    res = {}
    for name, flag in (("two launches", 0), ("row kernel", _lib.GQA_FORCE_ROW)):
        layers = []
        for _ in range(L):
            lc = make_layer_cache(cfg, B, nh, D, T0 + 256, "cuda", num_heads=nh)
            lc.prefill(torch.randn((B, nh, T0, D), device="cuda", dtype=torch.float16), torch.randn((B, nh, T0, D), device="cuda", dtype=torch.float16))
            lc.flags = flag
            layers.append(lc)
        q = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16); k = torch.randn_like(q); v = torch.randn_like(q)
        out = torch.empty_like(q)
        def steps(n):
            for _ in range(n):
                for lc in layers:
                    kivi_attention_decode(q, k, v, lc, out=out)
        steps(3); torch.cuda.synchronize()
        t = time.perf_counter(); steps(20); torch.cuda.synchronize(); res[name] = (time.perf_counter() - t) / 20 * 1e3
        del layers
    print(f"B={B} T0={T0} ({B * nh} rows): " + "  ".join(f"{n} {ms:.3f} ms/step" for n, ms in res.items()))
