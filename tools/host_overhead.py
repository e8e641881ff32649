#!/usr/bin/env python3
"""Host cost of one layer step of the matrix-pipe cache (Python bookkeeping + the ctypes call), with the library call
replaced by a no-op and with the real call: what B <= 4 steps are bound by (bench.py: host_enqueue_ms_per_step)."""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
cfg = KiviConfig(2, 2, 32, 32)
B, nh, D, T0, L = 2, 32, 128, 1024, 32
layers = []
for _ in range(L):
    lc = make_layer_cache(cfg, B, nh, D, T0 + 4096, "cuda", num_heads=nh)
    lc.prefill(torch.randn((B, nh, T0, D), device="cuda", dtype=torch.float16), torch.randn((B, nh, T0, D), device="cuda", dtype=torch.float16))
    layers.append(lc)
q = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16); k = torch.randn_like(q); v = torch.randn_like(q)
out = torch.empty_like(q)
def steps(n):
    for _ in range(n):
        for lc in layers:
            kivi_attention_decode(q, k, v, lc, out=out)
steps(3); torch.cuda.synchronize()
t = time.perf_counter(); steps(20); dt = time.perf_counter() - t; torch.cuda.synchronize()
print(f"real call: {dt / 20 * 1e3:.3f} ms per 32-layer step enqueue ({dt / 20 / L * 1e6:.1f} us per layer)")
pr = cProfile.Profile(); pr.enable(); steps(20); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
