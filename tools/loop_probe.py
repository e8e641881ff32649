#!/usr/bin/env python3
"""Why are the fused GEMVs slower inside bench.py's decode loop than in the isolated sweep?
Times per-dispatch kernel durations of the default K / V kernels on REAL KiviLayerCache objects
(paged K, capacity-strided V) under different launch patterns."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kivi_amd import _lib  # noqa: E402
from kivi_amd.attention import KiviConfig, KiviLayerCache  # noqa: E402
from kivi_amd.quant import matmul  # noqa: E402

lib = _lib.load()
B, nh, D, T0, L = 32, 32, 128, 4096, int(os.environ.get("LAYERS", "16"))
cfg = KiviConfig(2, 2, 32, 32)
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = []
for _ in range(L):
    lc = KiviLayerCache(cfg, B, nh, D, T0 + 33, dev)
    lc.prefill(torch.randn((B, nh, T0, D), device=dev, dtype=torch.float16),
               torch.randn((B, nh, T0 + 32, D), device=dev, dtype=torch.float16)[:, :, :T0])
    lc.v_quant_len = T0 - 32
    layers.append(lc)
q = torch.randn((B, nh, 1, D), device=dev, dtype=torch.float16)
kv = T0 + 13
attn = torch.softmax(torch.randn((B, nh, 1, kv), device=dev), -1).half()
scores = torch.empty((B, nh, 1, 4136), device=dev, dtype=torch.float16)
attn_c = torch.softmax(torch.randn((B, nh, 1, 4064), device=dev), -1).half()
big = torch.randn((B, nh, 1, kv), device=dev, dtype=torch.float16)


KVAR = -1


def run(pattern, reps=3):
    evk, evv = [], []
    for _ in range(reps):
        for lc in layers:
            for op in pattern:
                if op == "k":
                    e = (lib.kivi_event_create(), lib.kivi_event_create())
                    lib.kivi_set_launch_events(*e)
                    matmul.gemv_k_paged(32, q, lc.k_code, lc.k_scale, lc.k_mn, lc.k_quant_len, 2, out=scores[..., :lc.k_quant_len], variant=KVAR)
                    evk.append(e)
                elif op in ("v", "vc"):
                    e = (lib.kivi_event_create(), lib.kivi_event_create())
                    lib.kivi_set_launch_events(*e)
                    Tv = lc.v_quant_len
                    a = attn[..., :Tv] if op == "v" else attn_c[..., :Tv]
                    matmul.cuda_bmm_fA_qB_outer(32, a, lc.v_code[:, :, :Tv], lc.v_scale[:, :, :Tv], lc.v_mn[:, :, :Tv], 2)
                    evv.append(e)
                elif op == "s":
                    torch.softmax(big, dim=-1, dtype=torch.float32)
                elif op == "m":
                    torch.matmul(q.view(B, nh, 1, D), lc.k_res.transpose(2, 3))
    torch.cuda.synchronize()
    def stats(ev):
        if not ev:
            return "-"
        ts = sorted(lib.kivi_event_elapsed_us(a, b) for a, b in ev)
        return f"median {ts[len(ts)//2]:6.1f} min {ts[0]:6.1f} max {ts[-1]:6.1f} (n={len(ts)})"
    print(f"pattern {''.join(pattern):10s}  K: {stats(evk):48s}  V: {stats(evv)}")


names = {n: i for k, i, n in matmul.bmm_variants() if k == "k"}
for vn in sys.argv[1:] or ["k_b2_g32_w2_ds4_r1_u4_m2_nt1"]:
    KVAR = names[vn]
    print("==", vn)
    for pat in (["k"], ["s", "k"], ["m", "s", "k"]):
        run(pat)
        run(pat)
