#!/usr/bin/env python3
"""Where does the time of the fused "attend" launch go?  Per-dispatch durations of the sV launch at four fusion
levels on real KiviLayerCache objects, each preceded by the packed qK^T GEMV of the same layer (bench.py's pattern):
  L0 plain packed sV (external probabilities)         kivi_gemv_v
  L1 + fp16 window + V append + flush                 kivi_decode_output
  L2 + scale/mask/softmax of complete score rows      kivi_decode_softmax_output
  L3 + residual scores + K append                     kivi_decode_attend
The cache lengths are not advanced, so every repetition sees the same state."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kivi_amd import _lib  # noqa: E402
from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode  # noqa: E402
from kivi_amd.quant import fused, matmul  # noqa: E402

lib = _lib.load()
B, nh, D, T0 = int(os.environ.get("B", "32")), 32, 128, int(os.environ.get("T0", "4096"))
nh_kv = int(os.environ.get("NH_KV", "32"))
RES = int(os.environ.get("RES", "32"))
L = int(os.environ.get("LAYERS", "12"))
cfg = KiviConfig(2, 2, 32, RES)
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = []
for _ in range(L):
    lc = KiviLayerCache(cfg, B, nh_kv, D, T0 + 64, dev)
    lc.prefill(torch.randn((B, nh_kv, T0, D), device=dev, dtype=torch.float16),
               torch.randn((B, nh_kv, T0, D), device=dev, dtype=torch.float16))
    layers.append(lc)
q = torch.randn((B, nh, 1, D), device=dev, dtype=torch.float16)
k = torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16)
v = torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16)
for _ in range(5):
    for lc in layers:
        kivi_attention_decode(q, k, v, lc)
torch.cuda.synchronize()
lc0 = layers[0]
kv = lc0.kv_seq_len + 1
print("state: k_quant", lc0.k_quant_len, "k_res", lc0.k_res_len, "v_quant", lc0.v_quant_len, "v_win", lc0.v_res_len, "kv", kv)
pitch = (kv + 64) // 8 * 8
scores = torch.randn((B, nh, 1, pitch), device=dev, dtype=torch.float16)
probs = torch.softmax(scores[..., :kv].float(), -1).half()
probs_buf = torch.zeros((B, nh, 1, pitch), device=dev, dtype=torch.float16)
probs_buf[..., :kv] = probs
out = torch.empty((B, nh, 1, D), device=dev, dtype=torch.float16)
inv = 1.0 / math.sqrt(D)


class Shim:
    """the layer with a shorter fp16 window (no flush below R+1 tokens)"""
    def __init__(self, lc, v_res_len):
        self.__dict__["_lc"], self.__dict__["_n"] = lc, v_res_len
    def __getattr__(self, name):
        return self._n if name == "v_res_len" else getattr(self._lc, name)


def level(lc, lv):
    Tv = lc.v_quant_len
    if lv == 10:
        return fused.decode_output(Shim(lc, 0), probs_buf, v, out)
    if lv == 11:
        return fused.decode_output(Shim(lc, RES - 1), probs_buf, v, out)
    if lv == 0:
        matmul.cuda_bmm_fA_qB_outer(32, probs_buf[..., :Tv], lc.v_code[:, :, :Tv], lc.v_scale[:, :, :Tv], lc.v_mn[:, :, :Tv], 2)
    elif lv == 1:
        fused.decode_output(lc, probs_buf, v, out)
    elif lv == 2:
        fused.decode_output(lc, scores, v, out, softmax_inv_scale=inv)
    else:
        fused.decode_attend(lc, q, k, v, scores, out, inv)


def run(lv, with_k=True, reps=4):
    ev = []
    for _ in range(reps):
        for lc in layers:
            if with_k:
                matmul.gemv_k_paged(32, q, lc.k_code, lc.k_scale, lc.k_mn, lc.k_quant_len, 2, out=scores[..., :lc.k_quant_len])
            e = (lib.kivi_event_create(), lib.kivi_event_create())
            lib.kivi_set_launch_events(*e)
            level(lc, lv)
            ev.append(e)
    torch.cuda.synchronize()
    ts = sorted(lib.kivi_event_elapsed_us(a, b) for a, b in ev)
    print(f"L{lv:<2d} {'after K' if with_k else 'alone  '}: median {ts[len(ts)//2]:6.1f} min {ts[0]:6.1f} max {ts[-1]:6.1f} (n={len(ts)})")


for rnd in range(2):
    for lv in (0, 10, 11, 1, 2, 3):
        run(lv, True)
    for lv in (0, 3):
        run(lv, False)
