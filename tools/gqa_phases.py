#!/usr/bin/env python3
"""Per-wave phase times of gqa_v_kernel (kivi_debug_set_stamps): slots 0 entry, 2 softmax constants done, 3 stream loop /
window accumulation done, 4 block partial ready, 5 after arrival (+ combine by the last block); 1 / 12 = 100 MHz realtime."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kivi_amd import _lib
from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
lib = _lib.load()
B, nh, kv, T, R = int(os.environ.get("B", 64)), 32, 8, int(os.environ.get("T", 8064)), 128
cfg = KiviConfig(2, 2, 32, R)
layers = []
for _ in range(4):
    lc = make_layer_cache(cfg, B, kv, 128, T + 64, "cuda", num_heads=nh)
    lc.prefill(torch.randn((B, kv, T, 128), device="cuda", dtype=torch.float16), torch.randn((B, kv, T, 128), device="cuda", dtype=torch.float16))
    layers.append(lc)
q = torch.randn((B, nh, 1, 128), device="cuda", dtype=torch.float16)
k = torch.randn((B, kv, 1, 128), device="cuda", dtype=torch.float16)
v = torch.randn((B, kv, 1, 128), device="cuda", dtype=torch.float16)
for _ in range(3):
    for lc in layers:
        kivi_attention_decode(q, k, v, lc)
torch.cuda.synchronize()
nblk = 8192
st = torch.zeros((nblk, 4, 16), dtype=torch.int64, device="cuda")
lib.kivi_debug_set_stamps(st.data_ptr())
for lc in layers:
    st.zero_()
    kivi_attention_decode(q, k, v, lc)
torch.cuda.synchronize()
lib.kivi_debug_set_stamps(None)
s = st.cpu().numpy()
used = s[:, 0, 0] != 0
nb = int(used.sum())
units = B * kv
print(f"blocks stamped {nb}")
dt_rt = (s[used][:, :, 12] - s[used][:, :, 1]).astype(np.float64) * 0.01
dt_sh = (s[used][:, :, 5] - s[used][:, :, 0]).astype(np.float64)
mhz = float(np.median(dt_sh / np.maximum(dt_rt, 1e-3)))
rt0 = s[used][:, :, 1].min()
for name, sel in (("stream blocks", slice(0, nb)),):
    x = s[sel]
    beg = (x[:, :, 1] - rt0) * 0.01
    print(f"{name}: entry median {np.median(beg):.1f} us p90 {np.percentile(beg, 90):.1f};  exit median {np.median((x[:, :, 12] - rt0) * 0.01):.1f} max {((x[:, :, 12] - rt0) * 0.01).max():.1f} us   (clock {mhz:.0f} MHz)")
    prev = 0
    for i, nm in ((2, "softmax constants"), (3, "stream loop"), (4, "window share + fold + block partial"), (5, "arrival (+ combine)")):
        d = (x[:, :, i] - x[:, :, prev]).reshape(-1) / mhz
        print(f"   {nm:26s} median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f} us")
        prev = i
