#!/usr/bin/env python3
"""Per-wave phase timeline of mf_row_kernel (the round-3 one-launch MHA step on the matrix-pipe layout) at the bench shape
(kivi_debug_set_stamps): every wave of every block stamps the shader clock at its phase boundaries; this prints, per
phase, the median / p10 / p90 duration over all waves and the spread of the boundary inside a block.

stamp slots: 0 entry, 1 realtime(100 MHz), 3 packed qK^T of the wave's super-blocks done (V ring requested right after),
4 residual scores done, 5 after the barrier, 7 softmax done (2 block reductions + barrier), 8 window / flush done (V stream
starts), 9 V stream loop done, 10 per-wave result + combine barrier, 11 end."""
import os
os.environ.setdefault("KIVI_TUNING", "1")   # the knobs below are honoured in tuning sessions only (kivi_amd/_tuning.py)
import sys

_TUNING = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kivi_amd", "_variants", "libkivi_tuning.so")
if os.path.exists(_TUNING):      # phase stamps / environment knobs exist in the -DKIVI_TUNING build only (tools/build_variant.sh)
    os.environ.setdefault("KIVI_HIP_LIB", _TUNING)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kivi_amd import _lib  # noqa: E402
from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache  # noqa: E402

lib = _lib.load()
B, nh, D = int(os.environ.get("B", "32")), 32, 128
nh_kv = int(os.environ.get("NHKV", "32"))      # NHKV=8: the grouped-query row kernel (mf_row4_kernel), e.g. B=64 T0=8064 R=128
T0 = int(os.environ.get("T0", "4080"))
STEPS = int(os.environ.get("STEPS", "4"))
L = int(os.environ.get("LAYERS", "8"))
cfg = KiviConfig(2, 2, 32, int(os.environ.get("R", "32")))
dev = torch.device("cuda:0")
torch.manual_seed(0)
layers = []
for _ in range(L):
    lc = make_layer_cache(cfg, B, nh_kv, D, T0 + 64, dev, num_heads=nh)
    lc.prefill(torch.randn((B, nh_kv, T0, D), device=dev, dtype=torch.float16),
               torch.randn((B, nh_kv, T0, D), device=dev, dtype=torch.float16))
    layers.append(lc)
q = torch.randn((B, nh, 1, D), device=dev, dtype=torch.float16)
k = torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16)
v = torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16)
for _ in range(STEPS):
    for lc in layers:
        kivi_attention_decode(q, k, v, lc)
torch.cuda.synchronize()
nblk = B * nh_kv
NW = int(os.environ.get("NWAVES", "4"))   # waves per block of the instantiation under test (KIVI_ROW_X=nw8ds4: 8)
stamps = torch.zeros((nblk, NW, 16), dtype=torch.int64, device=dev)
lib.kivi_debug_set_stamps(stamps.data_ptr())
# the stamped launch is the LAST layer of a full pass, so it runs behind a warm stream like in the bench
for lc in layers:
    kivi_attention_decode(q, k, v, lc)
torch.cuda.synchronize()
lib.kivi_debug_set_stamps(None)
s = stamps.cpu().numpy().astype(np.int64)
t0 = s[:, :, 0].min()
rt = s[:, :, 1]
span_rt = (rt.max() - rt.min()) * 10e-3   # us between first and last wave entry (100 MHz ticks)
end = s[:, :, 11].max() - t0
# clock: per-dispatch event time of the same launch is not available here; assume the shader clock from the entry spread
print(f"blocks {nblk}, entry spread {span_rt:.2f} us (realtime), kernel span {end} shader ticks")
names = {3: "packed qK^T (prologue + stream)", 4: "V ring request + residual scores", 5: "barrier (wait for the slowest wave)",
         7: "softmax", 8: "window PV + flush", 9: "V stream loop",
         10: "per-wave result + combine barrier", 11: "final sum + store"}
slots = (3, 4, 5, 7, 8, 9, 10, 11)
if (s[:, :, 6] != 0).all():     # mf_row4_kernel (round 5): no softmax phase -- statistics merge [+ exchange between slices], then the window's probabilities
    names[3] = "packed qK^T + statistics per segment"
    names[6] = "statistics merge [+ slice exchange]"
    names[7] = "window probabilities + barrier"
    names[9] = "V stream loop (p'' made on the fly)"
    slots = (3, 4, 5, 6, 7, 8, 9, 10, 11)
mhz = float(os.environ.get("CLOCK_MHZ", "0"))
if not mhz:
    # calibrate the shader clock against the 100 MHz realtime counter (slot 1 at entry, slot 12 at exit of every wave)
    dt_rt = (s[:, :, 12] - s[:, :, 1]).reshape(-1).astype(np.float64) * 0.01   # us
    dt_sh = (s[:, :, 11] - s[:, :, 0]).reshape(-1).astype(np.float64)
    mhz = float(np.median(dt_sh / np.maximum(dt_rt, 1e-3)))
prev = 0
print(f"{'phase':38s} {'median':>9s} {'p10':>9s} {'p90':>9s}   (us at {mhz:.0f} MHz)   in-block spread of the boundary (median / p90 us)")
for i in slots:
    d = (s[:, :, i] - s[:, :, prev]).reshape(-1) / mhz
    spread = (s[:, :, i].max(axis=1) - s[:, :, i].min(axis=1)) / mhz
    print(f"{names[i]:38s} {np.median(d):9.2f} {np.percentile(d, 10):9.2f} {np.percentile(d, 90):9.2f}"
          f"{'':24s}{np.median(spread):6.2f} / {np.percentile(spread, 90):6.2f}")
    prev = i
tot = (s[:, :, 11] - s[:, :, 0]).reshape(-1) / mhz
print(f"{'wave total':38s} {np.median(tot):9.2f} {np.percentile(tot, 10):9.2f} {np.percentile(tot, 90):9.2f}")
# global time of a stamp = entry on the chip-wide 100 MHz counter + shader ticks since entry (the shader counters of
# different XCDs need not agree)
beg_us = (rt - rt.min()).astype(np.float64) * 0.01
def at(i):
    return (beg_us + (s[:, :, i] - s[:, :, 0]) / mhz).reshape(-1)
rel_end = at(11)
rel_beg = at(0)
print(f"entry   (shader clock, rel.): median {np.median(rel_beg):.2f}  p90 {np.percentile(rel_beg, 90):.2f}  max {rel_beg.max():.2f} us")
print(f"finish  (shader clock, rel.): p10 {np.percentile(rel_end, 10):.2f}  median {np.median(rel_end):.2f}  p90 {np.percentile(rel_end, 90):.2f}  max {rel_end.max():.2f} us")
# global timeline: how many waves are inside a streaming phase at each instant
edges = np.linspace(0, rel_end.max(), 41)
k_in = at(0), at(3)
v_in = at(8), at(9)
print("t (us): waves in K stream / in V stream / elsewhere (of %d)" % (nblk * NW))
for e in edges[:-1]:
    nk = int(((k_in[0] <= e) & (e < k_in[1])).sum())
    nv = int(((v_in[0] <= e) & (e < v_in[1])).sum())
    alive = int(((rel_beg <= e) & (e < rel_end)).sum())
    print(f"  {e:6.1f}: {nk:5d} {nv:5d} {alive - nk - nv:5d}")

# who lags?  K-stream duration and finish time by dispatch position and by hardware placement
kd = ((s[:, :, 3] - s[:, :, 0]) / mhz)
fin = at(11).reshape(nblk, NW)
hw = s[:, :, 13]
groups = {
    "block index >> 8 (dispatch quarter)": (np.arange(nblk)[:, None] >> 8) + 0 * hw,
    "block index & 7 (XCD by dispatch order)": (np.arange(nblk)[:, None] & 7) + 0 * hw,
    "XCC_ID": s[:, :, 14] & 15,
    "HW_ID wave slot [3:0]": hw & 15,
    "HW_ID SIMD [5:4]": (hw >> 4) & 3,
    "wave in block": np.arange(NW)[None, :] + 0 * hw,
}
for name, g in groups.items():
    print(name)
    for val in np.unique(g):
        m = g == val
        print(f"   {int(val):3d}: n {int(m.sum()):5d}  K stream median {np.median(kd[m]):6.2f} us   finish median {np.median(fin[m]):6.2f}  max {fin[m].max():6.2f}")
