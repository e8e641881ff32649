#!/usr/bin/env python3
"""Per-kernel medians of the SQ counters of one rocprofv3 --pmc pass (tools/session.sh sq): where the wave cycles of the
matrix-pipe kernels go.  SQ_BUSY_CYCLES counts per shader engine; the busy fractions below are relative to the wave cycles
(sum over all waves), i.e. 'of the cycles a wave is resident, how many did it spend issuing VALU / with the matrix pipe busy'."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if not any(t in k for t in ("mf_", "decode_row", "gemv_k_kernel", "gemv_v_kernel")):
        continue
    name = k.split("(anonymous namespace)::")[-1].split("(")[0][:70]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, " launches", len(next(iter(d.values()))))
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    wc = med.get("SQ_WAVE_CYCLES", 1) or 1
    for c, v in sorted(med.items()):
        print(f"   {c:28s} {v:16.0f}   {v / wc:6.3f} of WAVE_CYCLES")
    if med.get("SQ_INSTS_VALU") and med.get("SQ_INSTS_MFMA"):
        print(f"   VALU (non-MFMA) instructions per MFMA: {(med['SQ_INSTS_VALU'] - med['SQ_INSTS_MFMA']) / med['SQ_INSTS_MFMA']:.2f}")
