#!/bin/bash
# SQ counters of decode_row_kernel on the bench command (where do the wave cycles go: parked in s_waitcnt, issue-stalled,
# issuing VALU?).  One pass, 7 SQ counters; --kernel-trace only (gpurun rule).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf $O/sq_row
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq_row -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $O/run_row.log 2>&1
f=$(find $O/sq_row -name "*counter_collection.csv" | head -1); echo $f
python - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "decode_row" not in k and "gemv_k_kernel" not in k:
        continue
    agg[k.split("(anonymous namespace)::")[-1].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, " launches", len(next(iter(d.values()))))
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    wc = med.get("SQ_WAVE_CYCLES", 1)
    for c, v in med.items():
        print(f"   {c:28s} {v:16.0f}   {v / wc:6.3f} of WAVE_CYCLES")
PY
find $O/sq_row -name "*.csv" -size +3M -delete
