#!/bin/bash
# SQ counters of the two launches of the grouped-query decode step (where do the wave cycles go: parked in s_waitcnt,
# issue-stalled, issuing VALU / MFMA?).  One pass, 8 SQ counters; --kernel-trace only (gpurun rule).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02_pmc; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf $O/sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq -o p -- python $R/tools/gqa_step_time.py --child --layers 3 --iters 2 > $O/run.log 2>&1
f=$(find $O/sq -name "*counter_collection.csv" | head -1); echo $f
python - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "gqa_" not in k:
        continue
    agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    wc = med.get("SQ_WAVE_CYCLES", 1)
    for c, v in med.items():
        print(f"   {c:28s} {v:14.0f}   {v / wc:6.3f} of WAVE_CYCLES")
PY
