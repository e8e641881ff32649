#!/bin/bash
# kernel-trace medians of the grouped-query shapes (BASELINE config 4 and the config-5 per-GPU slice)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3tr; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4 -o b -- python $R/bench.py --no-cpu-baseline --no-hook-kgemv --batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3 > $O/c4_bench.json 2> $O/c4.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o b -- python $R/bench.py --no-cpu-baseline --no-hook-kgemv --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2 > $O/c5_bench.json 2> $O/c5.err
cd $R
python tools/trace_median.py $(find $O/c4 -name "*kernel_trace.csv" | head -1) --skip 96 --match mf_ kt_pack --json $O/trace_median_config4.json > $O/c4_median.log 2>&1
python tools/trace_median.py $(find $O/c5 -name "*kernel_trace.csv" | head -1) --skip 64 --match mf_ kt_pack --json $O/trace_median_config5slice.json > $O/c5_median.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
python - <<'PY'
import json
for c in ("config4","config5slice"):
    j=json.load(open(f"gpurun_out/r3tr/trace_median_{c}.json"))
    for k,v in j.items(): print(c, k[:60], v["calls"], v["median_us"], v["p10_us"], v["p90_us"])
for c in ("c4","c5"):
    j=json.loads(open(f"gpurun_out/r3tr/{c}_bench.json").read().strip().splitlines()[-1]); print(c, j["value"], j["ms_per_step"])
PY
