#!/bin/bash
# past-the-end ring requests out of range (no traffic) instead of repeating the last block: parity, A/B, PMC traffic
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3dead; mkdir -p $O; rm -rf $O/*
export PYTHONUNBUFFERED=1
cd $R
timeout 900 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu > $O/t.log 2>&1; echo "t rc=$?" >> $O/status.log; tail -2 $O/t.log
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
for i in 1 2; do
  timeout 300 $BN > $O/hl_new_$i.json 2>> $O/b.err
done
timeout 300 $BN --batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 6 --warmup 2 > $O/c4_new.json 2>> $O/b.err
timeout 300 $BN --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2 > $O/c5_new.json 2>> $O/b.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-hook-kgemv > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-hook-kgemv > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $(find $O/f -name "*counter_collection.csv" | head -1) $(find $O/w -name "*counter_collection.csv" | head -1) --skip 32 \
   --config '{"B": 32, "nh": 32, "nh_kv": 32, "prompt": 4080, "bits": 2, "group": 32, "residual": 32}' --out $O/pmc_traffic.json 2>&1 | tail -3
find $O -name "*.csv" -size +4M -delete
cat $O/status.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3dead/*_new*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
    print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"))
PY
