#!/bin/bash
# PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the bench command at BASELINE config 4 -> merged into profiles/r03_pmc_traffic.json
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3pmc4; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-hook-kgemv"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o b -- python $R/bench.py $C4 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o b -- python $R/bench.py $C4 > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py $(find $O/f -name "*counter_collection.csv" | head -1) $(find $O/w -name "*counter_collection.csv" | head -1) --skip 32 \
   --config '{"B": 64, "nh": 32, "nh_kv": 8, "prompt": 8064, "bits": 2, "group": 32, "residual": 128}' --out $O/pmc_traffic_c4.json > $O/pmc.log 2>&1
cat $O/pmc.log
find $O -name "*.csv" -size +4M -delete
