#!/bin/bash
# HBM traffic of the dominant kernels from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 --pmc passes (kernel trace only), plus the calibration pass on a kernel with a known byte count.
# Condense with:  python tools/summarize_profiles.py r02   (reads gpurun_out/prof/, writes profiles/r02_kgemv_pmc.json)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out; mkdir -p $O/prof; cd /tmp && export TMPDIR=/tmp
KV=k_b2_g32_w2_ds4_r1_u4_m2_nt1
rm -rf $O/prof/bench_trace $O/prof/pmc_fetch $O/prof/pmc_write $O/prof/pmc_calib $O/prof/pmc_fetch_row $O/prof/pmc_write_row
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_fetch -o k -- python $R/tools/gpu_sweep.py --nbuf 12 --iters 12 --skip_pack --only $KV > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_write -o k -- python $R/tools/gpu_sweep.py --nbuf 12 --iters 12 --skip_pack --only $KV > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_fetch_row -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_write_row -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_calib -o c -- $R/tools/hbm_read_bw.bin 2 > $O/hbm_bw.log 2>&1
# keep only the counter tables (the merge back is capped at 64 MiB)
find $O/prof -name "*kernel_trace.csv" -delete; find $O/prof -name "*agent_info.csv" -delete
du -sh $O/prof; find $O/prof -name "*counter_collection.csv" | head
