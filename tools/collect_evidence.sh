#!/bin/bash
# One GPU session that produces everything profiles/ is condensed from (tools/summarize_profiles.py <round>).
# usage (on the GPU box, from the repo root):  tools/collect_evidence.sh
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out; mkdir -p $O/prof; cd /tmp && export TMPDIR=/tmp
KV=k_b2_g32_w2_ds4_r1_u4_m2_nt1
python -m pytest $R/tests -m gpu -x -q > $O/pytest_final.log 2>&1; tail -1 $O/pytest_final.log
python $R/bench.py --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; cut -c1-200 $O/bench_final.json
rm -rf $O/prof/bench_trace $O/prof/pmc_fetch $O/prof/pmc_write $O/prof/pmc_calib
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/bench_trace -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > $O/prof/bench_trace.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_fetch -o k -- python $R/tools/gpu_sweep.py --nbuf 12 --iters 12 --skip_pack --only $KV > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_write -o k -- python $R/tools/gpu_sweep.py --nbuf 12 --iters 12 --skip_pack --only $KV > /dev/null 2>&1
rm -rf $O/prof/pmc_fetch_row $O/prof/pmc_write_row
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_fetch_row -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_write_row -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/prof/pmc_calib -o c -- $R/tools/hbm_read_bw.bin 2 > $O/hbm_bw.log 2>&1
python $R/tools/gpu_sweep.py 2>&1 | grep -v "^/opt" > $O/sweep_final.log
cd $R
{
  tools/gqa_ab.sh "--bits 4" -
  tools/gqa_ab.sh "--batch 64 --kv-heads 8 --prompt 8192 --residual 128" -
  tools/gqa_ab.sh "--batch 16 --kv-heads 8 --prompt 32768 --residual 128" -
  tools/gqa_ab.sh "--batch 1 --prompt 32768" -
} > $O/shapes.log 2>&1; cat $O/shapes.log
tools/prof_shapes.sh c4 c5 b1 > $O/prof_shapes.log 2>&1
{ python examples/mem_spd_test.py --batch 32 --prompt 2048 --gen 512; python examples/mem_spd_test.py --batch 32 --prompt 2048 --gen 512 --graphs; python examples/mem_spd_test.py --batch 32 --prompt 2048 --gen 512 --baseline; } 2>/dev/null > $O/e2e.log; cat $O/e2e.log | cut -c1-400
