#!/bin/bash
# GQA shapes (BASELINE configs 4 / 5): tests of the matrix-pipe layout, bench lines, per-kernel medians from a kernel trace.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02_gqa; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest tests/test_mfma_gpu.py -m gpu -q 2>&1 | tail -6 ) | tee $O/pytest_mfma.log
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --no-cpu-baseline"
C5="--batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --no-cpu-baseline"
for tag in c4 c5; do
  [ $tag = c4 ] && A="$C4" || A="$C5"
  timeout 600 python $R/bench.py $A > $O/bench_$tag.json 2> $O/bench_$tag.err; cut -c1-330 $O/bench_$tag.json; tail -2 $O/bench_$tag.err
  KIVI_NO_MFMA_LAYOUT=1 timeout 600 python $R/bench.py $A > $O/bench_${tag}_valu.json 2> $O/bench_${tag}_valu.err; cut -c1-330 $O/bench_${tag}_valu.json
  rm -rf $O/trace_$tag
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -o t -- python $R/bench.py $A --steps 10 --warmup 3 --no-kernel-events > /dev/null 2> $O/rocprof_$tag.err
  f=$(find $O/trace_$tag -name "*kernel_trace.csv" | head -1)
  python $R/tools/trace_median.py $f --skip 96 --json $O/trace_${tag}_median.json | grep -E '"median_us"|": \{|calls|vgpr' | paste - - - - | head -8
  find $O/trace_$tag -name "*.csv" -size +5M -delete
done
