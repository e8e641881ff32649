#!/bin/bash
# Round-2 final evidence: full GPU suite, the bench line, a kernel trace of the bench command with per-kernel medians.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02_final; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1 ); tail -4 $O/pytest.log
( cd $R && timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 ); tail -1 $O/smoke.log
timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json; tail -2 $O/bench.err
rm -rf $O/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2>$O/rocprof.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_median.py $f --skip 160 --json $O/bench_trace_median.json | grep -E '"median_us"|": \{|calls' | paste - - - | head -6
find $O/trace -name "*kernel_trace.csv" -size +5M -delete
