#!/bin/bash
# Round-2 GPU session A: new parity tests, the bench line, a kernel trace of the bench command, the MFMA probe.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02a; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/dev.log 2>&1
( cd $R && timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 ); tail -5 $O/pytest.log
( cd $R && timeout 600 python -m pytest tests/test_fullsize_gpu.py -m gpu -q --durations=10 > $O/pytest_fullsize.log 2>&1 ); tail -15 $O/pytest_fullsize.log
timeout 600 python $R/bench.py > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json; tail -3 $O/bench.err
[ -x $R/tools/mfma_f16_probe.bin ] && timeout 300 $R/tools/mfma_f16_probe.bin > $O/mfma_probe.log 2>&1
rm -rf $O/trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2>$O/rocprof.err
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); echo trace=$f
python $R/tools/trace_median.py $f --skip 160 --json $O/bench_trace_summary.json | head -40
# keep the merged output small
find $O/trace -name "*.csv" -size +20M -delete
