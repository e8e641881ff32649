#!/bin/bash
# per-kernel times (rocprofv3) of bench.py at the small-batch / long-context shapes
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  case $tag in
    b1) A="--batch 1 --prompt 32768";;
    c5) A="--batch 16 --kv-heads 8 --prompt 32768 --residual 128";;
    c4) A="--batch 64 --kv-heads 8 --prompt 8192 --residual 128";;
    c2) A="";;
  esac
  mkdir -p $R/gpurun_out/prof4; rm -rf $R/gpurun_out/prof4/$tag
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof4/$tag -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events $A > $R/gpurun_out/prof4/$tag.json 2>/dev/null
  python $R/tools/prof_shapes.py $R/gpurun_out/prof4/$tag $tag
done
