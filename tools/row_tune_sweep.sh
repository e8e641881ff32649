#!/bin/bash
# A/B of decode_row_kernel instantiations on the bench command; one line per setting.  Arguments: "TAG:ENV=VAL[,ENV=VAL]" ...
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02_tune; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
run() {
  local tag=$1; shift
  env "$@" timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > $O/$tag.json 2> $O/$tag.err
  python - $tag $O/$tag.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d["roofline"]
    print(f"{sys.argv[1]:>12}: {d['ms_per_step']:.4f} ms/step  {d['value']:9.1f} tok/s  row kernel median {r['median_launch_us']:.2f} us min {r['min_launch_us']:.2f}  frac {r['frac']:.4f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run base KIVI_NOP=1
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  run $tag ${envs//,/ }
done
run base2 KIVI_NOP=1
