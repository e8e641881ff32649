#!/bin/bash
# A/B of the three-deep V ring (default) against the two-deep one (KIVI_ROW_X=d2) for the other decode_row instantiations:
# 4-bit g=32, 2-bit g=64, 2-bit g=128.  One line per setting (bench.py at the C2 shape with the given bits / group).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp && export TMPDIR=/tmp
CFGS=("--bits 4 --group 32" "--bits 2 --group 64 --residual 128" "--bits 2 --group 128 --residual 128")
[ -n "$ONLY" ] && CFGS=("${CFGS[@]:$ONLY:1}")
for cfg in "${CFGS[@]}"; do
  for x in d3 d2 d3 d2; do
    KIVI_ROW_X=$x timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg > /tmp/o.json 2>/tmp/o.err
    python - "$cfg" $x <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/o.json')); r = d["roofline"]
    print(f"{sys.argv[1]:>22} {sys.argv[2]}: {d['ms_per_step']:.4f} ms/step  row kernel median {r['median_launch_us']:.2f} us  frac {r['frac']:.4f}")
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e, open('/tmp/o.err').read()[-300:])
PY
  done
done
