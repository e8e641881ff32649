#!/bin/bash
# usage: tools/ab_bench.sh VAR v1 v2 ...   -> runs bench.py with VAR=value, two rounds each, prints tok/s, ms/step, K-GEMV us
VAR=$1; shift
for round in 1 2; do
  for v in "$@"; do
    env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null > /tmp/ab.json
    python - "$VAR" "$v" <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print(sys.argv[1], sys.argv[2], "tok/s", d["value"], "ms/step", d["ms_per_step"], "K us", d["roofline"]["avg_launch_us"])
PY
  done
done
