#!/bin/bash
# A/B of the sV launch with the fp16 window in dedicated blocks at the tail of the grid (default) against shares inside the
# stream blocks (KIVI_GQA_WIN_TAIL=0): bench lines at the config-4 shape and the config-5 slice, two rounds.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp && export TMPDIR=/tmp
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --no-cpu-baseline --steps 12 --warmup 4"
C5="--batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --no-cpu-baseline --steps 12 --warmup 4"
for round in 1 2; do
  for tag in c4 c5; do
    [ $tag = c4 ] && A="$C4" || A="$C5"
    for wt in 1 0; do
      KIVI_GQA_WIN_TAIL=$wt timeout 300 python $R/bench.py $A > /tmp/o.json 2>/tmp/o.err
      python - $tag $wt <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/o.json'))
    print(f"{sys.argv[1]} win_tail={sys.argv[2]}: {d['ms_per_step']:.4f} ms/step  {d['value']:9.1f} tok/s")
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e, open('/tmp/o.err').read()[-400:])
PY
    done
  done
done
