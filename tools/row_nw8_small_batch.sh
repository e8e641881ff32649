#!/bin/bash
# A/B of the eight-waves-per-row decode_row instantiation (KIVI_ROW_X=nw8ds4) at batch sizes whose rows do not fill the chip
# with four-wave blocks (B = 8 / 16 / 24 at 32 heads).
cd /tmp && export TMPDIR=/tmp
for b in ${BATCHES:-16 8 24}; do for x in ${VARIANTS:-d3 nw8ds4 d3 nw8ds4}; do
  KIVI_ROW_X=$x timeout 300 python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 12 --warmup 4 --batch $b > /tmp/o.json 2>/tmp/o.err
  python - $b $x <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/o.json')); r = d["roofline"]
    print(f"B={sys.argv[1]:3s} {sys.argv[2]:7s} {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step   {r['kernel'][:24]:24s} {r.get('median_launch_us', 0):7.2f} us  frac {r['frac']:.3f}")
except Exception as e:
    print(sys.argv[1], sys.argv[2], "FAILED", e, open('/tmp/o.err').read()[-300:])
PY
done; done
