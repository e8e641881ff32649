#!/bin/bash
# bench.py at the other shapes of DESIGN.md section 5 (one line each): 4-bit, other batch sizes, BASELINE configs 4 and 5
# (per-GPU slice), a single long sequence.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd /tmp && export TMPDIR=/tmp
run() {
  local tag="$1"; shift
  timeout 300 python $R/bench.py --no-cpu-baseline --steps 12 --warmup 4 "$@" > /tmp/o.json 2>/tmp/o.err
  python - "$tag" <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/o.json')); r = d["roofline"]
    print(f"{sys.argv[1]:34s} {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step   dominant kernel {r['kernel'][:28]:28s} {r.get('median_launch_us', 0):7.2f} us  frac {r['frac']:.3f}")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open('/tmp/o.err').read()[-300:])
PY
}
run "C2 (headline)"
run "C2 4-bit K/V" --bits 4
run "C2 shape, B=16" --batch 16
run "C2 shape, B=64" --batch 64
run "C2 shape, B=128" --batch 128
run "config 4: B=64 32/8 heads 8k R=128" --batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128
run "config 5 slice: B=16 32/8 32k R=128" --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128
run "B=1, T=32768 (MHA)" --batch 1 --prompt 32752
