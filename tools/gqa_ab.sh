#!/bin/bash
# usage: tools/gqa_ab.sh "<bench flags>" variant1 variant2 ...   ("-" = library default) -> tok/s, ms/step per forced sV variant
FLAGS=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset KIVI_GEMV_V_VARIANT; else export KIVI_GEMV_V_VARIANT=$v; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-events $FLAGS 2>/dev/null > /tmp/ab.json
  python - "$v" "$KIVI_V_SPLIT" <<'PY'
import json,sys
d=json.load(open('/tmp/ab.json'))
print("V", sys.argv[1], "split", sys.argv[2] or "auto", "tok/s", d["value"], "ms/step", d["ms_per_step"])
PY
done
