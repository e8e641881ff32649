# usage: c5ab.sh "<bench args>" VAR v1 v2 ...
ARGS="$1"; VAR=$2; shift; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events $ARGS 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$VAR=$v',d['value'],d['ms_per_step'])"
done
