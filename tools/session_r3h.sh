#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu -k "row_kernel or flushes" > $O/t_row.log 2>&1; echo "t_row rc=$?" >> $O/status.log
BN="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hook-kgemv"
timeout 300 $BN > $O/b_prio.json 2> $O/b.err
KIVI_MF_ROW_NOPRIO=1 timeout 300 $BN > $O/b_noprio.json 2>> $O/b.err
for r in 24 23 22 42; do KIVI_MF_ROW_RINGS=$r timeout 300 $BN > $O/b_rings$r.json 2>> $O/b.err; done
KIVI_NO_MFMA_MHA=1 timeout 300 $BN > $O/b_old.json 2>> $O/b.err
python tools/mf_row_phases.py > $O/row_phases_prio.log 2>&1
KIVI_MF_ROW_NOPRIO=1 python tools/mf_row_phases.py > $O/row_phases_noprio.log 2>&1
tail -n 3 $O/t_row.log; cat $O/status.log; head -14 $O/row_phases_prio.log; head -14 $O/row_phases_noprio.log | tail -12
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3h/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
