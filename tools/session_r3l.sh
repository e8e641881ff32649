#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu -k "row_kernel" > $O/t_row.log 2>&1; echo "t_row rc=$?" >> $O/status.log
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3 --no-cpu-baseline"
for c in 823 822 444 443; do KIVI_MF_ROW4=$c timeout 300 python bench.py $C4 > $O/c4_row4_$c.json 2>> $O/c4.err; done
KIVI_MF_NO_ROW=1 KIVI_MF_RING=2 KIVI_GQA_V_BLOCKS=512 timeout 300 python bench.py $C4 > $O/c4_split.json 2>> $O/c4.err
KIVI_MF_OLD=1 timeout 300 python bench.py $C4 > $O/c4_old.json 2>> $O/c4.err
tail -n 4 $O/t_row.log; cat $O/status.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3l/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
