#!/bin/bash
# same-box A/B: the library at the last commit (kivi_amd/_variants/libkivi_head.so) against the working tree, config 4 and the headline
O=gpurun_out/r3t; mkdir -p $O
export PYTHONUNBUFFERED=1
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 6 --warmup 2"
for i in 1 2; do
  KIVI_HIP_LIB=kivi_amd/_variants/libkivi_head.so timeout 300 $BN $C4 > $O/c4_head_$i.json 2>> $O/b.err
  timeout 300 $BN $C4 > $O/c4_new_$i.json 2>> $O/b.err
  KIVI_HIP_LIB=kivi_amd/_variants/libkivi_head.so timeout 300 $BN > $O/hl_head_$i.json 2>> $O/b.err
  timeout 300 $BN > $O/hl_new_$i.json 2>> $O/b.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3t/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
