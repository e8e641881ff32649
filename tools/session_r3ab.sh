#!/bin/bash
# same-box A/B of the bench command: library of the previous commit (kivi_amd/_variants/libkivi_prev.so) vs the working tree
O=gpurun_out/r3ab; mkdir -p $O; rm -f $O/*
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
for i in 1 2 3; do
  KIVI_HIP_LIB=kivi_amd/_variants/libkivi_prev.so timeout 300 $BN > $O/prev_$i.json 2>> $O/b.err
  timeout 300 $BN > $O/new_$i.json 2>> $O/b.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3ab/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get("roofline") or {}; m=j.get("roofline_single_layer_kgemv") or {}
    print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"), "| mf kgemv", m.get("median_launch_us"), m.get("frac"))
PY
