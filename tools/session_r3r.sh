#!/bin/bash
# lean row softmax (scaled scores + running maxima from the qK^T sinks, packed conversions) + window loads requested before it:
# parity of both row kernels, phase timelines, bench at the headline and at config 4
O=gpurun_out/r3r; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu -k "row_kernel or decode_steps" > $O/t.log 2>&1; echo "t rc=$?" >> $O/status.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_hook_gpu.py -q -x -m gpu > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/status.log
T=kivi_amd/_variants/libkivi_tuning.so
BN="python bench.py --no-cpu-baseline"
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 6 --warmup 2"
KIVI_HIP_LIB=$T B=64 NHKV=8 T0=8064 R=128 LAYERS=6 timeout 300 python tools/mf_row_phases.py > $O/row4_phases.log 2>&1
KIVI_HIP_LIB=$T timeout 300 python tools/mf_row_phases.py > $O/row_phases.log 2>&1
timeout 300 $BN $C4 --no-hook-kgemv > $O/c4_product.json 2>> $O/b.err
timeout 300 $BN --no-hook-kgemv > $O/b_headline.json 2>> $O/b.err
timeout 300 $BN --no-hook-kgemv > $O/b_headline2.json 2>> $O/b.err
for b in 8 16; do timeout 300 $BN --no-hook-kgemv --batch $b --steps 10 --warmup 3 > $O/b${b}.json 2>> $O/b.err; done
tail -n 3 $O/t.log $O/t2.log; cat $O/status.log
head -14 $O/row4_phases.log; head -14 $O/row_phases.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3r/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 $O/b.err
