#!/bin/bash
# two-launch grouped-query path after: continuous multi-super-block qK^T walk for R = 4 (mf_k_seq4), lean statistics,
# prefetched + lean probability stage of the sV launch.  Parity first, then per-launch medians at the config-5 slice and config 4.
O=gpurun_out/r3w; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu > $O/t.log 2>&1; echo "t rc=$?" >> $O/status.log
T=kivi_amd/_variants/libkivi_tuning.so
C5="--batch 16 --heads 32 --kv-heads 8 --tokens 32640 --residual 128 --layers 6"
C4="--batch 64 --heads 32 --kv-heads 8 --tokens 8064 --residual 128 --layers 6"
for cfg in "" "KIVI_MF_RING=2" "KIVI_MF_RING=8 KIVI_MF_VRING=2" "KIVI_MF_VRING=4" "KIVI_MF_SPW=2" "KIVI_MF_SPW=8" "KIVI_GQA_V_BLOCKS=1024"; do
  echo "== c5 $cfg" >> $O/steps.log
  env $cfg timeout 300 python tools/gqa_step_time.py $C5 >> $O/steps.log 2>&1
done
echo "== c4 split (KIVI_MF_NO_ROW=1)" >> $O/steps.log
KIVI_MF_NO_ROW=1 timeout 300 python tools/gqa_step_time.py $C4 >> $O/steps.log 2>&1
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
timeout 300 $BN --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2 > $O/c5.json 2>> $O/b.err
KIVI_HIP_LIB=kivi_amd/_variants/libkivi_head.so timeout 300 $BN --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2 > $O/c5_head.json 2>> $O/b.err
timeout 300 $BN --batch 1 --prompt 32752 --steps 10 --warmup 3 > $O/b1_32k.json 2>> $O/b.err
tail -n 3 $O/t.log; cat $O/status.log; grep -v "amdgpu.ids" $O/steps.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3w/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j.get("roofline") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"))
    except Exception as e: print(f, "ERR", e)
PY
