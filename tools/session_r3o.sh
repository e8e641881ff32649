#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu -k "row_kernel or decode_steps" > $O/t.log 2>&1; echo "t rc=$?" >> $O/status.log
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -m gpu -k "mf_row or small_batch" > $O/t2.log 2>&1; echo "t2 rc=$?" >> $O/status.log
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
timeout 300 $BN > $O/b_ring_1.json 2>> $O/b.err
KIVI_MF_NO_RING=1 timeout 300 $BN > $O/b_linear.json 2>> $O/b.err
timeout 300 $BN > $O/b_ring_2.json 2>> $O/b.err
KIVI_NO_MFMA_MHA=1 timeout 300 $BN > $O/b_hooklayout.json 2>> $O/b.err
T=kivi_amd/_variants/libkivi_tuning.so
for b in 8 16 24; do
  KIVI_HIP_LIB=$T KIVI_MF_ROW_NW8=0 timeout 300 $BN --batch $b --steps 10 --warmup 3 > $O/b${b}_nw4.json 2>> $O/b.err
  KIVI_HIP_LIB=$T KIVI_MF_ROW_NW8=1 timeout 300 $BN --batch $b --steps 10 --warmup 3 > $O/b${b}_nw8.json 2>> $O/b.err
done
for b in 4 2; do timeout 300 $BN --batch $b --steps 10 --warmup 3 > $O/b${b}.json 2>> $O/b.err; done
timeout 300 $BN --batch 1 --prompt 32752 --steps 10 --warmup 3 > $O/b1_32k.json 2>> $O/b.err
timeout 300 $BN --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2 > $O/c5.json 2>> $O/b.err
tail -n 2 $O/t.log $O/t2.log; cat $O/status.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3o/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"), "host", j.get("host_enqueue_ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
