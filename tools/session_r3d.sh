#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu -k "pack or relayout or scores or output" > $O/t1_gemv.log 2>&1; echo "t1 rc=$?" >> $O/status.log
timeout 900 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu -k "not pack and not relayout and not scores and not output" > $O/t2_decode.log 2>&1; echo "t2 rc=$?" >> $O/status.log
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -x -m gpu -k "mf_row or small_batch" > $O/t3_fullsize.log 2>&1; echo "t3 rc=$?" >> $O/status.log
T="python tools/gqa_time.py --batch 32 --heads 32 --kv-heads 32 --tokens 4096 --nbuf 8 --iters 5"
$T > $O/k_c2.log 2>&1
KIVI_MF_SPW=1 $T > $O/k_c2_spw1.log 2>&1
KIVI_MF_SPW=4 $T > $O/k_c2_spw4.log 2>&1
KIVI_MF_RING=2 $T > $O/k_c2_ring2.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b_new.json 2> $O/b_new.err; echo "b_new rc=$?" >> $O/status.log
for r in 22 43 23; do KIVI_MF_ROW_RINGS=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hook-kgemv > $O/b_new_rings$r.json 2>> $O/b_new.err; done
KIVI_MF_NO_ROW=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hook-kgemv > $O/b_new_split.json 2>> $O/b_new.err
KIVI_NO_MFMA_MHA=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b_old.json 2> $O/b_old.err
for b in 16 8 64; do timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-hook-kgemv > $O/b${b}_new.json 2>> $O/b_new.err; done
tail -n 3 $O/t1_gemv.log $O/t2_decode.log $O/t3_fullsize.log; cat $O/status.log; grep -h "mfma qK" $O/k_c2*.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3f/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline") or {}; s=j.get("roofline_single_layer_kgemv") or {}; h=j.get("roofline_single_layer_kgemv_hook_layout") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"), "| kgemv", s.get("kernel"), s.get("median_launch_us"), s.get("frac"), "| hook", h.get("median_launch_us"), h.get("frac"), "| host", j.get("host_enqueue_ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
