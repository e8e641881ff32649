#!/bin/bash
# round 3, GPU session A: parity of the round-3 matrix-pipe kernels, then A/B timings against the round-2 kernels
O=gpurun_out/r3a; mkdir -p $O
export PYTHONUNBUFFERED=1
python -c "import torch; print(torch.cuda.get_device_name(0))" > $O/env.log 2>&1
timeout 900 python -m pytest tests/test_mfma_gpu.py -q -x -m gpu -k "scores or output" > $O/t1_gemv.log 2>&1; echo "t1 rc=$?" >> $O/status.log
timeout 900 python -m pytest tests/test_mfma_gpu.py -q -m gpu -k "not scores and not output" > $O/t2_decode.log 2>&1; echo "t2 rc=$?" >> $O/status.log
timeout 600 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "mf_row or small_batch or config4" > $O/t3_fullsize.log 2>&1; echo "t3 rc=$?" >> $O/status.log
# headline A/B on this box: new (matrix pipe, one launch) vs round-2 (hook layout, VALU)
timeout 300 python bench.py --steps 20 --warmup 5 > $O/b_new.json 2> $O/b_new.err; echo "b_new rc=$?" >> $O/status.log
KIVI_NO_MFMA_MHA=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b_old.json 2> $O/b_old.err; echo "b_old rc=$?" >> $O/status.log
for r in 22 43 23; do KIVI_MF_ROW_RINGS=$r timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hook-kgemv > $O/b_new_rings$r.json 2>> $O/b_new.err; done
KIVI_MF_NO_ROW=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hook-kgemv > $O/b_new_split.json 2>> $O/b_new.err
KIVI_MF_RING=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hook-kgemv > $O/b_new_kring2.json 2>> $O/b_new.err
# config 4 (Llama-3-8B attention shape): new vs round-2 matrix-pipe kernels
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3 --no-cpu-baseline"
timeout 300 python bench.py $C4 > $O/c4_new.json 2> $O/c4_new.err
KIVI_MF_OLD=1 timeout 300 python bench.py $C4 > $O/c4_old.json 2> $O/c4_old.err
KIVI_MF_RING=2 timeout 300 python bench.py $C4 > $O/c4_new_ring2.json 2>> $O/c4_new.err
# config-5 slice + small batches
timeout 300 python bench.py --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2 --no-cpu-baseline > $O/c5_new.json 2> $O/c5_new.err
KIVI_MF_OLD=1 timeout 300 python bench.py --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2 --no-cpu-baseline > $O/c5_old.json 2> $O/c5_old.err
for b in 16 8 64; do timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-hook-kgemv > $O/b${b}_new.json 2>> $O/b_new.err; done
timeout 300 python bench.py --batch 1 --prompt 32752 --steps 10 --warmup 3 --no-cpu-baseline --no-hook-kgemv > $O/b1_32k_new.json 2>> $O/b_new.err
tail -3 $O/t1_gemv.log $O/t2_decode.log $O/t3_fullsize.log; cat $O/status.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3a/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline") or {}; s=j.get("roofline_single_layer_kgemv") or {}; h=j.get("roofline_single_layer_kgemv_hook_layout") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"), "| kgemv", s.get("kernel"), s.get("median_launch_us"), s.get("frac"), "| hook", h.get("median_launch_us"), h.get("frac"), "| host", j.get("host_enqueue_ms_per_step"))
    except Exception as e:
        print(f, "ERR", e)
PY
