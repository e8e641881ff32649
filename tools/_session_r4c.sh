R=$PWD; O=$R/gpurun_out/r4c; mkdir -p $O; export PYTHONUNBUFFERED=1
T=$R/kivi_amd/_variants/libkivi_tuning.so
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], j["value"], "tok/s", j["ms_per_step"], "ms", r.get("kernel"), r.get("median_launch_us"), "us frac", r.get("frac"), "host", j.get("host_enqueue_ms_per_step"), j.get("hipgraph"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_mfma_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
for sh in "b1_4k:--batch 1" "b4_4k:--batch 4" "b1_32k:--batch 1 --prompt 32752" "b32:" "c5:--batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128" "b16_70b:--batch 16 --heads 64 --kv-heads 8 --prompt 8064 --residual 128"; do
  n=${sh%%:*}; a=${sh#*:}
  timeout 300 $BN $a --steps 20 --warmup 6 > $O/eager_$n.json 2>> $O/err.log; line $O/eager_$n.json
  timeout 300 $BN $a --steps 20 --warmup 6 --graph > $O/graph_$n.json 2>> $O/err.log; line $O/graph_$n.json
done
C70B="--batch 64 --heads 64 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3"
for i in 1 2; do
  KIVI_TUNING=1 KIVI_HIP_LIB=$T timeout 300 $BN $C70B > $O/r8_ring4_$i.json 2>> $O/err.log; line $O/r8_ring4_$i.json
  KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_VRING=2 timeout 300 $BN $C70B > $O/r8_ring2_$i.json 2>> $O/err.log; line $O/r8_ring2_$i.json
done
C5="--batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2"
for i in 1 2; do
  KIVI_TUNING=1 KIVI_HIP_LIB=$T timeout 300 $BN $C5 > $O/c5_base_$i.json 2>> $O/err.log; line $O/c5_base_$i.json
  KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_VHL=4 timeout 300 $BN $C5 > $O/c5_vhl4_$i.json 2>> $O/err.log; line $O/c5_vhl4_$i.json
  KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_VHL=2 timeout 300 $BN $C5 > $O/c5_vhl2_$i.json 2>> $O/err.log; line $O/c5_vhl2_$i.json
done
KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_VHL=4 timeout 600 python -m pytest tests/test_mfma_gpu.py tests/test_hook_gpu.py -m gpu -x -q -k "(split and fixtures) or (mf_decode_steps and split)" > $O/vhl_parity.log 2>&1; echo "vhl parity rc=$?"; tail -3 $O/vhl_parity.log
timeout 900 python examples/mem_spd_test.py --graphs > $O/e2e_graphs.log 2>&1; echo "e2e rc=$?"; tail -3 $O/e2e_graphs.log
timeout 900 python examples/mem_spd_test.py > $O/e2e_eager.log 2>&1; echo "e2e eager rc=$?"; tail -2 $O/e2e_eager.log
tail -5 $O/err.log
