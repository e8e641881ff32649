R=$PWD; O=$R/gpurun_out/r4h; mkdir -p $O; export PYTHONUNBUFFERED=1
T=$R/kivi_amd/_variants/libkivi_tuning.so
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], j["value"], "tok/s", j["ms_per_step"], "ms", r.get("kernel"), r.get("median_launch_us"), "us frac", r.get("frac"), "host", j.get("host_enqueue_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
timeout 900 python -m pytest tests/test_mfma_gpu.py tests/test_hook_gpu.py tests/test_graph_gpu.py -m gpu -x -q -k "(row and (fixtures or mf_decode_steps or dynamic_range)) or matches_two_launch or dyn" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for sh in "b16:--batch 16" "b64:--batch 64"; do
  n=${sh%%:*}; a=${sh#*:}
  for i in 1 2; do
    KIVI_TUNING=1 KIVI_HIP_LIB=$T timeout 300 $BN $a --heads 64 --kv-heads 8 --prompt 4000 --residual 128 --steps 10 --warmup 3 > $O/r8_4k_${n}_row_$i.json 2>> $O/err.log; line $O/r8_4k_${n}_row_$i.json
    KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_NO_ROW=1 timeout 300 $BN $a --heads 64 --kv-heads 8 --prompt 4000 --residual 128 --steps 10 --warmup 3 > $O/r8_4k_${n}_split_$i.json 2>> $O/err.log; line $O/r8_4k_${n}_split_$i.json
  done
done
tail -3 $O/err.log
