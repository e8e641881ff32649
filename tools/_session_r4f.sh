R=$PWD; O=$R/gpurun_out/r4f; mkdir -p $O; export PYTHONUNBUFFERED=1
T=$R/kivi_amd/_variants/libkivi_tuning.so
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3"
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], j["value"], "tok/s", j["ms_per_step"], "ms", r.get("kernel"), r.get("median_launch_us"), "us frac", r.get("frac"), "host", j.get("host_enqueue_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for i in 1 2; do
  for cfg in 443 436 236 1436 1236 1226 1238 1228; do
    KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4=$cfg timeout 300 $BN $C4 > $O/c4_${cfg}_$i.json 2>> $O/err.log; line $O/c4_${cfg}_$i.json
  done
done
for cfg in 1236 1238; do
  KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4=$cfg timeout 600 python -m pytest tests/test_mfma_gpu.py tests/test_hook_gpu.py -m gpu -x -q -k "(row and fixtures) or matches_two_launch" > $O/parity_$cfg.log 2>&1; echo "parity $cfg rc=$?"; tail -3 $O/parity_$cfg.log
done
tail -3 $O/err.log
