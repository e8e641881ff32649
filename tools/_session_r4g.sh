R=$PWD; O=$R/gpurun_out/r4g; mkdir -p $O; export PYTHONUNBUFFERED=1
T=$R/kivi_amd/_variants/libkivi_tuning.so
BN="python bench.py --no-cpu-baseline --no-hook-kgemv"
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j.get("roofline") or {}; m = j.get("roofline_single_layer_kgemv_mf_layout") or {}
    print(sys.argv[1].split("/")[-1], j["value"], "tok/s", j["ms_per_step"], "ms", r.get("kernel"), r.get("median_launch_us"), "us frac", r.get("frac"), "host", j.get("host_enqueue_ms_per_step"), "| kgemv mf", m.get("median_launch_us"), m.get("frac"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
for b in 1 2 4 8; do
  for deep in 1 0; do
    KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW_DEEP=$deep timeout 300 $BN --batch $b --steps 20 --warmup 6 > $O/b${b}_deep$deep.json 2>> $O/err.log; line $O/b${b}_deep$deep.json
  done
done
KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW_DEEP=1 timeout 300 $BN --batch 16 --steps 20 --warmup 6 > $O/b16_deep1.json 2>> $O/err.log; line $O/b16_deep1.json
KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW_DEEP=0 timeout 300 $BN --batch 16 --steps 20 --warmup 6 > $O/b16_deep0.json 2>> $O/err.log; line $O/b16_deep0.json
timeout 300 $BN > $O/b32.json 2>> $O/err.log; line $O/b32.json
timeout 300 $BN --batch 64 > $O/b64.json 2>> $O/err.log; line $O/b64.json
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
