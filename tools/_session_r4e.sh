R=$PWD; O=$R/gpurun_out/r4e; mkdir -p $O; export PYTHONUNBUFFERED=1
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
for i in 1 2 3; do
  KIVI_TUNING=1 KIVI_HIP_LIB=$T timeout 300 $BN $C4 > $O/c4_wsm_$i.json 2>> $O/err.log; line $O/c4_wsm_$i.json
  KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4=2443 timeout 300 $BN $C4 > $O/c4_blocksm_$i.json 2>> $O/err.log; line $O/c4_blocksm_$i.json
done
KIVI_TUNING=1 KIVI_HIP_LIB=$T B=64 NHKV=8 T0=8064 R=128 LAYERS=6 timeout 300 python tools/mf_row_phases.py > $O/row4_phases.log 2>&1; sed -n 2,14p $O/row4_phases.log
timeout 900 python -m pytest tests/test_mfma_gpu.py tests/test_hook_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q \
   -k "(row and (fixtures or mf_decode_steps or dynamic_range)) or matches_two_launch or config4" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
