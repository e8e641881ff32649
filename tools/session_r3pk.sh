#!/bin/bash
# group statistics of the packed quantisers through v_pk_minimum3_f16 / v_pk_maximum3_f16 + tree-form code comparisons:
# bit-exactness (pack tests, golden fixtures, matrix-pipe layout tests), then the prefill kernels' times, old library vs new
O=gpurun_out/r3pk; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_pack_gpu.py tests/test_mfma_gpu.py -q -x -m gpu -k "pack or layout or relayout or equals or golden or exact" > $O/t.log 2>&1; echo "t rc=$?" >> $O/status.log
tail -n 3 $O/t.log
for lib in head new; do
  L=kivi_amd/libkivi_hip.so; [ $lib = head ] && L=kivi_amd/_variants/libkivi_head.so
  echo "== $lib" >> $O/times.log
  KIVI_HIP_LIB=$L timeout 300 python tools/pack_time.py 2>&1 | grep -v amdgpu.ids >> $O/times.log
  KIVI_HIP_LIB=$L timeout 300 python tools/mf_prefill_time.py 2>&1 | grep -v amdgpu.ids >> $O/times.log
done
cat $O/times.log
python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from kivi_amd.quant import new_pack
x = torch.randn((1, 1, 4, 64), dtype=torch.float16, device="cuda")
x[0, 0, 0, 3] = float("nan"); x[0, 0, 1, 40] = -float("nan"); x[0, 0, 2, 5] = 0.0; x[0, 0, 2, 6] = -0.0
c, s, m = new_pack.triton_quantize_and_pack_along_last_dim(x, 32, 2)
print("NaN-input groups: scale", s.flatten().tolist(), "mn", m.flatten().tolist(), "codes of the NaN groups", c[0, 0, 0, :2].tolist(), c[0, 0, 1, 2:].tolist())
PY
cat $O/status.log
