#!/bin/bash
# config 4 (B=64, 32/8 heads, 8k, R=128): phase timeline of mf_row4_kernel and K-ring depth variants; bench sanity at the headline
O=gpurun_out/r3q; mkdir -p $O
export PYTHONUNBUFFERED=1
T=kivi_amd/_variants/libkivi_tuning.so
BN="python bench.py --no-cpu-baseline"
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 6 --warmup 2"
KIVI_HIP_LIB=$T B=64 NHKV=8 T0=8064 R=128 LAYERS=6 timeout 300 python tools/mf_row_phases.py > $O/row4_phases.log 2>&1
for cfg in 443 423 483 484; do
  KIVI_HIP_LIB=$T KIVI_MF_ROW4=$cfg timeout 300 $BN --no-hook-kgemv $C4 > $O/c4_row4_$cfg.json 2>> $O/b.err
done
timeout 300 $BN $C4 --no-hook-kgemv > $O/c4_product.json 2>> $O/b.err
timeout 300 $BN > $O/b_headline.json 2>> $O/b.err
timeout 300 $BN --batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2 --no-hook-kgemv > $O/c5.json 2>> $O/b.err
head -40 $O/row4_phases.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3q/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
        r=j.get("roofline") or {}; s=j.get("roofline_single_layer_kgemv") or {}; m=j.get("roofline_single_layer_kgemv_mf_layout") or {}
        print(f.split("/")[-1], j["value"], j["ms_per_step"], r.get("kernel"), r.get("median_launch_us"), r.get("frac"), "| kgemv", s.get("kernel"), s.get("median_launch_us"), s.get("frac"), "| mf", m.get("median_launch_us"), m.get("frac"), "flush", j["config"].get("k_flush_launch_us_per_layer"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 $O/b.err
