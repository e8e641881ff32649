#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3e; mkdir -p $O
export PYTHONUNBUFFERED=1
T="python $R/tools/gqa_time.py --batch 32 --heads 32 --kv-heads 32 --tokens 4096 --nbuf 8 --iters 5"
$T > $O/k_base.log 2>&1
KIVI_HIP_LIB=$R/kivi_amd/_variants/libkivi_dense.so $T > $O/k_dense.log 2>&1
KIVI_HIP_LIB=$R/kivi_amd/_variants/libkivi_nosm.so $T > $O/k_nosm.log 2>&1
python $R/tools/gqa_time.py --batch 32 --heads 32 --kv-heads 32 --tokens 4096 --nbuf 1 --iters 20 > $O/k_mall.log 2>&1
python $R/tools/gqa_time.py --batch 4 --heads 32 --kv-heads 32 --tokens 4096 --nbuf 1 --iters 20 > $O/k_l2.log 2>&1
grep -h "qK" $O/k_*.log | paste <(ls $O/k_*.log | xargs -n1 basename | sed 's/$/\n/' | tr '\n' '\n') - 2>/dev/null
for f in $O/k_*.log; do echo "== $f"; cat $f; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq -o p -- $T > $O/pmc_run.log 2>&1
rm -rf $O/sq2
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/sq2 -o p -- $T > $O/pmc_run2.log 2>&1
rm -rf $O/tc
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $O/tc -o p -- $T > $O/pmc_run3.log 2>&1
for dd in sq sq2 tc; do
f=$(find $O/$dd -name "*counter_collection.csv" | head -1)
python - $f <<'PY'
import csv, sys, collections
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no csv", e); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "mf_k_kernel" not in k and "gemv_k_kernel" not in k:
        continue
    agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    for c, v in med.items():
        print(f"   {c:28s} {v:14.0f}")
PY
done > $O/pmc_summary.log 2>&1
cat $O/pmc_summary.log
