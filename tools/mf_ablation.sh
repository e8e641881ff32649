#!/bin/bash
# Where does the time of the round-3 matrix-pipe qK^T go?  (1) MFMA issue rate with subnormal / normal B operands,
# (2) DIAG builds of mf_k_kernel at BASELINE configs[1], (3) SQ counters of the same launch.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3b; mkdir -p $O
export PYTHONUNBUFFERED=1
$R/tools/mfma_rate_probe.bin > $O/mfma_rate.log 2>&1
T="python $R/tools/gqa_time.py --batch 32 --heads 32 --kv-heads 32 --tokens 4096 --nbuf 8 --iters 5"
$T > $O/abl_0_full.log 2>&1
for d in 1 2 3 4 5; do KIVI_MF_DIAG=$d $T > $O/abl_${d}.log 2>&1; done
KIVI_MF_RING=2 $T > $O/abl_ring2.log 2>&1
grep -h "mfma qK" $O/abl_*.log | paste <(ls $O/abl_*.log | xargs -n1 basename) - 
cd /tmp && export TMPDIR=/tmp
rm -rf $O/sq
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/sq -o p -- $T > $O/pmc_run.log 2>&1
rm -rf $O/sq2
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/sq2 -o p -- $T > $O/pmc_run2.log 2>&1
for dd in sq sq2; do
f=$(find $O/$dd -name "*counter_collection.csv" | head -1)
python - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "mf_k_kernel" not in k and "gemv_k_kernel" not in k:
        continue
    agg[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    med = {c: sorted(v)[len(v) // 2] for c, v in d.items()}
    wc = med.get("SQ_WAVE_CYCLES", 1)
    for c, v in med.items():
        print(f"   {c:28s} {v:14.0f}   {v / wc:6.3f} of WAVE_CYCLES")
PY
done > $O/pmc_summary.log 2>&1
cat $O/mfma_rate.log $O/pmc_summary.log
