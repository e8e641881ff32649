#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r3c; mkdir -p $O
T="python $R/tools/gqa_time.py --batch 32 --heads 32 --kv-heads 32 --tokens 4096 --nbuf 8 --iters 5"
for d in 0 6 2; do KIVI_MF_DIAG=$d $T > $O/abl_${d}.log 2>&1; done
T2="python $R/tools/gqa_time.py --batch 64 --heads 32 --kv-heads 32 --tokens 4096 --nbuf 6 --iters 5"
for d in 0 6 2; do KIVI_MF_DIAG=$d $T2 > $O/abl_b64_${d}.log 2>&1; done
grep -h "qK" $O/abl_*.log
