#!/bin/bash
# where do the 57 us of the R = 4 qK^T launch go?  (the same bytes stream in ~37 us inside mf_row4_kernel)
O=gpurun_out/r3x; mkdir -p $O
export PYTHONUNBUFFERED=1
C4="--batch 64 --heads 32 --kv-heads 8 --tokens 8064 --residual 128 --layers 6"
for cfg in "KIVI_MF_NO_ROW=1" "KIVI_MF_NO_ROW=1 KIVI_MF_K_DIAG=1" "KIVI_MF_NO_ROW=1 KIVI_MF_K_DIAG=2" "KIVI_MF_NO_ROW=1 KIVI_MF_K_DIAG=1 KIVI_MF_SPW=1" "KIVI_MF_NO_ROW=1 KIVI_MF_K_DIAG=1 KIVI_MF_SPW=2"; do
  echo "== c4 $cfg" >> $O/steps.log
  env $cfg timeout 300 python tools/gqa_step_time.py $C4 2>&1 | grep mf_k >> $O/steps.log
done
cat $O/steps.log
