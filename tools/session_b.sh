#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02b; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 python -m pytest tests/test_mfma_gpu.py -m gpu -q > $O/pytest_mfma.log 2>&1 ); tail -25 $O/pytest_mfma.log
( cd $R && timeout 300 python tools/gqa_time.py > $O/gqa_time.log 2>&1; KIVI_GQA_NO_HILO=1 timeout 300 python tools/gqa_time.py >> $O/gqa_time.log 2>&1; timeout 300 python tools/gqa_time.py --batch 16 --tokens 32768 >> $O/gqa_time.log 2>&1; timeout 300 python tools/gqa_time.py --batch 1 --tokens 32768 >> $O/gqa_time.log 2>&1 ); grep -v amdgpu.ids $O/gqa_time.log
