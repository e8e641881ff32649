#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02c; mkdir -p $O; cd $R
timeout 300 python tools/gqa_debug.py > $O/debug.log 2>&1; cat $O/debug.log
