#!/bin/bash
# The reference's end-to-end recipe (mem_spd_test.py:8-12, :53-70) on random-weight models: Llama-2-7B shape (MHA) and
# Mistral-7B shape (32 / 8 heads), KIVI hook vs fp16 KV cache.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
O=$R/gpurun_out/r02_e2e; mkdir -p $O; cd $R
run() { timeout 900 python examples/mem_spd_test.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
{
echo "# llama-2-7b shape, reference recipe (B=96, prompt 160, gen 338, R=128, 3 repeats)"
run --recipe
run --recipe --baseline
echo "# mistral-7b shape (32 query heads / 8 kv heads, ffn 14336), same recipe"
run --recipe --kv-heads 8 --intermediate 14336
run --recipe --kv-heads 8 --intermediate 14336 --baseline
echo "# llama-3-8b-like GQA, long prompt: B=32, prompt 8192, gen 128, R=128"
run --batch 32 --prompt 8192 --gen 128 --residual 128 --kv-heads 8 --intermediate 14336
run --batch 32 --prompt 8192 --gen 128 --residual 128 --kv-heads 8 --intermediate 14336 --baseline
} | tee $O/e2e.log
