#!/bin/bash
# ONE parameterised GPU session script (replaces the per-experiment session_*.sh of rounds 1-3).  Run through gpurun from the
# repo root, e.g.   gpurun --timeout 900 -- 'bash tools/session.sh tests bench shapes'
# Every stage writes under gpurun_out/<tag>/ (tag = $SESSION_TAG, default "s"); copy what is to be judged into profiles/.
#
#   tests            pytest -m gpu (+ smoke)                                  -> pytest_gpu.log
#   bench            the driver's command (default workload)                  -> bench.json
#   shapes           the other shapes of DESIGN section 5 through bench.py    -> shape_*.json + a one-line summary each
#   trace            rocprofv3 --kernel-trace --stats of the bench command + per-kernel medians (tools/trace_median.py)
#   trace_c4         the same for BASELINE config 4 only
#   trace_gqa        the same for BASELINE config 4, the config-5 per-GPU slice and the 70B-like (64 / 8 heads) slice
#   pmc              FETCH_SIZE / WRITE_SIZE passes (separate, kernel trace only) of the bench command + calibration
#   pmc_c4           the same at BASELINE config 4
#   e2e              examples/mem_spd_test.py --recipe and the configs[2] decode loop
#   ab <lib.so>...   same-box A/B of bench.py (headline + config 4) between the in-tree library and the given builds
#                    (tools/build_variant.sh; KIVI_TUNING=1 KIVI_HIP_LIB=...)
#   phases           per-wave phase timeline of mf_row_kernel / mf_row4_kernel (tuning build, tools/mf_row_phases.py)
#   row4ab           mf_row4_kernel ring variants of the in-stream flow at BASELINE config 4 (tuning build: KIVI_MF_ROW4 = <K ring><V ring><waves>,
#                    + 1000: chained hi / lo sV; KIVI_MF_ROW4_FLOW=stream): parity of the variants through the row-form tests, then same-box bench lines
#   xcd              two-launch form with a unit's blocks on one XCD (tuning build, KIVI_MF_XCD) against the plain block order
#   mf4              4-bit K / V on the matrix pipe: parity tests, then config 4 at --bits 4 against the VALU path
#   mf4prof          config 4 at --bits 4: kernel trace medians + HBM traffic; the config-5 slice at --bits 4 against the VALU path
#   mf4pmc           ... its first half only (trace medians + HBM traffic at config 4, --bits 4)
#   forms            round 5: the library's launch plan against forced forms (two launches / a block per row / N slices per row) at BASELINE
#                    config 4, the config-5 slice, the 70B-like slice, R = 8 at B = 64 and small grouped-query batches (bench.py --form)
#   flows            round 5: phase-softmax vs in-stream flow of mf_row4_kernel (and the round-4 tree from a worktree _r4/, if present) at BASELINE
#                    config 4 + headline, phase timelines
#   packs            kt_pack / vt_pack at 2 and 4 bits on 1 GiB of fp16
#   shapes_g         round 6: g = 64 / 128, D = 64, nh / nh_kv = 2 (hook-state layout, VALU kernels) through bench.py
#   fuzz             round 6: tools/fuzz_decode.py for FUZZ_SECONDS (600) with FUZZ_SEED (1)
#   sq6              round 6: SQ counters of mf_row4_kernel at config 4: four-wave product, six-wave block, three blocks per CU (B=96 x 6k)
#   sq <name> <args> SQ counters (wave cycles, VALU / MFMA instructions and busy cycles, waits) of one bench command
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
TAG=${SESSION_TAG:-s}
O=$R/gpurun_out/$TAG; mkdir -p $O
export PYTHONUNBUFFERED=1
BN="python $R/bench.py --no-cpu-baseline --no-hook-kgemv --no-extra-configs"
C4="--batch 64 --heads 32 --kv-heads 8 --prompt 8064 --residual 128"
C5="--batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128"
C70="--batch 16 --heads 64 --kv-heads 8 --prompt 8064 --residual 128"

line() {  # one-line summary of a bench JSON line
python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], j["value"], "tok/s", j["ms_per_step"], "ms", r.get("kernel"), r.get("median_launch_us"), "us frac", r.get("frac"), "host", j.get("host_enqueue_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}

trace_one() {  # <name> <skip> <bench args...>
    local name=$1 skip=$2; shift 2
    cd /tmp && export TMPDIR=/tmp
    rm -rf $O/trace_$name
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o b -- $BN "$@" > $O/trace_${name}_bench.json 2> $O/trace_$name.err
    cd $R
    python tools/trace_median.py $(find $O/trace_$name -name "*kernel_trace.csv" | head -1) --skip $skip --skip-for gemv_k_kernel=12 quant_pack=0 mf_row4_kernel=${ROW4_SKIP:-$skip} --match mf_ decode_row gemv_ kt_pack vt_pack quant_pack \
        --json $O/trace_median_$name.json > $O/trace_median_$name.log 2>&1
    cp $(find $O/trace_$name -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$name.csv 2>/dev/null
    rm -rf $O/trace_$name
    head -12 $O/trace_median_$name.log; line $O/trace_${name}_bench.json
}

pmc_one() {  # <name> <config json> <bench args...>
    local name=$1 cfg=$2; shift 2
    cd /tmp && export TMPDIR=/tmp
    rm -rf $O/pmc_f_$name $O/pmc_w_$name
    timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f_$name -o b -- $BN --steps 3 --warmup 1 --no-kernel-events "$@" > /dev/null 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w_$name -o b -- $BN --steps 3 --warmup 1 --no-kernel-events "$@" > /dev/null 2>&1
    [ -f $O/pmc_calib_done ] || { timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_calib -o c -- $R/tools/hbm_read_bw.bin 2 > $O/hbm_bw.log 2>&1; touch $O/pmc_calib_done; }
    cd $R
    python tools/pmc_traffic.py $(find $O/pmc_f_$name -name "*counter_collection.csv" | head -1) $(find $O/pmc_w_$name -name "*counter_collection.csv" | head -1) \
        $(find $O/pmc_calib -name "*counter_collection.csv" | head -1) --skip 32 --config "$cfg" --out $O/pmc_traffic_$name.json > $O/pmc_traffic_$name.log 2>&1
    find $O/pmc_f_$name $O/pmc_w_$name -name "*.csv" -size +4M -delete
    cat $O/pmc_traffic_$name.log
}

cd $R
while [ $# -gt 0 ]; do
    stage=$1; shift
    case $stage in
    r6new)
        # round 6: the new parity evidence first -- every-unit fp64 reference, the timeout path, the graph-vs-eager bands, LongChat shapes
        rm -f $R/gpurun_out/gemv_ratios.log
        timeout 1500 python -m pytest tests/test_fullcover_gpu.py tests/test_timeout_gpu.py tests/test_graph_gpu.py tests/test_fullsize_gpu.py tests/test_mfma4_gpu.py -m gpu -q -s --tb=short --maxfail=12 \
            -k "fullcover or timeout or graph or longchat or small_batch or mfma4" > $O/r6new.log 2>&1; echo "r6new rc=$?" | tee -a $O/status.log
        grep -E "passed|failed|worst ratio|Error|error" $O/r6new.log | tail -40 | cut -c1-300
        cp $R/gpurun_out/gemv_ratios.log $O/gemv_ratios_r6new.log 2>/dev/null ;;
    row6)
        # round 6: mf_row4_kernel with SIX waves per block (three per SIMD; tuning build: KIVI_MF_ROW4_NW=6, rings KIVI_MF_ROW4_6=<K ring><V ring>) against
        # the product's four-wave block, BASELINE config 4 + shorter rows; 4-bit codes likewise (KIVI_MF_ROW4_46); one box.  (Session r6c ran this stage
        # with the six-wave block as the product default and KIVI_MF_ROW4_NW=4 as the A/B side: profiles/r06_six_wave.log.)
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        for cfg in 43 23 22; do
            KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_NW=6 KIVI_MF_ROW4_6=$cfg timeout 900 python -m pytest tests/test_mfma_gpu.py tests/test_hook_gpu.py -m gpu -x -q \
                -k "(row and fixtures) or matches_two_launch or (decode_steps_match and row)" > $O/row6_parity_$cfg.log 2>&1; echo "row6 parity $cfg rc=$?" | tee -a $O/status.log; tail -2 $O/row6_parity_$cfg.log
        done
        for i in 1 2 3; do
            timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/row6_c4_product_$i.json 2>> $O/row6.err; line $O/row6_c4_product_$i.json
            for cfg in 43 23 42 22; do
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_NW=6 KIVI_MF_ROW4_6=$cfg timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/row6_c4_nw6_r${cfg}_$i.json 2>> $O/row6.err; line $O/row6_c4_nw6_r${cfg}_$i.json
            done
        done
        for i in 1 2; do      # 4-bit codes: the four-wave product block against six-wave variants (these spill 36-52 bytes)
            timeout 300 $BN $C4 --bits 4 --steps 10 --warmup 3 > $O/row6_c4b4_product_$i.json 2>> $O/row6.err; line $O/row6_c4b4_product_$i.json
            for cfg in 22 23; do
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_46=$cfg timeout 300 $BN $C4 --bits 4 --steps 10 --warmup 3 > $O/row6_c4b4_r${cfg}_$i.json 2>> $O/row6.err; line $O/row6_c4b4_r${cfg}_$i.json
            done
        done
        for sh in "32 8064" "64 4032" "128 2048"; do
            lb=${sh% *}; lt=${sh#* }
            timeout 300 $BN --batch $lb --heads 32 --kv-heads 8 --prompt $lt --residual 128 --steps 10 --warmup 3 > $O/row6_b${lb}_t${lt}_product.json 2>> $O/row6.err; line $O/row6_b${lb}_t${lt}_product.json
            KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_NW=6 timeout 300 $BN --batch $lb --heads 32 --kv-heads 8 --prompt $lt --residual 128 --steps 10 --warmup 3 > $O/row6_b${lb}_t${lt}_nw6.json 2>> $O/row6.err; line $O/row6_b${lb}_t${lt}_nw6.json
        done ;;
    compat)
        # round 6: the literal pybind twin (kivi_gemv.gemv_forward_cuda_outer_dim on the reference's kernel-input layout) at C2, both widths
        timeout 600 python -m pytest tests/test_gemv_gpu.py -m gpu -q -k "compat" > $O/compat_tests.log 2>&1; echo "compat tests rc=$?" | tee -a $O/status.log; tail -3 $O/compat_tests.log
        timeout 600 python tools/compat_time.py > $O/compat_time.log 2>&1; BITS=4 timeout 600 python tools/compat_time.py >> $O/compat_time.log 2>&1; cat $O/compat_time.log
        # the general kernel (one wave per packed row: what rounds 1-5 shipped) on the same box: tuning build, KIVI_COMPAT_OLD=1
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_COMPAT_CH=16 timeout 600 python tools/compat_time.py > $O/compat_time_ch16.log 2>&1; KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_COMPAT_CH=16 BITS=4 timeout 600 python tools/compat_time.py >> $O/compat_time_ch16.log 2>&1; grep "gemv_outer_dim" $O/compat_time_ch16.log
        KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_COMPAT_CH=16 timeout 300 python -m pytest tests/test_gemv_gpu.py -m gpu -q -k "compat" > $O/compat_tests_ch16.log 2>&1; echo "compat ch16 tests rc=$?" | tee -a $O/status.log
        KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_COMPAT_NO_ROWS=1 timeout 600 python tools/compat_time.py > $O/compat_time_wide.log 2>&1; KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_COMPAT_NO_ROWS=1 BITS=4 timeout 600 python tools/compat_time.py >> $O/compat_time_wide.log 2>&1; grep "gemv_outer_dim" $O/compat_time_wide.log
        KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_COMPAT_OLD=1 timeout 600 python tools/compat_time.py > $O/compat_time_old.log 2>&1; KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_COMPAT_OLD=1 BITS=4 timeout 600 python tools/compat_time.py >> $O/compat_time_old.log 2>&1; grep "gemv_outer_dim" $O/compat_time_old.log ;;
    occ3)
        # round 6: four-wave blocks of mf_row4_kernel compiled for THREE waves per SIMD (167 registers, rings 2 / 2; tuning build, KIVI_MF_ROW4_OCC3=1) where the
        # LDS allows three blocks per CU (rows of <= 6.5k keys) against the product, one box
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        for i in 1 2; do
            for sh in "64 4032" "128 2048" "96 6016" "64 8064"; do
                lb=${sh% *}; lt=${sh#* }
                timeout 300 $BN --batch $lb --heads 32 --kv-heads 8 --prompt $lt --residual 128 --steps 10 --warmup 3 > $O/occ3_b${lb}_t${lt}_product_$i.json 2>> $O/occ3.err; line $O/occ3_b${lb}_t${lt}_product_$i.json
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_OCC3=1 timeout 300 $BN --batch $lb --heads 32 --kv-heads 8 --prompt $lt --residual 128 --steps 10 --warmup 3 > $O/occ3_b${lb}_t${lt}_occ3_$i.json 2>> $O/occ3.err; line $O/occ3_b${lb}_t${lt}_occ3_$i.json
            done
        done ;;
    confirm)
        # round 6, last session: the library after the hand-off sizing fix (nh / nh_kv = 8 back at two blocks per CU): the ratio-8 shapes, one launch vs two
        timeout 300 $BN $C70 --steps 10 --warmup 3 > $O/confirm_70b_slice.json 2>> $O/confirm.err; line $O/confirm_70b_slice.json
        timeout 300 $BN $C70 --steps 10 --warmup 3 --form split > $O/confirm_70b_slice_split.json 2>> $O/confirm.err; line $O/confirm_70b_slice_split.json
        timeout 300 $BN --batch 64 --heads 64 --kv-heads 8 --prompt 8064 --residual 128 --steps 6 --warmup 2 > $O/confirm_r8_b64_8k.json 2>> $O/confirm.err; line $O/confirm_r8_b64_8k.json
        timeout 300 $BN --batch 64 --heads 64 --kv-heads 8 --prompt 4000 --residual 32 --steps 10 --warmup 3 > $O/confirm_r8_b64_4k.json 2>> $O/confirm.err; line $O/confirm_r8_b64_4k.json
        timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/confirm_config4.json 2>> $O/confirm.err; line $O/confirm_config4.json
        timeout 300 $BN $C5 --steps 6 --warmup 2 > $O/confirm_config5_slice.json 2>> $O/confirm.err; line $O/confirm_config5_slice.json
        cd /tmp && export TMPDIR=/tmp
        timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_calib -o c -- $R/tools/hbm_read_bw.bin 2 > $O/hbm_bw.log 2>&1; tail -4 $O/hbm_bw.log
        python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/pmc_calib/**/*counter_collection.csv", recursive=True)
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == "FETCH_SIZE"] if f else []
print("calibration: FETCH_SIZE units per launch", v[:4], "-> bytes per unit for 2 GiB of reads:", [round(2 * 2**30 / x, 2) for x in v[:4] if x])
PY
        cd $R ;;
    socc3)
        # round 6: the config-5 slice with MORE, shorter slices whose blocks fit three per CU, compiled for three waves per SIMD (in-stream flow: 168 registers,
        # 68 bytes spilled; tuning build, KIVI_MF_ROW4_SOCC3=1) against the plan (4 slices, two blocks per CU), one box
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        for i in 1 2; do
            timeout 300 $BN $C5 --steps 6 --warmup 2 > $O/socc3_c5_plan_$i.json 2>> $O/socc3.err; line $O/socc3_c5_plan_$i.json
            for sl in 6 8; do
                timeout 300 $BN $C5 --steps 6 --warmup 2 --form slices$sl > $O/socc3_c5_s${sl}_$i.json 2>> $O/socc3.err; line $O/socc3_c5_s${sl}_$i.json
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_SOCC3=1 timeout 300 $BN $C5 --steps 6 --warmup 2 --form slices$sl > $O/socc3_c5_s${sl}_occ3_$i.json 2>> $O/socc3.err; line $O/socc3_c5_s${sl}_occ3_$i.json
            done
        done ;;
    tickets)
        # round 6: eight ticket counters (blockIdx % 8; product) against ONE (tuning build, KIVI_MF_ONE_TICKET=1) for the sliced one-launch forms, one box
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        for i in 1 2 3; do
            for sh in "70b $C70 --steps 10 --warmup 3" "c5 $C5 --steps 6 --warmup 2" "b4_8k --batch 4 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3"; do
                n=${sh%% *}; a=${sh#* }
                timeout 300 $BN $a > $O/tickets_${n}_eight_$i.json 2>> $O/tickets.err; line $O/tickets_${n}_eight_$i.json
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ONE_TICKET=1 timeout 300 $BN $a > $O/tickets_${n}_one_$i.json 2>> $O/tickets.err; line $O/tickets_${n}_one_$i.json
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_NO_TICKET=1 timeout 300 $BN $a > $O/tickets_${n}_none_$i.json 2>> $O/tickets.err; line $O/tickets_${n}_none_$i.json
            done
        done ;;
    mf41)
        # round 6: 4-bit multi-head K / V on the matrix pipe (nh == nh_kv) against the VALU kernels of the hook-state layout, one box, alternating:
        # C2 at 4 bits (B = 8 / 32 / 64), LongChat-7B-32K + KIVI-4 rows (B = 8 x 16k / 32k)
        for i in 1 2; do
            for b in 32 8 64; do
                timeout 300 $BN --bits 4 --batch $b --steps 10 --warmup 3 > $O/mf41_c2_b${b}_mf_$i.json 2>> $O/mf41.err; line $O/mf41_c2_b${b}_mf_$i.json
                KIVI_TUNING=1 KIVI_NO_MFMA_MHA=1 timeout 300 $BN --bits 4 --batch $b --steps 10 --warmup 3 > $O/mf41_c2_b${b}_valu_$i.json 2>> $O/mf41.err; line $O/mf41_c2_b${b}_valu_$i.json
            done
            for t in 16256 32640; do
                timeout 300 $BN --bits 4 --batch 8 --prompt $t --residual 128 --steps 6 --warmup 2 > $O/mf41_b8_t${t}_mf_$i.json 2>> $O/mf41.err; line $O/mf41_b8_t${t}_mf_$i.json
                KIVI_TUNING=1 KIVI_NO_MFMA_MHA=1 timeout 300 $BN --bits 4 --batch 8 --prompt $t --residual 128 --steps 6 --warmup 2 > $O/mf41_b8_t${t}_valu_$i.json 2>> $O/mf41.err; line $O/mf41_b8_t${t}_valu_$i.json
            done
        done ;;
    long1)
        # round 6: multi-head rows beyond 8192 keys (LongChat-7B-32K shape): the plan's one-launch sliced form against two launches, one box
        for i in 1 2; do
            for f in auto split; do
                for sh in "8 32640" "16 32640" "8 16256" "16 16256" "1 32752"; do
                    lb=${sh% *}; lt=${sh#* }
                    timeout 300 $BN --batch $lb --heads 32 --kv-heads 32 --prompt $lt --residual 128 --steps 6 --warmup 2 --form $f > $O/long1_b${lb}_t${lt}_${f}_$i.json 2>> $O/long1.err; line $O/long1_b${lb}_t${lt}_${f}_$i.json
                done
            done
        done ;;
    tests)
        rm -f $R/gpurun_out/gemv_ratios.log
        timeout 2000 python -m pytest tests -m gpu -q --maxfail=${MAXFAIL:-1} --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/status.log
        cp $R/gpurun_out/gemv_ratios.log $O/gemv_ratios.log 2>/dev/null
        tail -15 $O/pytest_gpu.log
        timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" >> $O/pytest_gpu.log 2>&1; echo "smoke rc=$?" | tee -a $O/status.log ;;
    bench)
        timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/status.log
        head -c 2500 $O/bench.json; echo; line $O/bench.json ;;
    shapes)
        for b in 16 8 64 128; do timeout 300 $BN --batch $b --steps 10 --warmup 3 > $O/shape_b$b.json 2>> $O/shapes.err; line $O/shape_b$b.json; done
        timeout 300 $BN --batch 1 --prompt 32752 --steps 10 --warmup 3 > $O/shape_b1_32k.json 2>> $O/shapes.err; line $O/shape_b1_32k.json
        timeout 300 $BN --batch 4 --steps 10 --warmup 3 > $O/shape_b4.json 2>> $O/shapes.err; line $O/shape_b4.json
        timeout 300 $BN --bits 4 --steps 10 --warmup 3 > $O/shape_4bit.json 2>> $O/shapes.err; line $O/shape_4bit.json
        timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/shape_config4.json 2>> $O/shapes.err; line $O/shape_config4.json
        timeout 300 $BN $C4 --bits 4 --steps 10 --warmup 3 > $O/shape_config4_4bit.json 2>> $O/shapes.err; line $O/shape_config4_4bit.json
        timeout 300 $BN $C5 --steps 6 --warmup 2 > $O/shape_config5_slice.json 2>> $O/shapes.err; line $O/shape_config5_slice.json
        timeout 300 $BN $C70 --steps 10 --warmup 3 > $O/shape_70b_slice.json 2>> $O/shapes.err; line $O/shape_70b_slice.json
        KIVI_TUNING=1 KIVI_NO_MFMA_MHA=1 timeout 300 $BN > $O/shape_headline_hook_layout.json 2>> $O/shapes.err; line $O/shape_headline_hook_layout.json
        # round 6: LongChat-7B-32K shapes (multi-head rows beyond 8192 keys, KIVI-2 and KIVI-4), three blocks per CU on short grouped-query rows
        for bt in "8 32640" "16 16256"; do
            lb=${bt% *}; lt=${bt#* }
            timeout 300 $BN --batch $lb --prompt $lt --residual 128 --steps 6 --warmup 2 > $O/shape_longchat_b${lb}_t${lt}.json 2>> $O/shapes.err; line $O/shape_longchat_b${lb}_t${lt}.json
            timeout 300 $BN --bits 4 --batch $lb --prompt $lt --residual 128 --steps 6 --warmup 2 > $O/shape_longchat_b${lb}_t${lt}_4bit.json 2>> $O/shapes.err; line $O/shape_longchat_b${lb}_t${lt}_4bit.json
        done
        timeout 300 $BN --batch 96 --heads 32 --kv-heads 8 --prompt 6016 --residual 128 --steps 10 --warmup 3 > $O/shape_gqa_b96_6k.json 2>> $O/shapes.err; line $O/shape_gqa_b96_6k.json ;;
    shapes_g)
        # group sizes / head dims off the matrix-pipe layout (g = 64 is what the reference's own test procedures use, quant/test.py:21-54): the VALU
        # kernels on the hook-state layout (decode_row_kernel)
        timeout 300 $BN --group 64 --residual 64 --steps 10 --warmup 3 > $O/shape_g64_r64.json 2>> $O/shapes.err; line $O/shape_g64_r64.json
        timeout 300 $BN --group 64 --residual 128 --steps 10 --warmup 3 > $O/shape_g64_r128.json 2>> $O/shapes.err; line $O/shape_g64_r128.json
        timeout 300 $BN --group 64 --residual 128 --bits 4 --steps 10 --warmup 3 > $O/shape_g64_r128_4bit.json 2>> $O/shapes.err; line $O/shape_g64_r128_4bit.json
        timeout 300 $BN --group 128 --residual 128 --steps 10 --warmup 3 > $O/shape_g128_r128.json 2>> $O/shapes.err; line $O/shape_g128_r128.json
        timeout 300 $BN --head-dim 64 --heads 64 --kv-heads 64 --steps 10 --warmup 3 > $O/shape_d64.json 2>> $O/shapes.err; line $O/shape_d64.json
        timeout 300 $BN --heads 32 --kv-heads 16 --steps 10 --warmup 3 > $O/shape_ratio2.json 2>> $O/shapes.err; line $O/shape_ratio2.json ;;
    fuzz)
        # randomised parity sweep against the fp64 torch reference (tools/fuzz_decode.py); FUZZ_SECONDS / FUZZ_SEED
        timeout $(( ${FUZZ_SECONDS:-600} + 240 )) python tools/fuzz_decode.py --seconds ${FUZZ_SECONDS:-600} --seed ${FUZZ_SEED:-1} > $O/fuzz_seed${FUZZ_SEED:-1}.log 2> $O/fuzz_seed${FUZZ_SEED:-1}.err
        echo "fuzz rc=$?" | tee -a $O/status.log; grep -c " ok " $O/fuzz_seed${FUZZ_SEED:-1}.log; grep "FAIL\|ERROR\|^#" $O/fuzz_seed${FUZZ_SEED:-1}.log | cut -c1-400 | tail -30 ;;
    trace)
        # the driver's command incl. the BASELINE configs[1] loop through the reference's operator (cuda_bmm_fA_qB_outer -> gemv_k_kernel),
        # so that the kernel stats / trace medians carry a gemv_k_kernel row (the kernel the north-star target is written about)
        # round 6: ... and the three extra BASELINE configurations of the default invocation (roofline_config4 / _config4_4bit / _config5_slice:
        # 8 layers x (2 warm-up + 14 timed) steps each -> their mf_row4_kernel instantiations skip 16 warm-up launches)
        BN_SAVE=$BN; BN="python $R/bench.py --no-cpu-baseline"
        ROW4_SKIP=16 trace_one bench 160
        BN=$BN_SAVE ;;
    trace_c4) trace_one config4 96 $C4 --steps 10 --warmup 3 ;;
    trace_gqa)
        trace_one config4 96 $C4 --steps 10 --warmup 3
        trace_one config5slice 64 $C5 --steps 6 --warmup 2
        trace_one slice70b 96 $C70 --steps 10 --warmup 3 ;;
    pmc) pmc_one bench '{"B": 32, "nh": 32, "nh_kv": 32, "prompt": 4080, "bits": 2, "group": 32, "residual": 32}' ;;
    pmc_c4b4) pmc_one config4_4bit '{"B": 64, "nh": 32, "nh_kv": 8, "prompt": 8064, "bits": 4, "group": 32, "residual": 128}' $C4 --bits 4 ;;
    pmc_c5) pmc_one config5slice '{"B": 16, "nh": 32, "nh_kv": 8, "prompt": 32640, "bits": 2, "group": 32, "residual": 128}' $C5 ;;
    pmc_c4) pmc_one config4 '{"B": 64, "nh": 32, "nh_kv": 8, "prompt": 8064, "bits": 2, "group": 32, "residual": 128}' $C4 ;;
    e2e)
        timeout 900 python examples/mem_spd_test.py --recipe > $O/e2e_mem_spd_recipe.log 2>&1; echo "recipe rc=$?" | tee -a $O/status.log; tail -12 $O/e2e_mem_spd_recipe.log
        timeout 900 python examples/mem_spd_test.py --graphs > $O/e2e_llama7b_shape.log 2>&1; echo "e2e rc=$?" | tee -a $O/status.log; tail -8 $O/e2e_llama7b_shape.log ;;
    ab)
        libs=("")
        while [ $# -gt 0 ] && [[ $1 == *.so ]]; do libs+=("$1"); shift; done
        for i in 1 2; do
            for l in "${libs[@]}"; do
                n=$(basename "${l:-intree}" .so)
                KIVI_TUNING=1 KIVI_HIP_LIB=${l:-$R/kivi_amd/libkivi_hip.so} timeout 300 $BN > $O/ab_${n}_hl_$i.json 2>> $O/ab.err; line $O/ab_${n}_hl_$i.json
                KIVI_TUNING=1 KIVI_HIP_LIB=${l:-$R/kivi_amd/libkivi_hip.so} timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/ab_${n}_c4_$i.json 2>> $O/ab.err; line $O/ab_${n}_c4_$i.json
            done
        done ;;
    phases)
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        KIVI_TUNING=1 KIVI_HIP_LIB=$T timeout 300 python tools/mf_row_phases.py > $O/row_phases.log 2>&1; tail -30 $O/row_phases.log
        KIVI_TUNING=1 KIVI_HIP_LIB=$T B=64 NHKV=8 T0=8064 R=128 LAYERS=6 timeout 300 python tools/mf_row_phases.py > $O/row4_phases.log 2>&1; tail -30 $O/row4_phases.log ;;
    row4ab)
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        for cfg in 434 444 1434; do
            KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4=$cfg KIVI_MF_ROW4_FLOW=stream timeout 600 python -m pytest tests/test_mfma_gpu.py tests/test_hook_gpu.py -m gpu -x -q \
                -k "(row and fixtures) or matches_two_launch" > $O/row4_parity_$cfg.log 2>&1; echo "parity $cfg rc=$?" | tee -a $O/status.log; tail -3 $O/row4_parity_$cfg.log
        done
        for i in 1 2; do
            for cfg in 434 234 834 444 844 424 1434; do
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4=$cfg KIVI_MF_ROW4_FLOW=stream timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/c4_${cfg}_$i.json 2>> $O/row4ab.err; line $O/c4_${cfg}_$i.json
            done
        done ;;
    xcd)
        # two-launch form with a unit's blocks on ONE XCD (tuning build, KIVI_MF_XCD=1) against the plain block order: parity of the
        # renumbered launches through the two-launch tests, then alternating bench lines on one box
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_XCD=1 timeout 600 python -m pytest tests/test_mfma_gpu.py -m gpu -x -q -k "gqa_scores or gqa_output or decode_steps" \
            > $O/xcd_parity.log 2>&1; echo "xcd parity rc=$?" | tee -a $O/status.log; tail -3 $O/xcd_parity.log
        for i in 1 2; do
            for x in 0 1; do
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_XCD=$x timeout 300 $BN $C5 --steps 6 --warmup 2 > $O/xcd${x}_c5_$i.json 2>> $O/xcd.err; line $O/xcd${x}_c5_$i.json
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_XCD=$x timeout 300 $BN $C70 --steps 10 --warmup 3 > $O/xcd${x}_70b_$i.json 2>> $O/xcd.err; line $O/xcd${x}_70b_$i.json
                KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_XCD=$x timeout 300 $BN --batch 1 --prompt 32752 --steps 10 --warmup 3 > $O/xcd${x}_b1_32k_$i.json 2>> $O/xcd.err; line $O/xcd${x}_b1_32k_$i.json
            done
        done ;;
    mf4)
        # 4-bit K / V on the matrix pipe (nh / nh_kv = 4): its parity tests + the reference-class fixtures of that shape, then BASELINE
        # config 4 at --bits 4 on the matrix pipe against the VALU kernels of the hook-state layout (same box)
        timeout 1200 python -m pytest tests/test_mfma4_gpu.py tests/test_hook_gpu.py -m gpu -q --tb=short --maxfail=40 -k "mfma4 or b4_r" > $O/mf4_tests.log 2>&1
        echo "mf4 tests rc=$?" | tee -a $O/status.log; tail -60 $O/mf4_tests.log | cut -c1-400
        for i in 1 2; do
            timeout 300 $BN $C4 --bits 4 --steps 10 --warmup 3 > $O/mf4_c4_$i.json 2>> $O/mf4.err; line $O/mf4_c4_$i.json
            KIVI_TUNING=1 KIVI_NO_MFMA_LAYOUT=1 timeout 300 $BN $C4 --bits 4 --steps 10 --warmup 3 > $O/mf4_c4_valu_$i.json 2>> $O/mf4.err; line $O/mf4_c4_valu_$i.json
        done ;;
    mf4prof)
        # BASELINE config 4 at --bits 4 on the matrix pipe: kernel trace medians + HBM traffic; the config-5 slice at --bits 4, matrix pipe
        # (two launches) against the VALU kernels
        trace_one config4_4bit 96 $C4 --bits 4 --steps 10 --warmup 3
        pmc_one config4_4bit '{"B": 64, "nh": 32, "nh_kv": 8, "prompt": 8064, "bits": 4, "group": 32, "residual": 128}' $C4 --bits 4
        timeout 300 $BN $C5 --bits 4 --steps 6 --warmup 2 > $O/mf4_c5.json 2>> $O/mf4.err; line $O/mf4_c5.json
        KIVI_TUNING=1 KIVI_NO_MFMA_LAYOUT=1 timeout 300 $BN $C5 --bits 4 --steps 6 --warmup 2 > $O/mf4_c5_valu.json 2>> $O/mf4.err; line $O/mf4_c5_valu.json ;;
    mf4pmc)
        # the first half of mf4prof only: BASELINE config 4 at --bits 4, kernel trace medians + HBM traffic
        trace_one config4_4bit 96 $C4 --bits 4 --steps 10 --warmup 3
        pmc_one config4_4bit '{"B": 64, "nh": 32, "nh_kv": 8, "prompt": 8064, "bits": 4, "group": 32, "residual": 128}' $C4 --bits 4 ;;
    forms)
        # round 5: the library's launch plan (auto) against forced forms on ONE box: BASELINE config 4, the config-5 per-GPU slice, the 70B-like
        # slice, R = 8 at B = 64 (1024 blocks: ticket ids), grouped-query rows of few units
        for f in auto split; do
            timeout 300 $BN $C5 --steps 6 --warmup 2 --form $f > $O/forms_c5_$f.json 2>> $O/forms.err; line $O/forms_c5_$f.json
            timeout 300 $BN $C70 --steps 10 --warmup 3 --form $f > $O/forms_c70_$f.json 2>> $O/forms.err; line $O/forms_c70_$f.json
            timeout 300 $BN --batch 64 --heads 64 --kv-heads 8 --prompt 8064 --residual 128 --steps 6 --warmup 2 --form $f > $O/forms_r8b64_$f.json 2>> $O/forms.err; line $O/forms_r8b64_$f.json
        done
        for f in auto split row; do
            timeout 300 $BN $C4 --steps 10 --warmup 3 --form $f > $O/forms_c4_$f.json 2>> $O/forms.err; line $O/forms_c4_$f.json
            for b in 4 16 32; do
                timeout 300 $BN --batch $b --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3 --form $f > $O/forms_b${b}_8k_$f.json 2>> $O/forms.err; line $O/forms_b${b}_8k_$f.json
            done
        done ;;
    flows)
        # round 5: the two flows of mf_row4_kernel for unsliced rows at BASELINE config 4 (phase softmax = product, in-stream = KIVI_MF_ROW4_FLOW=stream in
        # the tuning build) and, when a worktree of the round-4 tree exists (git worktree add _r4 96aba04 && (cd _r4 && python -m kivi_amd.build)), that
        # tree from its own directory; one box, alternating; then the phase timelines of both flows and of mf_row_kernel
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        for i in 1 2 3; do
            [ -d $R/_r4 ] && { ( cd $R/_r4 && timeout 300 python bench.py --no-cpu-baseline --no-hook-kgemv $C4 --steps 10 --warmup 3 > $O/flows_c4_r4tree_$i.json 2>> $O/flows.err ); line $O/flows_c4_r4tree_$i.json; }
            timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/flows_c4_psm_$i.json 2>> $O/flows.err; line $O/flows_c4_psm_$i.json
            KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_FLOW=stream timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/flows_c4_stream_$i.json 2>> $O/flows.err; line $O/flows_c4_stream_$i.json
            [ -d $R/_r4 ] && { ( cd $R/_r4 && timeout 300 python bench.py --no-cpu-baseline --no-hook-kgemv > $O/flows_hl_r4tree_$i.json 2>> $O/flows.err ); line $O/flows_hl_r4tree_$i.json; }
            timeout 300 $BN > $O/flows_hl_new_$i.json 2>> $O/flows.err; line $O/flows_hl_new_$i.json
        done
        KIVI_TUNING=1 KIVI_HIP_LIB=$T B=64 NHKV=8 T0=8064 R=128 LAYERS=6 timeout 300 python tools/mf_row_phases.py > $O/row4_phases_psm.log 2>&1; sed -n 2,14p $O/row4_phases_psm.log
        KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_FLOW=stream B=64 NHKV=8 T0=8064 R=128 LAYERS=6 timeout 300 python tools/mf_row_phases.py > $O/row4_phases_stream.log 2>&1; sed -n 2,14p $O/row4_phases_stream.log
        KIVI_TUNING=1 KIVI_HIP_LIB=$T timeout 300 python tools/mf_row_phases.py > $O/row_phases.log 2>&1; sed -n 2,14p $O/row_phases.log ;;
    packs)
        # prompt-pass packers of the matrix-pipe layout on 1 GiB of fp16 (2- and 4-bit), same box
        BITS=2 timeout 200 python tools/mf_prefill_time.py > $O/pack2_time.log 2>&1; tail -4 $O/pack2_time.log
        BITS=4 timeout 200 python tools/mf_prefill_time.py > $O/pack4_time.log 2>&1; tail -4 $O/pack4_time.log ;;
    sq)
        name=$1; shift
        extra=()
        while [ $# -gt 0 ] && [[ $1 == --* || $1 =~ ^[0-9]+$ ]]; do extra+=("$1"); shift; done
        cd /tmp && export TMPDIR=/tmp
        rm -rf $O/sq_$name
        timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv \
            -d $O/sq_$name -o p -- $BN --steps 2 --warmup 1 --no-kernel-events "${extra[@]}" > $O/sq_$name.run.log 2>&1
        cd $R
        python tools/sq_counters.py $(find $O/sq_$name -name "*counter_collection.csv" | head -1) > $O/sq_$name.log 2>&1; cat $O/sq_$name.log
        rm -rf $O/sq_$name ;;
    sq6)
        # round 6: SQ counters of mf_row4_kernel at BASELINE config 4 -- the product's four-wave block (two blocks per CU), the six-wave block
        # (tuning build, KIVI_MF_ROW4_NW=6, rings 4 / 3), and the three-blocks-per-CU instantiation at B=96 x 6k -- one box
        T=$R/kivi_amd/_variants/libkivi_tuning.so
        sq_pass() {  # <name> <bench args...>   (environment of the caller)
            local name=$1; shift
            cd /tmp && export TMPDIR=/tmp
            rm -rf $O/sq_$name
            timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv \
                -d $O/sq_$name -o p -- $BN --steps 2 --warmup 1 --no-kernel-events "$@" > $O/sq_$name.run.log 2>&1
            cd $R
            echo "== $name" >> $O/sq6.log
            python tools/sq_counters.py $(find $O/sq_$name -name "*counter_collection.csv" | head -1) >> $O/sq6.log 2>&1
            rm -rf $O/sq_$name
        }
        rm -f $O/sq6.log
        sq_pass config4_four_waves $C4
        KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_NW=6 KIVI_MF_ROW4_6=43 sq_pass config4_six_waves $C4
        sq_pass b96_6k_three_blocks --batch 96 --heads 32 --kv-heads 8 --prompt 6016 --residual 128
        for i in 1 2; do
            timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/sq6_c4_four_$i.json 2>> $O/sq6.err; line $O/sq6_c4_four_$i.json
            KIVI_TUNING=1 KIVI_HIP_LIB=$T KIVI_MF_ROW4_NW=6 KIVI_MF_ROW4_6=43 timeout 300 $BN $C4 --steps 10 --warmup 3 > $O/sq6_c4_six_$i.json 2>> $O/sq6.err; line $O/sq6_c4_six_$i.json
        done
        cat $O/sq6.log ;;
    *) echo "unknown stage $stage" ;;
    esac
done
cat $O/status.log 2>/dev/null
