R=$PWD; O=$R/gpurun_out/r5sweep; mkdir -p $O
BN="python $R/bench.py --no-cpu-baseline --no-hook-kgemv"
line() { python - "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = j.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], j["value"], "tok/s", j["ms_per_step"], "ms", r.get("kernel"), r.get("median_launch_us"), "us frac", r.get("frac"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
run() { n=$1; shift; timeout 200 $BN "$@" > $O/$n.json 2>> $O/err.log; line $O/$n.json; }
C5="--batch 16 --heads 32 --kv-heads 8 --prompt 32640 --residual 128 --steps 6 --warmup 2"
C70="--batch 16 --heads 64 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3"
B4="--batch 4 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3"
B16="--batch 16 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3"
B8="--batch 8 --heads 32 --kv-heads 8 --prompt 8064 --residual 128 --steps 10 --warmup 3"
for i in 1 2; do
run c5_auto_$i $C5; run c5_s8_$i $C5 --form slices8
run c70_auto_$i $C70; run c70_s8_$i $C70 --form slices8; run c70_s2_$i $C70 --form slices2
run b4_auto_$i $B4; run b4_s8_$i $B4 --form slices8; run b4_s16_$i $B4 --form slices16
run b8_auto_$i $B8; run b8_s2_$i $B8 --form slices2; run b8_s8_$i $B8 --form slices8; run b8_split_$i $B8 --form split
run b16_auto_$i $B16; run b16_s4_$i $B16 --form slices4; run b16_split_$i $B16 --form split
done
tail -3 $O/err.log
