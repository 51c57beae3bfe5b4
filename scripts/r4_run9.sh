#!/bin/bash
# round-4 GPU session 9: single-stream kernel statistics (BatchNormalization / depthwise totals without side-stream inflation),
# lane-size and rows-per-run A/Bs in alternating repetitions
set -u
OUT=gpurun_out/r4i; mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit"
SG_SIDE_WGRAD=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single -- $B > $OUT/prof_single.json 2> $OUT/prof_single.err; echo "prof single rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_single -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_f32_single_stream.csv \;
find $OUT -name '*kernel_trace*.csv' -delete; find $OUT -name '*.db' -delete
SG_SIDE_WGRAD=0 SG_BN_SUMS=0 SG_BN_DEFER=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_single_r3 -- $B > $OUT/prof_single_r3.json 2> $OUT/prof_single_r3.err; echo "prof single r3-like rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_single_r3 -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_f32_single_stream_nosums_nodefer.csv \;
find $OUT -name '*kernel_trace*.csv' -delete; find $OUT -name '*.db' -delete
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2 3; do
  EXTRA="--no-jit" run rr_default_$rep A=1
  EXTRA="--no-jit" run rr2_$rep SG_DW_RR=2
  EXTRA="--no-jit" run fin4_$rep SG_FINALIZE_LANES=4
done
EXTRA="--no-jit --dtype bf16" run bf16_fin16 A=1
EXTRA="--no-jit --dtype bf16" run bf16_fin4 SG_FINALIZE_LANES=4
for rep in 1 2; do
  EXTRA="--jit" run lanes16_$rep SG_JIT_LANE_BLOCKS=16
  EXTRA="--jit" run lanes24_$rep SG_JIT_LANE_BLOCKS=24
  EXTRA="--jit" run lanes32_$rep SG_JIT_LANE_BLOCKS=32
done
echo done | tee -a $OUT/summary.txt
