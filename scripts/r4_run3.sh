#!/bin/bash
# round-4 GPU session 3: full suite with the BatchNormalization-sums fusion, BN_DEFER default, lanes; A/B benches
set -u
OUT=gpurun_out/r4c; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -n 5 $OUT/tests.log
B="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $B ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
EXTRA="--no-jit" run eager A=1
EXTRA="--no-jit" run eager_nosums SG_BN_SUMS=0
EXTRA="--no-jit" run eager_nodefer SG_BN_DEFER=0
EXTRA="--no-jit" run eager_nosums_nodefer SG_BN_SUMS=0 SG_BN_DEFER=0
EXTRA="--jit" run jit A=1
EXTRA="--no-jit --dtype bf16" run bf16_eager A=1
EXTRA="--no-jit --dtype bf16" run bf16_eager_nosums_nodefer SG_BN_SUMS=0 SG_BN_DEFER=0
EXTRA="--jit --dtype bf16" run bf16_jit A=1
EXTRA="--jit --dtype bf16" run bf16_jit_lanes4 SG_JIT_LANE_BLOCKS=4
EXTRA="--jit --dtype bf16" run bf16_jit_nolanes SG_JIT_LANES=0
timeout -k 10 500 python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?" | tee -a $OUT/summary.txt
echo done | tee -a $OUT/summary.txt
