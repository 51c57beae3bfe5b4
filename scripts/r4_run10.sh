#!/bin/bash
# round-4 GPU session 10: two-row stencil runs by default, four-lane split reduce (A/B), full suite, the default bench line
set -u
OUT=gpurun_out/r4j; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -n 6 $OUT/tests.log
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit"
run() { name=$1; shift; env "$@" $BB ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2 3; do
  EXTRA="" run z4_$rep A=1
  EXTRA="" run z1_$rep SG_REDUCE_Z4=0
done
EXTRA="--dtype bf16" run bf16_z4 A=1
EXTRA="--dtype bf16" run bf16_z1 SG_REDUCE_Z4=0
timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?" | tee -a $OUT/summary.txt
echo done | tee -a $OUT/summary.txt
