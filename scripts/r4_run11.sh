#!/bin/bash
# round-4 GPU session 11: the other models with the join limited to once per 16 side blocks; memory test; final suite of the end build
set -u
OUT=gpurun_out/r4k; mkdir -p $OUT
BB="timeout -k 10 500 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit"
run() { name=$1; shift; env "$@" $BB ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for m in res34 hrnet scse bam; do
  EXTRA="--model $m" run ${m}_default A=1
  EXTRA="--model $m" run ${m}_nojoin SG_SIDE_KEEP_GIB=1000
  EXTRA="--model $m" run ${m}_noside SG_SIDE_WGRAD=0
done
EXTRA="" run v3plus_default A=1
EXTRA="" run v3plus_nojoin SG_SIDE_KEEP_GIB=1000
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -n 5 $OUT/tests.log
echo done | tee -a $OUT/summary.txt
