#!/bin/bash
# round-4 GPU session 13: the planes-in x6 kernel (conv_x6w.h): focused tests, dilated-set A/B, step A/B, then the full suite
set -u
OUT=gpurun_out/r4m; mkdir -p $OUT
T="timeout -k 10 900 python -m pytest -q -p no:cacheprovider"
$T tests/test_ops_gpu.py -m gpu -x -k "planes_in or aspp or x6_at_least or sk_d6" > $OUT/t_ops.log 2>&1; echo "ops rc=$?" | tee -a $OUT/summary.txt
tail -n 4 $OUT/t_ops.log
ONLY_DILATED=1 SG_X6_WIDE=0 timeout -k 10 200 python scripts/dilated_bench.py > $OUT/dilated_f32_wide0.txt 2>&1
ONLY_DILATED=1 SG_X6_WIDE=1 timeout -k 10 200 python scripts/dilated_bench.py > $OUT/dilated_f32_wide1.txt 2>&1
grep -h "aspp\|dilated set" $OUT/dilated_f32_wide0.txt $OUT/dilated_f32_wide1.txt
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit"
run() { name=$1; shift; env "$@" $BB ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  EXTRA="" run wide1_$rep A=1
  EXTRA="" run wide0_$rep SG_X6_WIDE=0
done
EXTRA="--model bam" run bam_wide1 A=1
EXTRA="--model bam" run bam_wide0 SG_X6_WIDE=0
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -n 8 $OUT/tests.log
echo done | tee -a $OUT/summary.txt
