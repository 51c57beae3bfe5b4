#!/bin/bash
# round-4 GPU session 5: full suite (stem kernels, wide bf16 kernel, BN sums with two-row runs), stem A/B, benches
set -u
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -n 8 $OUT/tests.log
SG_STEM3=1 timeout -k 10 200 python scripts/stem_ab.py > $OUT/stem_on.txt 2>&1; echo "stem on rc=$?" | tee -a $OUT/summary.txt
SG_STEM3=0 timeout -k 10 200 python scripts/stem_ab.py > $OUT/stem_off.txt 2>&1; echo "stem off rc=$?" | tee -a $OUT/summary.txt
cat $OUT/stem_on.txt $OUT/stem_off.txt | grep -v amdgpu.ids
timeout -k 10 500 python bench.py --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?" | tee -a $OUT/summary.txt
B="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $B ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
EXTRA="--no-jit" run f32_eager A=1
EXTRA="--jit" run f32_jit A=1
EXTRA="--jit --force-dp" run f32_dp1_jit A=1
EXTRA="--no-jit --dtype bf16" run bf16_eager A=1
EXTRA="--jit --dtype bf16" run bf16_jit A=1
EXTRA="--jit --force-dp --dtype bf16" run bf16_dp1_jit A=1
EXTRA="--no-jit --model res34" run res34_eager A=1
EXTRA="--no-jit --model res34" run res34_eager_nostem SG_STEM3=0
EXTRA="--no-jit --model scse" run scse_eager A=1
EXTRA="--no-jit --model hrnet" run hrnet_eager A=1
EXTRA="--no-jit --model bam" run bam_eager A=1
echo done | tee -a $OUT/summary.txt
