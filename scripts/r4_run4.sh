#!/bin/bash
# round-4 GPU session 4: the 256-wide bf16 kernel (tests, census A/B, step A/B), BatchNormalization sums re-measured
set -u
OUT=gpurun_out/r4d; mkdir -p $OUT
T="timeout -k 10 900 python -m pytest -q -p no:cacheprovider"
$T tests/test_bf16_gpu.py -m gpu -x -s > $OUT/t_bf16.log 2>&1; echo "bf16 tests rc=$?" | tee -a $OUT/summary.txt
tail -n 3 $OUT/t_bf16.log
$T tests/test_block_chains_gpu.py tests/test_ops_gpu.py -m gpu -s -k "bf16 or sums or middle" > $OUT/t_chains.log 2>&1; echo "chains/sums rc=$?" | tee -a $OUT/summary.txt
$T tests/test_fullsize_gpu.py -m gpu -s -k "config3" > $OUT/t_config3.log 2>&1; echo "config3 rc=$?" | tee -a $OUT/summary.txt
tail -n 3 $OUT/t_config3.log
DTYPE=bf16 SG_B16_WIDE=0 timeout -k 10 400 python scripts/conv_census.py > $OUT/census_bf16_wide0.txt 2>&1; echo "census0 rc=$?" | tee -a $OUT/summary.txt
DTYPE=bf16 SG_B16_WIDE=1 timeout -k 10 400 python scripts/conv_census.py > $OUT/census_bf16_wide1.txt 2>&1; echo "census1 rc=$?" | tee -a $OUT/summary.txt
B="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $B ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
EXTRA="--no-jit --dtype bf16" run bf16_wide1_nosums SG_BN_SUMS=0
EXTRA="--no-jit --dtype bf16" run bf16_wide0_nosums SG_BN_SUMS=0 SG_B16_WIDE=0
EXTRA="--no-jit --dtype bf16" run bf16_wide1_sums A=1
EXTRA="--no-jit" run f32_sums A=1
EXTRA="--no-jit" run f32_nosums SG_BN_SUMS=0
EXTRA="--jit" run f32_jit_nosums_l24 SG_BN_SUMS=0 SG_JIT_LANE_BLOCKS=24
EXTRA="--jit" run f32_jit_nosums_l12 SG_BN_SUMS=0
EXTRA="--jit --dtype bf16" run bf16_jit_nosums_l24 SG_BN_SUMS=0 SG_JIT_LANE_BLOCKS=24
EXTRA="--jit --dtype bf16" run bf16_jit_nosums_l48 SG_BN_SUMS=0 SG_JIT_LANE_BLOCKS=48
echo done | tee -a $OUT/summary.txt
