#!/bin/bash
# round-4 GPU session 2: new tests, side-graph replay, write-through fused reductions (A/B benches)
set -u
OUT=gpurun_out/r4b; mkdir -p $OUT
T="timeout -k 10 900 python -m pytest -q -p no:cacheprovider -x"
$T tests/test_block_chains_gpu.py -m gpu -s > $OUT/t_chains.log 2>&1; echo "chains rc=$?" | tee -a $OUT/summary.txt
$T tests/test_ops_gpu.py -m gpu -k "maxpool or collected or add2_bn" > $OUT/t_ops.log 2>&1; echo "ops rc=$?" | tee -a $OUT/summary.txt
$T tests/test_models_gpu.py tests/test_dist_gpu.py -m gpu -k "jit or side or captured or prepared" > $OUT/t_jit.log 2>&1; echo "jit rc=$?" | tee -a $OUT/summary.txt
SG_SEG_FUSED=3 $T tests/test_ops_gpu.py tests/test_models_gpu.py -m gpu -k "bn or batchnorm or dw or depthwise or gates or loss or parity or golden" > $OUT/t_fused3.log 2>&1; echo "fused3 rc=$?" | tee -a $OUT/summary.txt
B="timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $B ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
EXTRA="" run auto A=1
EXTRA="--jit" run jit_lanes A=1
EXTRA="--jit" run jit_nolanes SG_JIT_LANES=0
EXTRA="--jit" run jit_lanes12 SG_JIT_LANE_BLOCKS=12
EXTRA="--jit" run jit_lanes3 SG_JIT_LANE_BLOCKS=3
EXTRA="--no-jit" run eager_fused3 SG_SEG_FUSED=3
EXTRA="--jit" run jit_fused3 SG_SEG_FUSED=3
EXTRA="--jit" run jit_fused3_bndefer SG_SEG_FUSED=3 SG_BN_DEFER=1
EXTRA="--jit --force-dp" run dp1_jit A=1
EXTRA="--jit --dtype bf16" run bf16_jit_fused3 SG_SEG_FUSED=3
EXTRA="--jit --dtype bf16" run bf16_jit_fused3_bndefer SG_SEG_FUSED=3 SG_BN_DEFER=1
EXTRA="--no-jit --dtype bf16" run bf16_eager A=1
echo done | tee -a $OUT/summary.txt
