#!/bin/bash
# round-4 GPU session 7: A/B of the depthwise kernels' register budgets
set -u
OUT=gpurun_out/r4g; mkdir -p $OUT
B="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit"
run() { name=$1; shift; env "$@" $B ${EXTRA:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
EXTRA="" run f32_base A=1
EXTRA="" run f32_occ2 SG_DW_STRIP_OCC2=1
EXTRA="" run f32_rr2 SG_DW_RR=2
EXTRA="" run f32_rr2_occ2 SG_DW_RR=2 SG_DW_STRIP_OCC2=1
EXTRA="" run f32_base2 A=1
EXTRA="--dtype bf16" run bf16_base A=1
EXTRA="--dtype bf16" run bf16_occ2 SG_DW_STRIP_OCC2=1
EXTRA="--dtype bf16" run bf16_rr2 SG_DW_RR=2
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider -k "dw or depthwise or gather" > $OUT/t_dw.log 2>&1; echo "dw tests rc=$?" | tee -a $OUT/summary.txt
SG_DW_STRIP_OCC2=1 timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_models_gpu.py -m gpu -q -p no:cacheprovider -k "dw or depthwise or gather" > $OUT/t_dw_occ2.log 2>&1; echo "dw tests occ2 rc=$?" | tee -a $OUT/summary.txt
echo done | tee -a $OUT/summary.txt
