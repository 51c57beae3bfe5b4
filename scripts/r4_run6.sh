#!/bin/bash
# round-4 GPU session 6: evidence (kernel statistics, marker-cut trace, PMC traffic fp32 + bf16) and the woven-DMA variant of conv_b16w
set -u
OUT=gpurun_out/r4f; mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 10 600 python -m pytest -q -p no:cacheprovider"
$T tests/test_dist_gpu.py -m gpu > $OUT/t_dist.log 2>&1; echo "dist rc=$?" | tee -a $OUT/summary.txt
SG_B16W_VAR=1 $T tests/test_bf16_gpu.py -m gpu -k "conv or wide" > $OUT/t_bf16_var1.log 2>&1; echo "bf16 var1 rc=$?" | tee -a $OUT/summary.txt
$T tests/test_block_chains_gpu.py -m gpu -s -k "bf16" > $OUT/t_chains_bf16.log 2>&1; echo "chains bf16 rc=$?" | tee -a $OUT/summary.txt
$T tests/test_bf16_gpu.py -m gpu -s -k "convergence" > $OUT/t_conv.log 2>&1; echo "convergence rc=$?" | tee -a $OUT/summary.txt
DTYPE=bf16 ONLY_DILATED=1 SG_B16W_VAR=0 timeout -k 10 200 python scripts/dilated_bench.py > $OUT/dilated_bf16_var0.txt 2>&1
DTYPE=bf16 ONLY_DILATED=1 SG_B16W_VAR=1 timeout -k 10 200 python scripts/dilated_bench.py > $OUT/dilated_bf16_var1.txt 2>&1
DTYPE=bf16 ONLY_DILATED=1 SG_B16_WIDE=0 timeout -k 10 200 python scripts/dilated_bench.py > $OUT/dilated_bf16_wide0.txt 2>&1
grep -h "dilated set" $OUT/dilated_bf16_*.txt
# kernel statistics of the eager step, both precisions
B="python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f32 -- $B > $OUT/prof_f32.json 2> $OUT/prof_f32.err; echo "prof f32 rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bf16 -- $B --dtype bf16 > $OUT/prof_bf16.json 2> $OUT/prof_bf16.err; echo "prof bf16 rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_f32 -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_f32.csv \;
find $OUT/prof_bf16 -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_bf16.csv \;
# marker-cut trace of the dilated set (fp32)
SG_TRACE_MARK=1 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_f32 -- $B > $OUT/trace_f32.json 2> $OUT/trace_f32.err; echo "trace rc=$?" | tee -a $OUT/summary.txt
python scripts/trace_dilated.py $OUT/trace_f32 6 $OUT/trace_dilated.json $OUT/trace_dilated_launches.csv > $OUT/trace_dilated.txt 2>&1; echo "trace parse rc=$?" | tee -a $OUT/summary.txt
find $OUT -name '*kernel_trace*.csv' -delete
find $OUT -name '*.db' -delete
# PMC traffic passes: fp32 set, bf16 set
export ONLY_DILATED=1 ITERS=1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 scripts/dilated_bench.py > $OUT/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 scripts/dilated_bench.py > $OUT/pmc_write.log 2>&1; echo "pmc write rc=$?" | tee -a $OUT/summary.txt
mkdir -p $OUT/bf16
DTYPE=bf16 timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/bf16/pmc_fetch -- python3 scripts/dilated_bench.py > $OUT/bf16/pmc_fetch.log 2>&1; echo "pmc bf16 fetch rc=$?" | tee -a $OUT/summary.txt
DTYPE=bf16 timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/bf16/pmc_write -- python3 scripts/dilated_bench.py > $OUT/bf16/pmc_write.log 2>&1; echo "pmc bf16 write rc=$?" | tee -a $OUT/summary.txt
unset ONLY_DILATED ITERS
python scripts/pmc_traffic.py $OUT $OUT/pmc_traffic.json --prepared > $OUT/pmc_traffic.txt 2>&1; echo "pmc parse rc=$?" | tee -a $OUT/summary.txt
python scripts/pmc_traffic.py $OUT/bf16 $OUT/pmc_traffic_bf16.json --prepared --bf16 > $OUT/pmc_traffic_bf16.txt 2>&1; echo "pmc bf16 parse rc=$?" | tee -a $OUT/summary.txt
find $OUT -name '*kernel_trace*.csv' -size +5M -delete
du -sh $OUT
echo done | tee -a $OUT/summary.txt
