#!/bin/bash
# round-4 GPU session 39: END build (pointwise tile width by rounds x width) - full GPU suite, smoke, default bench line, kernel statistics + marker trace
set -u
OUT=gpurun_out/r4M; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc" | tee -a $OUT/summary.txt
tail -n 5 $OUT/tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stop" | tee -a $OUT/summary.txt; exit 1; fi
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
timeout -k 10 580 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?" | tee -a $OUT/summary.txt
B="python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f32 -- $B > $OUT/prof_f32.json 2> $OUT/prof_f32.err; echo "prof f32 rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_f32 -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_f32.csv \;
find $OUT -name '*kernel_trace*.csv' -delete; find $OUT -name '*.db' -delete
SG_TRACE_MARK=1 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_f32 -- $B > $OUT/trace_f32.json 2> $OUT/trace_f32.err; echo "trace rc=$?" | tee -a $OUT/summary.txt
python scripts/trace_dilated.py $OUT/trace_f32 6 $OUT/trace_dilated.json $OUT/trace_dilated_launches.csv > $OUT/trace_dilated.txt 2>&1; echo "trace parse rc=$?" | tee -a $OUT/summary.txt
find $OUT -name '*kernel_trace*.csv' -delete; find $OUT -name '*.db' -delete
B16="$B --dtype bf16"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bf16 -- $B16 > $OUT/prof_bf16.json 2> $OUT/prof_bf16.err; echo "prof bf16 rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_bf16 -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats_bf16.csv \;
find $OUT -name "*kernel_trace*.csv" -delete; find $OUT -name "*.db" -delete
SG_SIDE_WGRAD=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_f32_1s -- $B > $OUT/prof_f32_1s.json 2> $OUT/prof_f32_1s.err; echo "prof f32 single stream rc=$?" | tee -a $OUT/summary.txt
find $OUT/prof_f32_1s -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_f32_single_stream.csv \;
find $OUT -name '*kernel_trace*.csv' -delete; find $OUT -name '*.db' -delete
du -sh $OUT; echo done | tee -a $OUT/summary.txt
