#!/bin/bash
# round-4 GPU session 8: the end-of-round record - full GPU suite, smoke, the default bench line (as the driver runs it)
set -u
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -n 6 $OUT/tests.log
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -n 2 $OUT/smoke.log
timeout -k 10 500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?" | tee -a $OUT/summary.txt
echo done | tee -a $OUT/summary.txt
