#!/bin/bash
# round-4 GPU session 36: where the GPU suite's time goes
set -u
OUT=gpurun_out/r4J; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=30 > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
grep -A40 "slowest" $OUT/tests.log | head -45
tail -2 $OUT/tests.log
