#!/bin/bash
# round-4 GPU session 26: the bit-identity test of the two barrier placements (child processes)
set -u
OUT=gpurun_out/r4z; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_schedules_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/t.log 2>&1; echo "rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/t.log
