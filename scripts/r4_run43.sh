#!/bin/bash
# round-4 GPU session 43: the full GPU suite with torch's CPU pool capped at the box's share; smoke
set -u
OUT=gpurun_out/r4Q; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
grep -A14 "slowest" $OUT/tests.log
tail -2 $OUT/tests.log
( time timeout -k 10 300 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/smoke.log
