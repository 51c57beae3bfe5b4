#!/bin/bash
# Generic interleaved A/B of an environment switch on the conv micro-benchmark: ab_env.sh <tag> VAR "v1 v2 ..." [iters]
OUT=gpurun_out/${1:-abenv}; VAR=$2; VALS=$3; IT=${4:-20}
mkdir -p $OUT
for round in 1 2; do
  for v in $VALS; do
    echo "== round $round $VAR=$v" >> $OUT/ab.log
    if [ "$v" = "unset" ]; then ITERS=$IT timeout -k 10 300 python scripts/dilated_bench.py >> $OUT/ab.log 2>&1 || exit 1
    else env $VAR=$v ITERS=$IT timeout -k 10 300 python scripts/dilated_bench.py >> $OUT/ab.log 2>&1 || exit 1; fi
  done
done
grep -v amdgpu.ids $OUT/ab.log
