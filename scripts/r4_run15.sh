#!/bin/bash
# round-4 GPU session 15: conv_x6w mid-stage barrier (SG_X6W_VAR=1) against the end-of-stage form (0): parity, A/B, per-kernel stats
set -u
OUT=gpurun_out/r4o; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "planes_in or aspp or x6 or dilat" > $OUT/t_ops.log 2>&1; echo "ops rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/t_ops.log
for rep in 1 2; do
for v in 0 1; do
  echo "== SG_X6W_VAR=$v rep $rep" >> $OUT/ab.txt
  ONLY_DILATED=1 SG_X6W_VAR=$v timeout -k 10 200 python scripts/dilated_bench.py 2>&1 | grep "aspp\|dilated set" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt | cut -c1-120
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  ONLY_DILATED=1 SG_X6W_VAR=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_v$v -o st -- python3 $GRAFT_REPO_ROOT/scripts/dilated_bench.py > $GRAFT_REPO_ROOT/$OUT/prof_v$v.log 2>&1
  echo "prof v$v rc=$?" | tee -a $GRAFT_REPO_ROOT/$OUT/summary.txt
done
ONLY_DILATED=1 SG_X6W_VAR=0 SG_X6W_ABLATE=7 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_a7 -o st -- python3 $GRAFT_REPO_ROOT/scripts/dilated_bench.py > $GRAFT_REPO_ROOT/$OUT/prof_a7.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_v0 prof_v1 prof_a7; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d" >> $OUT/stats.txt; head -8 "$f" | cut -c1-160 >> $OUT/stats.txt
  find $OUT/$d -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
cat $OUT/stats.txt
echo done | tee -a $OUT/summary.txt
