#!/bin/bash
# round-4 GPU session 14: timing ablations of conv_x6w_kernel (what bounds its stage loop)
set -u
OUT=gpurun_out/r4n; mkdir -p $OUT
for a in 0 1 2 3 4 5 6 7; do
  echo "== SG_X6W_ABLATE=$a" >> $OUT/ablate.txt
  ONLY_DILATED=1 SG_X6W_ABLATE=$a timeout -k 10 200 python scripts/dilated_bench.py 2>&1 | grep "aspp" >> $OUT/ablate.txt
done
cat $OUT/ablate.txt | cut -c1-110
echo done | tee -a $OUT/summary.txt
