#!/bin/bash
# round-4 GPU session 33: channel-block run length of the K walk (SG_CONV_CB) with the planes-in kernel, dilated set
set -u
OUT=gpurun_out/r4G; mkdir -p $OUT
for rep in 1 2; do for v in 2 4 8 16 64; do
  echo "== SG_CONV_CB=$v rep $rep" >> $OUT/cb.txt
  ONLY_DILATED=1 SG_CONV_CB=$v timeout -k 10 200 python scripts/dilated_bench.py 2>&1 | grep "aspp\|dilated set" >> $OUT/cb.txt
done; done
cut -c1-125 $OUT/cb.txt
