#!/bin/bash
# Where does the x6 forward/dgrad kernel spend its time?  SG_X6_ABLATE: 1 no loads, 2 no LDS store/barriers, 4 no MFMAs
OUT=gpurun_out/${1:-ablx6}
mkdir -p $OUT
for v in ${VARIANTS:-0 2}; do
for a in ${ABLS:-0 3 5 6}; do
  echo "== SG_X6_VARIANT=$v SG_X6_ABLATE=$a" >> $OUT/abl.log
  ONLY_DILATED=1 ITERS=20 SG_X6_VARIANT=$v SG_X6_ABLATE=$a timeout -k 10 300 python scripts/dilated_bench.py >> $OUT/abl.log 2>&1 || exit 1
done
done
grep "==\|aspp" $OUT/abl.log
