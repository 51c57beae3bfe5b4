#!/bin/bash
# round-4 GPU session 44: the GPU suite on the OTHER forms of this round's kernels (end-of-stage barriers, strips on every map,
# 384-wide pointwise tiles only, dilated-only planes-in rule)
set -u
OUT=gpurun_out/r4R; mkdir -p $OUT
SG_X6W_VAR=0 SG_PW_VAR=0 SG_WPW_VAR=0 SG_DW_FSTRIP=2 SG_PW_WIDE=3 SG_X6_WIDE=1 timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/tests_alt.log 2>&1; echo "alt tests rc=$?" | tee -a $OUT/summary.txt
tail -4 $OUT/tests_alt.log
