#!/bin/bash
# round-4 GPU session 32: the depthwise variants of the step, run kernel against strips, stand-alone
set -u
OUT=gpurun_out/r4F; mkdir -p $OUT
for v in 0 2 0 2; do SG_DW_FSTRIP=$v timeout -k 10 200 python scripts/dw_variants_bench.py >> $OUT/variants.txt 2>&1; done
for hs in 4 16; do echo "HS=$hs" >> $OUT/variants.txt; SG_DW_FSTRIP=2 SG_DW_FSTRIP_HS=$hs timeout -k 10 200 python scripts/dw_variants_bench.py >> $OUT/variants.txt 2>&1; done
grep -v amdgpu.ids $OUT/variants.txt | cut -c1-200
