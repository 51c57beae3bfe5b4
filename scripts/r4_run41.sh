#!/bin/bash
# round-4 GPU session 41: census of the GEMM shapes on the END build (stand-alone times, sorted by time lost against 200 TFLOP/s)
set -u
OUT=gpurun_out/r4O; mkdir -p $OUT
RATE=200 timeout -k 10 600 python scripts/conv_census.py v3plus 16 512 > $OUT/census_f32.txt 2>&1; echo "census rc=$?" | tee -a $OUT/summary.txt
grep -v amdgpu.ids $OUT/census_f32.txt | cut -c1-150
