#!/bin/bash
# round-4 GPU session 42: the DeepLab parity cases at 96 x 96 instead of 128 x 128 (CPU oracle time)
set -u
OUT=gpurun_out/r4P; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_models_gpu.py -x -q -m gpu -p no:cacheprovider -k "parity" --durations=8 > $OUT/t.log 2>&1; echo "rc=$?" | tee -a $OUT/summary.txt
tail -16 $OUT/t.log
