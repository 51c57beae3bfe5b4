#!/bin/bash
# Repeat the fit_generator trajectory test with the multi-chunk patch kernel on / off: is the step-3 loss a property of the
# build or of the run?  usage: bash scripts/fit_repeat.sh <tag>
out=gpurun_out/$1; mkdir -p $out
for v in 1 0 1 0; do
  echo "== SG_X6P_CHUNKS=$v" | tee -a $out/fit.txt
  SG_X6P_CHUNKS=$v timeout -k 10 300 python -m pytest tests/test_models_gpu.py -q -p no:cacheprovider -k fit_generator -rP 2>&1 \
    | grep -E "loss per step|passed|failed" | tee -a $out/fit.txt
done
true
