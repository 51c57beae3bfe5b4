#!/bin/bash
# A/B of the x6 (bf16 split) convolution variants (SG_X6_VARIANT: bit 0 two-slab prefetch, bit 1 4-wave workgroups)
OUT=gpurun_out/${1:-abx6}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "conv" > $OUT/tests.log 2>&1 || { tail -n 40 $OUT/tests.log; exit 1; }
tail -n 2 $OUT/tests.log
for round in 1 2; do
  for v in ${VARIANTS:-0 1 2 3}; do
    echo "== round $round SG_X6_VARIANT=$v" >> $OUT/ab.log
    ITERS=20 SG_X6_VARIANT=$v timeout -k 10 300 python scripts/dilated_bench.py >> $OUT/ab.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $OUT/ab.log
