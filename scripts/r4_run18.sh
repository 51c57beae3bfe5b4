#!/bin/bash
# round-4 GPU session 18: conv_b16w with the scalar walk (tap mask, no LDS table / divisions per stage): parity, bf16 dilated set
set -u
OUT=gpurun_out/r4r; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv or wide or b16" > $OUT/t_bf16.log 2>&1; echo "bf16 tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/t_bf16.log
for rep in 1 2; do
for v in 0 1; do
  echo "== DTYPE=bf16 SG_B16W_VAR=$v rep $rep" >> $OUT/ab.txt
  DTYPE=bf16 ONLY_DILATED=1 SG_B16W_VAR=$v timeout -k 10 200 python scripts/dilated_bench.py 2>&1 | grep "aspp\|sk \|dilated set" >> $OUT/ab.txt
done; done
cat $OUT/ab.txt | cut -c1-120
echo done | tee -a $OUT/summary.txt
