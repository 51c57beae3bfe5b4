#!/bin/bash
# The other BASELINE.json configurations on one GPU: C4 (SCSE-UNet, DeepLab-BAM 512x512 bs16 train step), C1 shape
# (Res34-UNet 256x256 bs2 train step) and C5 (5-model ensemble inference 1024x1024 bs8, hipGraph).
OUT=gpurun_out/${1:-cfg}
mkdir -p $OUT
for m in scse bam; do
  timeout -k 10 500 python bench.py --model $m --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_$m.log 2>&1 || exit 1
  tail -n 1 $OUT/bench_$m.log
done
timeout -k 10 300 python bench.py --model res34 --size 256 --batch 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_res34_c1.log 2>&1 || exit 1
tail -n 1 $OUT/bench_res34_c1.log
timeout -k 10 600 python scripts/bench_infer.py > $OUT/bench_infer.log 2>&1 || exit 1
tail -n 3 $OUT/bench_infer.log
