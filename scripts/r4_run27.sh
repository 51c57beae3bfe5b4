#!/bin/bash
# round-4 GPU session 27: idle time inside the step (replay and eager forms), from full kernel traces
set -u
OUT=gpurun_out/r4A; mkdir -p $OUT
export TMPDIR=/tmp
for form in jit no-jit; do
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$form -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-bf16-leg --$form > $OUT/bench_$form.json 2> $OUT/bench_$form.err; echo "trace $form rc=$?" | tee -a $OUT/summary.txt
  f=$(find $OUT/tr_$form -name '*kernel_trace.csv' | head -1)
  python scripts/idle_trace.py "$f" 4 > $OUT/idle_$form.txt 2>&1; echo "parse rc=$?" | tee -a $OUT/summary.txt
  python scripts/overlap_trace.py "$f" 4 > $OUT/overlap_$form.txt 2>&1
  find $OUT -name '*kernel_trace*.csv' -delete; find $OUT -name '*.db' -delete
  echo "=== $form"; cat $OUT/idle_$form.txt
done
echo done | tee -a $OUT/summary.txt
