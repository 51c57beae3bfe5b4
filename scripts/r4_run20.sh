#!/bin/bash
# round-4 GPU session 20: the planes-in kernel on the non-dilated long-K 3x3 convolutions too (SG_X6_WIDE=2)? step A/B, alternating
set -u
OUT=gpurun_out/r4t; mkdir -p $OUT
timeout -k 10 400 env SG_X6_WIDE=2 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider -k "conv" > $OUT/t_ops.log 2>&1; echo "ops rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/t_ops.log
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run wide2_$rep SG_X6_WIDE=2
  run wide1_$rep SG_X6_WIDE=1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4t/bench_wide*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "family", d["roofline"]["family"]["frac"], "loss", d["config"]["final_loss"])
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
