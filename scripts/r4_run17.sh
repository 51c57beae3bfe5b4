#!/bin/bash
# round-4 GPU session 17: the woven conv_x6w in the training step (SG_X6W_VAR A/B, alternating), then the counter passes of the
# dilated set with the final kernels (fabric traffic; matrix-pipe / LDS counters)
set -u
OUT=gpurun_out/r4q; mkdir -p $OUT
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run var1_$rep SG_X6W_VAR=1
  run var0_$rep SG_X6W_VAR=0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4q/bench_var*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "frac", d["roofline"]["frac"], "family", d["roofline"]["family"]["frac"], "jit", d["config"].get("captured_step"))
    except Exception as e: print(f, "unreadable", e)
PY
bash scripts/gpu_ci.sh r4q pmc pmc2 > $OUT/pmc_stages.log 2>&1; echo "pmc rc=$?" | tee -a $OUT/summary.txt
python scripts/pmc_traffic.py $OUT $OUT/pmc_traffic.json --prepared > $OUT/pmc_traffic.txt 2>&1; echo "traffic parse rc=$?" | tee -a $OUT/summary.txt
cat $OUT/pmc_traffic.txt
# keep the counter tables small: the per-dispatch csv of the four passes (a few hundred rows each)
find $OUT -name '*kernel_trace*.csv' -delete; find $OUT -name '*.db' -delete
du -sh $OUT; echo done | tee -a $OUT/summary.txt
