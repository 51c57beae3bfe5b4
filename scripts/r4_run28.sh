#!/bin/bash
# round-4 GPU session 28: bf16 wide pointwise kernel with the barrier inside the stage: bit identity (schedule test), bf16 tests, bf16 step A/B
set -u
OUT=gpurun_out/r4B; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_schedules_gpu.py tests/test_bf16_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/t.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -4 $OUT/t.log
BB="timeout -k 10 400 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run var1_$rep SG_PW_VAR=1
  run var0_$rep SG_PW_VAR=0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4B/bench_var*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "family", d["roofline"]["family"]["frac"], "loss", d["config"]["final_loss"], (d["config"].get("train_step_choice") or {}).get("chosen"))
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
