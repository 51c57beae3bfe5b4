#!/bin/bash
# round-4 GPU session 40: world-1 data-parallel form and the other models on the END build
set -u
OUT=gpurun_out/r4N; mkdir -p $OUT
BB="timeout -k 10 500 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; $BB "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
run f32_single
run f32_dp1 --force-dp
run bf16_single --dtype bf16
run bf16_dp1 --dtype bf16 --force-dp
run bam --model bam
run scse --model scse
run hrnet --model hrnet
run res34 --model res34
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4N/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]
        print("%-22s %8.3f ms %8.2f tiles/s probe %s family %s host %s mem %s %s"%(f.split("/")[-1], d["ms_per_step"], d["value"], r.get("ms_per_step"), (r.get("family") or {}).get("frac"), c.get("host_enqueue_ms_per_step"), c.get("peak_device_memory_gib"), (c.get("train_step_choice") or {}).get("chosen")))
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
