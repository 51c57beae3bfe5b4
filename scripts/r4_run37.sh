#!/bin/bash
# round-4 GPU session 37: 512-wide tiles of the wide pointwise kernel where they make one round (1024 columns at 16384 rows):
# bit identity, scan, step A/B (SG_PW_512=1 default / 0)
set -u
OUT=gpurun_out/r4K; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_schedules_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/t.log 2>&1; echo "schedule tests rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/t.log
for rep in 1 2; do for v in 0 1; do
  echo "== SG_PW_512=$v rep $rep" >> $OUT/scan.txt
  SG_PW_512=$v timeout -k 10 300 python scripts/pw_scan.py 2>&1 | grep -- "-> 1024" >> $OUT/scan.txt
done; done
cut -c1-130 $OUT/scan.txt
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run w512_$rep SG_PW_512=1
  run w256_$rep SG_PW_512=0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4K/bench_w*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "family", d["roofline"]["family"]["frac"], "loss", d["config"]["final_loss"])
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
