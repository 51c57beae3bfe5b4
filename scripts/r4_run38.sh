#!/bin/bash
# round-4 GPU session 38: tile width of the wide pointwise kernel by rounds x width: 1024 columns on 512-wide (default) / 256-wide
# (SG_PW_512=0) / 384-wide (SG_PW_WIDE=3) tiles; bit identity; step A/B
set -u
OUT=gpurun_out/r4L; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_schedules_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/t.log 2>&1; echo "schedule tests rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/t.log
for rep in 1 2; do for v in "SG_PW_WIDE=3" "SG_PW_512=0" "SG_PW_512=1"; do
  echo "== $v rep $rep" >> $OUT/scan.txt
  env $v timeout -k 10 300 python scripts/pw_scan.py 2>&1 | grep -- "-> 1024\|->  728" >> $OUT/scan.txt
done; done
grep -- "==\| 728-> 1024\|1024-> 1024\|2048-> 1024\| 728->  728" $OUT/scan.txt | cut -c1-130
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run w512_$rep SG_PW_512=1
  run w256_$rep SG_PW_512=0
  run w384_$rep SG_PW_WIDE=3
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4L/bench_w*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "family", d["roofline"]["family"]["frac"], "loss", d["config"]["final_loss"])
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
