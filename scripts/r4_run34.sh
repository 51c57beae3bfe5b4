#!/bin/bash
# round-4 GPU session 34: 256-wide tiles of the wide pointwise kernel (1024 / 2048 / 256 columns): bit identity against the
# 128-wide kernels and between the barrier placements, conv tests, pointwise scan, step A/B (SG_PW_WIDE=1 default / 3 = 384 only)
set -u
OUT=gpurun_out/r4H; mkdir -p $OUT
SG_PW_WIDE=3 timeout -k 10 300 python scripts/pw_bn_check.py > $OUT/dig_w3.txt 2>&1; echo "dig w3 rc=$?" | tee -a $OUT/summary.txt
SG_PW_WIDE=1 timeout -k 10 300 python scripts/pw_bn_check.py > $OUT/dig_w1.txt 2>&1; echo "dig w1 rc=$?" | tee -a $OUT/summary.txt
SG_PW_WIDE=1 SG_PW_VAR=0 timeout -k 10 300 python scripts/pw_bn_check.py > $OUT/dig_w1v0.txt 2>&1; echo "dig w1 var0 rc=$?" | tee -a $OUT/summary.txt
for a in dig_w1 dig_w1v0; do if diff <(grep -- "->" $OUT/dig_w3.txt) <(grep -- "->" $OUT/$a.txt) > $OUT/$a.diff; then echo "$a identical to the 128-wide kernels ($(grep -c -- '->' $OUT/$a.txt) lines)" | tee -a $OUT/summary.txt; else echo "$a DIFFERS" | tee -a $OUT/summary.txt; cat $OUT/$a.diff | head; fi; done
tail -2 $OUT/dig_w1.txt
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py tests/test_schedules_gpu.py -x -q -m gpu -p no:cacheprovider > $OUT/t.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/t.log
for rep in 1 2; do for v in 3 1; do
  echo "== SG_PW_WIDE=$v rep $rep" >> $OUT/scan.txt
  SG_PW_WIDE=$v timeout -k 10 300 python scripts/pw_scan.py 2>&1 | grep -- "->" >> $OUT/scan.txt
done; done
grep -- "== \|-> 1024\|->  256" $OUT/scan.txt | cut -c1-130
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run w1_$rep SG_PW_WIDE=1
  run w3_$rep SG_PW_WIDE=3
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4H/bench_w*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "family", d["roofline"]["family"]["frac"], "loss", d["config"]["final_loss"])
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
