#!/bin/bash
# round-4 GPU session 23: pw_wide_kernel<3,float> with the barrier in the middle of the k-step (SG_PW_VAR=1) against the end (0):
# bit identity, the pointwise tests, the 728-wide scan, step A/B (alternating)
set -u
OUT=gpurun_out/r4w; mkdir -p $OUT
SG_PW_WIDE=2 SG_PW_VAR=0 SG_WPW_VAR=0 timeout -k 10 200 python scripts/pw_var_check.py > $OUT/digest_var0.txt 2>&1; echo "digest0 rc=$?" | tee -a $OUT/summary.txt
SG_PW_WIDE=2 SG_PW_VAR=1 SG_WPW_VAR=1 timeout -k 10 200 python scripts/pw_var_check.py > $OUT/digest_var1.txt 2>&1; echo "digest1 rc=$?" | tee -a $OUT/summary.txt
if diff <(grep -- "->" $OUT/digest_var0.txt) <(grep -- "->" $OUT/digest_var1.txt) > $OUT/digest_diff.txt; then echo "digests identical" | tee -a $OUT/summary.txt; else echo "DIGESTS DIFFER" | tee -a $OUT/summary.txt; cat $OUT/digest_diff.txt; fi
grep -c -- "->" $OUT/digest_var1.txt
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -p no:cacheprovider -k "pointwise or pw or wide or conv2d" > $OUT/t_ops.log 2>&1; echo "ops rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/t_ops.log
for rep in 1 2; do for v in 0 1; do
  echo "== SG_PW_VAR=$v rep $rep" >> $OUT/scan.txt
  SG_PW_VAR=$v timeout -k 10 300 python scripts/pw_scan.py 2>&1 | grep -- "728->\|1024->\| 728:\|cout" >> $OUT/scan.txt
done; done
grep -- "== \| 728->  728\|1456->  728\| 728-> 1024" $OUT/scan.txt | cut -c1-130
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run var1_$rep SG_PW_VAR=1
  run var0_$rep SG_PW_VAR=0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4w/bench_var*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "family", d["roofline"]["family"]["frac"], "loss", d["config"]["final_loss"])
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
