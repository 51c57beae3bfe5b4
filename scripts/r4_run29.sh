#!/bin/bash
# round-4 GPU session 29: the depthwise stencil as column strips (dw_strip_kernel, SG_DW_FSTRIP=1) against the run kernel (0):
# bit identity, the depthwise / BatchNormalization-sums / block-chain tests, stand-alone timing, step A/B (alternating)
set -u
OUT=gpurun_out/r4C; mkdir -p $OUT
SG_DW_FSTRIP=0 timeout -k 10 200 python scripts/dw_var_check.py > $OUT/digest_0.txt 2>&1; echo "digest0 rc=$?" | tee -a $OUT/summary.txt
SG_DW_FSTRIP=1 timeout -k 10 200 python scripts/dw_var_check.py > $OUT/digest_1.txt 2>&1; echo "digest1 rc=$?" | tee -a $OUT/summary.txt
if diff <(grep "x" $OUT/digest_0.txt | grep ":") <(grep "x" $OUT/digest_1.txt | grep ":") > $OUT/digest_diff.txt; then echo "digests identical ($(grep -c ':' $OUT/digest_1.txt) lines)" | tee -a $OUT/summary.txt; else echo "DIGESTS DIFFER" | tee -a $OUT/summary.txt; head -20 $OUT/digest_diff.txt; tail -3 $OUT/digest_1.txt; fi
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_block_chains_gpu.py -x -q -m gpu -p no:cacheprovider -k "depthwise or dw or separable or sums or chain or batchnorm" > $OUT/t_ops.log 2>&1; echo "ops rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/t_ops.log
for v in 0 1; do
  echo "== SG_DW_FSTRIP=$v" >> $OUT/bw.txt
  SG_DW_FSTRIP=$v timeout -k 10 300 python scripts/bw_bench.py 2>&1 | grep -i "dw\|depthwise\|copy" >> $OUT/bw.txt
done
cat $OUT/bw.txt | cut -c1-150
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run strip1_$rep SG_DW_FSTRIP=1
  run strip0_$rep SG_DW_FSTRIP=0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4C/bench_strip*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "family", d["roofline"]["family"]["frac"], "loss", d["config"]["final_loss"])
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
