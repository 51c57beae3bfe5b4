#!/bin/bash
# round-4 GPU session 30: strip height of the depthwise stencil (SG_DW_FSTRIP_HS = 4 / 8 / 16 / default) - stand-alone and in the step
set -u
OUT=gpurun_out/r4D; mkdir -p $OUT
for hs in 4 8 16; do
  echo "== SG_DW_FSTRIP_HS=$hs" >> $OUT/bw.txt
  SG_DW_FSTRIP_HS=$hs timeout -k 10 300 python scripts/bw_bench.py 2>&1 | grep -i "dw 3x3 fwd\|^ *[0-9]*x\|shape\|==" >> $OUT/bw.txt
done
cat $OUT/bw.txt | cut -c1-150
BB="timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-leg"
run() { name=$1; shift; env "$@" $BB > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "bench $name rc=$?" | tee -a $OUT/summary.txt; }
for rep in 1 2; do
  run hs4_$rep SG_DW_FSTRIP_HS=4
  run hs8_$rep SG_DW_FSTRIP_HS=8
  run hs16_$rep SG_DW_FSTRIP_HS=16
  run run_$rep SG_DW_FSTRIP=0
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4D/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["ms_per_step"], "probe", d["roofline"]["ms_per_step"], "family", d["roofline"]["family"]["frac"])
    except Exception as e: print(f, "unreadable", e)
PY
echo done | tee -a $OUT/summary.txt
