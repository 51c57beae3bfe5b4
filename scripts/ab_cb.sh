#!/bin/bash
# A/B of SG_CONV_CB (32-deep slabs per channel block of the K order) on the dilated-conv micro-benchmark: timing interleaved
# over two rounds, then one FETCH_SIZE pass per variant (fabric-side read bytes summed over the set's kernels).
OUT=gpurun_out/${1:-abcb}
mkdir -p $OUT
export ONLY_DILATED=1
for round in 1 2; do
  for v in ${VARIANTS:-1 2 4}; do
    echo "== round $round SG_CONV_CB=$v" >> $OUT/ab.log
    ITERS=30 SG_CONV_CB=$v timeout -k 10 300 python scripts/dilated_bench.py >> $OUT/ab.log 2>&1 || exit 1
  done
done
grep -v amdgpu.ids $OUT/ab.log | grep "==\|dilated set\|aspp\|sk"
export TMPDIR=/tmp
for v in ${VARIANTS:-1 2 4}; do
  SG_CONV_CB=$v ITERS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cb_$v -- python3 scripts/dilated_bench.py > $OUT/pmc_cb_$v.log 2>&1 || exit 1
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/pmc_cb_$v/*/*counter_collection.csv")[0]
tot = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    key = "conv" if "conv_x6_kernel" in k else ("wgrad" if "wgrad_x6_kernel" in k else "other")
    tot[key] = tot.get(key, 0.0) + float(r["Counter_Value"]) * 2048
print("SG_CONV_CB=$v fetch MiB (all launches of the run):", {k: round(v / 2**20) for k, v in tot.items()})
PY
done
echo done
