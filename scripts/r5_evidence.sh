#!/bin/bash
# Round-5 end-of-round evidence, ONE session ("committed once"): GPU suite, the default bench line, kernel statistics
# of the two-stream step and of the single-stream step, the marker-cut trace of the roofline set, the other four models, the
# world-1 data-parallel form.  Every stage under its own timeout; a stage that times out ends the session.
#   usage: bash scripts/r5_evidence.sh <tag>
set -u
TAG=${1:-r5end}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
stage() {  # name timeout cmd...
  local name=$1 tmo=$2; shift 2
  echo "== stage $name" | tee -a "$OUT/summary.txt"
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"
  local rc=$?
  echo "== stage $name exit $rc" | tee -a "$OUT/summary.txt"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== timed out: aborting" | tee -a "$OUT/summary.txt"; exit $rc; fi
}
stage tests 900 python -m pytest tests -m gpu -q -p no:cacheprovider
tail -n 3 "$OUT/tests.log"
stage bench 600 python bench.py
tail -n 1 "$OUT/bench.log" | cut -c1-400
stage stats2 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rp2" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit
find "$OUT/rp2" -name '*kernel_stats*.csv' -exec cp {} "$OUT/kernel_stats_two_streams.csv" \; ; rm -rf "$OUT/rp2"
SG_SIDE_WGRAD=0 stage stats1 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rp1" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit
find "$OUT/rp1" -name '*kernel_stats*.csv' -exec cp {} "$OUT/kernel_stats_single_stream.csv" \; ; rm -rf "$OUT/rp1"
stage statsbf 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rpb" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 --no-jit
find "$OUT/rpb" -name '*kernel_stats*.csv' -exec cp {} "$OUT/kernel_stats_bf16.csv" \; ; rm -rf "$OUT/rpb"
SG_TRACE_MARK=1 stage trace 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr" -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-bf16-leg --no-jit
python scripts/trace_dilated.py "$OUT/tr" 5 "$OUT/trace_dilated.json" > "$OUT/trace_dilated.txt" 2>&1; rm -rf "$OUT/tr"
tail -n 3 "$OUT/trace_dilated.txt"
for m in bam scse hrnet res34; do
  stage bench_$m 500 python bench.py --model $m --steps 6 --warmup 3 --no-cpu-baseline
  tail -n 1 "$OUT/bench_$m.log" | cut -c1-260
done
stage dp1 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg --force-dp
tail -n 1 "$OUT/dp1.log" | cut -c1-260
stage single 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg
tail -n 1 "$OUT/single.log" | cut -c1-260
stage dp1bf 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 --force-dp
stage singlebf 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16
echo "== done" | tee -a "$OUT/summary.txt"
