#!/bin/bash
# Round-2 evidence run on one GPU box: full parity suite, smoke, the headline bench in both storage modes, and the
# rocprofv3 kernel trace + stats of the same command (fp32 with SG_TRACE_MARK=1 so that scripts/trace_dilated.py can cut the
# roofline kernel set out of the trace).  usage: scripts/r02_final.sh <tag>
set -u
TAG=${1:-R2final}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
step() { echo "== $*" | tee -a "$OUT/summary.txt"; }
step tests; timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/tests.log" 2>&1; rc=$?; tail -n 3 "$OUT/tests.log"
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
step smoke; timeout -k 10 300 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1 || exit 1; tail -n 2 "$OUT/smoke.log"
step bench_f32; timeout -k 10 600 python bench.py > "$OUT/bench_f32.json" 2> "$OUT/bench_f32.err" || exit 1
step bench_bf16; timeout -k 10 600 python bench.py --dtype bf16 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err" || exit 1
cd /tmp
step prof_f32; SG_TRACE_MARK=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_f32" -o f32 -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline > "$OUT/prof_f32.log" 2>&1 || exit 1
step prof_bf16; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bf16" -o bf16 -- python3 "$ROOT/bench.py" --steps 6 --warmup 2 --dtype bf16 --no-cpu-baseline > "$OUT/prof_bf16.log" 2>&1 || exit 1
cd "$ROOT"
step trace_dilated; python scripts/trace_dilated.py "$OUT/prof_f32" 6 "$OUT/trace_dilated.json" "$OUT/trace_dilated_launches.csv" > "$OUT/trace_dilated.log" 2>&1; tail -n 4 "$OUT/trace_dilated.log"
gzip -f "$OUT"/prof_f32/*kernel_trace.csv 2>/dev/null
rm -f "$OUT"/prof_bf16/*kernel_trace.csv
python - <<PY
import json
for f in ("f32", "bf16"):
    d = json.loads(open("$OUT/bench_%s.json" % f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_step"], d["roofline"].get("family", {}).get("frac"),
          d["config"].get("host_enqueue_ms_per_step"))
PY
step done
