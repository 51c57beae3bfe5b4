#!/bin/bash
# bench.py --jit: are the family / dilated-set figures of a replayed run in line with an eager run's, now that one untimed eager step
# precedes the bracketed ones?  Also the data-parallel form (world 1) with the replay.   usage: ... <tag>
set -u
OUT=gpurun_out/$1; mkdir -p "$OUT"; cd "$(dirname "$0")/.."
show() { python - "$1" "$2" <<'PY'
import sys, json
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = j["roofline"]
print(f"{sys.argv[2]:22s} ms_per_step {j['ms_per_step']:7.3f}  family {r['family']['ms_per_step']:7.3f} ms frac {r['family']['frac']}  dilated set {r['ms_per_step']:6.3f} ms frac {r['frac']}  step: {j['config']['train_step'][:28]}")
PY
}
for name in "no-jit:--no-jit" "jit:--jit" "jit_again:--jit" "dp1_jit:--force-dp --jit"; do
  n=${name%%:*}; a=${name#*:}
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg $a > "$OUT/$n.json" 2>> "$OUT/err.txt" || { echo "$n failed"; tail -5 "$OUT/err.txt"; exit 1; }
  show "$OUT/$n.json" "$n" | tee -a "$OUT/summary.txt"
done
