#!/bin/bash
# A/B of one environment switch through bench.py with the step form PINNED (LAB_NOTEBOOK 12.4: the default line picks eager or
# replay per run, which confounds an A/B).  Alternates the two settings, both forms, keeps every JSON line.
#   usage: bash scripts/ab_pinned.sh <tag> <ENV_NAME> <value_a> <value_b> [repeats=2]
set -u
TAG=$1; VAR=$2; A=$3; B=$4; REP=${5:-2}
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd "$(dirname "$0")/.."
for form in --no-jit --jit; do
  for r in $(seq 1 "$REP"); do
    for v in "$A" "$B"; do
      f="$OUT/bench_${VAR}_${v}_${form#--}_$r.json"
      env "$VAR=$v" timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-leg $form > "$f" 2>> "$OUT/err.txt" || { echo "bench failed: $f" | tee -a "$OUT/ab.txt"; exit 1; }
      python - "$f" "$VAR" "$v" "$form" <<'PY' | tee -a "$OUT/ab.txt"
import sys, json
f, var, v, form = sys.argv[1:5]
j = json.loads([l for l in open(f) if l.startswith("{")][-1])
r = j["roofline"]
print(f"{var}={v} {form:8s} ms_per_step {j['ms_per_step']:7.3f}  family {r['family']['ms_per_step']:7.3f} ms  dilated set {r['ms_per_step']:6.3f} ms  "
      f"final_loss {j['config']['final_loss']:.9f}  step: {j['config']['train_step']}")
PY
    done
  done
done
