#!/bin/bash
# Does test_dgrad_kernel_choice_does_not_depend_on_a_collected_gradient catch the bug it was written for?  Runs it on the build as
# it is, then on a scratch rebuild with round 5's first form of x6p_ok (the x6w exclusion asked with `res` still set), and
# records both.  The source tree on the GPU box is a throw-away copy; only gpurun_out/ comes back.   usage: ... <tag>
set -u
OUT=gpurun_out/$1; mkdir -p "$OUT"
T=tests/test_ops_gpu.py::test_dgrad_kernel_choice_does_not_depend_on_a_collected_gradient
# (one file per stage: two tees on one file - one truncating, one appending - interleaved the first version's output so that the
# second stage's error text sat above the first stage's header)
timeout -k 10 300 python -m pytest "$T" -q -p no:cacheprovider > "$OUT/as_built.txt" 2>&1
grep -E "^E .*(Error|assert)|passed|failed|^FAILED" "$OUT/as_built.txt" | cut -c1-300
H=building_detection_amd/csrc/conv_x6p.h
python - "$H" <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
i = s.index("  if (p.C > 64) {   // (the planes-in kernel declines"); j = s.index("  return true;\n}", i)
s = s[:i] + "  if (p.C > 64 && x6w_plan(p) > 0) return false;   // round 5's first form: depends on p.res through x6w_plan\n" + s[j:]
open(p, "w").write(s)
PY
echo "== rebuilt with the first form of x6p_ok"
timeout -k 10 900 make -C building_detection_amd/csrc > "$OUT/rebuild.log" 2>&1 || { echo "rebuild failed"; exit 1; }
timeout -k 10 300 python -m pytest "$T" -q -p no:cacheprovider > "$OUT/with_first_form.txt" 2>&1
grep -E "^E .*(Error|assert)|passed|failed|^FAILED" "$OUT/with_first_form.txt" | cut -c1-300
true
