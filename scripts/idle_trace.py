#!/usr/bin/env python3
"""Where is the GPU idle inside a training step?  From a rocprofv3 --kernel-trace CSV of bench.py: one steady-state step (between
two adam_kernel launches, the third from the end by default), the union of all kernels' intervals over all queues, the total idle
time, the idle time by gap size, the largest gaps with the kernels on either side, and the time during which only the second queue
ran.  Use: python scripts/idle_trace.py <kernel_trace.csv> [steps back from the end]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = adam[-back - 1], adam[-back]
step = rows[a + 1:b + 1]
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], r["Queue_Id"]) for r in step)
t0, t1 = iv[0][0], max(e for _, e, _, _ in iv)
byq = defaultdict(float)
for s, e, _, q in iv:
    byq[q] += e - s
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, {len(iv)} dispatches; kernel time per queue: " + ", ".join(f"{q}: {v / 1e6:.2f} ms" for q, v in byq.items()))
gaps = []
cur_end, last_name = iv[0][1], iv[0][2]
for s, e, n, q in iv[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, last_name, n))
    if e > cur_end:
        cur_end, last_name = e, n
idle = sum(g for g, _, _ in gaps)
print(f"idle (no kernel on any queue): {idle / 1e6:.3f} ms in {len(gaps)} gaps; mean gap {idle / max(len(gaps), 1) / 1e3:.2f} us")
for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)):
    sel = [g for g, _, _ in gaps if lo * 1e3 <= g < hi * 1e3]
    print(f"  gaps {lo}-{hi if hi < 1e9 else 'inf'} us: {len(sel)} gaps, {sum(sel) / 1e6:.3f} ms")
print("largest gaps:")
for g, before, after in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:7.1f} us   after {before}   before {after}")
