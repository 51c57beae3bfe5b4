#!/usr/bin/env python3
"""How much of a training step runs on two streams at once: from a rocprofv3 --kernel-trace CSV of bench.py, takes the
dispatches between two consecutive adam_kernel launches (one steady-state step of the timed region: the third from the end
by default, bench.py's last two steps are the bracketed single-stream ones) and prints the wall time, the summed kernel
time per queue, the time during which kernels of BOTH queues were running, and per kernel family the mean duration
(to compare with a SG_SIDE_WGRAD=0 trace).  Use: python scripts/overlap_trace.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3   # bench.py: the last two steps are the bracketed (single-stream) ones
a, b = adam[-back - 1], adam[-back]
step = rows[a + 1:b + 1]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
byq = defaultdict(list)
for r in step:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, {len(step)} dispatches on {len(byq)} queue(s)")
for q, v in byq.items():
    print(f"  queue {q}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) / 1e6:.2f} ms")
if len(byq) >= 2:
    qs = sorted(byq, key=lambda q: -len(byq[q]))
    main, side = byq[qs[0]], byq[qs[1]]
    ev = []
    for s, e, _ in main:
        ev += [(s, 0, 1), (e, 0, -1)]
    for s, e, _ in side:
        ev += [(s, 1, 1), (e, 1, -1)]
    ev.sort()
    cnt = [0, 0]
    last = ev[0][0]
    both = only_side = 0
    for t, w, d in ev:
        if cnt[0] > 0 and cnt[1] > 0:
            both += t - last
        elif cnt[1] > 0:
            only_side += t - last
        last = t
        cnt[w] += d
    print(f"  both queues busy {both / 1e6:.2f} ms, side queue alone {only_side / 1e6:.2f} ms")
    # what runs on the main queue while the side queue is busy
    fam = defaultdict(float)
    for s, e, n in main:
        ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, _ in side if s2 < e and e2 > s)
        key = re.sub(r"\(anonymous namespace\)::", "", n)
        key = re.sub(r"[<(].*$", "", key).replace("void ", "")
        fam[key] += ov
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:10]:
        print(f"    main-queue time beside a side kernel: {v / 1e6:6.2f} ms  {k}")
dur = defaultdict(lambda: [0, 0])
for r in step:
    key = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    key = re.sub(r"\(.*$", "", key).replace("void ", "")[:70]
    dur[key][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    dur[key][1] += 1
for k, (t, n) in sorted(dur.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"  {t / 1e6:6.2f} ms {n:4d} x {t / n / 1e3:7.1f} us  {k}")
