#!/usr/bin/env python3
"""Re-derive bench.py's `roofline.ms_per_step` from a rocprofv3 per-launch kernel trace of the same run.

    SG_TRACE_MARK=1 rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps K --warmup W ...
    python scripts/trace_dilated.py DIR K profiles/rNN_trace_dilated.json [profiles/rNN_trace_dilated_launches.csv]

With SG_TRACE_MARK=1 every launch group that bench.py brackets with HIP events is also bracketed by two empty marker
kernels (sg_trace_mark_kernel<0,0> ... <0,1> = the dilated-convolution set).  This script walks the trace in start
order, sums the durations of the kernels between a begin and an end marker, divides by the number of timed steps and
writes (a) the per-step sum, per kernel name, and (b) optionally the individual launches (name, grid, duration) of the
first timed step so that each of the 18 launch groups (6 convs x fwd / wgrad / dgrad) can be read off."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short_name(name: str) -> str:
    """'void (anonymous namespace)::conv_x6_kernel<128, 2, 4, 1>(Args...)' -> 'conv_x6_kernel<128, 2, 4, 1>'"""
    n = name.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n)
    depth, out = 0, []
    for ch in n:  # cut at the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()[:100]

src, steps, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
rows_out = sys.argv[4] if len(sys.argv) > 4 else None
f = sorted(glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))))
rows.sort()
inside = False
groups = []  # list of lists of (name, grid, wg, dur_ns)
for s, e, name, grid, wg in rows:
    if "sg_trace_mark_kernel<0, 0>" in name or "sg_trace_mark_kernel<0,0>" in name:
        inside, cur = True, []
        continue
    if "sg_trace_mark_kernel<0, 1>" in name or "sg_trace_mark_kernel<0,1>" in name:
        if inside:
            groups.append(cur)
        inside = False
        continue
    if inside and "sg_trace_mark_kernel" not in name:
        cur.append((name, grid, wg, e - s))
# bench.py runs 2 more steps after the timed ones with EVERY conv bracketed (roofline.family): they carry the same
# dilated markers, so the trace holds steps + 2 steps' worth of groups; only the first `steps` are the timed region
per_step_groups = len(groups) // (steps + 2) if len(groups) % (steps + 2) == 0 else len(groups) // steps
groups = groups[:per_step_groups * steps]
by_name = defaultdict(lambda: [0, 0])
tot = 0
for g in groups:
    for name, grid, wg, d in g:
        short = short_name(name)
        by_name[short][0] += 1
        by_name[short][1] += d
        tot += d
out = {"source": f"{f} (rocprofv3 --kernel-trace, SG_TRACE_MARK=1), {steps} timed steps",
       "groups_per_step": per_step_groups, "launches_per_step": sum(v[0] for v in by_name.values()) / steps,
       "ms_per_step": tot / steps / 1e6,
       "kernels": {k: {"calls_per_step": v[0] / steps, "ms_per_step": v[1] / steps / 1e6, "avg_us": v[1] / v[0] / 1e3}
                   for k, v in sorted(by_name.items(), key=lambda kv: -kv[1][1])}}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("groups_per_step", "launches_per_step", "ms_per_step")}))
if rows_out:
    with open(rows_out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["group", "kernel", "grid", "workgroup", "duration_us"])
        for gi, g in enumerate(groups[:per_step_groups]):
            for name, grid, wg, d in g:
                w.writerow([gi, short_name(name), grid, wg, round(d / 1e3, 2)])
