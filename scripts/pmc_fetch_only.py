#!/usr/bin/env python3
"""Fabric-side fetch bytes per launch of the dilated-set kernels from ONE rocprofv3 pass (--pmc FETCH_SIZE over
scripts/dilated_bench.py with ONLY_DILATED=1 ITERS=1): the quick form of scripts/pmc_traffic.py for tile-order experiments.
Use: python scripts/pmc_fetch_only.py <rocprof output dir>"""
import csv
import glob
import sys

f = glob.glob(f"{sys.argv[1]}/*/*counter_collection.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
ci = wi = 0
per = {"fwd": {}, "dgrad": {}, "wgrad": {}}
helpers = 0.0
for r in rows:
    k, v = r["Kernel_Name"], float(r["Counter_Value"]) * 2 * 1024
    if "conv_x6_kernel" in k or "igemm_conv_kernel" in k:
        per["fwd" if ci % 5 < 3 else "dgrad"].setdefault(ci // 5, []).append(v)
        ci += 1
    elif "wgrad_x6_kernel" in k or "igemm_wgrad_kernel" in k:
        per["wgrad"].setdefault(wi // 2, []).append(v)
        wi += 1
    elif "copyBuffer" not in k and "split3" not in k:
        helpers += v
tot = 0.0
for t in per:
    vals = [sum(x) / len(x) / 2 ** 20 for _, x in sorted(per[t].items())]
    tot += sum(vals)
    print(t, " ".join(f"{v:.0f}" for v in vals), "MiB fetched per launch")
print(f"helpers {helpers / 2 / 2 ** 20:.0f} MiB; fetch total {tot + helpers / 2 / 2 ** 20:.0f} MiB")
