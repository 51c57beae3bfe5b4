#!/usr/bin/env python3
"""Matrix-pipe and LDS counters of the roofline kernel set from the two `scripts/gpu_ci.sh <tag> pmc2` passes.
Per kernel family (x6 forward/dgrad, x6 wgrad): MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x
SIMDs) as rocprofv3's own MfmaUtil expression defines it (SQ counters are sums over the chip's SIMDs), the LDS
bank-conflict share of LDS-active cycles, and the bf16 MFMA operation count against the launch's nominal FLOPs.
Use: python scripts/pmc_mfma.py gpurun_out/<tag> profiles/r01_pmc_mfma.json"""
import csv
import glob
import json
import sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
bench = sys.argv[3] if len(sys.argv) > 3 else "scripts/dilated_bench.py (ONLY_DILATED=1 ITERS=1)"
SIMDS = 256 * 4
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for sub in ("pmc_mfma", "pmc_lds"):
    f = glob.glob(f"{src}/{sub}/*/*counter_collection.csv")[0]
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = ("x6_fwd_dgrad" if ("conv_x6_kernel" in k or "conv_x6w_kernel" in k) else "x6_wgrad" if "wgrad_x6_kernel" in k
               else "x6_patch" if "conv_x6p_kernel" in k else None)
        if fam is None:
            continue
        agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
        if sub == "pmc_mfma" and r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            calls[fam] += 1
out = {"source": f"{src}/pmc_mfma + pmc_lds: rocprofv3 --kernel-trace --pmc ... -- python3 {bench}",
       "note": "MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
       "kernels": {}}
for fam, c in agg.items():
    e = {"launches": calls[fam]}
    if c.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * SIMDS), 4)
        e["sq_busy_share"] = round(c["SQ_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"], 4) if c.get("SQ_BUSY_CYCLES") else None
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        e["bf16_mfma_flops"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
    e["raw"] = dict(c)
    out["kernels"][fam] = e
    print(fam, {k: v for k, v in e.items() if k != "raw"})
json.dump(out, open(dst, "w"), indent=1)
