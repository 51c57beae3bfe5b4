#!/usr/bin/env python3
"""Matrix-pipe and LDS counters of the roofline kernel set from the two `scripts/gpu_ci.sh <tag> pmc2` passes.
Per kernel family (x6 forward/dgrad, x6 wgrad): MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (chip cycles x 1024
SIMDs), the LDS bank-conflict share of LDS-active cycles, and the bf16 MFMA operation count against the launch's
nominal FLOPs.  Units (MI355X_MICROARCH.md, "DVFS give-back" and the PMC unit table): the SQ counters are sums over
the chip's SIMDs; rocprofv3 reports GRBM_GUI_ACTIVE as the SUM OVER THE 8 XCDs, so chip cycles = GRBM_GUI_ACTIVE / 8
(rounds 1-4 divided by the raw sum and printed shares 8x too small - VERDICT r4 weak #4).
Use: python scripts/pmc_mfma.py gpurun_out/<tag> profiles/r05_pmc_mfma.json
     python scripts/pmc_mfma.py --renormalise profiles/r04_pmc_mfma.json   (recompute the shares of a committed record from its raw block)"""
import csv
import glob
import json
import sys
from collections import defaultdict

XCDS = 8
SIMDS = 256 * 4


def shares(c, e):
    if c.get("GRBM_GUI_ACTIVE"):
        cyc = c["GRBM_GUI_ACTIVE"] / XCDS
        e["chip_cycles"] = cyc
        e["mfma_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * SIMDS), 4)
        e["sq_busy_share"] = round(c["SQ_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"], 4) if c.get("SQ_BUSY_CYCLES") else None
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        e["bf16_mfma_flops"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512
    return e


NOTE = ("MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) - rocprofv3 sums GRBM_GUI_ACTIVE "
        "over the 8 XCDs; sq_busy_share = SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE (both per-XCD sums; a value of 3.6 = the SQ's "
        "shader engines counted separately); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")

if sys.argv[1] == "--renormalise":
    rec = json.load(open(sys.argv[2]))
    rec["note"] = NOTE + "; shares recomputed from the raw block (the first print divided by the 8-XCD sum: 8x too small)"
    for fam, e in rec["kernels"].items():
        shares(e["raw"], e)
        print(fam, {k: v for k, v in e.items() if k != "raw"})
    json.dump(rec, open(sys.argv[2], "w"), indent=1)
    sys.exit(0)

src, dst = sys.argv[1], sys.argv[2]
bench = sys.argv[3] if len(sys.argv) > 3 else "scripts/dilated_bench.py (ONLY_DILATED=1 ITERS=1)"
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
       "note": NOTE,
       "kernels": {}}
for fam, c in agg.items():
    e = {"launches": calls[fam]}
    shares(c, e)
    e["raw"] = dict(c)
    out["kernels"][fam] = e
    print(fam, {k: v for k, v in e.items() if k != "raw"})
json.dump(out, open(dst, "w"), indent=1)
