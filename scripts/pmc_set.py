#!/usr/bin/env python3
"""Round 5: counters of the roofline kernel set as the step runs it (scripts/dilated_step.py, ITERS=1 = two runs of the set), summed
over EVERY kernel of the script by name and halved - no positional bookkeeping.  From `scripts/gpu_ci.sh <tag> pmc5`:
  pmc_fetch / pmc_write  -> profiles/r05_pmc_traffic.json  (FETCH_SIZE x 2 for gfx950's 64-byte tally of 128-byte requests, both
                            counters in KiB: MI355X_MICROARCH.md "HBM"; fabric side, Infinity-Cache hits included)
  pmc_mfma / pmc_lds     -> profiles/r05_pmc_mfma.json     (executed bf16 MFMA FLOPs = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512; MFMA busy
                            share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); LDS conflict share)
Use: python scripts/pmc_set.py gpurun_out/<tag> profiles/r05_pmc_traffic.json profiles/r05_pmc_mfma.json"""
import csv
import glob
import json
import sys
from collections import defaultdict

src, dst_t, dst_m = sys.argv[1], sys.argv[2], sys.argv[3]
MiB = 2 ** 20


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0][:70]


def by_kernel(sub):
    f = glob.glob(f"{src}/{sub}/*/*counter_collection.csv")[0]
    agg = defaultdict(lambda: defaultdict(float))
    n = defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "copyBuffer" in k or "fill" in k.lower() and "Kernel" in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in n.items()}


fetch, nf = by_kernel("pmc_fetch")
write, _ = by_kernel("pmc_write")
per = {}
tot = 0.0
for k in sorted(set(fetch) | set(write)):
    fb = fetch[k].get("FETCH_SIZE", 0.0) * 2 * 1024 / 2
    wb = write[k].get("WRITE_SIZE", 0.0) * 1024 / 2
    per[k] = {"launches_per_step": nf.get(k, 0) / 2, "fetch_bytes_per_step": fb, "write_bytes_per_step": wb}
    tot += fb + wb
    print(f"{k:72s} {nf.get(k, 0) / 2:5.1f} launches  fetch {fb / MiB:8.1f} MiB  write {wb / MiB:8.1f} MiB")
# each operand once: ASPP conv 128 MiB x + 18 MiB w + 16 MiB y per pass, SK conv 16 + 2.25 + 16 MiB; three passes
alg = (3 * 3 * (128 + 18 + 16) + 3 * 3 * (16 + 2.25 + 16)) * MiB
# the weight planes are prepared once per step for the whole model (sg_prepare_planes), not by this script's per-launch split:
# the six layers' share of that launch is added, the script's own split3_weights launches are left out
N_WEIGHTS = 3 * 9 * 2048 * 256 + 3 * 9 * 256 * 256
split = sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for k, v in per.items() if "split3_weights" in k)
prep = N_WEIGHTS * (4 + 12)
set_bytes = tot - split + prep
json.dump({"source": f"{src}/pmc_fetch + pmc_write: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 scripts/dilated_step.py (ITERS=1: the set twice, sums halved)",
           "correction": "FETCH_SIZE x2 (gfx950) x1024; WRITE_SIZE x1024; fabric-side bytes, Infinity-Cache hits included",
           "dtype": "fp32", "per_kernel": per, "per_launch_weight_split_not_in_the_step": split, "prepare_share_per_step": prep,
           "set_bytes_per_step": set_bytes, "algorithmic_bytes_per_step": alg, "ratio": set_bytes / alg}, open(dst_t, "w"), indent=1)
print(f"set total per step {set_bytes / MiB:.0f} MiB = {set_bytes / alg:.2f} x the algorithmic {alg / MiB:.0f} MiB")

mf, nm = by_kernel("pmc_mfma")
ld, _ = by_kernel("pmc_lds")
XCDS, SIMDS = 8, 1024
kern = {}
mops = busy = gui = 0.0
for k in sorted(set(mf) | set(ld)):
    c = dict(mf.get(k, {}))
    c.update(ld.get(k, {}))
    e = {"launches_per_step": nm.get(k, 0) / 2, "raw_two_runs": c}
    if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        e["mfma_busy_share"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / XCDS * SIMDS), 4)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"):
        e["bf16_mfma_flops_per_step"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / 2
        mops += e["bf16_mfma_flops_per_step"]
        busy += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        gui += c.get("GRBM_GUI_ACTIVE", 0.0)
    kern[k] = e
    print(k, {a: b for a, b in e.items() if a != "raw_two_runs"})
nominal = 97.84e9 * 16 * 6   # bf16 MFMA FLOPs of the set at six passes per fp32 product, nominal (padding taps counted)
json.dump({"source": f"{src}/pmc_mfma + pmc_lds: rocprofv3 --kernel-trace --pmc ... -- python3 scripts/dilated_step.py (ITERS=1)",
           "note": "MFMA busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs",
           "kernels": kern, "bf16_mfma_flops_per_step": mops, "executed_share_of_nominal": mops / nominal,
           "mfma_busy_share_of_the_mfma_kernels": round(busy / (gui / XCDS * SIMDS), 4) if gui else None}, open(dst_m, "w"), indent=1)
print(f"executed bf16 MFMA FLOPs per step {mops:.4e} = {mops / nominal:.4f} of nominal; MFMA busy share of the matrix kernels {busy / (gui / XCDS * SIMDS) if gui else 0:.3f}")
