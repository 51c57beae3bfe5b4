#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes over scripts/dilated_bench.py (ONLY_DILATED=1 ITERS=1: FETCH_SIZE pass,
WRITE_SIZE pass; scripts/gpu_ci.sh pmc) into profiles/<name>.json: fabric-side bytes per launch of every kernel
of the roofline set and their per-step total.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B,
MI355X_MICROARCH.md "HBM") and both counters are KiB.  Note: these are L2-miss (fabric) bytes; Infinity-Cache
hits are counted, so for operands that fit the 256 MiB cache this is an upper bound on HBM traffic.
Use: python scripts/pmc_traffic.py gpurun_out/<tag> profiles/r01_pmc_traffic.json"""
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
# round 2: in the training step the weight planes are prepared ONCE per step for the whole model (sg_prepare_planes), the
# per-launch split3_weights_kernel of this micro-benchmark no longer runs there.  --prepared leaves its bytes out of the
# helpers and adds the six layers' share of the batched preparation instead (4 B read + 12 B written per weight: three bf16
# planes for the forward and three for the dgrad orientation).
PREPARED = "--prepared" in sys.argv
BF16 = "--bf16" in sys.argv   # DTYPE=bf16 passes: conv_b16w / conv_b16 forward and dgrad kernels, every activation 2 bytes
N_WEIGHTS = 3 * 9 * 2048 * 256 + 3 * 9 * 256 * 256
names = ["aspp_d6", "aspp_d12", "aspp_d18", "sk_d6", "sk_d12", "sk_d18"]
out = {}
for kind, mul in (("fetch", 2 * 1024), ("write", 1024)):
    f = glob.glob(f"{src}/pmc_{kind}/*/*counter_collection.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    ci = wi = 0
    per = {"fwd": {}, "dgrad": {}, "wgrad": {}}
    helpers = split = 0.0
    for r in rows:
        k, v = r["Kernel_Name"], float(r["Counter_Value"]) * mul
        if ("igemm_conv_kernel" in k or "conv_x6_kernel" in k or "conv_x6w_kernel" in k or "conv_b16_kernel" in k
                or "conv_b16w_kernel" in k):  # per case: 3 forward launches (1 + warm-up + 1 timed), then 2 dgrad
            per["fwd" if ci % 5 < 3 else "dgrad"].setdefault(ci // 5, []).append(v)
            ci += 1
        elif "igemm_wgrad_kernel" in k or "wgrad_x6_kernel" in k:
            per["wgrad"].setdefault(wi // 2, []).append(v)
            wi += 1
        elif PREPARED and "split3_weights" in k:
            split += v
        elif "copyBuffer" not in k:
            helpers += v  # kernel transpose, split reduce, bias column sum, round 4: x6w_split (activation planes): each runs twice
    o = {t: [sum(x) / len(x) for _, x in sorted(per[t].items())] for t in per}
    o["helpers_per_step"] = helpers / 2
    if PREPARED:
        o["per_launch_split_not_in_the_step"] = split / 2
        o["prepare_share_per_step"] = N_WEIGHTS * (4 if kind == "fetch" else 12)
    out[kind] = o
tot = sum(sum(out[k][t]) for k in out for t in ("fwd", "dgrad", "wgrad")) + out["fetch"]["helpers_per_step"] + out["write"]["helpers_per_step"]
if PREPARED:
    tot += out["fetch"]["prepare_share_per_step"] + out["write"]["prepare_share_per_step"]
MiB = 2 ** 20
alg = (3 * 3 * (128 + 18 + 16) + 3 * 3 * (16 + 2.25 + 16)) * MiB
if BF16:   # each operand once, activations and weight planes in bf16 (the filter gradient itself stays fp32)
    alg = (3 * (2 * (64 + 9 + 8) + (64 + 8 + 18)) + 3 * (2 * (8 + 1.125 + 8) + (8 + 8 + 2.25))) * MiB
json.dump({"source": f"{src}/pmc_fetch + pmc_write: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 scripts/dilated_bench.py (ONLY_DILATED=1 ITERS=1)",
           "correction": "FETCH_SIZE x2 (gfx950) x1024; WRITE_SIZE x1024; fabric-side bytes, Infinity-Cache hits included",
           "dtype": "bf16 storage" if BF16 else "fp32", "cases": names, "per_launch_bytes": out, "set_bytes_per_step": tot, "algorithmic_bytes_per_step": alg}, open(dst, "w"), indent=1)
for t in ("fwd", "dgrad", "wgrad"):
    print(t, " ".join(f"{n}: {out['fetch'][t][i] / MiB:.0f}+{out['write'][t][i] / MiB:.0f}" for i, n in enumerate(names)), "MiB (fetch+write)")
print(f"set total per step {tot / MiB:.0f} MiB; algorithmic {alg / MiB:.0f} MiB")
