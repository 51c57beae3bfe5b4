#!/usr/bin/env python3
"""Bandwidth kernels of the training step one by one (BatchNormalization apply / backward, add, depthwise, copy), at the
step's own tensor shapes, as hipGraph replays of ROT launches over ROT distinct buffer sets (ROT = 1: every launch on the same,
cache-warm buffers; ROT = 8: ~0.4-1.5 GB rotate, nothing survives in the 256 MB Infinity Cache).  Prints the time per launch and the
algorithmic bytes / time.  torch's own copy_ / add on the same buffers is the yardstick for what this size of tensor can reach.
   DTYPE=f32|bf16 ROT=8 python scripts/bw_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine, _ptr, _dt  # noqa: E402

e = get_engine(0)
bf = os.environ.get("DTYPE", "f32") == "bf16"
ROT = int(os.environ.get("ROT", "8"))
dt = torch.bfloat16 if bf else torch.float32
es = 2 if bf else 4
SHAPES = [(16, 32, 32, 728), (16, 64, 64, 256), (16, 128, 128, 128), (16, 256, 256, 64), (16, 32, 32, 2048)]
if os.environ.get("ONLY"):
    SHAPES = SHAPES[:int(os.environ["ONLY"])]


def timed(fns, reps=6):
    """fns: ROT closures (one per buffer set).  Capture them in one graph, replay reps times."""
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fns:
            f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for f in fns:
                f()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * len(fns)) * 1e3


print(f"dtype={'bf16' if bf else 'f32'} ROT={ROT}")
for shp in SHAPES:
    n, h, w, c = shp
    rows = n * h * w
    nbytes = rows * c * es
    sets = []
    for i in range(ROT):
        x = torch.randn(*shp, device="cuda").to(dt)
        dy = torch.randn(*shp, device="cuda").to(dt)
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        sets.append((x, dy, y, dx))
    gamma = torch.rand(c, device="cuda") + 0.5
    beta = torch.randn(c, device="cuda") * 0.1
    mm, mv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    _, mean, invstd = e.bn_train_fwd(sets[0][0], gamma, beta, mm, mv, relu=True, out=sets[0][2])
    dg, db = e.empty(c), e.empty(c)
    wdw = torch.randn(3, 3, c, 1, device="cuda") * 0.2
    dd = e.conv_desc(shp, c, 3, 3, 1, 1, "same")

    res = {}
    res["torch copy (1r+1w)"] = (timed([lambda s=s: s[2].copy_(s[0]) for s in sets]), 2)
    res["torch add (2r+1w)"] = (timed([lambda s=s: torch.add(s[0], s[1], out=s[2]) for s in sets]), 3)
    res["bn_fwd stats+apply (2r+1w)"] = (timed([lambda s=s: e.bn_train_fwd(s[0], gamma, beta, mm, mv, relu=True, out=s[2]) for s in sets]), 3)
    res["bn_apply (1r+1w)"] = (timed([lambda s=s: e.lib.sg_bn_apply(e.h, e.stream, _dt(s[0]), rows, c, _ptr(s[0]), _ptr(gamma), _ptr(beta), _ptr(mean),
                                                                   _ptr(invstd), _ptr(s[2]), 1)
                                      for s in sets]), 2)
    res["bn_bwd reduce+apply (4r+1w)"] = (timed([lambda s=s: e.bn_train_bwd(s[0], None, s[1], gamma, mean, invstd, relu=True, out=s[3], dgamma=dg,
                                                                            dbeta=db, beta=beta) for s in sets]), 5)
    res["add_n 2 (2r+1w)"] = (timed([lambda s=s: e.add_n([s[0], s[1]], out=s[2]) for s in sets]), 3)
    res["dw 3x3 fwd (1r+1w)"] = (timed([lambda s=s: e.dwconv_fwd(s[0], wdw, out=s[2], desc=dd) for s in sets]), 2)
    dwg = e.empty(3, 3, c, 1)
    res["dw 3x3 wgrad (2r)"] = (timed([lambda s=s: e.dwconv_wgrad(s[0], s[1], dd, True, dw=dwg) for s in sets]), 2)
    print(f"shape {shp} tensor {nbytes / 1e6:.1f} MB")
    for k, (t, passes) in res.items():
        print(f"   {k:30s} {t:8.1f} us   {passes * nbytes / t / 1e6:6.2f} TB/s", flush=True)
