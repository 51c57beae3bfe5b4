#!/usr/bin/env python3
"""Planes-in filter gradient (sg_conv2d_wgrad_planes, csrc/conv_x6.h wgrad_x6_kernel<.., PIN>) against the fp32-operand kernel on
the step's long-K shapes: bit-identity and time (the split of x, the split of dy and the gradient itself timed apart).
Use: python scripts/wgrad_planes_bench.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
e = get_engine(0)
g = torch.Generator().manual_seed(2)
SHAPES = [(32, 2048, 256, 3, 6), (32, 2048, 256, 3, 12), (32, 2048, 256, 3, 18), (32, 2048, 256, 3, 1), (32, 256, 256, 3, 6), (32, 256, 256, 3, 1),
          (32, 512, 256, 3, 1), (64, 512, 256, 3, 1), (64, 256, 256, 3, 1), (128, 256, 128, 3, 1), (128, 128, 128, 3, 1), (256, 128, 64, 3, 1),
          (32, 1280, 256, 1, 1), (32, 2048, 256, 1, 1)]
if os.environ.get("SHAPES") == "pw":   # the wide pointwise layers: an experiment of round 5 (LAB_NOTEBOOK 12.8, not in the tree) - the
    # library declines them (conv2d_wgrad_planes_ok false) and the loop below prints so
    SHAPES = [(32, 728, 728, 1, 1), (32, 728, 1024, 1, 1), (32, 1024, 1536, 1, 1), (32, 1536, 1536, 1, 1), (32, 1536, 2048, 1, 1),
              (64, 728, 728, 1, 1), (32, 1024, 1024, 1, 1), (32, 1280, 256, 1, 1), (32, 2048, 256, 1, 1), (64, 256, 728, 1, 1)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    z.record()
    torch.cuda.synchronize()
    return a.elapsed_time(z) / reps * 1e3


print(f"{'HxW':>8s} {'cin':>5s} {'cout':>5s} k dil | fp32-operand us | planes-in us  split x  split dy | same bits")
tot = [0.0, 0.0, 0.0, 0.0]
for hw, cin, cout, k, dil in SHAPES:
    x = torch.randn(n, hw, hw, cin, generator=g).cuda()
    dy = torch.randn(n, hw, hw, cout, generator=g).cuda()
    d = e.conv_desc((n, hw, hw, cin), cout, k, k, 1, dil, "same")
    if not e.conv2d_wgrad_planes_ok(d):
        print(f"{hw:4d}x{hw:<3d} {cin:5d} {cout:5d} {k} {dil:3d} | not on the planes-in kernel")
        continue
    dw0, _ = e.conv2d_wgrad(x, dy, d, want_bias=False)
    xp, yp = e.split_planes(x), e.split_planes(dy)
    dw1 = e.conv2d_wgrad_planes(xp, yp, d)
    same = torch.equal(dw0, dw1)
    t0 = timed(lambda: e.conv2d_wgrad(x, dy, d, want_bias=False, dw=dw0))
    t1 = timed(lambda: e.conv2d_wgrad_planes(xp, yp, d, dw=dw1))
    tx = timed(lambda: e.split_planes(x, out=xp))
    ty = timed(lambda: e.split_planes(dy, out=yp))
    for i, v in enumerate((t0, t1, tx, ty)):
        tot[i] += v
    print(f"{hw:4d}x{hw:<3d} {cin:5d} {cout:5d} {k} {dil:3d} | {t0:15.1f} | {t1:12.1f} {tx:8.1f} {ty:9.1f} | {same}")
    del x, dy, xp, yp
print(f"sum: fp32-operand {tot[0]:.0f} us, planes-in {tot[1]:.0f} us (+ split x {tot[2]:.0f}, split dy {tot[3]:.0f})")
