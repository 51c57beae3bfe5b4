#!/usr/bin/env python3
"""A/B of the LDS-patch x6 kernel (conv_x6p.h) against the im2col x6 kernel on the low-channel 3x3 convolutions
of the five models (batch 16): forward and dgrad time and TFLOP/s.  SG_X6_NOPATCH=1 selects the im2col form."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
N = int(os.environ.get("BATCH", "16"))
iters = int(os.environ.get("ITERS", "10"))
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


print("patch kernel:", "off" if os.environ.get("SG_X6_NOPATCH") else "on")
tot = 0.0
for h, cin, cout in [(512, 32, 32), (512, 64, 32), (512, 64, 64), (256, 32, 64), (256, 64, 64), (256, 64, 128), (128, 64, 64), (256, 32, 32)]:
    x = (torch.rand(N, h, h, cin, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(3, 3, cin, cout, generator=g) * 2 - 1) * 0.05).cuda()
    d = e.conv_desc(tuple(x.shape), cout, 3, 3, 1, 1, "same")
    y = e.conv2d_fwd(x, w, None, desc=d)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda()
    dx = e.empty(*x.shape)
    fl = 2.0 * N * h * h * cout * 9 * cin / 1e12
    tf_ = timed(lambda: e.conv2d_fwd(x, w, None, desc=d, out=y))
    td = timed(lambda: e.conv2d_dgrad(dy, w, d, out=dx))
    tot += tf_ + td
    print(f"{h:4d}x{h:<4d} {cin:3d}->{cout:3d}: fwd {tf_:7.3f} ms {fl / tf_ * 1e3:6.1f} TF | dgrad {td:7.3f} ms {fl / td * 1e3:6.1f} TF", flush=True)
    del x, y, dy, dx
print(f"sum {tot:.2f} ms")
