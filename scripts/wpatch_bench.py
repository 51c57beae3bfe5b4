#!/usr/bin/env python3
"""Filter-gradient micro-benchmark of the low-channel 3x3 convolutions (DeepLabv3+ entry / decoder layers), fp32 (x6) and
bf16 storage: per-launch time of sg_conv2d_wgrad (partial slabs + reduce, no bias gradient) from HIP events.  A/B:
SG_X6_NOWPATCH=1 runs the slab kernel (wgrad_x6_kernel) instead of the patch form (conv_x6wp.h)."""
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


cases = [(512, 64, 32), (512, 32, 32), (256, 32, 64), (256, 64, 64)]
for dt in (torch.float32, torch.bfloat16):
    tot = 0.0
    for h, cin, cout in cases:
        x = (torch.rand(N, h, h, cin, generator=g) * 2 - 1).cuda().to(dt)
        dy = (torch.rand(N, h, h, cout, generator=g) * 2 - 1).cuda().to(dt)
        d = e.conv_desc(tuple(x.shape), cout, 3, 3, 1, 1, "same")
        dw = e.empty(3, 3, cin, cout)
        t = timed(lambda: e.conv2d_wgrad(x, dy, d, want_bias=False, dw=dw))
        fl = 2.0 * N * h * h * cout * 9 * cin / 1e12
        gb = (x.numel() + dy.numel()) * x.element_size() / 1e9
        tot += t
        print(f"{str(dt)[6:]:9s} {h:3d}^2 {cin:2d}->{cout:2d}: {t:7.3f} ms {fl / t * 1e3:6.0f} TF  ({gb / t * 1e3:5.0f} GB/s of x + dy once)", flush=True)
    print(f"{str(dt)[6:]} sum {tot:.3f} ms")
