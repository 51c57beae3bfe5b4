#!/usr/bin/env python3
"""A/B of the two x6 structures (SG_X6_VARIANT=0: two workgroups per CU, single LDS buffer; 1: one workgroup per CU,
double-buffered, interleaved step) on the launches of the DeepLabv3+ step that have 769..1024 tiles (batch 16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print("SG_X6_VARIANT =", os.environ.get("SG_X6_VARIANT", "auto"))
tot = 0.0
for h, cin, cout, k in [(64, 512, 256, 3), (64, 256, 256, 3), (64, 256, 256, 1), (32, 728, 1024, 1), (32, 1024, 1024, 1), (32, 1024, 1536, 1),
                        (128, 128, 128, 3)]:
    x = (torch.rand(16, h, h, cin, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(k, k, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
    d = e.conv_desc(tuple(x.shape), cout, k, k, 1, 1, "same")
    y = e.conv2d_fwd(x, w, None, desc=d)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda()
    dx = e.empty(*x.shape)
    tf_ = timed(lambda: e.conv2d_fwd(x, w, None, desc=d, out=y))
    td = timed(lambda: e.conv2d_dgrad(dy, w, d, out=dx))
    tiles_f = (16 * h * h // 128) * ((cout + 127) // 128)
    tiles_d = (16 * h * h // 128) * ((cin + 127) // 128)
    tot += tf_ + td
    print(f"{h:3d}x{h:<3d} {cin:4d}->{cout:4d} k{k}: fwd {tf_:7.1f} us ({tiles_f} tiles) | dgrad {td:7.1f} us ({tiles_d} tiles)", flush=True)
print(f"sum {tot:.0f} us")
