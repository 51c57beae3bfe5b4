#!/usr/bin/env python3
"""Stand-alone time of the depthwise stencil's variants the training step uses, at the middle flow's shape (16 x 32 x 32 x 728) and a
large map: forward with the BatchNormalization (+ReLU) in the gather, dgrad with the BatchNormalization sums (with / without ReLU
mask).  hipGraph-free: 30 launches between two events over 4 rotating buffer sets.  Use: SG_DW_FSTRIP=0|2 python scripts/dw_variants_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
g = torch.Generator(device="cpu").manual_seed(3)


def timed(fn, sets, iters=32):
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(sets[i % len(sets)])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print("SG_DW_FSTRIP =", os.environ.get("SG_DW_FSTRIP", "default"))
for (n, h, w_, c) in ((16, 32, 32, 728), (16, 64, 64, 256), (16, 128, 128, 128)):
    sets = []
    for _ in range(4):
        x = (torch.rand(n, h, w_, c, generator=g) * 2 - 1).cuda()
        dy = (torch.rand(n, h, w_, c, generator=g) * 2 - 1).cuda()
        sets.append((x, dy, torch.empty_like(x), torch.empty_like(x)))
    wt = (torch.rand(3, 3, c, 1, generator=g) * 2 - 1).cuda()
    bn = tuple(t.cuda() for t in (torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5, torch.rand(c, generator=g) - 0.5,
                                  torch.rand(c, generator=g) + 0.5))
    dg, db = torch.empty(c).cuda(), torch.empty(c).cuda()
    d = e.conv_desc((n, h, w_, c), c, 3, 3, 1, 1, "same")
    t_plain = timed(lambda s: e.dwconv_fwd(s[0], wt, 1, desc=d, out=s[2]), sets)
    t_bn = timed(lambda s: e.dwconv_fwd(s[0], wt, 1, desc=d, out=s[2], bn=bn + (True,)), sets)
    t_dg = timed(lambda s: e.dwconv_dgrad(s[1], wt, d, out=s[3]), sets)
    t_sums = timed(lambda s: e.dwconv_dgrad_bnsums(s[1], wt, d, s[0], bn[2], bn[3], bn[0], bn[1], True, dg, db, out=s[3]), sets)
    t_sums_m = timed(lambda s: e.dwconv_dgrad_bnsums(s[1], wt, d, s[0], bn[2], bn[3], bn[0], bn[1], True, dg, db, x=s[0], pre_relu=True, out=s[3]), sets)
    mb = n * h * w_ * c * 4 / 1e6
    print(f"{n}x{h}x{w_}x{c} ({mb:.1f} MB): fwd {t_plain:6.1f} us | fwd BN+ReLU gather {t_bn:6.1f} | dgrad {t_dg:6.1f} | dgrad + BN sums {t_sums:6.1f} | "
          f"dgrad + mask + BN sums {t_sums_m:6.1f}", flush=True)
