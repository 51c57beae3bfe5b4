#!/usr/bin/env python3
"""UpSampling2D(2) -> Conv2D(32, 3) on 64 channels at the step's size (16 x 256 x 256 x 64 -> 512 x 512): the fused kernels
(csrc/conv_x6p.h: sub-pixel forward, dgrad with the 2x2 sum, filter gradient gathering the source) against the unfused launches.
Use: python scripts/up2_bench.py [batch] [source size]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
hs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
e = get_engine(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(n, hs, hs, 64, generator=g).cuda()
w = (torch.randn(3, 3, 64, 32, generator=g) * 0.04).cuda()
b = torch.randn(32, generator=g).cuda()
dy = torch.randn(n, 2 * hs, 2 * hs, 32, generator=g).cuda()
d = e.conv_desc((n, 2 * hs, 2 * hs, 64), 32, 3, 3, 1, 1, "same")
assert e.conv2d_up2_ok(d)
up = e.upsample_fwd(x, 2)


def timed(name, fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    z.record()
    torch.cuda.synchronize()
    print(f"{name:58s} {a.elapsed_time(z) / reps * 1e3:9.1f} us")


y = e.empty(n, 2 * hs, 2 * hs, 32)
dxs = e.empty(n, hs, hs, 64)
dxu = e.empty(n, 2 * hs, 2 * hs, 64)
timed("up-sampling forward (materialise 4x)", lambda: e.upsample_fwd(x, 2))
timed("conv forward on the materialised tensor (+stats)", lambda: e.conv2d_fwd(up, w, b, desc=d, want_stats=True, out=y))
timed("FUSED sub-pixel forward (+stats)", lambda: e.conv2d_fwd(x, w, b, desc=d, want_stats=True, out=y, up2=True))
timed("dgrad on the up-sampled grid", lambda: e.conv2d_dgrad(dy, w, d, out=dxu))
timed("up-sampling backward (2x2 sums)", lambda: e.upsample_bwd(dxu, tuple(x.shape), 2))
timed("FUSED dgrad with the 2x2 sum in the epilogue", lambda: e.conv2d_dgrad(dy, w, d, out=dxs, down2=True))
timed("filter gradient on the materialised tensor", lambda: e.conv2d_wgrad(up, dy, d))
timed("FUSED filter gradient gathering the source", lambda: e.conv2d_wgrad(x, dy, d, x_up2=True))

# the input gradient as ONE 4x4 stride-2 convolution of dy with summed taps (the 2x2 sum taken before the products: 4/9 of them)
S = {-1: [2], 0: [1, 2], 1: [0, 1], 2: [0]}
wd = torch.zeros(4, 4, 32, 64, device="cuda")
for u in range(-1, 3):
    for v in range(-1, 3):
        for kh in S[u]:
            for kw in S[v]:
                wd[u + 1, v + 1] += w[kh, kw].t()
ref = e.conv2d_dgrad(dy, w, d, down2=True)
got = e.conv2d_fwd(dy, wd, None, stride=2, padding="same")
print("4x4 stride-2 form vs fused dgrad: max |diff|", float((got - ref).abs().max()), "of", float(ref.abs().max()))
timed("input gradient as a 4x4 stride-2 convolution of dy", lambda: e.conv2d_fwd(dy, wd, None, stride=2, padding="same", out=dxs))
