#!/usr/bin/env python3
"""Census of the bandwidth-bound layer kernels of one model: every distinct depthwise-conv and BatchNormalization
shape is timed stand-alone (forward / backward) and listed with the effective GB/s against its algorithmic traffic
(depthwise: read x + write y; dgrad the same; wgrad: read x + dy.  BN forward: 3 tensor passes, backward: 5).
Use: python scripts/bw_census.py [model] [batch] [size]"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd import zoo  # noqa: E402
from building_detection_amd.ops import get_engine  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "v3plus"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
size = int(sys.argv[3]) if len(sys.argv) > 3 else 512
iters = int(os.environ.get("ITERS", "10"))
e = get_engine(0)
model = zoo.BUILDERS[name]((size, size, 3))


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


dws, bns = OrderedDict(), OrderedDict()
for n in model.nodes:
    if n.op == "separable_conv2d":
        _, h, w, c = n.inputs[0].shape
        k = (h, w, c, n.stride)
        dws[k] = dws.get(k, 0) + 1
    elif n.op == "batch_normalization" and len(n.output.shape) == 4:
        k = tuple(n.output.shape[1:])
        bns[k] = bns.get(k, 0) + 1

g = torch.Generator().manual_seed(0)
tot = 0.0
print(f"{name} bs{N} {size}: depthwise 3x3 (count | fwd ms GB/s | dgrad ms GB/s | wgrad ms GB/s)")
for (h, w, c, s), cnt in dws.items():
    x = (torch.rand(N, h, w, c, generator=g) - 0.5).cuda()
    wt = (torch.rand(3, 3, c, 1, generator=g) - 0.5).cuda()
    d = e.conv_desc(tuple(x.shape), c, 3, 3, s, 1, "same")
    y = e.dwconv_fwd(x, wt, s)
    dy = torch.rand_like(y)
    dx, dw = torch.empty_like(x), torch.empty_like(wt)
    bx, by = x.numel() * 4, y.numel() * 4
    tf = timed(lambda: e.dwconv_fwd(x, wt, s, out=y))
    td = timed(lambda: e.dwconv_dgrad(dy, wt, d, x=x, out=dx))
    tw = timed(lambda: e.dwconv_wgrad(x, dy, d, dw=dw))
    tot += cnt * (tf + td + tw)
    print(f"  {h:4d}x{w:<4d} c{c:5d} s{s} x{cnt:3d} | {tf:6.3f} {(bx + by) / tf / 1e6:6.0f} | {td:6.3f} {(bx + by) / td / 1e6:6.0f} | {tw:6.3f} {(bx + by) / tw / 1e6:6.0f}",
          flush=True)
    del x, y, dy, dx
print(f"  depthwise total {tot:.2f} ms per step")
tot = 0.0
print("BatchNormalization 4-D (count | train fwd ms GB/s(3 passes) | train bwd ms GB/s(5 passes))")
for shp, cnt in bns.items():
    x = (torch.rand(N, *shp, generator=g) - 0.5).cuda()
    c = shp[-1]
    gam, bet, mm, mv = torch.ones(c).cuda(), torch.zeros(c).cuda(), torch.zeros(c).cuda(), torch.ones(c).cuda()
    y, mean, inv = e.bn_train_fwd(x, gam, bet, mm, mv, relu=True)
    dy = torch.rand_like(y)
    dx = torch.empty_like(x)
    b = x.numel() * 4
    tf = timed(lambda: e.bn_train_fwd(x, gam, bet, mm, mv, relu=True, out=y))
    tb = timed(lambda: e.bn_train_bwd(x, y, dy, gam, mean, inv, relu=True, out=dx))
    tot += cnt * (tf + tb)
    print(f"  {str(shp):18s} x{cnt:3d} | {tf:6.3f} {3 * b / tf / 1e6:6.0f} | {tb:6.3f} {5 * b / tb / 1e6:6.0f}", flush=True)
    del x, y, dy, dx
print(f"  BN total {tot:.2f} ms per step")
