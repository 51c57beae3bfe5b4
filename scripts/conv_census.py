#!/usr/bin/env python3
"""Census of the MFMA work of one training step: every distinct implicit-GEMM shape of a model (Conv2D, the
pointwise half of SeparableConv2D, Conv2DTranspose is left out) is timed stand-alone, forward / dgrad / wgrad,
and listed with its count, time share and achieved TFLOP/s, sorted by the time it loses against RATE TFLOP/s
(130 in fp32, 800 with DTYPE=bf16).
Use: python scripts/conv_census.py [model] [batch] [size]."""
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
iters = int(os.environ.get("ITERS", "5"))
bf = os.environ.get("DTYPE", "f32") == "bf16"            # DTYPE=bf16: activations in bf16 storage (conv_b16 kernels)
RATE = float(os.environ.get("RATE", "800" if bf else "130")) * 1e-3   # yardstick of the "lost" column, TFLOP/s
e = get_engine(0)
model = zoo.BUILDERS[name]((size, size, 3))

shapes = OrderedDict()
for node in model.nodes:
    if node.op == "conv2d":
        _, h, w, cin = node.inputs[0].shape
        key = (h, w, cin, node.filters, node.k, node.stride, node.dilation, node.padding)
    elif node.op == "separable_conv2d":
        _, ho, wo, co = node.output.shape
        key = (ho, wo, node.inputs[0].shape[-1], co, 1, 1, 1, "same")
    else:
        continue
    shapes[key] = shapes.get(key, 0) + 1


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


g = torch.Generator(device="cpu").manual_seed(0)
rows = []
for (h, w, cin, cout, k, s, dil, pad), cnt in shapes.items():
    x = (torch.rand(N, h, w, cin, generator=g) * 2 - 1).cuda()
    if bf:
        x = x.to(torch.bfloat16)
    wt = ((torch.rand(k, k, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
    d = e.conv_desc(tuple(x.shape), cout, k, k, s, dil, pad)
    y = e.conv2d_fwd(x, wt, None, desc=d)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda().to(y.dtype)
    fl = 2.0 * y.numel() * k * k * cin / 1e12
    dx, dw = e.empty(*x.shape, dtype=x.dtype), e.empty(*wt.shape)
    tf_ = timed(lambda: e.conv2d_fwd(x, wt, None, desc=d, out=y))
    td = timed(lambda: e.conv2d_dgrad(dy, wt, d, out=dx))
    tw = timed(lambda: e.conv2d_wgrad(x, dy, d, want_bias=False, dw=dw))
    rows.append(((h, w, cin, cout, k, s, dil), cnt, fl, tf_, td, tw))
    del x, wt, y, dy, dx, dw

tot = sum(c * (a + b + w_) for _, c, _, a, b, w_ in rows)
totfl = sum(c * 3 * f for _, c, f, *_ in rows)
print(f"{name} bs{N} {size}x{size}: {len(rows)} distinct GEMM shapes, {totfl:.2f} TFLOP, {tot:.1f} ms stand-alone = {totfl / tot * 1e3:.1f} TFLOP/s")
print(f"{'HxW':>9s} {'cin':>5s} {'cout':>5s} k s dil  cnt | {'fwd ms':>7s} {'TF':>5s} | {'dgrad':>7s} {'TF':>5s} | {'wgrad':>7s} {'TF':>5s} | {'sum ms':>7s} {'lost':>6s}")
rows.sort(key=lambda r: -r[1] * ((r[3] + r[4] + r[5]) - 3 * r[2] / RATE))
for (h, w, cin, cout, k, s, dil), cnt, fl, a, b, c in rows:
    lost = cnt * ((a + b + c) - 3 * fl / RATE)
    print(f"{h:4d}x{w:<4d} {cin:5d} {cout:5d} {k} {s} {dil:3d} {cnt:4d} | {a:7.3f} {fl / a * 1e3:5.1f} | {b:7.3f} {fl / b * 1e3:5.1f} | "
          f"{c:7.3f} {fl / c * 1e3:5.1f} | {cnt * (a + b + c):7.2f} {lost:6.2f}", flush=True)
