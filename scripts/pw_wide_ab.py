#!/usr/bin/env python3
"""A/B of the wide pointwise kernel (csrc/conv_pw.h) against the 128 x 128-tile kernels on the middle-flow shape and its
neighbours: time per launch (weight-plane conversion included in both arms: the call converts per launch here) and the error
against an fp64 product on sampled rows.   SG_PW_WIDE=0|1|2  DTYPE=f32|bf16  python scripts/pw_wide_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
iters = int(os.environ.get("ITERS", "50"))
bf = os.environ.get("DTYPE", "f32") == "bf16"
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


print(f"SG_PW_WIDE={os.environ.get('SG_PW_WIDE', 'default')} dtype={'bf16' if bf else 'f32'}")
SHAPES = [(16, 32, 728, 728), (16, 32, 728, 1024), (16, 32, 1536, 1536), (16, 32, 256, 728), (16, 64, 728, 728), (3, 32, 728, 728)]
if os.environ.get('MSCAN'):
    SHAPES = [(b, 32, 728, 728) for b in (1, 2, 4, 6, 8, 12, 16, 24, 32)]
if os.environ.get('ONLY'):
    SHAPES = SHAPES[:int(os.environ['ONLY'])]
print('SG_PW_ABLATE=' + os.environ.get('SG_PW_ABLATE', '0'))
for (n, h, cin, cout) in SHAPES:
    x = (torch.rand(n, h, h, cin, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(1, 1, cin, cout, generator=g) * 2 - 1) * 0.05).cuda()
    b = (torch.rand(cout, generator=g) - 0.5).cuda()
    if bf:
        x = x.to(torch.bfloat16)
    d = e.conv_desc(tuple(x.shape), cout, 1, 1, 1, 1, "same")
    y = e.conv2d_fwd(x, w, b, desc=d)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda().to(y.dtype)
    dx = e.conv2d_dgrad(dy, w, d)
    rows = torch.randint(0, n * h * h, (64,), generator=g).cuda()
    wr = w.to(torch.bfloat16).double() if bf else w.double()
    ref = x.view(-1, cin)[rows].double() @ wr.view(cin, cout) + b.double()
    refd = dy.view(-1, cout)[rows].double() @ wr.view(cin, cout).t()
    ey = float((y.view(-1, cout)[rows].double() - ref).abs().max() / ref.abs().max())
    ed = float((dx.view(-1, cin)[rows].double() - refd).abs().max() / refd.abs().max())
    tf_ = timed(lambda: e.conv2d_fwd(x, w, b, desc=d, out=y))
    td = timed(lambda: e.conv2d_dgrad(dy, w, d, out=dx))
    fl = 2.0 * n * h * h * cin * cout / 1e6
    dw, _ = e.conv2d_wgrad(x, dy, d, want_bias=False)
    cols = torch.randint(0, cout, (48,), generator=g).cuda()
    refw = x.view(-1, cin).double().t() @ dy.view(-1, cout)[:, cols].double()
    ew = float((dw.view(cin, cout)[:, cols].double() - refw).abs().max() / refw.abs().max())
    tw = timed(lambda: e.conv2d_wgrad(x, dy, d, want_bias=False, dw=dw))
    print(f"  M {n * h * h:6d} {cin:5d}->{cout:5d}: fwd {tf_:7.1f} us {fl / tf_:6.1f} TF err {ey:.1e} | dgrad {td:7.1f} us {fl / td:6.1f} TF err {ed:.1e}"
          f" | wgrad {tw:7.1f} us {fl / tw:6.1f} TF err {ew:.1e}", flush=True)
