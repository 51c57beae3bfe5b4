#!/usr/bin/env python3
"""Where does a short-K pointwise GEMM (the 48 middle-flow 728->728 convs: 19 ms of the step) lose its time?
Times the 1x1 conv at 32x32, batch 16 (M = 16384 rows) over a sweep of K (= Cin) for two widths: the slope of
time over K is the steady-state slab rate, the intercept the per-launch / per-tile fixed cost (prologue, epilogue,
launch ramp).  Use: python scripts/pw_scan.py   (SG_X6_VARIANT=0/1 forces the two structures)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
N, h = 16, 32
iters = int(os.environ.get("ITERS", "20"))
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
    return a.elapsed_time(b) / iters * 1e3  # microseconds


bf = os.environ.get("DTYPE", "f32") == "bf16"   # DTYPE=bf16: bf16 storage (pw_wide_kernel<1, bf16, ..>, 64-deep stages)
print(f"SG_X6_VARIANT={os.environ.get('SG_X6_VARIANT', 'auto')}  M={N * h * h}  dtype={'bf16' if bf else 'f32'}")
for cout in (256, 728, 1024):
    pts = []
    for cin in (128, 256, 512, 728, 1024, 1456, 2048):
        x = (torch.rand(N, h, h, cin, generator=g) * 2 - 1).cuda()
        if bf:
            x = x.to(torch.bfloat16)
        w = ((torch.rand(1, 1, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
        d = e.conv_desc(tuple(x.shape), cout, 1, 1, 1, 1, "same")
        y = e.conv2d_fwd(x, w, None, desc=d)
        dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda().to(y.dtype)
        dx, dw = e.empty(*x.shape, dtype=x.dtype), e.empty(*w.shape)
        tf_ = timed(lambda: e.conv2d_fwd(x, w, None, desc=d, out=y))
        td = timed(lambda: e.conv2d_dgrad(dy, w, d, out=dx))
        tw = timed(lambda: e.conv2d_wgrad(x, dy, d, want_bias=False, dw=dw))
        fl = 2.0 * N * h * h * cin * cout / 1e6  # MFLOP -> /us = TFLOP/s
        pts.append((cin, tf_))
        print(f"  {cin:5d}->{cout:5d}: fwd {tf_:7.1f} us {fl / tf_:6.1f} TF | dgrad {td:7.1f} us {fl / td:6.1f} TF | "
              f"wgrad {tw:7.1f} us {fl / tw:6.1f} TF", flush=True)
    (k1, t1), (k2, t2) = pts[1], pts[-1]
    slope = (t2 - t1) / ((k2 - k1) / 32)
    tiles = (N * h * h // 128) * ((cout + 127) // 128)
    print(f"  cout {cout}: {tiles} tiles; forward slope {slope:.3f} us per 32-deep slab per launch, intercept {t1 - slope * k1 / 32:.1f} us")
