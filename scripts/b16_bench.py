#!/usr/bin/env python3
"""bf16-storage convolution micro-benchmark: the dilated set, the middle-flow pointwise GEMM and a decoder conv, forward /
dgrad / wgrad, per-launch time from HIP events (A/B runs of SG_B16_DEEP, SG_B16_KS)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
N = int(os.environ.get("BATCH", "16"))
iters = int(os.environ.get("ITERS", "10"))
g = torch.Generator(device="cpu").manual_seed(0)
BF = torch.bfloat16


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


cases = [("aspp", 32, 2048, 256, 3, 6), ("aspp", 32, 2048, 256, 3, 18), ("sk", 32, 256, 256, 3, 12), ("pw728", 32, 728, 728, 1, 1),
         ("pw256", 128, 128, 256, 1, 1), ("dec304", 128, 304, 256, 3, 1), ("dec64", 256, 128, 64, 3, 1), ("ent64", 256, 64, 64, 3, 1)]
tot = {"f": 0.0, "d": 0.0, "w": 0.0}
for name, h, cin, cout, k, dil in cases:
    x = (torch.rand(N, h, h, cin, generator=g) * 2 - 1).cuda().to(BF)
    w = ((torch.rand(k, k, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
    b = torch.zeros(cout).cuda()
    d = e.conv_desc(tuple(x.shape), cout, k, k, 1, dil, "same")
    y = e.conv2d_fwd(x, w, b, desc=d)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda().to(BF)
    fl = 2.0 * N * h * h * cout * k * k * cin / 1e12
    t_f = timed(lambda: e.conv2d_fwd(x, w, b, desc=d, out=y))
    dx = torch.empty_like(x)
    t_d = timed(lambda: e.conv2d_dgrad(dy, w, d, out=dx))
    dw, db = e.empty(*w.shape), e.empty(cout)
    t_w = timed(lambda: e.conv2d_wgrad(x, dy, d, dw=dw, db=db))
    tot["f"] += t_f; tot["d"] += t_d; tot["w"] += t_w
    print(f"{name:6s} {h:3d}^2 d={dil:2d} {cin:4d}->{cout:4d} k{k}: fwd {t_f:7.3f} ms {fl / t_f * 1e3:6.0f} TF | dgrad {t_d:7.3f} ms "
          f"{fl / t_d * 1e3:6.0f} TF | wgrad {t_w:7.3f} ms {fl / t_w * 1e3:6.0f} TF", flush=True)
print(f"sum: fwd {tot['f']:.3f} dgrad {tot['d']:.3f} wgrad {tot['w']:.3f} ms")
