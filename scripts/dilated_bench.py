#!/usr/bin/env python3
"""Micro-benchmark of the north_star kernel set: the six dilated 3x3 convs of DeepLabv3+ (3 ASPP 2048->256,
3 SK 256->256, 32x32 maps, batch 16) forward / dgrad / wgrad, plus the middle-flow pointwise 728->728 GEMMs.
Prints per-launch time (HIP events on the launch stream) and achieved TFLOP/s; also the command the PMC
passes profile (scripts/gpu_ci.sh pmc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
N = int(os.environ.get("BATCH", "16"))
iters = int(os.environ.get("ITERS", "5"))
BF = os.environ.get("DTYPE", "f32") == "bf16"   # DTYPE=bf16: activations in bf16 storage (BASELINE configs[2]'s kernels)
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


cases = [("aspp", 2048, 256, 3, d) for d in (6, 12, 18)] + [("sk", 256, 256, 3, d) for d in (6, 12, 18)]
if not os.environ.get("ONLY_DILATED"):  # the PMC traffic pass profiles exactly the roofline kernel set
    cases += [("pw728", 728, 728, 1, 1), ("dec64", 128, 64, 3, 1)]
tot_ms = tot_fl = 0.0
for name, cin, cout, k, dil in cases:
    h = 32 if name != "dec64" else 256
    x = (torch.rand(N, h, h, cin, generator=g) * 2 - 1).cuda()
    if BF:
        x = x.to(torch.bfloat16)
    w = ((torch.rand(k, k, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
    b = torch.zeros(cout).cuda()
    d = e.conv_desc(tuple(x.shape), cout, k, k, 1, dil, "same")
    y = e.conv2d_fwd(x, w, b, desc=d)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda().to(y.dtype)
    fl = 2.0 * N * h * h * cout * k * k * cin / 1e12
    t_f = timed(lambda: e.conv2d_fwd(x, w, b, desc=d, out=y))
    dx = e.empty(*x.shape, dtype=x.dtype)
    t_d = timed(lambda: e.conv2d_dgrad(dy, w, d, out=dx))
    dw, db = e.empty(*w.shape), e.empty(cout)
    t_w = timed(lambda: e.conv2d_wgrad(x, dy, d, dw=dw, db=db))
    print(f"{name:6s} d={dil:2d} {cin:4d}->{cout:4d} k{k}: fwd {t_f:7.3f} ms {fl / t_f * 1e3:6.1f} TF | dgrad {t_d:7.3f} ms "
          f"{fl / t_d * 1e3:6.1f} TF | wgrad {t_w:7.3f} ms {fl / t_w * 1e3:6.1f} TF", flush=True)
    if name in ("aspp", "sk"):
        tot_ms += t_f + t_d + t_w
        tot_fl += 3 * fl
print(f"dilated set: {tot_fl:.3f} TFLOP in {tot_ms:.3f} ms = {tot_fl / tot_ms * 1e3:.1f} TFLOP/s = "
      + (f"{tot_fl / tot_ms * 1e3 / 2500.0:.3f} of the bf16 MFMA peak" if BF else f"{tot_fl / tot_ms * 1e3 / 157.3:.3f} of the fp32 MFMA peak"))
