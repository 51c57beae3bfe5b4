#!/usr/bin/env python3
"""The RGB stem kernels (csrc/conv_stem.h) against the kernels they replace: run once with SG_STEM3=1 and once with SG_STEM3=0
(the switch is read once per process); prints the time per launch and a hash of the fp32 forward output - the two runs must
print the SAME hash (the stencil kernel adds the 27 products in the fp32-MFMA kernel's order)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
g = torch.Generator().manual_seed(5)


def timed(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


print("SG_STEM3 =", os.environ.get("SG_STEM3", "1"))
for n, hw, cout, stride in ((16, 512, 64, 1), (16, 512, 32, 2), (2, 64, 64, 1), (3, 50, 32, 2)):
    x = (torch.rand(n, hw, hw, 3, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(3, 3, 3, cout, generator=g) * 2 - 1) * 0.3).cuda()
    b = (torch.rand(cout, generator=g) - 0.5).cuda()
    d = e.conv_desc(tuple(x.shape), cout, 3, 3, stride, 1, "same")
    y = e.conv2d_fwd(x, w, b, desc=d)
    h = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda()
    dw, db = e.empty(*w.shape), e.empty(cout)
    t_f = timed(lambda: e.conv2d_fwd(x, w, b, desc=d, out=y))
    t_w = timed(lambda: e.conv2d_wgrad(x, dy, d, dw=dw, db=db))
    print(f"  {n} x {hw}^2 x 3 -> {cout} s{stride}: fwd {t_f * 1e3:8.1f} us  wgrad {t_w * 1e3:8.1f} us  fwd sha {h}  dw sum {float(dw.double().sum()):.6f}")
