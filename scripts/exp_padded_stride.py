#!/usr/bin/env python3
"""Experiment for DESIGN 10.8: does a padded pixel stride (x_ld = 2048 + PAD channels instead of 2048) change the fabric
fetch volume / the time of the ASPP forward?  Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  and read the
conv_x6_kernel rows; PAD=0 is the dense tensor.   PAD=32 DIL=6 python scripts/exp_padded_stride.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
pad, dil = int(os.environ.get("PAD", "0")), int(os.environ.get("DIL", "6"))
n, h, cin, cout = 16, 32, 2048, 256
ld = cin + pad
g = torch.Generator().manual_seed(0)
xw = torch.zeros(n, h, h, ld)
xw[..., :cin] = torch.rand(n, h, h, cin, generator=g) * 2 - 1
xw = xw.cuda()
w = ((torch.rand(3, 3, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
d = e.conv_desc((n, h, h, cin), cout, 3, 3, 1, dil, "same", x_ld=ld if pad else 0)
y = e.conv2d_fwd(xw, w, None, desc=d)
ref = e.conv2d_fwd(xw[..., :cin].contiguous(), w, None, dilation=dil)
print("max |y - y_dense|", float((y - ref).abs().max()))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    e.conv2d_fwd(xw, w, None, desc=d, out=y)
b.record()
torch.cuda.synchronize()
print(f"PAD={pad} d={dil}: forward {a.elapsed_time(b) / 5 * 1e3:.1f} us")
