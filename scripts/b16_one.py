#!/usr/bin/env python3
"""One bf16-storage convolution shape, forward / dgrad / filter gradient, stand-alone (weights converted per call), for a
rocprofv3 --kernel-trace --stats pass: which kernels does it launch?  usage: b16_one.py H CIN COUT K DIL [BATCH]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

h, cin, cout, k, dil = (int(a) for a in sys.argv[1:6])
N = int(sys.argv[6]) if len(sys.argv) > 6 else 16
e = get_engine(0)
g = torch.Generator(device="cpu").manual_seed(0)
x = (torch.rand(N, h, h, cin, generator=g) * 2 - 1).cuda().to(torch.bfloat16)
w = ((torch.rand(k, k, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
d = e.conv_desc(tuple(x.shape), cout, k, k, 1, dil, "same")
y = e.conv2d_fwd(x, w, None, desc=d)
dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda().to(y.dtype)
dx, dw = e.empty(*x.shape, dtype=x.dtype), e.empty(*w.shape)
for _ in range(3):
    e.conv2d_fwd(x, w, None, desc=d, out=y)
    e.conv2d_dgrad(dy, w, d, out=dx)
    e.conv2d_wgrad(x, dy, d, want_bias=False, dw=dw)
torch.cuda.synchronize()
print("done")
