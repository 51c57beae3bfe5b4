#!/usr/bin/env python3
"""Digest of pointwise convolutions whose column count takes the 256-wide tile of pw_wide_kernel (1024, 2048, 256 columns at many
rows): forward with bias + BatchNormalization statistics, dgrad, in fp32 and bf16 storage.  With SG_PW_WIDE=3 (384-wide tiles
only) the same layers run on conv_x6_kernel / conv_b16_kernel: both add the same products in the same order, so the lines must
agree.  Use: SG_PW_WIDE=1|3 python scripts/pw_bn_check.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
g = torch.Generator(device="cpu").manual_seed(5)
for dt in (torch.float32, torch.bfloat16):
    for (n, h, cin, cout) in ((16, 32, 728, 1024), (16, 32, 1536, 2048), (9, 32, 1024, 1024), (4, 128, 256, 256), (16, 32, 1000, 2048 - 64)):
        x = (torch.rand(n, h, h, cin, generator=g) * 2 - 1).cuda().to(dt)
        w = ((torch.rand(1, 1, cin, cout, generator=g) * 2 - 1) * 0.05).cuda()
        b = (torch.rand(cout, generator=g) - 0.5).cuda()
        d = e.conv_desc(tuple(x.shape), cout, 1, 1, 1, 1, "same")
        y, st = e.conv2d_fwd(x, w, b, desc=d, want_stats=True)
        dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda().to(dt)
        dx = e.conv2d_dgrad(dy, w, d)
        torch.cuda.synchronize()
        dig = hashlib.sha256()
        for t in (y.float(), st[0] if st is not None else y[:0].float(), dx.float()):
            dig.update(t.detach().cpu().numpy().tobytes())
        print(f"{str(dt)[6:]} {n}x{h}x{h} {cin}->{cout}: {dig.hexdigest()[:24]}", flush=True)
