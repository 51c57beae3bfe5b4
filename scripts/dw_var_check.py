#!/usr/bin/env python3
"""Digest of the stride-1 3x3 depthwise stencil's outputs - forward (plain, pre-activation ReLU, BatchNormalization in the gather
with and without ReLU), dgrad (plain, ReLU mask, a collected gradient riding along) - in fp32 and bf16 storage over maps of several
sizes.  Run once with SG_DW_FSTRIP=0 (run kernel) and once with 1 (strip kernel) and compare the lines: same products in the same
order, the bits must agree.  (The BatchNormalization sums of sg_dwconv2d_dgrad_bnsums are added in another order: the op test
holds them to the oracle.)  Use: SG_DW_FSTRIP=0|1 python scripts/dw_var_check.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
g = torch.Generator(device="cpu").manual_seed(11)
for dt in (torch.float32, torch.bfloat16):
    for (n, h, w_, c) in ((2, 32, 32, 728), (1, 64, 64, 256), (3, 12, 16, 36), (2, 8, 8, 128), (1, 128, 128, 128), (2, 20, 24, 64)):
        x = (torch.rand(n, h, w_, c, generator=g) * 2 - 1).cuda().to(dt)
        wt = (torch.rand(3, 3, c, 1, generator=g) * 2 - 1).cuda()
        dy = (torch.rand(n, h, w_, c, generator=g) * 2 - 1).cuda().to(dt)
        res = (torch.rand(n, h, w_, c, generator=g) * 2 - 1).cuda().to(dt)
        bn = tuple(t.cuda() for t in (torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) - 0.5, torch.rand(c, generator=g) - 0.5,
                                      torch.rand(c, generator=g) + 0.5))
        d = e.conv_desc(tuple(x.shape), c, 3, 3, 1, 1, "same")
        outs = []
        outs.append(e.dwconv_fwd(x, wt, 1, pre_relu=False, desc=d))
        outs.append(e.dwconv_fwd(x, wt, 1, pre_relu=True, desc=d))
        outs.append(e.dwconv_fwd(x, wt, 1, desc=d, bn=bn + (False,)))
        outs.append(e.dwconv_fwd(x, wt, 1, desc=d, bn=bn + (True,)))
        outs.append(e.dwconv_dgrad(dy, wt, d))
        outs.append(e.dwconv_dgrad(dy, wt, d, x=x, pre_relu=True))
        if e.dwconv_dgrad_acc_ok(d):
            outs.append(e.dwconv_dgrad(dy, wt, d, x=x, pre_relu=True, res=res))
        torch.cuda.synchronize()
        dig = hashlib.sha256()
        for t in outs:
            dig.update(t.float().detach().cpu().numpy().tobytes())
        print(f"{str(dt)[6:]} {n}x{h}x{w_}x{c}: {dig.hexdigest()[:24]}", flush=True)
