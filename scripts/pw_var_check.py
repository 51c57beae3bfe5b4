#!/usr/bin/env python3
"""Digest of the wide pointwise kernels' outputs (forward with bias + statistics, dgrad; the wide filter gradient) over a sweep
of reduction depths that covers every tail of their k loops (1, 2, 3, 4 and many k-steps, ragged last k-steps) - run once per
SG_PW_VAR / SG_WPW_VAR and compare the lines: the two schedules must give the same bits.
The planes-in kernel's two stage loops (SG_X6W_VAR) are covered the same way.
Use: SG_PW_WIDE=2 SG_PW_VAR=0|1 SG_WPW_VAR=0|1 SG_X6W_VAR=0|1 python scripts/pw_var_check.py"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
g = torch.Generator(device="cpu").manual_seed(7)
for (n, h, cin, cout) in ((2, 32, 16, 384), (2, 32, 32, 384), (1, 32, 48, 400), (2, 32, 64, 728), (3, 32, 728, 728), (16, 32, 728, 728),
                          (2, 32, 1000, 1536), (1, 64, 256, 2048 - 128)):
    x = (torch.rand(n, h, h, cin, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(1, 1, cin, cout, generator=g) * 2 - 1) * 0.05).cuda()
    b = (torch.rand(cout, generator=g) - 0.5).cuda()
    d = e.conv_desc(tuple(x.shape), cout, 1, 1, 1, 1, "same")
    y, st = e.conv2d_fwd(x, w, b, desc=d, want_stats=True)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda()
    dx = e.conv2d_dgrad(dy, w, d)
    torch.cuda.synchronize()
    dig = hashlib.sha256()
    for t in (y, st[0] if st is not None else y[:0], dx):
        dig.update(t.detach().cpu().numpy().tobytes())
    print(f"{n}x{h}x{h} {cin}->{cout}: {dig.hexdigest()[:24]}", flush=True)

# the wide filter gradient (P >= 6144 pixels, Cin >= 256, tiles >= 3/4 full): whole and ragged pixel counts, few and many shares
for (n, h, w_, cin, cout) in ((6, 32, 32, 728, 728), (16, 32, 32, 728, 728), (7, 30, 30, 256, 384), (7, 30, 31, 1024, 1536), (2, 64, 64, 384, 2048 - 128)):
    x = (torch.rand(n, h, w_, cin, generator=g) * 2 - 1).cuda()
    dy = (torch.rand(n, h, w_, cout, generator=g) * 2 - 1).cuda()
    d = e.conv_desc(tuple(x.shape), cout, 1, 1, 1, 1, "same")
    dw = e.empty(1, 1, cin, cout)
    e.conv2d_wgrad(x, dy, d, want_bias=False, dw=dw)
    torch.cuda.synchronize()
    print(f"wgrad {n}x{h}x{w_} {cin}->{cout}: {hashlib.sha256(dw.detach().cpu().numpy().tobytes()).hexdigest()[:24]}", flush=True)

# conv_x6w_kernel: long-K multi-tap convolutions (two K shares, whole K, an odd stage count, padding taps skipped)
for (n, h, cin, cout, dil) in ((2, 32, 2048, 256, 6), (1, 64, 256, 512, 2), (2, 32, 480, 192, 1), (2, 32, 1024, 256, 18)):
    x = (torch.rand(n, h, h, cin, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(3, 3, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
    b = (torch.rand(cout, generator=g) - 0.5).cuda()
    d = e.conv_desc(tuple(x.shape), cout, 3, 3, 1, dil, "same")
    y, st = e.conv2d_fwd(x, w, b, desc=d, want_stats=True)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda()
    dx = e.conv2d_dgrad(dy, w, d)
    torch.cuda.synchronize()
    dig = hashlib.sha256()
    for t in (y, st[0] if st is not None else y[:0], dx):
        dig.update(t.detach().cpu().numpy().tobytes())
    print(f"x6w {n}x{h}x{h} {cin}->{cout} d{dil}: {dig.hexdigest()[:24]}", flush=True)

# pw_wide_kernel<1, bf16>: bf16 storage, 64-deep stages (1, 2, 3 and many stages, a ragged last stage)
for (n, h, cin, cout) in ((2, 32, 64, 384), (2, 32, 128, 384), (1, 32, 192, 400), (6, 32, 728, 728), (2, 32, 1000, 1536)):
    x = (torch.rand(n, h, h, cin, generator=g) * 2 - 1).cuda().bfloat16()
    w = ((torch.rand(1, 1, cin, cout, generator=g) * 2 - 1) * 0.05).cuda()
    b = (torch.rand(cout, generator=g) - 0.5).cuda()
    d = e.conv_desc(tuple(x.shape), cout, 1, 1, 1, 1, "same")
    y, st = e.conv2d_fwd(x, w, b, desc=d, want_stats=True)
    dy = (torch.rand(*y.shape, generator=g) * 2 - 1).cuda().bfloat16()
    dx = e.conv2d_dgrad(dy, w, d)
    torch.cuda.synchronize()
    dig = hashlib.sha256()
    for t in (y.float(), st[0] if st is not None else y[:0].float(), dx.float()):
        dig.update(t.detach().cpu().numpy().tobytes())
    print(f"bf16 {n}x{h}x{h} {cin}->{cout}: {dig.hexdigest()[:24]}", flush=True)
