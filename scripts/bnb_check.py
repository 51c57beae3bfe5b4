#!/usr/bin/env python3
"""sg_conv2d_dgrad_bnb (csrc/conv_pw.h, BNB form): the BatchNormalization backward apply evaluated in the A path of the pointwise
dgrad, against sg_bn_train_bwd_apply + sg_conv2d_dgrad - results and time, at the middle flow's shape and two others.
Use: python scripts/bnb_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
g = torch.Generator().manual_seed(4)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    z.record()
    torch.cuda.synchronize()
    return a.elapsed_time(z) / reps * 1e3


for n, h, cin, cout, relu in ((16, 32, 728, 728, True), (16, 32, 728, 728, False), (16, 32, 728, 1024, True), (16, 32, 1536, 2048, True), (16, 64, 256, 728, True)):
    t = torch.randn(n, h, h, cin, generator=g).cuda()
    w = (torch.randn(1, 1, cin, cout, generator=g) / cin ** 0.5).cuda()
    d = e.conv_desc(tuple(t.shape), cout, 1, 1, 1, 1, "same")
    y = e.conv2d_fwd(t, w, None, desc=d)
    gam, bet = (torch.rand(cout, generator=g) + 0.5).cuda(), (torch.randn(cout, generator=g) * 0.3).cuda()
    z, mean, inv = e.bn_train_fwd(y, gam, bet, torch.zeros(cout).cuda(), torch.ones(cout).cuda(), relu=relu)
    dyb = torch.randn(n, h, h, cout, generator=g).cuda()
    dz_ref, dgam, dbet = e.bn_train_bwd(y, z, dyb, gam, mean, inv, relu=relu, beta=bet)
    dx_ref = e.conv2d_dgrad(dz_ref, w, d)
    ok = e.conv2d_dgrad_bnb_ok(d)
    print(f"{n}x{h}x{h} {cin}->{cout} relu={relu}: fused form available: {ok}")
    if not ok:
        continue
    dx, dz = e.conv2d_dgrad_bnb(dyb, y, w, d, gam, bet, mean, inv, dgam, dbet, relu)
    ez = float((dz - dz_ref).abs().max()) / float(dz_ref.abs().max())
    ex = float((dx - dx_ref).abs().max()) / float(dx_ref.abs().max())
    same_z, same_x = torch.equal(dz, dz_ref), torch.equal(dx, dx_ref)
    t_apply = timed(lambda: e.bn_train_bwd_apply(y, dyb, gam, bet, mean, inv, dgam, dbet, relu=relu))
    t_dgrad = timed(lambda: e.conv2d_dgrad(dz_ref, w, d))
    t_fused = timed(lambda: e.conv2d_dgrad_bnb(dyb, y, w, d, gam, bet, mean, inv, dgam, dbet, relu))
    print(f"    dz: max rel diff {ez:.2e} (same bits: {same_z}); dx: {ex:.2e} (same bits: {same_x}); "
          f"apply {t_apply:.1f} + dgrad {t_dgrad:.1f} = {t_apply + t_dgrad:.1f} us, fused {t_fused:.1f} us")
