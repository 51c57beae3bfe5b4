#!/usr/bin/env python3
"""Diagnostic: conv3x3 -> BN(train)+ReLU -> conv1x1(2) -> softmax -> edge_focal_loss, run op by op through the
engine and compared tensor by tensor with torch autograd (fp64) to localise a backward bug."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from building_detection_amd.ops import get_engine  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402
from oracle import tfops as T  # noqa: E402
from oracle import models as M  # noqa: E402

e = get_engine(0)
g = torch.Generator().manual_seed(0)
N, H, W, C0, C1 = 2, 64, 64, 128, 64


def rel(a, b, name):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    s = b.abs().max().item()
    err = (a - b).abs().max().item()
    print(f"{name:28s} rel {err / max(s, 1e-30):.3e}  (scale {s:.3e})")


for trial, (relu_in, cin) in enumerate([(True, C0), (False, C0)]):
    print(f"--- trial {trial}: input {'post-ReLU (positive mean)' if relu_in else 'zero-mean'}")
    x = torch.randn(N, H, W, cin, generator=g)
    if relu_in:
        x = torch.relu(x) + 0.1
    w1 = torch.randn(3, 3, cin, C1, generator=g) * 0.05
    b1 = torch.zeros(C1)
    gam, bet = torch.ones(C1), torch.zeros(C1)
    w2 = torch.randn(1, 1, C1, 2, generator=g) * 0.3
    b2 = torch.zeros(2)
    _, y = synthetic_batch(N, H, W, seed=5)
    yt = torch.from_numpy(y)

    # fp64 reference with autograd, keeping intermediates
    D = torch.float64
    xr = x.to(D)
    ps = [t.to(D).clone().requires_grad_() for t in (w1, b1, gam, bet, w2, b2)]
    z1 = T.conv2d(xr, ps[0], ps[1]); z1.retain_grad()
    a1, _, _ = T.batch_norm(z1, ps[2], ps[3], torch.zeros(C1, dtype=D), torch.ones(C1, dtype=D), True)
    y1 = torch.relu(a1); y1.retain_grad()
    z2 = T.conv2d(y1, ps[4], ps[5]); z2.retain_grad()
    p = torch.softmax(z2, -1); p.retain_grad()
    loss = M.loss_fn("edge_focal_loss", yt.to(D), p)
    loss.backward()

    # engine, op by op
    xd = x.cuda()
    w1d, b1d, gd, bd, w2d, b2d = [t.cuda() for t in (w1, b1, gam, bet, w2, b2)]
    mm, mv = torch.zeros(C1).cuda(), torch.ones(C1).cuda()
    z1g = e.conv2d_fwd(xd, w1d, b1d)
    rel(z1g, z1, "z1 = conv3x3")
    y1g, mean, invstd = e.bn_train_fwd(z1g, gd, bd, mm, mv, relu=True)
    rel(y1g, y1, "y1 = relu(bn(z1))")
    z2g = e.conv2d_fwd(y1g, w2d, b2d)
    pg = e.softmax2_fwd(z2g)
    rel(pg, p, "p")
    ytd = yt.cuda()
    rel(e.loss_fwd(2, pg, ytd), loss.reshape(1), "loss")
    dp = e.loss_bwd(2, pg, ytd)
    rel(dp, p.grad, "dL/dp")
    dz2 = e.softmax2_bwd(pg, dp)
    rel(dz2, z2.grad, "dL/dz2")
    d2 = e.conv_desc(tuple(y1g.shape), 2, 1, 1)
    dw2, db2 = e.conv2d_wgrad(y1g, dz2, d2)
    rel(dw2, ps[4].grad, "dw2"); rel(db2, ps[5].grad, "db2")
    dy1 = e.conv2d_dgrad(dz2, w2d, d2)
    rel(dy1, y1.grad, "dL/dy1 (head dgrad)")
    dz1, dgam, dbet = e.bn_train_bwd(z1g, y1g, dy1, gd, mean, invstd, relu=True)
    rel(dgam, ps[2].grad, "dgamma"); rel(dbet, ps[3].grad, "dbeta")
    rel(dz1, z1.grad, "dL/dz1 (bn bwd dx)")
    # same BN backward but fed the exact upstream gradient
    dz1b, dgamb, dbetb = e.bn_train_bwd(z1g, y1g, y1.grad.float().cuda(), gd, mean, invstd, relu=True)
    rel(dbetb, ps[3].grad, "dbeta (exact dy1 in)"); rel(dz1b, z1.grad, "bn dx (exact dy1 in)")
    d1 = e.conv_desc(tuple(xd.shape), C1, 3, 3)
    dw1, db1g = e.conv2d_wgrad(xd, dz1, d1)
    rel(dw1, ps[0].grad, "dw1 (engine chain)")
    dw1b, _ = e.conv2d_wgrad(xd, z1.grad.float().cuda(), d1)
    rel(dw1b, ps[0].grad, "dw1 (exact dz1 in)")
    # fp32 torch for reference noise level
    ps32 = [t.clone().requires_grad_() for t in (w1, b1, gam, bet, w2, b2)]
    z = T.conv2d(x, ps32[0], ps32[1])
    a, _, _ = T.batch_norm(z, ps32[2], ps32[3], torch.zeros(C1), torch.ones(C1), True)
    l32 = M.loss_fn("edge_focal_loss", yt, torch.softmax(T.conv2d(torch.relu(a), ps32[4], ps32[5]), -1))
    l32.backward()
    rel(ps32[0].grad, ps[0].grad, "dw1 torch-fp32 vs fp64")
    rel(ps32[3].grad, ps[3].grad, "dbeta torch-fp32 vs fp64")
