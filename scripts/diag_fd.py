#!/usr/bin/env python3
"""Directional derivative of the training loss by central differences over a range of step sizes, fp32 and bf16 storage:
which h resolves <g, d> in each mode?   SIZE=512 BATCH=16 python scripts/diag_fd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd import zoo, mixed_precision as MP  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402
from building_detection_amd.losses import edge_focal_loss  # noqa: E402

size, bs = int(os.environ.get("SIZE", "512")), int(os.environ.get("BATCH", "16"))
x, y = synthetic_batch(bs, size, size, seed=1103)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
ws = None
for policy in ("float32", "mixed_bfloat16"):
    MP.set_global_policy(policy)
    try:
        m = zoo.Xception_DeepLabV3_Plus((size, size, 3), 2, aspp_pool=size // 16)
    finally:
        MP.set_global_policy("float32")
    m.compile(optimizer="adam", loss=edge_focal_loss, metrics=[])
    if ws is None:
        ws = m.get_weights()
    m.set_weights(ws)
    rt = m._runtime()
    w0, f0 = rt.w_train.clone(), rt.w_frozen.clone()
    loss, _ = m.train_on_batch(xd, yd, return_device_scalars=True)
    g = rt.g_train.clone()
    gen = torch.Generator(device="cpu").manual_seed(7)
    d = torch.randn(w0.numel(), generator=gen).abs().cuda()
    d *= (w0.abs() + 1e-3) * torch.sign(g)
    gd = float((g.double() * d.double()).sum().item())

    def loss_at(w):
        rt.w_train.copy_(w)
        rt.weights_changed()
        rt.w_frozen.copy_(f0)
        pr = rt.forward(xd, training=True)
        val = float(rt.eng.loss_fwd(m.loss_kind, pr, yd).item())
        rt.release()
        return val

    print(f"{policy}: loss {float(loss.item()):.6f}  <g,d> = {gd:.5e}", flush=True)
    # per layer group: the same direction restricted to the group's parameters (h = 3e-5 and 1.5e-5, extrapolated to 0)
    import importlib.util
    spec = importlib.util.spec_from_file_location("dg", os.path.join(os.path.dirname(os.path.abspath(__file__)), "diag_bf16_groups.py"))
    dg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dg)
    grp = dg.groups_of(m)
    tr = [q for q in m.params if q.trainable]
    for gname in dict.fromkeys(grp):
        mask = torch.zeros_like(w0)
        for q, gq in zip(tr, grp):
            if gq == gname:
                mask[q.offset:q.offset + q.size] = 1.0
        dgm = d * mask
        gdg = float((g.double() * dgm.double()).sum().item())
        f = []
        for h in (3e-5, 1.5e-5):
            f.append((loss_at(w0 + h * dgm) - loss_at(w0 - h * dgm)) / (2 * h))
        f0 = f[1] + (f[1] - f[0])
        print(f"   group {gname:8s}: <g,d_g> {gdg:.4e}  fd(3e-5) {f[0]:.4e} fd(1.5e-5) {f[1]:.4e} -> h=0 {f0:.4e}  ratio {f0 / gdg:.4f}", flush=True)
    for h in (1.6e-2, 8e-3, 4e-3, 2e-3, 1e-3, 5e-4, 2.5e-4, 1.25e-4, 6e-5, 3e-5, 1.5e-5):
        lp, lm = loss_at(w0 + h * d), loss_at(w0 - h * d)
        print(f"   h {h:8.2e}: L+ {lp:.7f} L- {lm:.7f}  fd {(lp - lm) / (2 * h):.5e}  ratio to <g,d> {(lp - lm) / (2 * h) / gd:.4f}", flush=True)
    del m, rt, g, d, w0, f0
    torch.cuda.empty_cache()
