#!/usr/bin/env python3
"""Diagnostic: per-parameter gradient error of the engine vs the fp64 oracle, listed in backward order
(output side first) so the first bad layer localises a backward bug.  usage: diag_grads.py <model> [size]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from building_detection_amd import zoo  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402
from building_detection_amd.losses import edge_focal_loss  # noqa: E402
from oracle import models as M  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hrnet"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kw = {"aspp_pool": size // 16} if name in ("v3plus", "bam") else {}
model = zoo.BUILDERS[name]((size, size, 3), 2, **kw) if kw else zoo.BUILDERS[name]((size, size, 3))
x, y = synthetic_batch(2, size, size, seed=23)
ws0 = model.get_weights()
model.compile(optimizer="adam", loss=edge_focal_loss, metrics=[])
model.train_on_batch(x, y)
gg = model.get_gradients()
P = M.Params(weights=ws0, dtype=torch.float64)
p = M.BUILDERS[name](P, torch.from_numpy(x).double(), training=True, **kw)
M.loss_fn("edge_focal_loss", torch.from_numpy(y).double(), p).backward()
g64 = [t.grad.numpy() for t in P.trainable_tensors()]
specs = [(n, prm) for n in model.nodes for prm in n.params if prm.trainable]
rows = []
for (node, prm), a, b in zip(specs, gg, g64):
    scale = float(np.abs(b).max())
    err = float(np.abs(a - b).max())
    rows.append((node.index, node.name, node.op, prm.kind, prm.shape, err / max(scale, 1e-30), scale,
                 tuple(node.inputs[0].shape), getattr(node, "stride", ""), getattr(node, "relu", ""), getattr(node, "pre_relu", "")))
print(f"{name} {size}: {len(rows)} trainable tensors; listing backward order, rel err > 1e-3 flagged")
for r in sorted(rows, key=lambda r: -r[0]):
    flag = "  <<<<" if r[5] > 1e-3 and r[6] > 1e-9 else ""
    print(f"{r[0]:4d} {r[1]:28s} {r[2]:18s} {r[3]:18s} {str(r[4]):22s} rel {r[5]:.2e} scale {r[6]:.2e} in {r[7]} s={r[8]} relu={r[9]} pre={r[10]}{flag}")
