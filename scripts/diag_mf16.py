#!/usr/bin/env python3
"""Gradients of one HRNet 32x32 bs2 training step under SG_X6_MF16=0 / 1 (run once per setting, then with 'cmp')."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = "/tmp/R3t"
os.makedirs(out, exist_ok=True)
if len(sys.argv) > 1 and sys.argv[1] == "cmp":
    sd = sys.argv[2]
    a, b = np.load(f"{out}/g0_{sd}.npz"), np.load(f"{out}/g1_{sd}.npz")
    r = np.load(f"{out}/ref_{sd}.npz")
    rows = []
    for k in a.files:
        if k == "loss":
            continue
        ref = r[k].astype(np.float64)
        n = np.linalg.norm(ref) + 1e-30
        if n < 1e-9:
            continue
        rows.append((np.linalg.norm(a[k] - ref) / n, np.linalg.norm(b[k] - ref) / n, k, a[k].shape))
    print("loss mf0", a["loss"], "mf1", b["loss"], "fp64", r["loss"])
    rows.sort(key=lambda t: -t[1])
    e0s, e1s = np.array([t[0] for t in rows]), np.array([t[1] for t in rows])
    print(f"seed {sd}: median rel-L2 vs fp64 mf0 {np.median(e0s):.2e} mf1 {np.median(e1s):.2e}; max mf0 {e0s.max():.2e} mf1 {e1s.max():.2e}")
    sys.exit(0)
import torch
from building_detection_amd import zoo
from building_detection_amd.data import synthetic_batch
from building_detection_amd.losses import edge_focal_loss
size = int(os.environ.get("SIZE", "32"))
seed = int(os.environ.get("SEED", "100"))
model = zoo.BUILDERS["hrnet"]((size, size, 3))
model.compile(optimizer="adam", loss=edge_focal_loss, metrics=[])
x, y = synthetic_batch(2, size, size, seed=seed)
ws0 = model.get_weights()
logs = model.train_on_batch(x, y)
g = model.get_gradients()
names = [p.name for p in model.params if p.trainable]
tag = os.environ.get("SG_X6_MF16", "1")
np.savez(f"{out}/g{tag}_{seed}.npz", loss=logs["loss"], **{n: a for n, a in zip(names, g)})
if tag == "0":
    from oracle import models as M
    P = M.Params(weights=ws0, dtype=torch.float64)
    p = M.hrnet(P, torch.from_numpy(x).double(), training=True)
    loss = M.loss_fn("edge_focal_loss", torch.from_numpy(y).double(), p)
    loss.backward()
    np.savez(f"{out}/ref_{seed}.npz", loss=loss.item(), **{n: t.grad.numpy() for n, t in zip(names, P.trainable_tensors())})
print("done", tag, logs["loss"])
