#!/usr/bin/env python3
"""Where does the bf16-storage gradient differ from the fp32 one?  Per-tensor relative error and cosine, DeepLabv3+ 128x128,
for (a) mixed_bfloat16 storage, (b) fp32 storage with one-pass bf16 products (sg_set_conv_x6(2))."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd import zoo, mixed_precision as MP  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402
from building_detection_amd.losses import edge_focal_loss  # noqa: E402
from building_detection_amd.ops import get_engine  # noqa: E402

name = os.environ.get("MODEL", "v3plus")
size = int(os.environ.get("SIZE", "128"))
bs = int(os.environ.get("BATCH", "2"))
kw = {"aspp_pool": size // 16} if name in ("v3plus", "bam") else {}


def build(policy):
    MP.set_global_policy(policy)
    m = zoo.BUILDERS[name]((size, size, 3), 2, **kw) if kw else zoo.BUILDERS[name]((size, size, 3))
    MP.set_global_policy("float32")
    m.compile(optimizer="adam", loss=edge_focal_loss, metrics=[])
    return m


x, y = synthetic_batch(bs, size, size, seed=11)
eng = get_engine(0)
m32 = build("float32")
ws = m32.get_weights()
l32 = m32.train_on_batch(x, y)["loss"]
g32 = m32.get_gradients()
m16 = build("mixed_bfloat16")
m16.set_weights(ws)
l16 = m16.train_on_batch(x, y)["loss"]
g16 = m16.get_gradients()
prev = eng.lib.sg_set_conv_x6(2)
m2 = build("float32")
m2.set_weights(ws)
l2 = m2.train_on_batch(x, y)["loss"]
g2 = m2.get_gradients()
eng.lib.sg_set_conv_x6(prev)
names = [p.name for p in m32.params if p.trainable]
tot = sum(float(np.square(g.astype(np.float64)).sum()) for g in g32)


def summary(tag, gs, loss):
    num = sum(float(np.square(a.astype(np.float64) - b).sum()) for a, b in zip(gs, g32))
    dot = sum(float((a.astype(np.float64) * b).sum()) for a, b in zip(gs, g32))
    na = sum(float(np.square(a.astype(np.float64)).sum()) for a in gs)
    print(f"{tag}: loss {loss:.6f} (fp32 {l32:.6f}); rel-L2 {np.sqrt(num / tot):.3e}, cosine {dot / np.sqrt(na * tot):.4f}")


summary("bf16 storage", g16, l16)
summary("fp32 storage, bf16 products", g2, l2)
rows = []
for n, a, b, c in zip(names, g32, g16, g2):
    e = float(np.square(a.astype(np.float64)).sum())
    if e <= 0:
        continue
    r16 = float(np.sqrt(np.square(b.astype(np.float64) - a).sum() / e))
    r2 = float(np.sqrt(np.square(c.astype(np.float64) - a).sum() / e))
    rows.append((e / tot, n, r16, r2))
rows.sort(reverse=True)
print("share of |g|^2   rel err bf16-storage   rel err bf16-products   tensor")
for sh, n, r16, r2 in rows[:40]:
    print(f"{sh:10.3e}   {r16:10.3e}   {r2:10.3e}   {n}")
