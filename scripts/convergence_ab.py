#!/usr/bin/env python3
"""fp32 vs mixed_bfloat16 convergence A/B of DeepLabv3+ on a fixed set of synthetic tiles, longer than the test in
tests/test_bf16_gpu.py: EPOCHS passes over NB batches of BATCH tiles of SIZE x SIZE, same initial weights, Adam 1e-3, captured
train step.  Prints the epoch means of loss and MIoU side by side.   SIZE=256 BATCH=8 NB=8 EPOCHS=40 python scripts/convergence_ab.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd import mixed_precision as MP, zoo  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402
from building_detection_amd.losses import edge_focal_loss, PA, IoU, MIoU, F1_score  # noqa: E402

size, batch = int(os.environ.get("SIZE", "256")), int(os.environ.get("BATCH", "8"))
nb, epochs = int(os.environ.get("NB", "8")), int(os.environ.get("EPOCHS", "40"))
dev = [tuple(torch.from_numpy(a).cuda() for a in synthetic_batch(batch, size, size, seed=900 + i)) for i in range(nb)]
curves, ws0 = {}, None
for dt in ("float32", "mixed_bfloat16"):
    MP.set_global_policy(dt)
    try:
        m = zoo.Xception_DeepLabV3_Plus((size, size, 3), 2, aspp_pool=size // 16)
    finally:
        MP.set_global_policy("float32")
    if ws0 is None:
        ws0 = m.get_weights()
    m.set_weights(ws0)
    m.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score], jit_compile=True)
    logs = [m.train_on_batch(*dev[s % nb]) for s in range(nb * epochs)]
    curves[dt] = (np.array([l["loss"] for l in logs]).reshape(epochs, nb).mean(1),
                  np.array([l["MIoU"] for l in logs]).reshape(epochs, nb).mean(1))
(la, ma), (lb, mb) = curves["float32"], curves["mixed_bfloat16"]
print(f"DeepLabv3+ {size}x{size}, {nb} batches of {batch} tiles, {epochs} epochs ({nb * epochs} Adam steps), epoch means")
print("epoch   loss fp32   loss bf16   MIoU fp32   MIoU bf16")
for e in range(epochs):
    print(f"{e + 1:5d}   {la[e]:9.5f}   {lb[e]:9.5f}   {ma[e]:9.4f}   {mb[e]:9.4f}")
print(f"final loss fp32 {la[-1]:.5f} ({la[-1] / la[0] * 100:.2f} % of the first epoch), bf16 {lb[-1]:.5f} ({lb[-1] / lb[0] * 100:.2f} %); "
      f"largest |MIoU gap| {float(np.max(np.abs(ma - mb))):.4f}, final MIoU fp32 {ma[-1]:.4f} bf16 {mb[-1]:.4f}")
