"""Seeded synthetic tiles shaped like the reference's generator output (train_model/DeepLabv3plus.py:32-107).

x: uint8 RGB noise -> `/127.5 - 1` float32 [N,H,W,3] (decode_img, :36-37).
y: [N,H,W,4] = one-hot(background, building), f_edge weight, p_edge weight, built like train_data_gen
(:70-100): the binary mask comes from random filled rectangles; `erode`/`dilate` with a 3x3 kernel, 5
iterations, restated with scipy min/max filters (OpenCV is not available): cv.erode pads with +inf, cv.dilate
with -inf; p_edge = 2 where mask - erode == 1 (inner building rim), f_edge = 2 where dilate - mask == 1 (outer
rim), else 1; channel order (one_hot, f_edge, p_edge) as in `np.concatenate` at :100.
"""
from __future__ import annotations

import numpy as np


def edge_weight_channels(mask: np.ndarray):
    from scipy import ndimage
    m = mask.astype(np.float32)
    er, di = m, m
    for _ in range(5):
        er = ndimage.minimum_filter(er, size=3, mode="constant", cval=np.inf)
        di = ndimage.maximum_filter(di, size=3, mode="constant", cval=-np.inf)
    p_edge = np.where((m - er) == 1, 2.0, 1.0)
    f_edge = np.where((di - m) == 1, 2.0, 1.0)
    return f_edge, p_edge


def synthetic_batch(n: int, h: int = 512, w: int = 512, seed: int = 1103):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8).astype(np.float32) / 127.5 - 1.0
    y = np.empty((n, h, w, 4), np.float32)
    for i in range(n):
        mask = np.zeros((h, w), np.float32)
        for _ in range(int(rng.integers(3, 13))):
            rh = int(rng.integers(max(h // 32, 2), max(h // 4, 3)))
            rw = int(rng.integers(max(w // 32, 2), max(w // 4, 3)))
            r0, c0 = int(rng.integers(0, h - rh)), int(rng.integers(0, w - rw))
            mask[r0:r0 + rh, c0:c0 + rw] = 1.0
        f_edge, p_edge = edge_weight_channels(mask)
        y[i, ..., 0], y[i, ..., 1], y[i, ..., 2], y[i, ..., 3] = 1.0 - mask, mask, f_edge, p_edge
    return x.astype(np.float32), y
