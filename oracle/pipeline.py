"""CPU restatement of the inference pipeline (oracle — test infrastructure only).

`detection_ref` follows predict.py:90-116 line by line (including its column loop over `new_h`, :106) with the
model replaced by any `predict_fn(tile[1,512,512,3] float64) -> probs[1,512,512,2]`; `vote_ref` follows
model_fuse.py:315-323.  File I/O (cv.imread / cv.imwrite) is outside the restated path.
"""
from __future__ import annotations

import math

import numpy as np


def detection_ref(img_rgb_u8: np.ndarray, predict_fn) -> np.ndarray:
    img = img_rgb_u8 / 127.5 - 1                                     # predict.py:93
    h, w, c = img.shape
    h_num = math.ceil((h - 152) / 360)
    w_num = math.ceil((w - 152) / 360)
    new_h = h_num * 360 + 152
    new_w = w_num * 360 + 152
    tmp_img = np.zeros((max(new_h, 512), max(new_w, 512), 3))
    pred_result = np.zeros((max(new_h, 512), max(new_w, 512)), np.int8)
    tmp_img[:h, :w, :] = img
    for i in range(0, new_h - 152, 360):
        for j in range(0, new_h - 152, 360):                          # predict.py:106 (new_h, as in the reference)
            test_part = tmp_img[i:i + 512, j:j + 512, :]
            test_part = np.expand_dims(test_part, axis=0)
            pred_part = predict_fn(test_part)
            pred_part = np.argmax(pred_part, axis=-1)                 # tf.argmax: lowest index on ties
            pred_part = np.squeeze(pred_part)
            pred_result[i:i + 512, j:j + 512] += pred_part.astype(np.int8)
    pred_result = np.where(pred_result >= 1, 255, 0)
    return pred_result[:h, :w].astype(np.uint8)


def vote_ref(masks, k: int = 3) -> np.ndarray:
    final = sum(m // 255 for m in masks)                              # model_fuse.py:315
    return np.array(np.where(final >= k, 255, 0), np.uint8)           # :323-324
