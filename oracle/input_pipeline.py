"""CPU restatement of the reference's training input pipeline - TEST INFRASTRUCTURE ONLY (nothing under
building_detection_amd/ imports this; tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may).

Follows /root/reference/train_model/DeepLabv3plus.py (the other four training scripts carry the same text):
    decode_img      :32-39    cv.imread -> BGR2RGB -> cv.resize(512,512) -> float32 / 127.5 - 1
    decode_lbel     :42-50    cv.imread -> BGR2GRAY -> cv.resize(512,512) -> [...,None] float32 / 255
    train_data_gen  :53-107   sort, itertools.cycle, to_categorical(label, 2), 5 x erode / dilate edge bands,
                              np.concatenate((one_hot, f_edge, p_edge))
    val_data_gen    :110-153  the same text over the validation lists

PARITY UNPINNED: the arithmetic lives in OpenCV (opencv-python, version unpinned by the reference, absent from this
image: `import cv2` -> ModuleNotFoundError) and the reference holds no fixture for it.  What is restated here is
OpenCV's PUBLISHED algorithm for each call, written as plain integer numpy so that the GPU kernels can be held to it
bit for bit:

  cv.cvtColor(BGR2GRAY), 8-bit     (R*4899 + G*9617 + B*1868 + 8192) >> 14   (imgproc color_rgb: 14-bit weights)
  cv.resize(.., (w,h)) INTER_LINEAR, 8-bit
      coefficients  fx = float((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx; sx < 0 -> (0, fx = 0);
                    sx >= n-1 -> (n-1, fx = 0); a = round_half_even((1 - fx) * 2048), round_half_even(fx * 2048) as int16
                    rows: same, the two source rows clamped to [0, n-1] with the coefficients unchanged
      horizontal    D = S[sx] * a0 + S[sx+1] * a1                       (int32, scale 2^11)
      vertical      ((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2, saturated to 8 bits
                    - the arithmetic of the vectorised row kernel (VResizeLinearVec_32s8u) that every SIMD build runs
                    on whole vectors; the scalar tail formula (b0*D0 + b1*D1 + 2^21) >> 22 can differ from it by one
                    grey level and is only reached when (out_width * channels) is not a multiple of the vector length
                    (512 * 3 and 512 * 1 are multiples of 64)
      exact 2x downscale: resize() turns INTER_LINEAR into the fast INTER_AREA: (a + b + c + d + 2) >> 2 per 2x2 block
      same size: the coefficients are (2048, 0): a copy
  cv.erode / cv.dilate(label, ones((3,3)), iterations=5)   min / max over the 3x3 neighbourhood, five times; cells outside
                    the image do not take part (the default border value is +inf for erode, -inf for dilate)
  tf.keras.utils.to_categorical(label, 2)                  integer truncation: only label == 1.0 (grey 255) is class 1

File decoding itself (PNG / TIFF -> 8-bit RGB) goes through Pillow: there is no second decoder in this image.
Everything is written with explicit loops over the (small) filter windows, independent of the product's scipy / float code.
"""
from __future__ import annotations

import itertools

import numpy as np

SIZE = 512
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def imread_rgb(path) -> np.ndarray:
    """cv.cvtColor(cv.imread(path), cv.COLOR_BGR2RGB): uint8 [H,W,3]; 16-bit files reduced to 8 bits as imread's default flag does."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            a = np.asarray(im, np.uint32)
            im = Image.fromarray((a >> 8).astype(np.uint8) if a.max() > 255 else a.astype(np.uint8))
        return np.asarray(im.convert("RGB"), np.uint8)


def bgr2gray_u8(rgb: np.ndarray) -> np.ndarray:
    r, g, b = (rgb[..., i].astype(np.int64) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def _round_half_even_i16(v: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(v), -32768, 32767).astype(np.int64)  # np.rint rounds half to even, as cvRound does


def linear_coeffs(n_in: int, n_out: int):
    """(index of the first tap, int16 weight of tap 0, of tap 1) for every output position; float32 fx as in resize()."""
    scale = n_in / n_out
    d = np.arange(n_out, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def resize_linear_u8(img: np.ndarray, size=(SIZE, SIZE)) -> np.ndarray:
    """cv.resize(img, (ow, oh)) of an 8-bit image [H,W] or [H,W,C], default interpolation (header above)."""
    ow, oh = size
    h, w = img.shape[:2]
    a = img.reshape(h, w, -1).astype(np.int64)
    if h == 2 * oh and w == 2 * ow:  # INTER_LINEAR -> fast INTER_AREA
        out = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
        return out.astype(np.uint8).reshape((oh, ow) + img.shape[2:])
    sx, fx = linear_coeffs(w, ow)
    lo, hi = sx < 0, sx >= w - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, w - 1, sx))
    a0 = _round_half_even_i16((np.float32(1) - fx) * np.float32(COEF_SCALE))
    a1 = _round_half_even_i16(fx * np.float32(COEF_SCALE))
    sx1 = np.minimum(sx + 1, w - 1)          # never read with a non-zero weight when sx == w-1
    sy, fy = linear_coeffs(h, oh)
    b0 = _round_half_even_i16((np.float32(1) - fy) * np.float32(COEF_SCALE))
    b1 = _round_half_even_i16(fy * np.float32(COEF_SCALE))
    y0, y1 = np.clip(sy, 0, h - 1), np.clip(sy + 1, 0, h - 1)
    out = np.empty((oh, ow, a.shape[2]), np.uint8)
    for dy in range(oh):  # explicit row loop, as resizeGeneric_ walks it
        d0 = a[y0[dy]][sx] * a0[:, None] + a[y0[dy]][sx1] * a1[:, None]
        d1 = a[y1[dy]][sx] * a0[:, None] + a[y1[dy]][sx1] * a1[:, None]
        v = (((b0[dy] * (d0 >> 4)) >> 16) + ((b1[dy] * (d1 >> 4)) >> 16) + 2) >> 2
        out[dy] = np.clip(v, 0, 255).astype(np.uint8)
    return out.reshape((oh, ow) + img.shape[2:])


def decode_img(img_path) -> np.ndarray:
    img = resize_linear_u8(imread_rgb(img_path))
    return np.array(img, np.float32) / 127.5 - 1


def decode_lbel(label_path) -> np.ndarray:
    label = resize_linear_u8(bgr2gray_u8(imread_rgb(label_path)))
    return np.array(label[..., np.newaxis], np.float32) / 255


def to_categorical(label: np.ndarray, num_classes: int = 2) -> np.ndarray:
    y = np.array(label, dtype="int").reshape(label.shape[:-1] if label.shape[-1] == 1 else label.shape)
    out = np.zeros(y.shape + (num_classes,), np.float32)
    for c in range(num_classes):
        out[..., c] = (y == c)
    return out


def _morph3x3(a: np.ndarray, op, border) -> np.ndarray:
    """One pass of a 3x3 rectangular erode (op = np.minimum, border = +inf) / dilate (np.maximum, -inf)."""
    h, w = a.shape
    p = np.full((h + 2, w + 2), border, a.dtype)
    p[1:-1, 1:-1] = a
    out = a.copy()
    for dy in range(3):
        for dx in range(3):
            out = op(out, p[dy:dy + h, dx:dx + w])
    return out


def erode(a: np.ndarray, iterations: int = 5) -> np.ndarray:
    for _ in range(iterations):
        a = _morph3x3(a, np.minimum, np.inf)
    return a


def dilate(a: np.ndarray, iterations: int = 5) -> np.ndarray:
    for _ in range(iterations):
        a = _morph3x3(a, np.maximum, -np.inf)
    return a


def label_channels(label: np.ndarray, iterations: int = 5) -> np.ndarray:
    """label [H,W] float32 (grey / 255, not necessarily binary) -> [H,W,4] float64 = (one_hot, f_edge, p_edge), :70-100."""
    label = np.asarray(label, np.float32)
    one_hot = to_categorical(label[..., None], 2)
    p_edge = np.where((label - erode(label, iterations)) == 1, 2.0, 1.0)
    f_edge = np.where((dilate(label, iterations) - label) == 1, 2.0, 1.0)
    return np.concatenate((one_hot, f_edge[..., None], p_edge[..., None]), axis=-1)


def data_gen(img_path, lab_path, BATCH_SIZE, loss="edge_focal_loss"):
    """train_data_gen / val_data_gen (label_smooth=False: the True branch reads names the reference never defines)."""
    images, label = img_path, lab_path
    images.sort()
    label.sort()
    zipped = itertools.cycle(zip(images, label))
    while True:
        x_train, y_train = [], []
        for _ in range(BATCH_SIZE):
            img, seg = next(zipped)
            x = decode_img(img)
            lab = decode_lbel(seg)
            y = label_channels(np.squeeze(lab)) if loss == "edge_focal_loss" else to_categorical(lab, 2)
            x_train.append(x)
            y_train.append(y)
        yield np.array(x_train), np.array(y_train)
