"""Inference pipeline of predict.py:17-116 and the ensemble vote of model_fuse.py:315-323, on the engine.

    load_model()                 -> (res_model, hr_model, v3_model, unet_model, bam_model)   predict.py:17-54
    detection(img, user_path, model, save_name)  sliding 512-px window, stride 360, argmax, OR-merge   :90-116
    vote(masks, k=3)             -> 255 where at least k of the 5 cleaned masks agree          model_fuse.py:315-323

What changes against the reference: tiles of one image are predicted in batches on the GPU instead of one
`model.predict` per tile (BatchNorm runs on moving statistics in inference, so per-tile results do not depend
on the batching), the argmax / int8 accumulation / vote run as HIP kernels (sg_argmax_accumulate_i8,
sg_vote_ge), and images may be passed as arrays (OpenCV is not available here; PNG I/O uses Pillow).  The
contour clean-up around the vote (model_fuse.py:9-218) runs on the GPU as label-map kernels (cleanup.py, csrc/morph.hip:
`model_confuse` below); polygonisation (edge_3.py) is CPU OpenCV geometry and out of scope.

`reference_jloop=True` keeps the reference's column loop `for j in range(0, new_h-152, 360)` (predict.py:106
iterates the HEIGHT for columns): for landscape images the right-hand columns beyond new_h are never
predicted, for portrait images whose extra rows would index tiles past the canvas the reference fails inside
TensorFlow — here that case raises ValueError.  `reference_jloop=False` iterates new_w (the evident intent).
"""
from __future__ import annotations

import math
import os
from typing import List, Sequence

import numpy as np

TILE, STRIDE, OVERLAP = 512, 360, 152


def tile_origins(h: int, w: int, reference_jloop: bool = True):
    """Canvas size and tile origins exactly as predict.py:98-106 computes them."""
    h_num = math.ceil((h - OVERLAP) / STRIDE)
    w_num = math.ceil((w - OVERLAP) / STRIDE)
    new_h, new_w = h_num * STRIDE + OVERLAP, w_num * STRIDE + OVERLAP
    ch, cw = max(new_h, TILE), max(new_w, TILE)
    rows = list(range(0, new_h - OVERLAP, STRIDE))
    cols = list(range(0, (new_h if reference_jloop else new_w) - OVERLAP, STRIDE))
    if reference_jloop:
        for j in cols:
            if j + TILE > cw:
                raise ValueError(
                    f"image {h}x{w}: the reference's column loop (predict.py:106 uses new_h) indexes a tile at column {j} "
                    f"past the {cw}-px canvas; tf.keras would reject the truncated tile. Use reference_jloop=False.")
    return (ch, cw), [(i, j) for i in rows for j in cols]


def read_rgb(path: str) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"))


def detection(img, user_path=None, model=None, save_name="model", batch: int = 8, reference_jloop: bool = True):
    """predict.py:90-116 for one model.  `img`: path or uint8 RGB array [h,w,3].  Returns the uint8 mask
    (0/255) of shape [h,w]; writes `<user_path>/<save_name>.png` when user_path is given."""
    import torch
    from .ops import get_engine
    if isinstance(img, (str, os.PathLike)):
        img = read_rgb(str(img))
    arr = np.asarray(img)
    h, w = arr.shape[:2]
    x = arr.astype(np.float64) / 127.5 - 1                      # predict.py:93 (float64, like the reference)
    (ch, cw), origins = tile_origins(h, w, reference_jloop)
    canvas_img = np.zeros((ch, cw, 3))                           # zeros = mid-grey padding (:102)
    canvas_img[:h, :w, :] = x
    rt = model._runtime()
    eng = rt.eng
    pred = torch.zeros(ch, cw, dtype=torch.int8, device=eng.device)
    for s in range(0, len(origins), batch):
        chunk = origins[s:s + batch]
        tiles = np.stack([canvas_img[i:i + TILE, j:j + TILE, :] for i, j in chunk]).astype(np.float32)
        p = model.predict_device(torch.from_numpy(tiles).to(eng.device))
        for k, (i, j) in enumerate(chunk):
            eng.argmax_accumulate(p[k], pred, i, j)             # argmax (ties -> 0) and int8 `+=` (:110-113)
    out = np.where(pred.cpu().numpy() >= 1, 255, 0).astype(np.uint8)[:h, :w]   # :114
    if user_path is not None:
        from PIL import Image
        os.makedirs(user_path, exist_ok=True)
        Image.fromarray(out).save(os.path.join(user_path, f"{save_name}.png"), compress_level=0)
    return out


def vote(masks: Sequence[np.ndarray], k: int = 3) -> np.ndarray:
    """model_fuse.py:315-323: `final = sum(l_i // 255)`; `np.where(final >= 3, 255, 0)` as one HIP kernel."""
    import torch
    from .ops import get_engine
    eng = get_engine(0)
    dev = [torch.from_numpy(np.ascontiguousarray(m, dtype=np.uint8)).to(eng.device) for m in masks]
    return eng.vote_ge(dev, k).cpu().numpy()


def model_confuse(path, name: str = ""):
    """model_fuse.py:271-350: the five `*.png` masks in `path` (what run_model wrote) -> clean each, 3-of-5 vote, clean
    again -> `<path>/<name>_result.png`; returns the uint8 mask.  Also accepts a list of five arrays (then nothing is
    written).  Fewer / more than five images: prints 'no five images' and returns None, like the reference."""
    import glob
    from . import cleanup
    if isinstance(path, (str, os.PathLike)):
        files = sorted(glob.glob(os.path.join(str(path), "*.png")))
        files = [f for f in files if not f.endswith("_result.png")]
        if len(files) != 5:
            print("no five images")
            return None
        from PIL import Image
        masks = [np.asarray(Image.open(f).convert("L")) for f in files]
        out = cleanup.model_confuse(masks)
        Image.fromarray(out).save(os.path.join(str(path), f"{name}_result.png"), compress_level=0)
        return out
    if len(path) != 5:
        print("no five images")
        return None
    return cleanup.model_confuse(list(path))


def load_model(weight_dir: str = ".", shape=(512, 512, 3)):
    """predict.py:17-54: builds the five models, loads `<weight_dir>/{resnet34,hrnet,deep,scse,bam}.h5`; a
    missing file is reported and the model keeps its random initialisation, exactly like the reference."""
    from . import zoo
    specs = [("res_model", lambda: zoo.ResNetFamily(shape).run_model("res34"), "resnet34.h5"),
             ("hr_model", lambda: zoo.HRNet(shape), "hrnet.h5"),
             ("v3_model", lambda: zoo.Xception_DeepLabV3_Plus(shape), "deep.h5"),
             ("unet_model", lambda: zoo.UNet(2, shape), "scse.h5"),
             ("bam_model", lambda: zoo.Xception_DeepLabV3_Plus_bam(shape), "bam.h5")]
    models = []
    for i, (name, build, fname) in enumerate(specs, 1):
        m = build()
        try:
            m.load_weights(os.path.join(weight_dir, fname))
            print(f"load weights {name} {i}/5")
        except OSError as e:
            print(f"error while loading {name}: {e}")
        models.append(m)
    return tuple(models)


def run_model(img, user_path, models, name="", batch: int = 8, reference_jloop: bool = True) -> List[np.ndarray]:
    """predict.py:75-87: the five detections in the reference's order and file names."""
    prefixes = ["res34_", "hrnet_", "v3plus_", "scse_", "bam_"]
    return [detection(img, user_path, m, p + name, batch, reference_jloop) for m, p in zip(models, prefixes)]
