"""Mask clean-up and ensemble fusion of model_fuse.py on the GPU.

    fill_and_delete(mask)        model_fuse.py:9-32     -> (gray_label, object table)        sg_mask_objects
    eroede_dilate_process(...)   model_fuse.py:173-218  -> cleaned mask                      sg_mask_split
    clean(mask)                  the two in sequence, as model_confuse applies them to every mask
    model_confuse(masks)         model_fuse.py:271-350: clean x5 -> 3-of-5 vote (sg_vote_ge) -> clean

Masks are uint8 [H, W] (0 / 255) as numpy arrays or device tensors; the result has the type of the input.  What each
OpenCV call of the reference means on a mask (findContours RETR_EXTERNAL + fillPoly = fill holes of the top-level
8-connected objects; contourArea = the Green area of the border polygon = N4 + N3/2 over 2x2 pixel quads; erode / dilate
border rules) is restated on the CPU in oracle/cleanup.py, which the GPU tests compare against bit for bit.  There is no
CPU path here.  The host reads the object table once per clean-up (object count and bounding boxes size the per-object
scratch): this is post-processing, not the training hot path.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib

AREA_OBJECT = 1000  # model_fuse.py:22  `if area <= 1000`
AREA_PIECE = 500    # model_fuse.py:57  `if area <= 500`


def _dev(mask, eng):
    import torch
    if isinstance(mask, torch.Tensor):
        t = mask.to(eng.device)
    else:
        t = torch.from_numpy(np.ascontiguousarray(mask)).to(eng.device)
    if t.dim() == 3:  # label[:, :, 0] of a BGR image (model_fuse.py:10)
        t = t[..., 0]
    if t.dtype != torch.uint8:
        t = (t != 0).to(torch.uint8) * 255
    return t.contiguous()


def _back(t, like):
    import torch
    return t if isinstance(like, torch.Tensor) else t.cpu().numpy()


def fill_and_delete(mask, engine=None, max_objs: int = 1 << 16):
    """-> (gray_label u8 device tensor, labels i32 device tensor, object table as a numpy array [n, 8])."""
    import torch
    from .ops import get_engine
    eng = engine or get_engine(0)
    m = _dev(mask, eng)
    h, w = m.shape
    lib = eng.lib
    with eng.lock:
        ws = torch.empty(lib.sg_mask_objects_ws_bytes(h, w), dtype=torch.uint8, device=eng.device)
        labels = torch.empty(h, w, dtype=torch.int32, device=eng.device)
        kept = torch.empty(h, w, dtype=torch.uint8, device=eng.device)
        while True:
            table = torch.empty(max_objs, 8, dtype=torch.int32, device=eng.device)
            count = torch.zeros(1, dtype=torch.int32, device=eng.device)
            _lib.check(lib.sg_mask_objects(eng.h, eng.stream, h, w, C.c_void_p(m.data_ptr()), 2 * AREA_OBJECT,
                                           C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(labels.data_ptr()),
                                           C.c_void_p(table.data_ptr()), max_objs, C.c_void_p(count.data_ptr()),
                                           C.c_void_p(kept.data_ptr())), "sg_mask_objects")
            n = int(count.item())
            if n <= max_objs:
                break
            max_objs = 1 << int(np.ceil(np.log2(n + 1)))
        return kept, labels, table[:n].cpu().numpy(), table


def eroede_dilate_process(labels, table_host, table_dev, shape, engine=None):
    """The kept objects of `fill_and_delete` split / kept / dropped as model_fuse.py:173-218 does -> mask u8 device tensor."""
    import torch
    from .ops import get_engine
    eng = engine or get_engine(0)
    h, w = shape
    lib = eng.lib
    out = torch.zeros(h, w, dtype=torch.uint8, device=eng.device)
    objs = np.nonzero(table_host[:, 6])[0].astype(np.int32) if len(table_host) else np.zeros(0, np.int32)
    if len(objs) == 0:
        return out
    words = np.array([lib.sg_mask_split_words(h, w, int(r[2]), int(r[3]), int(r[4]), int(r[5])) for r in table_host[objs]], np.int64)
    offs = np.concatenate([[0], np.cumsum(words)[:-1]]).astype(np.int64)
    with eng.lock:
        ws = torch.empty(int(words.sum()), dtype=torch.int32, device=eng.device)
        objs_d = torch.from_numpy(objs).to(eng.device)
        offs_d = torch.from_numpy(offs).to(eng.device)
        _lib.check(lib.sg_mask_split(eng.h, eng.stream, h, w, C.c_void_p(labels.data_ptr()), C.c_void_p(table_dev.data_ptr()),
                                     C.c_void_p(objs_d.data_ptr()), C.c_void_p(offs_d.data_ptr()), len(objs), 2 * AREA_PIECE,
                                     C.c_void_p(ws.data_ptr()), C.c_void_p(out.data_ptr())), "sg_mask_split")
        torch.cuda.current_stream(eng.device).synchronize()  # ws / objs_d / offs_d are released when this returns
    return out


def clean(mask, engine=None):
    """fill_and_delete -> eroede_dilate_process -> redraw (what model_confuse does to each of its six masks)."""
    _, labels, table_host, table_dev = fill_and_delete(mask, engine)
    out = eroede_dilate_process(labels, table_host, table_dev, tuple(labels.shape), engine)
    return _back(out, mask)


def model_confuse(masks: Sequence, engine=None):
    """model_fuse.py:271-350 on five masks (arrays or device tensors, 0/255): clean each, vote >= 3, clean the vote."""
    from .ops import get_engine
    eng = engine or get_engine(0)
    if len(masks) != 5:
        raise ValueError("no five images")  # model_fuse.py:283-285 prints this and returns
    cleaned = [_dev(clean(_dev(m, eng), eng), eng) for m in masks]
    vote = eng.vote_ge([c.view(-1) for c in cleaned], 3).view(cleaned[0].shape)
    return _back(_dev(clean(vote, eng), eng), masks[0])
