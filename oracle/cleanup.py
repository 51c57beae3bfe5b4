"""CPU restatement of the ensemble's mask clean-up (oracle - test infrastructure only; the product is csrc/morph.hip).

Follows model_fuse.py of the reference object by object, with scipy.ndimage standing in for OpenCV (cv2 is not available
here, so this restatement is UNPINNED against OpenCV itself; what the tests pin is stated at each function):

    fill_and_delete(label)            model_fuse.py:9-32     fill every external contour, drop objects of contourArea <= 1000
    erode_process / erode_process1    model_fuse.py:65-115   split an object with a 1x5 / 5x1 erosion (5 iterations)
    fill_small_target                 model_fuse.py:50-62    fill the pieces, drop pieces of contourArea <= 500
    dilate_process                    model_fuse.py:35-47    dilate every piece on its own, take its external contour
    eroede_dilate_process             model_fuse.py:173-218  per object: keep, replace by its pieces, or drop
    model_confuse                     model_fuse.py:271-350  five masks -> clean -> 3-of-5 vote -> clean

OpenCV facts the restatement relies on (documented behaviour, not visible in the reference's text):
  * findContours(RETR_EXTERNAL) returns the outer border of every 8-connected component that is not enclosed by another
    one; components inside a hole of another are not reported (they are covered when the outer one is filled).
  * fillPoly / drawContours(FILLED) of such a border sets the component and everything it encloses (background is
    4-connected): "fill holes".
  * contourArea is the shoelace (Green) area of the polygon through the border pixels' centres - NOT the pixel count.
    For a hole-free 8-connected region it equals  N4 + N3 / 2  over all 2x2 pixel quads (N4: quads fully inside, N3: quads
    with three pixels inside); `contour_area` traces the border and applies the shoelace formula, `quad_area` counts quads;
    tests/test_cleanup_cpu.py checks them against each other on random shapes.
  * erode / dilate(kernel, iterations=5): anchor at the kernel centre; pixels outside the image never erode (erode's border
    value is +inf) and never dilate (-inf).
"""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
from scipy import ndimage as ndi

EIGHT = np.ones((3, 3), bool)
AREA_OBJECT = 1000   # model_fuse.py:22
AREA_PIECE = 500     # model_fuse.py:57


def contour_area(region: np.ndarray) -> float:
    """cv.contourArea of the external contour of ONE 8-connected region: Moore-neighbour border following (the outer
    border cv.findContours returns with CHAIN_APPROX_NONE: every border pixel, 8-connected steps, thin parts walked out
    and back), closed by Jacob's criterion, then the shoelace formula over the pixel centres."""
    ys, xs = np.nonzero(region)
    if len(ys) == 0:
        return 0.0
    h, w = region.shape
    r = np.zeros((h + 2, w + 2), bool)
    r[1:-1, 1:-1] = region
    y0 = int(ys.min())
    start = (y0 + 1, int(xs[ys == y0].min()) + 1)          # top-most, then left-most: its west neighbour is background
    ring = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]   # clockwise on screen, from west
    pts = [start]
    p, bd = start, 0                                         # bd: ring index of the backtrack (background) neighbour of p
    first = None
    for _ in range(16 * r.size):
        nxt = None
        for k in range(1, 9):
            d = (bd + k) % 8
            c = (p[0] + ring[d][0], p[1] + ring[d][1])
            if r[c]:
                prev = (p[0] + ring[(d - 1) % 8][0], p[1] + ring[(d - 1) % 8][1])
                nxt = (c, ring.index((prev[0] - c[0], prev[1] - c[1])))
                break
        if nxt is None:
            return 0.0                                       # an isolated pixel
        move = (p, nxt[0])
        if first is None:
            first = move
        elif move == first:
            break
        p, bd = nxt
        pts.append(p)
    else:
        raise RuntimeError("border following did not close")
    pts = pts[:-1]                                           # the walk ended back on `start`
    a = 0
    for (y1, x1), (y2, x2) in zip(pts, pts[1:] + pts[:1]):
        a += x1 * y2 - x2 * y1
    return abs(a) / 2.0


def quad_area(region: np.ndarray) -> float:
    """N4 + N3 / 2 over the 2x2 quads of a hole-free region (= contour_area, see the module text)."""
    r = np.pad(region.astype(np.int32), 1)
    s = r[:-1, :-1] + r[:-1, 1:] + r[1:, :-1] + r[1:, 1:]
    return float((s == 4).sum() + 0.5 * (s == 3).sum())


def fill_holes(mask: np.ndarray) -> np.ndarray:
    return ndi.binary_fill_holes(mask)  # background 4-connected (scipy's default cross structure)


def top_level_objects(mask: np.ndarray) -> List[np.ndarray]:
    """The filled regions findContours(RETR_EXTERNAL) + fillPoly produce, in raster order of their first pixel."""
    lab, n = ndi.label(fill_holes(mask), structure=EIGHT)
    return [lab == i for i in range(1, n + 1)]


def fill_and_delete(gray: np.ndarray):
    """model_fuse.py:9-32 -> (gray_label 0/255, list of object masks)."""
    objs = [o for o in top_level_objects(gray > 0) if quad_area(o) > AREA_OBJECT]
    out = np.zeros(gray.shape, np.uint8)
    for o in objs:
        out[o] = 255
    return out, objs


def _line(axis: int):
    return np.ones((1, 5), bool) if axis == 1 else np.ones((5, 1), bool)


def split_object(blob: np.ndarray, axis: int) -> Union[None, bool, List[np.ndarray]]:
    """erode_process (axis 1: 1x5 kernel) / erode_process1 (axis 0: 5x1) of model_fuse.py:65-115 for one object:
    None = the erosion leaves one piece (no overlap in that direction); False = pieces appeared but all were small;
    else the list of dilated-and-filled pieces (possibly empty when the erosion leaves nothing at all)."""
    st = _line(axis)
    er = ndi.binary_erosion(blob, structure=st, iterations=5, border_value=1)
    pieces = top_level_objects(er)
    if len(pieces) == 1:
        return None
    small = [quad_area(p) <= AREA_PIECE for p in pieces]
    if any(small):
        pieces = [p for p, s in zip(pieces, small) if not s]
        if not pieces:
            return False
    return [fill_holes(ndi.binary_dilation(p, structure=st, iterations=5, border_value=0)) for p in pieces]


def erode_dilate(objs: List[np.ndarray], shape) -> np.ndarray:
    """eroede_dilate_process + only_plt (model_fuse.py:173-218, 265-268) -> mask 0/255."""
    out = np.zeros(shape, np.uint8)
    for blob in objs:
        h, v = split_object(blob, 1), split_object(blob, 0)
        if h is False or v is False:
            continue
        if h is None and v is None:
            out[blob] = 255
            continue
        for part in ([] if h is None else h) + ([] if v is None else v):
            out[part] = 255
    return out


def clean(gray: np.ndarray) -> np.ndarray:
    """fill_and_delete followed by eroede_dilate_process and redrawing: what model_confuse applies to every mask."""
    _, objs = fill_and_delete(gray)
    return erode_dilate(objs, gray.shape)


def model_confuse(masks: List[np.ndarray]) -> np.ndarray:
    """model_fuse.py:271-350: clean each of the five masks, vote (>= 3 of 5), clean the vote."""
    cleaned = [clean(m) for m in masks]
    final = sum(c // 255 for c in cleaned)
    vote = np.where(final >= 3, 255, 0).astype(np.uint8)
    return clean(vote)
