"""oracle/cleanup.py - the CPU restatement of model_fuse.py's mask clean-up (the checker of tests/test_cleanup_gpu.py).
UNPINNED against OpenCV (cv2 is absent); pinned here: the contour-area identity the GPU relies on, and the documented
OpenCV semantics on hand-made cases (rectangle area (w-1)(h-1), hole filling, the drop / split / keep rules)."""
import numpy as np
import pytest
from scipy import ndimage as ndi

from oracle import cleanup as CL


def blobs(rng, h, w, sigma, thr, speckle=0.0):
    m = ndi.gaussian_filter(rng.standard_normal((h, w)), sigma) > thr
    if speckle:
        m |= rng.random((h, w)) > 1 - speckle
    return m


def test_contour_area_equals_quad_count_on_random_shapes():
    """cv.contourArea (border following + shoelace) == N4 + N3/2 for every hole-free 8-connected region: the identity
    that lets the GPU take the area from 2x2 pixel patterns with integer atomics."""
    rng = np.random.default_rng(0)
    n = 0
    for it in range(120):
        m = blobs(rng, 40, 48, rng.uniform(0.6, 3), rng.uniform(-0.05, 0.2), 0.03 if it % 3 == 0 else 0.0)
        for o in CL.top_level_objects(m):
            assert CL.contour_area(o) == CL.quad_area(o)
            n += 1
    assert n > 1500
    r = np.zeros((20, 20), bool)
    r[3:10, 4:16] = True
    assert CL.contour_area(r) == (7 - 1) * (12 - 1)           # OpenCV: a w x h pixel rectangle has contourArea (w-1)(h-1)
    line = np.zeros((9, 9), bool)
    line[4, 1:8] = True
    assert CL.contour_area(line) == 0 and CL.contour_area(np.eye(7, dtype=bool)) == 0
    one = np.zeros((5, 5), bool)
    one[2, 2] = True
    assert CL.contour_area(one) == 0


def test_fill_and_delete_rules():
    g = np.zeros((120, 160), np.uint8)
    g[10:50, 10:60] = 255          # 40 x 50: area 39*49 = 1911 > 1000, with a hole and an island inside the hole
    g[20:40, 20:50] = 0
    g[28:32, 30:36] = 255
    g[70:100, 10:45] = 255         # 30 x 35: area 29*34 = 986 <= 1000 -> deleted
    g[70:103, 60:93] = 255         # 33 x 33: 32*32 = 1024 > 1000 -> kept
    g[5:8, 100:150] = 255          # thin: area 2*49 = 98 -> deleted
    out, objs = CL.fill_and_delete(g)
    assert len(objs) == 2
    assert out[10:50, 10:60].min() == 255                      # hole (and island) filled
    assert out[70:100, 10:45].max() == 0 and out[5:8, 100:150].max() == 0
    assert out[70:103, 60:93].min() == 255 and out.sum() == 255 * (40 * 50 + 33 * 33)


def test_split_rules():
    # two 40x40 blocks joined by a 6-px-wide vertical corridor: the 1x5 erosion (x5 = 21 wide) cuts the corridor
    g = np.zeros((140, 80), bool)
    g[10:50, 10:50] = True
    g[50:80, 27:33] = True
    g[80:120, 10:50] = True
    h = CL.split_object(g, 1)
    assert isinstance(h, list) and len(h) == 2                  # two pieces, each dilated back to its block
    assert h[0].sum() == 40 * 40 and h[1].sum() == 40 * 40
    assert CL.split_object(g, 0) is None                        # vertically everything hangs together
    out = CL.erode_dilate([g], g.shape)
    assert out[55:75, 27:33].max() == 0 and out[10:50, 10:50].min() == 255   # the corridor is gone, the blocks stay
    # an object narrower than 21 px: the horizontal erosion leaves NOTHING -> an empty list (not None): the object vanishes
    thin = np.zeros((100, 60), bool)
    thin[10:90, 20:38] = True
    assert CL.split_object(thin, 1) == [] and CL.split_object(thin, 0) is None
    assert CL.erode_dilate([thin], thin.shape).max() == 0
    # pieces that are all small -> False -> the object is dropped
    dumb = np.zeros((80, 120), bool)
    dumb[10:34, 10:40] = True
    dumb[34:46, 22:28] = True
    dumb[46:70, 10:40] = True                                    # eroded blocks: 24 x 10 -> area 23*9 = 207 <= 500 each
    assert CL.split_object(dumb, 1) is False
    assert CL.erode_dilate([dumb], dumb.shape).max() == 0
    # a compact object is kept as it is
    sq = np.zeros((90, 90), bool)
    sq[20:70, 20:70] = True
    assert CL.split_object(sq, 1) is None and CL.split_object(sq, 0) is None
    assert np.array_equal(CL.erode_dilate([sq], sq.shape) > 0, sq)


def test_image_border_never_erodes():
    g = np.zeros((60, 100), bool)
    g[0:30, 0:30] = True                                        # touches the top-left corner
    er = ndi.binary_erosion(g, structure=np.ones((1, 5), bool), iterations=5, border_value=1)
    assert er[5, 0] and er[5, 19] and not er[5, 20]             # eroded from the right only
    assert CL.split_object(g, 1) is None


def test_model_confuse_votes_between_cleanings():
    rng = np.random.default_rng(3)
    base = np.zeros((200, 240), np.uint8)
    base[30:100, 30:120] = 255
    base[120:180, 140:220] = 255
    masks = []
    for i in range(5):
        m = base.copy()
        if i < 2:
            m[120:180, 140:220] = 0                              # only three of five models see the second building
        m[rng.integers(0, 190, 30), rng.integers(0, 230, 30)] = 255   # speckle: removed by the area rule
        masks.append(m)
    out = CL.model_confuse(masks)
    assert out[30:100, 30:120].min() == 255 and out[120:180, 140:220].min() == 255
    assert out.sum() == 255 * (70 * 90 + 60 * 80)
