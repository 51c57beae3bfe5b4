"""the mask clean-up of model_fuse.py on the GPU (csrc/morph.hip through building_detection_amd/cleanup.py)
against its CPU restatement (oracle/cleanup.py), bit for bit."""
import numpy as np
import pytest
import torch
from scipy import ndimage as ndi

from oracle import cleanup as CL

pytestmark = pytest.mark.gpu


def scene(rng, h, w, n_rect=14, speckle=0.002):
    """building-like masks: rotated / overlapping rectangles, thin corridors, holes, speckle, objects on the image border"""
    m = np.zeros((h, w), bool)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n_rect):
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        a, b = rng.uniform(8, 70), rng.uniform(8, 70)
        t = rng.uniform(0, np.pi)
        u = (xx - cx) * np.cos(t) + (yy - cy) * np.sin(t)
        v = -(xx - cx) * np.sin(t) + (yy - cy) * np.cos(t)
        m |= (np.abs(u) < a) & (np.abs(v) < b)
    for _ in range(6):   # corridors between things
        y, x = rng.integers(0, h - 5), rng.integers(0, w - 5)
        if rng.random() < 0.5:
            m[y:y + rng.integers(2, 9), x:min(w, x + rng.integers(20, 120))] = True
        else:
            m[y:min(h, y + rng.integers(20, 120)), x:x + rng.integers(2, 9)] = True
    for _ in range(8):   # holes, some with islands
        y, x = rng.integers(0, h - 30), rng.integers(0, w - 30)
        m[y:y + rng.integers(4, 28), x:x + rng.integers(4, 28)] = False
        if rng.random() < 0.4:
            m[y + 5:y + 9, x + 5:x + 9] = True
    m ^= rng.random((h, w)) < speckle
    return (m * 255).astype(np.uint8)


@pytest.mark.parametrize("seed,h,w", [(0, 256, 320), (1, 300, 257), (2, 512, 512), (3, 97, 1030), (4, 640, 200)])
def test_clean_matches_oracle(engine, seed, h, w):
    from building_detection_amd import cleanup as GC
    rng = np.random.default_rng(seed)
    g = scene(rng, h, w)
    kept, labels, table, _ = GC.fill_and_delete(g, engine)
    ref_label, ref_objs = CL.fill_and_delete(g)
    assert np.array_equal(kept.cpu().numpy(), ref_label), "fill_and_delete differs from the restatement"
    # contourArea per object: the table's area2 / 2 against border following + shoelace on the CPU
    lab = labels.cpu().numpy()
    checked = 0
    for k in np.nonzero(table[:, 6])[0][:12]:
        assert CL.contour_area(lab == k) == table[k, 1] / 2.0
        ys, xs = np.nonzero(lab == k)
        assert (xs.min(), ys.min(), xs.max(), ys.max()) == tuple(table[k, 2:6])
        checked += 1
    got = GC.clean(g, engine)
    ref = CL.clean(g)
    assert got.dtype == np.uint8 and set(np.unique(got)) <= {0, 255}
    diff = int((got != ref).sum())
    print(f"seed {seed} {h}x{w}: {len(table)} objects, {int(table[:, 6].sum())} kept, {checked} areas checked, "
          f"{int((ref > 0).sum())} px in the result, {diff} differ")
    assert diff == 0


def test_many_separate_buildings(engine):
    """A town: ~300 separate rectangles / L-shapes / dumbbells of 20-90 px on a 1400x1500 canvas - hundreds of object windows,
    many of them split, next to each other."""
    from building_detection_amd import cleanup as GC
    rng = np.random.default_rng(42)
    g = np.zeros((1400, 1500), np.uint8)
    for gy in range(0, 1400, 100):
        for gx in range(0, 1500, 100):
            y, x = gy + rng.integers(2, 12), gx + rng.integers(2, 12)
            hh, ww = rng.integers(20, 86), rng.integers(20, 86)
            g[y:y + hh, x:x + ww] = 255
            kind = rng.integers(0, 4)
            if kind == 0:    # L-shape
                g[y:y + hh // 2, x + ww // 2:x + ww] = 0
            elif kind == 1:  # dumbbell: a neck cut into the middle
                n0 = rng.integers(3, 12)
                g[y + hh // 2 - 4:y + hh // 2 + 4, x:x + ww // 2 - n0 // 2] = 0
                g[y + hh // 2 - 4:y + hh // 2 + 4, x + ww // 2 + n0 // 2:x + ww] = 0
            elif kind == 2:  # courtyard
                g[y + hh // 3:y + 2 * hh // 3, x + ww // 3:x + 2 * ww // 3] = 0
    got, ref = GC.clean(g, engine), CL.clean(g)
    n_in = ndi.label(g > 0, structure=np.ones((3, 3)))[1]
    n_out = ndi.label(ref > 0, structure=np.ones((3, 3)))[1]
    print(f"town: {n_in} objects in, {n_out} out, {int((got != ref).sum())} pixels differ")
    assert np.array_equal(got, ref) and n_out > 50


def test_split_drop_and_keep_rules_on_the_gpu(engine):
    """The hand-made cases of tests/test_cleanup_cpu.py::test_split_rules, placed in ONE image: corridor cut (list), narrow
    object (empty list -> vanishes), all pieces small (False -> dropped), compact (kept), an object in the image corner."""
    from building_detection_amd import cleanup as GC
    g = np.zeros((400, 520), np.uint8)
    g[10:50, 10:50] = 255; g[50:80, 27:33] = 255; g[80:120, 10:50] = 255          # two blocks + corridor
    g[10:90, 100:118] = 255                                                          # 18 px wide: vanishes
    g[150:174, 10:40] = 255; g[174:186, 22:28] = 255; g[186:210, 10:40] = 255        # small pieces: dropped
    g[150:200, 100:150] = 255                                                        # compact: kept
    g[340:400, 460:520] = 255                                                        # bottom-right corner: kept (border never erodes)
    g[250:300, 200:300] = 255; g[262:288, 215:285] = 0                               # ring: hole filled, then kept
    got = GC.clean(g, engine)
    ref = CL.clean(g)
    assert np.array_equal(got, ref)
    assert got[55:75, 27:33].max() == 0 and got[10:50, 10:50].min() == 255 and got[10:90, 100:118].max() == 0
    assert got[150:210, 10:40].max() == 0 and got[150:200, 100:150].min() == 255 and got[340:400, 460:520].min() == 255
    assert got[262:288, 215:285].min() == 255


def test_model_confuse_matches_oracle(engine):
    from building_detection_amd import cleanup as GC
    rng = np.random.default_rng(11)
    base = scene(rng, 384, 448, n_rect=10)
    masks = []
    for i in range(5):
        m = base.copy()
        flip = ndi.binary_dilation(rng.random(m.shape) < 0.0008, iterations=int(rng.integers(2, 9)))
        m[flip] = 255 - m[flip]                       # every model disagrees somewhere
        masks.append(m)
    got = GC.model_confuse(masks, engine)
    ref = CL.model_confuse(masks)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} pixels differ"
    dev = [torch.from_numpy(m).cuda() for m in masks]
    got_dev = GC.model_confuse(dev, engine)           # device tensors in, device tensor out
    assert isinstance(got_dev, torch.Tensor) and np.array_equal(got_dev.cpu().numpy(), ref)
    with pytest.raises(ValueError):
        GC.model_confuse(masks[:4], engine)
    from building_detection_amd import pipeline as PL
    assert PL.model_confuse(masks[:4]) is None                  # 'no five images', as the reference prints
    assert np.array_equal(PL.model_confuse(masks), ref)
