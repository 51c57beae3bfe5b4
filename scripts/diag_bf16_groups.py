#!/usr/bin/env python3
"""Per-layer-group agreement of the DeepLabv3+ gradient between arithmetic modes, at the BASELINE workload.

    SIZE=512 BATCH=16 [STEPS=0] python scripts/diag_bf16_groups.py

Modes compared with the fp32 engine (x6: exact fp32 products on the bf16 pipe) on the SAME weights and tiles:
    bf16     mixed_bfloat16 storage (BASELINE configs[2])
    native   fp32 storage, native fp32 MFMA (sg_set_conv_x6(0)): a second CORRECT fp32 evaluation = the fp32 noise floor
STEPS > 0: first train the fp32 model for that many Adam steps (on a fixed set of batches) and compare at those weights.
Groups: entry flow / middle flow / exit flow / SK block / ASPP / neck (1x1 + 3x3 x2 + scSE) / decoder (+head).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd import zoo, mixed_precision as MP  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402
from building_detection_amd.losses import edge_focal_loss  # noqa: E402
from building_detection_amd.ops import get_engine  # noqa: E402


def groups_of(model):
    """trainable-parameter index -> group name, from the creation order of zoo.deeplab.Xception_DeepLabV3_Plus."""
    order = [n for n in model.nodes if n.params]
    convs = [i for i, n in enumerate(order) if n.op == "conv2d"]             # k-th Conv2D / SeparableConv2D created, whatever
    seps = [i for i, n in enumerate(order) if n.op == "separable_conv2d"]    # uid offset the layer names carry
    assert len(seps) == 62 and len(convs) >= 47, (len(seps), len(convs))
    # separable 8 opens the middle flow; Conv2D 5 is the exit flow's shortcut, 6-17 the SK block, 18-22 ASPP, 23-28 the neck
    # (1x1, two 3x3, scSE), 29.. the decoder
    cuts = [("entry", 0), ("middle", seps[8]), ("exit", convs[5]), ("sk", convs[6]), ("aspp", convs[18]), ("neck", convs[23]),
            ("decoder", convs[29])]
    out = []
    for i, n in enumerate(order):
        g = [c[0] for c in cuts if c[1] <= i][-1]
        out += [g] * sum(1 for p in n.params if p.trainable)
    return out


def compare(tag, ga, gb, grp):
    """ga vs reference gb: per-group cosine and relative L2."""
    res = {}
    for g in dict.fromkeys(grp):
        a = np.concatenate([x.reshape(-1) for x, q in zip(ga, grp) if q == g]).astype(np.float64)
        b = np.concatenate([x.reshape(-1) for x, q in zip(gb, grp) if q == g]).astype(np.float64)
        res[g] = (float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300)), float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)))
    print(f"{tag}: " + "  ".join(f"{g} cos {c:.4f} rel {r:.2e}" for g, (c, r) in res.items()), flush=True)
    return res


def main():
    size, bs, steps = int(os.environ.get("SIZE", "512")), int(os.environ.get("BATCH", "16")), int(os.environ.get("STEPS", "0"))
    eng = get_engine(0)

    def build(policy):
        MP.set_global_policy(policy)
        try:
            m = zoo.Xception_DeepLabV3_Plus((size, size, 3), 2, aspp_pool=size // 16)
        finally:
            MP.set_global_policy("float32")
        m.compile(optimizer="adam", loss=edge_focal_loss, metrics=[])
        return m

    m32 = build("float32")
    grp = groups_of(m32)
    for s in range(steps):
        xb, yb = synthetic_batch(bs, size, size, seed=500 + s % 4)
        l = m32.train_on_batch(xb, yb)["loss"]
        if s % 10 == 0 or s == steps - 1:
            print(f"warm-up step {s}: loss {l:.5f}", flush=True)
    ws = m32.get_weights()
    x, y = synthetic_batch(bs, size, size, seed=11)
    l32 = m32.train_on_batch(x, y)["loss"]
    g32 = m32.get_gradients()
    del m32
    m16 = build("mixed_bfloat16")
    m16.set_weights(ws)
    l16 = m16.train_on_batch(x, y)["loss"]
    g16 = m16.get_gradients()
    del m16
    prev = eng.lib.sg_set_conv_x6(0)
    mn = build("float32")
    mn.set_weights(ws)
    ln = mn.train_on_batch(x, y)["loss"]
    gn = mn.get_gradients()
    eng.lib.sg_set_conv_x6(prev)
    del mn
    print(f"size {size} bs {bs} after {steps} fp32 steps: loss fp32(x6) {l32:.6f}  bf16 {l16:.6f}  fp32(native MFMA) {ln:.6f}")
    compare("bf16   vs fp32", g16, g32, grp)
    compare("native vs fp32", gn, g32, grp)


if __name__ == "__main__":
    main()
