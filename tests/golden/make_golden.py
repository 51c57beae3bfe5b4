#!/usr/bin/env python3
"""Generates the golden fixtures of tests/golden/ from the build's own CPU oracle (oracle/), fixed seeds.

The reference cannot run here (TensorFlow absent) and holds no golden vectors, so these fixtures do not pin the
oracle to the reference ("parity unpinned", oracle/__init__.py); they pin the oracle - and through it the HIP
engine - to ITSELF over time: a change in oracle/, in the engine or in the installed torch that moves any of
these numbers fails tests/test_oracle_cpu.py::test_golden_* (CPU) or tests/test_models_gpu.py::test_golden_* (GPU).
The format is the one SURVEY.md 8(c) asks for, so that a container with TensorFlow can regenerate the very same
files from the reference's builders (inputs, weights-by-seed, expected predict() output, loss and gradient norms).

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

Fixtures are data only (inputs and expected outputs); no reference source is stored."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import models as M  # noqa: E402
from oracle import tfops as T  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402  (host-side numpy only)

MODELS = [  # name, oracle builder, size, kwargs
    ("v3plus", "deeplab_v3plus", 64, {"aspp_pool": 4}),
    ("bam", "deeplab_v3plus_bam", 64, {"aspp_pool": 4}),
    ("scse", "scse_unet", 32, {}),
    ("res34", "res34_unet", 32, {}),
    ("hrnet", "hrnet", 32, {}),
]
SEED = 1103


def model_fixture(name, fn, size, kw):
    torch.manual_seed(0)
    x, y = synthetic_batch(2, size, size, seed=SEED)
    P = M.Params(seed=SEED)  # the oracle's own seeded Keras-default initialisers
    with torch.no_grad():
        probs = getattr(M, fn)(P, torch.from_numpy(x), training=False, **kw).numpy()
    ws = [t.detach().numpy().copy() for t in P.tensors]
    # one training step in float64 from the same weights: loss, metrics, gradient norms
    P64 = M.Params(weights=ws, dtype=torch.float64)
    p = getattr(M, fn)(P64, torch.from_numpy(x).double(), training=True, **kw)
    loss = M.loss_fn("edge_focal_loss", torch.from_numpy(y).double(), p)
    loss.backward()
    gn = np.array([float(t.grad.norm()) for t in P64.trainable_tensors()])
    cm = M.metrics_from_counts(*M.confusion(torch.from_numpy(y), p.detach().float()))
    return {
        "seed": np.int64(SEED), "size": np.int64(size), "x_sum": np.float64(x.astype(np.float64).sum()),
        "n_tensors": np.int64(len(ws)), "w_abs_sum": np.float64(sum(float(np.abs(w).sum()) for w in ws)),
        "probs": probs.astype(np.float32), "train_loss": np.float64(loss.item()),
        "train_probs_mean": np.float64(p.detach().mean().item()), "grad_norms": gn,
        "metrics": np.array([cm["PA"], cm["IoU"], cm["MIoU"], cm["F1_score"]], np.float64),
    }


def ops_fixture():
    g = np.random.default_rng(SEED)
    r = lambda *s: g.uniform(-1, 1, size=s).astype(np.float32)  # noqa: E731
    t = torch.from_numpy
    out = {}
    x = r(2, 8, 8, 4)
    w = r(3, 3, 4, 6)
    b = r(6)
    out.update(conv_x=x, conv_w=w, conv_b=b,
               conv_s2_even=T.conv2d(t(x), t(w), t(b), 2, 1, "same").numpy(),      # App. B-1: pad (0,1)
               conv_d2=T.conv2d(t(x), t(w), t(b), 1, 2, "same").numpy(),
               conv_1x1_s2=T.conv2d(t(x), t(w[1:2, 1:2]), None, 2, 1, "same").numpy())
    dw, pw = r(3, 3, 4, 1), r(1, 1, 4, 5)
    out.update(sep_dw=dw, sep_pw=pw, sep_y=T.separable_conv2d(t(x), t(dw), t(pw), t(b[:5]), 1).numpy(),
               sep_y_s2=T.separable_conv2d(t(x), t(dw), t(pw), t(b[:5]), 2).numpy())
    wt3, wt2 = r(3, 3, 5, 4), r(2, 2, 5, 4)  # Keras Conv2DTranspose kernel [kh, kw, Cout, Cin]
    out.update(convT_w3=wt3, convT_w2=wt2, convT_k3=T.conv2d_transpose(t(x), t(wt3), t(b[:5]), 2).numpy(),
               convT_k2=T.conv2d_transpose(t(x), t(wt2), None, 2).numpy())
    gam, bet = r(4) + 1.5, r(4)
    mm, mv = torch.zeros(4), torch.ones(4)
    y_bn, nm, nv = T.batch_norm(t(x), t(gam), t(bet), mm, mv, True)            # 4-D: moving var from the unbiased estimate
    x2 = r(6, 4)
    y_bn2, nm2, nv2 = T.batch_norm(t(x2), t(gam), t(bet), mm, mv, True)       # 2-D (after Dense): biased
    imean, ivar = r(4), np.abs(r(4)) + 0.5
    out.update(bn_gamma=gam, bn_beta=bet, bn_train_y=y_bn.numpy(), bn_new_mean=nm.numpy(), bn_new_var=nv.numpy(),
               bn2_x=x2, bn2_train_y=y_bn2.numpy(), bn2_new_var=nv2.numpy(), bn_imean=imean, bn_ivar=ivar,
               bn_infer_y=T.batch_norm(t(x), t(gam), t(bet), t(imean), t(ivar), False)[0].numpy())
    out.update(maxpool_3s2_same=T.max_pool(t(x), 3, 2, "same").numpy(), maxpool_2s4=T.max_pool(t(x), 2, 4).numpy(),
               maxpool_2s2=T.max_pool(t(x), 2).numpy(), avgpool_4=T.avg_pool(t(x), 4).numpy(),
               up_2=T.upsample_nearest(t(x), 2).numpy(), gap=T.global_avg_pool(t(x)).numpy())
    # losses / metrics / optimiser / schedule
    logits = r(2, 8, 8, 2) * 3
    yp = T.softmax(t(logits)).numpy()
    m = (g.uniform(size=(2, 8, 8)) > 0.5).astype(np.float32)
    yt = np.stack([1 - m, m, 1 + (g.uniform(size=m.shape) > 0.7), 1 + (g.uniform(size=m.shape) > 0.7)], -1).astype(np.float64)
    out.update(loss_y_true=yt, loss_y_pred=yp,
               loss_values=np.array([M.loss_fn(k, t(yt), t(yp)).item() for k in ("binary_crossentropy", "focal_loss", "edge_focal_loss")]))
    tp, tn, fp, fn = M.confusion(t(yt), t(yp))
    cm = M.metrics_from_counts(tp, tn, fp, fn)
    out.update(confusion=np.array([int(tp), int(tn), int(fp), int(fn)], np.int64),
               metrics=np.array([cm["PA"], cm["IoU"], cm["MIoU"], cm["F1_score"]]))
    p0, g0 = r(16), r(16)
    ps, ms, vs = [t(p0.copy())], [torch.zeros(16)], [torch.zeros(16)]
    for step in (1, 2, 3):
        M.adam_step(ps, [t(g0 * step)], ms, vs, step, 1e-3)
    out.update(adam_p0=p0, adam_g0=g0, adam_p3=ps[0].numpy(), adam_m3=ms[0].numpy(), adam_v3=vs[0].numpy())
    out["cosine_lr"] = np.array([M.cosine_decay_with_warmup(s, 1e-3, 1000, warmup_learning_rate=1e-5, warmup_steps=30)
                                 for s in (0, 1, 15, 29, 30, 31, 500, 999, 1000)], np.float64)
    return out


if __name__ == "__main__":
    torch.set_num_threads(8)
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **ops_fixture())
    for name, fn, size, kw in MODELS:
        np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **model_fixture(name, fn, size, kw))
        print("wrote", name)
