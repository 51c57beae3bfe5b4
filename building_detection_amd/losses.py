"""Losses and metrics of train_model/DeepLabv3plus.py:490-623 as the engine sees them.

`model.compile(loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score])` (DeepLabv3plus.py:834-837) passes
Python callables.  The engine does not trace tf ops: it recognises the three losses and four metrics of the
reference by name (these objects, or any callable with the same `__name__`, e.g. the reference's own
functions) and runs the fused HIP kernels `sg_loss_fwd/bwd` and `sg_confusion_counts` instead.
"""
from __future__ import annotations

import warnings

import numpy as np

from ._lib import SG_LOSS_CE2, SG_LOSS_FOCAL, SG_LOSS_EDGE_FOCAL

K_EPSILON = 1e-7

_LOSS_KINDS = {
    "binary_crossentropy": SG_LOSS_CE2,   # 2-class categorical CE on softmax outputs, eps inside the log (:490-499)
    "focal_loss": SG_LOSS_FOCAL,          # alpha = (.5,.5), gamma = 2 (:502-512)
    "edge_focal_loss": SG_LOSS_EDGE_FOCAL,  # alpha = (.35,.65), gamma = 2, edge weights y_true[...,2:4] (:515-527)
}
_METRICS = ("PA", "IoU", "MIoU", "F1_score")


class _Named:
    def __init__(self, name, doc):
        self.__name__ = name
        self.__doc__ = doc

    def __call__(self, y_true, y_pred):
        raise RuntimeError(
            f"{self.__name__} is evaluated by the engine's fused kernel inside Model.train_on_batch/test_on_batch; "
            "pass it to model.compile(...) rather than calling it")

    def __repr__(self):
        return f"<building_detection_amd {self.__name__}>"


binary_crossentropy = _Named("binary_crossentropy", "2-class CE on softmax probabilities (DeepLabv3plus.py:490-499)")
focal_loss = _Named("focal_loss", "focal loss, alpha=(.5,.5), gamma=2 (DeepLabv3plus.py:502-512)")
edge_focal_loss = _Named("edge_focal_loss", "edge-weighted focal loss (DeepLabv3plus.py:515-527)")
PA = _Named("PA", "pixel accuracy (DeepLabv3plus.py:530-553)")
IoU = _Named("IoU", "foreground IoU (DeepLabv3plus.py:556-575)")
MIoU = _Named("MIoU", "mean of foreground / background IoU (DeepLabv3plus.py:578-598)")
F1_score = _Named("F1_score", "F1 (DeepLabv3plus.py:601-623)")


class ForeignCallableWarning(UserWarning):
    """A loss / metric callable that is not the engine's own object was selected BY NAME."""


def _warn_foreign(fn, what, lines):
    """The reference's training scripts hand their own Python functions to compile() (DeepLabv3plus.py:834-837).  The engine
    never runs their bodies: it picks its fused kernel by `__name__`.  An EDITED edge_focal_loss would therefore train with
    the stock formula - say so, once per compile, instead of doing it silently."""
    if isinstance(fn, (str, _Named)):
        return
    warnings.warn(
        f"{what} {getattr(fn, '__name__', fn)!r} is a foreign callable ({getattr(fn, '__module__', '?')}): the engine selects its "
        f"fused kernel by NAME and evaluates the stock formula of train_model/DeepLabv3plus.py:{lines}; the body of the "
        "callable is never executed, so any edit to it has no effect", ForeignCallableWarning, stacklevel=4)


def resolve_loss(loss) -> int:
    name = loss if isinstance(loss, str) else getattr(loss, "__name__", None)
    if name not in _LOSS_KINDS:
        raise ValueError(f"loss {loss!r}: the engine implements {sorted(_LOSS_KINDS)} (train_model/DeepLabv3plus.py:490-527)")
    _warn_foreign(loss, "loss", "490-527")
    return _LOSS_KINDS[name]


def resolve_metric(metric) -> str:
    name = metric if isinstance(metric, str) else getattr(metric, "__name__", None)
    if name not in _METRICS:
        raise ValueError(f"metric {metric!r}: the engine implements {_METRICS} (train_model/DeepLabv3plus.py:530-623)")
    _warn_foreign(metric, "metric", "530-623")
    return name


def metrics_from_counts(tp: int, tn: int, fp: int, fn: int) -> dict:
    """The reference casts the int32 counts to float32 and divides with +epsilon (DeepLabv3plus.py:547-598,
    614-623); the same float32 arithmetic is done here on the host from the exact device counts."""
    f = np.float32
    tp, tn, fp, fn, e = f(tp), f(tn), f(fp), f(fn), f(K_EPSILON)
    pa = (tp + tn) / (tp + tn + fp + fn + e)
    iou = tp / (tp + fp + fn + e)
    miou = (tp / (tp + fp + fn + e) + tn / (tn + fp + fn + e)) / f(2)
    recall, precision = tp / (tp + fn + e), tp / (tp + fp + e)
    f1 = (f(2.0) * precision * recall) / (precision + recall + e)
    return {"PA": float(pa), "IoU": float(iou), "MIoU": float(miou), "F1_score": float(f1)}
