"""tf.keras layer semantics restated with torch CPU functional ops (oracle — test infrastructure only).

Layout convention is Keras': activations NHWC, conv kernels HWIO, depthwise kernels [kh,kw,C,1],
Conv2DTranspose kernels [kh,kw,Cout,Cin], Dense kernels [in,out].  Every function here is differentiable by
torch autograd, which supplies the oracle's backward pass.  The framework rules restated here are the ones
SURVEY.md App. B lists (they are tf.keras documented behaviour, not visible in the reference text); each
function names the reference call sites that depend on it.
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import torch
import torch.nn.functional as F

K_EPSILON = 1e-7  # tf.keras.backend.epsilon()


# ----------------------------------------------------------------------------------------------- padding
def same_pad(in_size: int, k: int, stride: int, dilation: int = 1) -> Tuple[int, int, int]:
    """TF `padding='same'`: returns (out, pad_before, pad_after)  (SURVEY App. B-1).

    out = ceil(in/stride); total = max((out-1)*stride + (k-1)*dilation + 1 - in, 0);
    before = total//2 (the smaller half goes first), after = total - before.
    """
    out = -(-in_size // stride)
    total = max((out - 1) * stride + (k - 1) * dilation + 1 - in_size, 0)
    before = total // 2
    return out, before, total - before


def _nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1)


# ------------------------------------------------------------------------------------------- convolutions
def conv2d(x, w, b=None, stride: int = 1, dilation: int = 1, padding: str = "same"):
    """`Conv2D(filters, k, strides, padding, dilation_rate)`; x NHWC, w HWIO.

    Call sites: predict_model/v3plus.py:173,177,185,289 (ASPP/SK dilated 3x3 at :83-91,:298-300);
    predict_model/scse.py:52-95; predict_model/res34.py:33,54; predict_model/hrnet.py:21.
    """
    kh, kw, cin, cout = w.shape
    xn = _nchw(x)
    if padding == "same":
        _, pt, pb = same_pad(x.shape[1], kh, stride, dilation)
        _, pl, pr = same_pad(x.shape[2], kw, stride, dilation)
        xn = F.pad(xn, (pl, pr, pt, pb))
    elif padding != "valid":
        raise ValueError(padding)
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=stride, dilation=dilation)
    return _nhwc(y)


def depthwise_conv2d(x, dw, stride: int = 1, padding: str = "same"):
    """Depthwise half of `SeparableConv2D` (depth_multiplier 1, no bias); dw [kh,kw,C,1]."""
    kh, kw, c, mult = dw.shape
    assert mult == 1
    xn = _nchw(x)
    if padding == "same":
        _, pt, pb = same_pad(x.shape[1], kh, stride)
        _, pl, pr = same_pad(x.shape[2], kw, stride)
        xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, dw.permute(2, 3, 0, 1), None, stride=stride, groups=c)
    return _nhwc(y)


def separable_conv2d(x, dw, pw, b=None, stride: int = 1):
    """`SeparableConv2D(filters, 3, strides, padding='same')`: depthwise (no bias, NO BN/activation in
    between) then pointwise 1x1 + one bias.  predict_model/v3plus.py:187-278."""
    return conv2d(depthwise_conv2d(x, dw, stride), pw, b, 1, 1, "same")


def conv2d_transpose(x, w, b=None, stride: int = 2, padding: str = "same"):
    """`Conv2DTranspose(filters, k, strides=2, padding='same')`; w [kh,kw,Cout,Cin].

    TF defines it as the input-gradient of the forward SAME conv that maps the (stride*in) grid back to
    `in`:  out[stride*i + a - pad_before] += x[i] * w[a], pad_before from `same_pad(stride*in, k, stride)`.
    predict_model/v3plus.py:328,335; predict_model/scse.py:71-89 (k=3); predict_model/res34.py:144 (k=2).
    """
    kh, kw, cout, cin = w.shape
    n, h, wd, _ = x.shape
    oh, ow = h * stride, wd * stride
    _, pt, _ = same_pad(oh, kh, stride)
    _, pl, _ = same_pad(ow, kw, stride)
    # torch conv_transpose2d weight is [Cin, Cout, kh, kw]; full output is (h-1)*s + k, crop [pt, pt+oh)
    full = F.conv_transpose2d(_nchw(x), w.permute(3, 2, 0, 1), None, stride=stride)
    fh, fw = full.shape[2], full.shape[3]
    need_h, need_w = pt + oh, pl + ow
    if need_h > fh or need_w > fw:
        full = F.pad(full, (0, max(need_w - fw, 0), 0, max(need_h - fh, 0)))
    y = full[:, :, pt:pt + oh, pl:pl + ow]
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return _nhwc(y)


def dense(x, w, b=None):
    """`Dense(units)`; w [in,out].  predict_model/bam.py (channel_gate), predict_model/res34.py:94,98."""
    y = x @ w
    return y if b is None else y + b


# ----------------------------------------------------------------------------------------- normalisation
BN_MOMENTUM = 0.99
BN_EPS = 1e-3


def batch_norm(x, gamma, beta, moving_mean, moving_var, training: bool):
    """`BatchNormalization()` with Keras defaults (axis -1, momentum .99, eps 1e-3)  (SURVEY App. B-4).

    Works on [N,H,W,C] and on [N,C].  Training mode normalises with the *biased* batch variance and
    returns the updated moving statistics; for 4-D inputs Keras uses the fused op whose moving-variance
    update takes the *unbiased* variance (n/(n-1)), the 2-D path keeps the biased one.
    Returns (y, new_moving_mean, new_moving_var)  (the moving stats are returned unchanged in inference).
    """
    red = tuple(range(x.dim() - 1))
    if training:
        mean = x.mean(dim=red)
        var = x.var(dim=red, unbiased=False)
        y = (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta
        n = x.numel() // x.shape[-1]
        upd_var = var * (n / max(n - 1, 1)) if x.dim() == 4 else var
        with torch.no_grad():
            nm = moving_mean * BN_MOMENTUM + mean.detach() * (1 - BN_MOMENTUM)
            nv = moving_var * BN_MOMENTUM + upd_var.detach() * (1 - BN_MOMENTUM)
        return y, nm, nv
    y = (x - moving_mean) * torch.rsqrt(moving_var + BN_EPS) * gamma + beta
    return y, moving_mean, moving_var


# ------------------------------------------------------------------------------------------------ pooling
def max_pool(x, pool: int = 2, stride: int | None = None, padding: str = "valid"):
    """`MaxPooling2D(pool, strides, padding)`; strides default to pool.

    3x3 s2 'same' (predict_model/v3plus.py:192) pads (0,1) with -inf; `MaxPool2D(strides=4)`
    (predict_model/res34.py:153) is a 2x2 window at stride 4, valid."""
    stride = pool if stride is None else stride
    xn = _nchw(x)
    if padding == "same":
        _, pt, pb = same_pad(x.shape[1], pool, stride)
        _, pl, pr = same_pad(x.shape[2], pool, stride)
        xn = F.pad(xn, (pl, pr, pt, pb), value=float("-inf"))
    return _nhwc(F.max_pool2d(xn, pool, stride))


def avg_pool(x, pool: int):
    """`AveragePooling2D(pool_size=pool)`: stride = pool, valid (predict_model/v3plus.py:302)."""
    return _nhwc(F.avg_pool2d(_nchw(x), pool, pool))


def global_avg_pool(x):
    """`GlobalAveragePooling2D()`: [N,H,W,C] -> [N,C]."""
    return x.mean(dim=(1, 2))


def upsample_nearest(x, size: int):
    """`UpSampling2D(size)` default interpolation='nearest': out[i,j] = in[i//s, j//s]."""
    return x.repeat_interleave(size, dim=1).repeat_interleave(size, dim=2)


# ------------------------------------------------------------------------------------------- activations
def relu(x):
    return torch.relu(x)


def sigmoid(x):
    return torch.sigmoid(x)


def softmax(x, axis: int = -1):
    return torch.softmax(x, dim=axis)


# ------------------------------------------------------------------------------------------ initialisers
def glorot_uniform_(shape: Sequence[int], gen: torch.Generator, dtype=torch.float32) -> torch.Tensor:
    """Keras default kernel initialiser; fan_in/fan_out as Keras computes them (receptive field * chans)."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = 1
        for s in shape[:-2]:
            rf *= s
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(tuple(shape), generator=gen, dtype=torch.float64) * 2 - 1).mul_(limit).to(dtype)
