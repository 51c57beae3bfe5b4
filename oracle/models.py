"""The five reference graphs restated on the CPU with oracle/tfops.py (oracle — test infrastructure only).

Each builder follows the layer-call sequence of the reference file it cites, creating parameters in the same
order (so a flat weight list in creation order is interchangeable with the engine's `get_weights()`), and
returns softmax probabilities [N,H,W,2].  Backward comes from torch autograd.  PARITY UNPINNED (see
oracle/__init__.py): TensorFlow is unavailable, so these restate documented tf.keras semantics.

Parameters live in a `Params` store: the first call of a builder creates them (Keras default initialisers,
seeded) and records their specs; later calls replay them, so the same store can be run in training mode
(batch statistics, moving-stat updates) and inference mode.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

from . import tfops as T


class Params:
    """Creation-ordered parameter store shared by all oracle builders."""

    def __init__(self, weights: Optional[Sequence] = None, seed: int = 1103, dtype=torch.float32):
        self.dtype = dtype
        self.gen = torch.Generator().manual_seed(seed)
        self.tensors: List[torch.Tensor] = []      # every weight incl. BN moving stats, creation order
        self.trainable: List[bool] = []
        self.kinds: List[str] = []
        self._given = None if weights is None else [torch.as_tensor(w).to(dtype).clone() for w in weights]
        self._cursor = 0
        self._replay = False

    # -- bookkeeping ------------------------------------------------------------------------------------
    def begin(self):
        """Call before every forward: rewinds the cursor; after the first pass parameters are replayed."""
        self._replay = len(self.tensors) > 0
        self._cursor = 0

    def _next(self, shape, kind, trainable, make):
        if self._replay:
            t = self.tensors[self._cursor]
            assert tuple(t.shape) == tuple(shape), (kind, tuple(t.shape), tuple(shape))
            self._cursor += 1
            return t
        if self._given is not None:
            t = self._given[len(self.tensors)]
            assert tuple(t.shape) == tuple(shape), (kind, len(self.tensors), tuple(t.shape), tuple(shape))
        else:
            t = make().to(self.dtype)
        if trainable:
            t.requires_grad_(True)
        self.tensors.append(t)
        self.trainable.append(trainable)
        self.kinds.append(kind)
        self._cursor += 1
        return t

    def _set(self, index_from_cursor_back: int, value: torch.Tensor):
        self.tensors[self._cursor - index_from_cursor_back] = value.detach().to(self.dtype)

    # -- initialisers ----------------------------------------------------------------
    def _glorot(self, shape):
        return lambda: T.glorot_uniform_(shape, self.gen, torch.float64)

    def _he_normal(self, shape):
        def make():
            rf = 1
            for s in shape[:-2]:
                rf *= s
            fan_in = shape[-2] * rf
            std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
            t = torch.empty(tuple(shape), dtype=torch.float64)
            torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=self.gen)
            return t
        return make

    def kernel(self, shape, init="glorot_uniform", kind="kernel"):
        mk = self._glorot(shape) if init == "glorot_uniform" else self._he_normal(shape)
        return self._next(shape, kind, True, mk)

    def bias(self, n):
        return self._next((n,), "bias", True, lambda: torch.zeros(n, dtype=torch.float64))

    def bn(self, c):
        g = self._next((c,), "gamma", True, lambda: torch.ones(c, dtype=torch.float64))
        b = self._next((c,), "beta", True, lambda: torch.zeros(c, dtype=torch.float64))
        m = self._next((c,), "moving_mean", False, lambda: torch.zeros(c, dtype=torch.float64))
        v = self._next((c,), "moving_var", False, lambda: torch.ones(c, dtype=torch.float64))
        return g, b, m, v

    # -- views ------------------------------------------------------------------------------------------
    def trainable_tensors(self):
        return [t for t, tr in zip(self.tensors, self.trainable) if tr]

    def count(self, trainable=True):
        return sum(t.numel() for t, tr in zip(self.tensors, self.trainable) if tr == trainable)

    def numpy_weights(self):
        return [t.detach().cpu().numpy().copy() for t in self.tensors]


class Net:
    """Layer helpers bound to a Params store and a training flag."""

    def __init__(self, P: Params, training: bool):
        self.P, self.training = P, training
        P.begin()

    def conv(self, x, filters, k=1, stride=1, dilation=1, relu=False, init="glorot_uniform"):
        w = self.P.kernel((k, k, x.shape[-1], filters), init)
        b = self.P.bias(filters)
        y = T.conv2d(x, w, b, stride, dilation, "same")
        return torch.relu(y) if relu else y

    def sepconv(self, x, filters, stride=1):
        c = x.shape[-1]
        dw = self.P.kernel((3, 3, c, 1), kind="depthwise_kernel")
        pw = self.P.kernel((1, 1, c, filters), kind="pointwise_kernel")
        b = self.P.bias(filters)
        return T.separable_conv2d(x, dw, pw, b, stride)

    def convT(self, x, filters, k, relu=False):
        w = self.P.kernel((k, k, filters, x.shape[-1]))
        b = self.P.bias(filters)
        y = T.conv2d_transpose(x, w, b, 2, "same")
        return torch.relu(y) if relu else y

    def dense(self, x, units):
        w = self.P.kernel((x.shape[-1], units))
        b = self.P.bias(units)
        return T.dense(x, w, b)

    def bn(self, x, relu=False):
        g, b, m, v = self.P.bn(x.shape[-1])
        y, nm, nv = T.batch_norm(x, g, b, m, v, self.training)
        if self.training:
            self.P._set(2, nm)
            self.P._set(1, nv)
        return torch.relu(y) if relu else y

    def conv_bn_relu(self, x, filters, k, stride=1, dilation=1, activate=True, init="glorot_uniform"):
        return self.bn(self.conv(x, filters, k, stride, dilation, init=init), relu=activate)

    # -- attention blocks -------------------------------------------------------------------------------
    def scse(self, x):
        """sSE_block + cSE + add: predict_model/v3plus.py:141-167 (= scse.py:20-46).  cSE has NO
        activation between its two 1x1 convs and always divides by 16."""
        c = x.shape[-1]
        s = torch.sigmoid(self.conv(x, 1, 1)) * x
        g = T.global_avg_pool(x).view(-1, 1, 1, c)
        g = self.conv(g, c // 16, 1)
        g = torch.sigmoid(self.conv(g, c, 1))
        return s + g * x

    def bam(self, x):
        """BAM_attention: predict_model/bam.py:20-71 (channel_gate, spatial_gate, combine)."""
        c = x.shape[-1]
        r = c // 16
        a = T.global_avg_pool(x)
        a = self.bn(self.dense(a, r), relu=True)
        a = self.bn(self.dense(a, r), relu=True)
        mc = self.dense(a, c)
        s = self.bn(self.conv(x, r, 1), relu=True)
        s = self.bn(self.conv(s, r, 3, dilation=4), relu=True)
        s = self.bn(self.conv(s, r, 3, dilation=4), relu=True)
        ms = self.conv(s, 1, 1)
        gate = torch.sigmoid(mc.view(-1, 1, 1, c) + ms)
        return gate * x + x

    def sk_block(self, x, reduce=16):
        """SKNet_block: predict_model/v3plus.py:74-138."""
        conv = self.conv_bn_relu(x, 256, 3)
        d1 = self.conv_bn_relu(conv, 256, 1)
        d6 = self.conv_bn_relu(conv, 256, 3, dilation=6)
        d12 = self.conv_bn_relu(conv, 256, 3, dilation=12)
        d18 = self.conv_bn_relu(conv, 256, 3, dilation=18)
        gap = T.global_avg_pool(conv).view(-1, 1, 1, 256)
        gap = self.conv_bn_relu(gap, 256, 1)
        gap = T.upsample_nearest(gap, conv.shape[1])
        tot = d1 + d6 + d12 + d18 + gap
        tot = T.global_avg_pool(tot).view(-1, 1, 1, 256)
        tot = self.conv_bn_relu(tot, 256 // reduce, 1)
        ws = [self.conv(tot, 256, 1) for _ in range(5)]
        wsm = torch.softmax(torch.cat(ws, dim=-2), dim=-2)        # [N,1,5,256], softmax over the 5 branches
        out = 0
        for i, br in enumerate((d1, d6, d12, d18, gap)):
            out = out + br * wsm[:, :, i:i + 1, :]
        return self.bn(out, relu=True)

    def aspp(self, x, pool=32):
        """inner ASPP: predict_model/v3plus.py:295-307 (AveragePooling2D(32) + UpSampling2D(32) literal)."""
        c1 = self.conv_bn_relu(x, 256, 1)
        p1 = self.conv_bn_relu(x, 256, 3, dilation=6)
        p2 = self.conv_bn_relu(x, 256, 3, dilation=12)
        p3 = self.conv_bn_relu(x, 256, 3, dilation=18)
        ap = T.avg_pool(x, pool)
        ap = self.conv_bn_relu(ap, 256, 1)
        ap = T.upsample_nearest(ap, pool)
        return torch.cat([c1, p1, p2, p3, ap], -1)

    # -- blocks of the U-Nets, callable on their own (tests/test_block_chains_gpu.py) -------------------------
    def res34_low_to_high(self, low, mid, high):
        """low_to_high_feature: predict_model/res34.py:151-159 (train_model/res34.py:292-300)."""
        low1, low2, mid1 = T.max_pool(low, 2), T.max_pool(low, 2, 4), T.max_pool(mid, 2)
        hi = torch.cat([high, mid1, low2], -1)
        hi = self.conv(hi, hi.shape[-1], 1, relu=True, init="he_normal")
        md = torch.cat([mid, low1], -1)
        md = self.conv(md, md.shape[-1], 1, relu=True, init="he_normal")
        return md, hi

    def res34_attention(self, t):
        """attention_demo: predict_model/res34.py:90-105 (train_model/res34.py:231-246)."""
        c = t.shape[-1]
        g = T.global_avg_pool(t)
        g = self.bn(self.dense(g, c // 2), relu=True)
        g = torch.sigmoid(self.bn(self.dense(g, c)))
        return t * g.view(-1, 1, 1, c)

    def hr_fuse2(self, b0, b1, b2):
        """fuse_block_2: predict_model/hrnet.py:114-139 (train_model/hrnet.py:245-298) - every branch to every resolution."""
        cbr, up = self.conv_bn_relu, T.upsample_nearest
        x12 = up(cbr(b1, 32, 1, activate=False), 2)
        x13 = up(cbr(b2, 32, 1, activate=False), 4)
        g0 = b0 + x12 + x13
        x21 = cbr(b0, 64, 3, 2, activate=False)
        x23 = up(cbr(b2, 64, 1, activate=False), 2)
        g1 = x21 + b1 + x23
        x31 = cbr(cbr(b0, 32, 3, 2), 128, 3, 2, activate=False)
        x32 = cbr(b1, 128, 3, 2, activate=False)
        g2 = x31 + x32 + b2
        return g0, g1, g2

    def xception_backbone(self, x, with_bam: bool):
        """Entry/middle/exit flow shared by predict_model/v3plus.py:173-282 and bam.py:173-279."""
        x = self.conv_bn_relu(x, 32, 3, stride=2)
        x = self.conv_bn_relu(x, 64, 3)
        if with_bam:
            x = self.bam(x)
        c = x
        res = self.bn(self.conv(x, 128, 1, stride=2))
        x = self.bn(self.sepconv(x, 128), relu=True)
        x = self.bn(self.sepconv(x, 128))
        x = T.max_pool(x, 3, 2, "same") + res
        c1 = x
        if with_bam:
            x = self.bam(x)
        for filters in (256, 728):
            res = self.bn(self.conv(x, filters, 1, stride=2))
            y = self.bn(self.sepconv(torch.relu(x), filters))
            y = self.bn(self.sepconv(torch.relu(y), filters))
            y = self.bn(self.sepconv(torch.relu(y), filters, stride=2))
            x = y + res
            if filters == 256:
                c2 = x
                if with_bam:
                    x = self.bam(x)
        for _ in range(16):
            y = x
            for _ in range(3):
                y = self.bn(self.sepconv(torch.relu(y), 728))
            x = y + x
        if with_bam:
            x = self.bam(x)
        res = self.bn(self.conv(x, 1024, 1))
        y = self.bn(self.sepconv(torch.relu(x), 728))
        y = self.bn(self.sepconv(torch.relu(y), 1024))
        y = self.bn(self.sepconv(torch.relu(y), 1024))
        x = y + res
        x = self.bn(self.sepconv(x, 1536), relu=True)
        x = self.bn(self.sepconv(x, 1536), relu=True)
        x = self.bn(self.sepconv(x, 2048), relu=True)
        return c, c1, c2, x


def deeplab_v3plus(P: Params, x, training=False, num_classes=2, aspp_pool=32):
    """Xception_DeepLabV3_Plus: predict_model/v3plus.py:170-350 (= train_model/DeepLabv3plus.py:306-485)."""
    n = Net(P, training)
    c, c1, c2, c5 = n.xception_backbone(x, with_bam=False)
    sk = n.sk_block(c5)
    a = n.aspp(c5, aspp_pool)
    y = n.conv_bn_relu(a, 256, 1)
    y = torch.cat([y, sk], -1)
    y = n.conv_bn_relu(y, 256, 3)
    y = n.conv_bn_relu(y, 256, 3)
    y = n.scse(y)
    y = torch.cat([T.upsample_nearest(y, 2), c2], -1)
    y = n.conv_bn_relu(y, 256, 3)
    y = n.conv_bn_relu(y, 256, 3)
    y = n.scse(y)
    y = torch.cat([n.convT(y, 128, 3), c1], -1)
    y = n.conv_bn_relu(y, 128, 3)
    y = n.conv_bn_relu(y, 128, 3)
    y = n.scse(y)
    y = torch.cat([c, n.convT(y, 64, 3)], -1)
    y = n.conv_bn_relu(y, 64, 3)
    y = n.conv_bn_relu(y, 64, 3)
    y = n.scse(y)
    y = T.upsample_nearest(y, 2)
    y = n.conv_bn_relu(y, 32, 3)
    y = n.conv_bn_relu(y, 32, 3)
    return torch.softmax(n.conv(y, num_classes, 1), -1)


def deeplab_v3plus_bam(P: Params, x, training=False, num_classes=2, aspp_pool=32):
    """Xception_DeepLabV3_Plus_bam: predict_model/bam.py:170-338."""
    n = Net(P, training)
    _, c1, c2, c5 = n.xception_backbone(x, with_bam=True)
    sk = n.sk_block(c5)
    a = n.aspp(c5, aspp_pool)
    y = n.conv_bn_relu(a, 256, 1)
    y = torch.cat([y, sk], -1)
    y = n.conv_bn_relu(y, 256, 3)
    y = n.conv_bn_relu(y, 256, 3)
    y = n.scse(y)
    y = torch.cat([c2, T.upsample_nearest(y, 2)], -1)
    y = n.conv_bn_relu(y, 128, 3)
    y = n.conv_bn_relu(y, 128, 3)
    y = n.scse(y)
    y = torch.cat([c1, T.upsample_nearest(y, 2)], -1)
    y = n.conv_bn_relu(y, 64, 3)
    y = n.conv_bn_relu(y, 64, 3)
    y = n.scse(y)
    y = T.upsample_nearest(y, 4)
    return torch.softmax(n.conv(y, num_classes, 1), -1)


def scse_unet(P: Params, x, training=False, num_classes=2):
    """UNet: predict_model/scse.py:49-97 (conv+ReLU, no BN; convT 3x3 s2 + ReLU; scSE per decoder stage)."""
    n = Net(P, training)
    skips = []
    y = x
    for f in (64, 128, 256, 512):
        y = n.conv(y, f, 3, relu=True)
        y = n.conv(y, f, 3, relu=True)
        skips.append(y)
        y = T.max_pool(y, 2)
    y = n.conv(y, 1024, 3, relu=True)
    y = n.conv(y, 1024, 3, relu=True)
    for f, s in zip((512, 256, 128, 64), reversed(skips)):
        up = n.convT(y, f, 3, relu=True)
        y = torch.cat([up, s], -1)
        y = n.conv(y, f, 3, relu=True)
        y = n.conv(y, f, 3, relu=True)
        y = n.scse(y)
    return torch.softmax(n.conv(y, num_classes, 1), -1)


def res34_unet(P: Params, x, training=False, backbone_only=False):
    """ResNetFamily(...).run_model('res34'): predict_model/res34.py:27-170."""
    n = Net(P, training)
    HE = "he_normal"

    def bn_conv_a(t, f):                       # res34.py:32-38
        return n.conv_bn_relu(t, f, 3, init=HE)

    def res_block(t, f):                       # res34.py:40-45
        y = bn_conv_a(bn_conv_a(t, f), f)
        return torch.relu(t + y)

    f0 = 64
    c1 = bn_conv_a(bn_conv_a(bn_conv_a(x, f0), f0), f0)
    feats = [c1]
    t = c1
    for mult, reps in ((1, 3), (2, 4), (4, 6), (8, 3)):      # res34.py:54-68 ("pool" = 1x1 stride-2 conv)
        t = n.conv(t, f0 * mult, 1, stride=2)
        for _ in range(reps):
            t = res_block(t, f0 * mult)
        feats.append(t)
    if backbone_only:
        return feats
    conv1, conv2, conv3, conv4, conv5 = feats

    low_to_high = n.res34_low_to_high          # res34.py:151-159
    attention = n.res34_attention              # attention_demo, res34.py:90-105

    def upsame(low, high):                     # res34.py:143-149
        c = low.shape[-1]
        up = n.convT(high, c, 2, relu=True)
        y = torch.cat([low, up], -1)
        y = n.conv(y, c, 1, relu=True, init=HE)
        return res_block(y, c)

    conv2, conv3 = low_to_high(conv1, conv2, conv3)
    conv3, conv4 = low_to_high(conv2, conv3, conv4)
    conv1, conv2, conv3, conv4, conv5 = [attention(t) for t in (conv1, conv2, conv3, conv4, conv5)]
    up = upsame(conv4, conv5)
    up = upsame(conv3, up)
    up = upsame(conv2, up)
    up = upsame(conv1, up)
    y = n.conv(up, 64, 3, relu=True, init=HE)
    return torch.softmax(n.conv(y, 2, 3, init=HE), -1)


def hrnet(P: Params, x, training=False, num_classes=2):
    """HRNet: predict_model/hrnet.py:20-203."""
    n = Net(P, training)
    cbr = n.conv_bn_relu

    def conv_block(t, f, stride=1):            # hrnet.py:28-38
        y = cbr(t, f // 4, 1, stride)
        y = cbr(y, f // 4, 3)
        y = cbr(y, f, 1, activate=False)
        sh = cbr(t, f, 1, stride, activate=False)
        return torch.relu(y + sh)

    def identity_block(t, f):                  # hrnet.py:41-49
        y = cbr(t, f // 4, 1)
        y = cbr(y, f // 4, 3)
        y = cbr(y, f, 1, activate=False)
        return torch.relu(y + t)

    def basic_block(t, f):                     # hrnet.py:52-59
        y = cbr(t, f, 3)
        y = cbr(y, f, 3, activate=False)
        return torch.relu(y + t)

    def branch(t, f):                          # hrnet.py:91-96
        for _ in range(4):
            t = basic_block(t, f)
        return t

    up = T.upsample_nearest
    y = cbr(x, 64, 3, 2)
    y = conv_block(y, 256)
    for _ in range(3):
        y = identity_block(y, 256)
    t0, t1 = cbr(y, 32, 3), cbr(y, 64, 3, 2)                       # transition_layer1
    b0, b1 = branch(t0, 32), branch(t1, 64)
    # fuse_block_1 (hrnet.py:99-111)
    u = up(cbr(b1, 32, 1, activate=False), 2)
    f0 = b0 + u
    f1 = cbr(b0, 64, 3, 2, activate=False) + b1
    t0, t1, t2 = cbr(f0, 32, 3), cbr(f1, 64, 3), cbr(f1, 128, 3, 2)  # transition_layer2
    b0, b1, b2 = branch(t0, 32), branch(t1, 64), branch(t2, 128)
    # fuse_block_2 (hrnet.py:114-139)
    g0, g1, g2 = n.hr_fuse2(b0, b1, b2)
    t0, t1, t2, t3 = cbr(g0, 32, 3), cbr(g1, 64, 3), cbr(g2, 128, 3), cbr(g2, 256, 3, 2)  # transition_layer3
    b0, b1, b2, b3 = branch(t0, 32), branch(t1, 64), branch(t2, 128), branch(t3, 256)
    # fuse_block_3 (hrnet.py:142-162)
    x1 = up(cbr(b1, 32, 1, activate=False), 2)
    x2 = up(cbr(b2, 32, 1, activate=False), 4)
    x3 = up(cbr(b3, 32, 1, activate=False), 8)
    y = torch.cat([b0, x1, x2, x3], -1)
    y = cbr(up(y, 2), 64, 3)
    return torch.softmax(n.conv(y, num_classes, 1), -1)


# ----------------------------------------------------------------------------------------- loss / metrics
def loss_fn(kind: str, y_true, y_pred):
    """binary_crossentropy / focal_loss / edge_focal_loss: train_model/DeepLabv3plus.py:490-527."""
    eps = T.K_EPSILON
    y = y_true[..., :2].to(y_pred.dtype)
    if kind == "binary_crossentropy":
        l = y * torch.log(y_pred + eps)
    elif kind == "focal_loss":
        l = torch.tensor([0.5, 0.5], dtype=y_pred.dtype) * y * (1 - y_pred) * (1 - y_pred) * torch.log(y_pred + eps)
    elif kind == "edge_focal_loss":
        wgt = y_true[..., 2:].to(y_pred.dtype)
        l = torch.tensor([0.35, 0.65], dtype=y_pred.dtype) * wgt * y * (1 - y_pred) * (1 - y_pred) * torch.log(y_pred + eps)
    else:
        raise ValueError(kind)
    return -(l[..., 0] + l[..., 1]).mean()


def confusion(y_true, y_pred):
    """TP, TN, FP, FN as in PA/IoU/MIoU/F1_score (DeepLabv3plus.py:530-623); argmax ties -> class 0."""
    t = (y_true[..., 1] > y_true[..., 0]).long()
    p = (y_pred[..., 1] > y_pred[..., 0]).long()
    return (int((t * p).sum()), int(((1 - t) * (1 - p)).sum()), int(((1 - t) * p).sum()), int((t * (1 - p)).sum()))


def metrics_from_counts(tp, tn, fp, fn):
    """float32 arithmetic as tf.cast(..., tf.float32) does (DeepLabv3plus.py:547-598,619-623)."""
    f = lambda v: torch.tensor(float(v), dtype=torch.float32)
    tp, tn, fp, fn = f(tp), f(tn), f(fp), f(fn)
    e = torch.tensor(T.K_EPSILON, dtype=torch.float32)
    pa = (tp + tn) / (tp + tn + fp + fn + e)
    iou = tp / (tp + fp + fn + e)
    miou = (tp / (tp + fp + fn + e) + tn / (tn + fp + fn + e)) / 2
    rec, prec = tp / (tp + fn + e), tp / (tp + fp + e)
    f1 = (2.0 * prec * rec) / (prec + rec + e)
    return {"PA": pa.item(), "IoU": iou.item(), "MIoU": miou.item(), "F1_score": f1.item()}


def adam_step(params, grads, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-7):
    """Keras-2 Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps)."""
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    with torch.no_grad():
        for p, g, mi, vi in zip(params, grads, m, v):
            mi.mul_(b1).add_(g, alpha=1 - b1)
            vi.mul_(b2).addcmul_(g, g, value=1 - b2)
            p.sub_(lr_t * mi / (vi.sqrt() + eps))


def cosine_decay_with_warmup(global_step, learning_rate_base, total_steps, warmup_learning_rate=0.0,
                             warmup_steps=0, min_learn_rate=0):
    """train_model/DeepLabv3plus.py:683-702."""
    import numpy as np
    if global_step >= warmup_steps:
        lr = 0.5 * learning_rate_base * (1 + np.cos(np.pi * (global_step - warmup_steps) / float(total_steps - warmup_steps)))
        return max(lr, min_learn_rate)
    k = (learning_rate_base - warmup_learning_rate) / warmup_steps
    return max(k * global_step + warmup_learning_rate, min_learn_rate)


BUILDERS = {
    "v3plus": deeplab_v3plus,
    "bam": deeplab_v3plus_bam,
    "scse": scse_unet,
    "res34": res34_unet,
    "hrnet": hrnet,
}
