"""DeepLabv3+ (Xception-65-like encoder + selective-kernel block + ASPP + scSE U-decoder) and its BAM variant.

Same graphs, layer order and `get_weights()` order as predict_model/v3plus.py:170-350 and
predict_model/bam.py:170-338 (= train_model/DeepLabv3plus.py:306-485, DeepLabv3plus_bam.py:310-475), written
table-driven and with the engine's fused attention nodes.  `aspp_pool` exposes the two hard-coded 32s of the
reference's ASPP (AveragePooling2D(32)/UpSampling2D(32), v3plus.py:302-304) so small test inputs can be used;
its default reproduces the reference literally.
"""
from __future__ import annotations

from .. import layers as L
from ..runtime import Model


def _cbr(x, filters, k, stride=1, dilate=1, activate=True):
    """conv_bn_relu of v3plus.py:288-293."""
    x = L.Conv2D(filters, k, stride, padding="same", dilation_rate=dilate)(x)
    x = L.BatchNormalization()(x)
    return L.Activation("relu")(x) if activate else x


def _sep_bn(x, filters, stride=1, pre_relu=False, post_relu=False):
    if pre_relu:
        x = L.Activation("relu")(x)
    x = L.SeparableConv2D(filters, 3, strides=stride, padding="same")(x)
    x = L.BatchNormalization()(x)
    return L.Activation("relu")(x) if post_relu else x


def _shortcut(x, filters, stride):
    return L.BatchNormalization()(L.Conv2D(filters, 1, strides=stride, padding="same")(x))


def _xception(inp, with_bam):
    """Entry / middle (16 x 3 separable convs @728) / exit flow; output stride 16.  Returns the decoder
    skips (c @1/2, c1 @1/4, c2 @1/8) and the 2048-channel feature map."""
    x = _cbr(inp, 32, 3, stride=2)
    x = _cbr(x, 64, 3)
    if with_bam:
        x = L.bam_block(x)
    c = x
    res = _shortcut(x, 128, 2)
    x = _sep_bn(x, 128, post_relu=True)
    x = _sep_bn(x, 128)
    x = L.MaxPooling2D(3, strides=2, padding="same")(x)
    x = L.add([x, res])
    c1 = x
    if with_bam:
        x = L.bam_block(x)
    c2 = None
    for filters in (256, 728):
        res = _shortcut(x, filters, 2)
        y = _sep_bn(x, filters, pre_relu=True)
        y = _sep_bn(y, filters, pre_relu=True)
        y = _sep_bn(y, filters, stride=2, pre_relu=True)
        x = L.add([y, res])
        if filters == 256:
            c2 = x
            if with_bam:
                x = L.bam_block(x)
    for _ in range(16):
        y = x
        for _ in range(3):
            y = _sep_bn(y, 728, pre_relu=True)
        x = L.add([y, x])
    if with_bam:
        x = L.bam_block(x)
    res = _shortcut(x, 1024, 1)
    y = _sep_bn(x, 728, pre_relu=True)
    y = _sep_bn(y, 1024, pre_relu=True)
    y = _sep_bn(y, 1024, pre_relu=True)
    x = L.add([y, res])
    for filters in (1536, 1536, 2048):
        x = _sep_bn(x, filters, post_relu=True)
    return c, c1, c2, x


def _sk_block(x, reduce=16):
    """SKNet_block (v3plus.py:74-138): 3x3 entry conv, branches {1x1, d6, d12, d18, GAP}, squeeze to C/16,
    five 1x1 excitation heads, softmax over the branches, weighted sum, BN, ReLU."""
    conv = _cbr(x, 256, 3)
    branches = [_cbr(conv, 256, 1)] + [_cbr(conv, 256, 3, dilate=d) for d in (6, 12, 18)]
    g = L.Reshape((1, 1, 256))(L.GlobalAvgPool2D()(conv))
    g = _cbr(g, 256, 1)
    branches.append(L.UpSampling2D(size=conv.shape[1])(g))
    t = L.add(branches)
    t = L.Reshape((1, 1, 256))(L.GlobalAvgPool2D()(t))
    t = _cbr(t, 256 // reduce, 1)
    logits = [L.Conv2D(256, 1, strides=1, padding="same")(t) for _ in range(5)]
    y = L.sk_fuse(branches, logits)
    return L.Activation("relu")(L.BatchNormalization()(y))


def _aspp(x, pool):
    """ASPP (v3plus.py:295-307): 1x1 + three dilated 3x3 (rates 6/12/18) + pooled branch, concatenated."""
    outs = [_cbr(x, 256, 1)] + [_cbr(x, 256, 3, dilate=d) for d in (6, 12, 18)]
    p = L.AveragePooling2D(pool_size=pool)(x)
    p = _cbr(p, 256, 1)
    outs.append(L.UpSampling2D(size=pool)(p))
    return L.concatenate(outs)


def _neck(c5, aspp_pool):
    sk = _sk_block(c5)
    a = _aspp(c5, aspp_pool)
    y = _cbr(a, 256, 1)
    y = L.concatenate([y, sk])
    y = _cbr(y, 256, 3)
    y = _cbr(y, 256, 3)
    return L.scse_block(y)


def _decode(y, filters):
    y = _cbr(y, filters, 3)
    y = _cbr(y, filters, 3)
    return L.scse_block(y)


def Xception_DeepLabV3_Plus(shape=(512, 512, 3), num_classes=2, aspp_pool=32):
    inp = L.Input(shape=shape)
    c, c1, c2, c5 = _xception(inp, with_bam=False)
    y = _neck(c5, aspp_pool)
    y = _decode(L.concatenate([L.UpSampling2D(size=2)(y), c2]), 256)
    y = _decode(L.concatenate([L.Conv2DTranspose(128, 3, strides=2, padding="same")(y), c1]), 128)
    y = _decode(L.concatenate([c, L.Conv2DTranspose(64, 3, strides=2, padding="same")(y)]), 64)
    y = L.UpSampling2D(size=2)(y)
    y = _cbr(y, 32, 3)
    y = _cbr(y, 32, 3)
    out = L.Conv2D(num_classes, 1, 1, activation="softmax")(y)
    return Model(inputs=inp, outputs=out, name="Xception_DeepLabV3_Plus")


def Xception_DeepLabV3_Plus_bam(shape=(512, 512, 3), num_classes=2, aspp_pool=32):
    inp = L.Input(shape=shape)
    _, c1, c2, c5 = _xception(inp, with_bam=True)
    y = _neck(c5, aspp_pool)
    y = _decode(L.concatenate([c2, L.UpSampling2D(size=2)(y)]), 128)
    y = _decode(L.concatenate([c1, L.UpSampling2D(size=2)(y)]), 64)
    y = L.UpSampling2D(size=4)(y)
    out = L.Conv2D(num_classes, 1, 1, activation="softmax")(y)
    return Model(inputs=inp, outputs=out, name="Xception_DeepLabV3_Plus_bam")
