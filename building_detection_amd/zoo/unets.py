"""SCSE-UNet (predict_model/scse.py:49-97), Res34-UNet (predict_model/res34.py:27-170) and HRNet
(predict_model/hrnet.py:20-203): same graphs and weight order as the reference builders, table-driven."""
from __future__ import annotations

from .. import layers as L
from ..runtime import Model


# ------------------------------------------------------------------------------------------------ SCSE-UNet
def UNet(num_classes=2, input_shape=(512, 512, 3)):
    """5-level VGG-style U-Net: conv3x3+ReLU pairs (no BN), 2x2 max-pool, Conv2DTranspose 3x3 s2 + ReLU,
    scSE after every decoder stage, 1x1 softmax head.  Note the (num_classes, input_shape) argument order."""
    def pair(t, f):
        t = L.Conv2D(f, 3, padding="same", activation="relu")(t)
        return L.Conv2D(f, 3, padding="same", activation="relu")(t)

    inp = L.Input(shape=input_shape)
    x, skips = inp, []
    for f in (64, 128, 256, 512):
        x = pair(x, f)
        skips.append(x)
        x = L.MaxPooling2D(pool_size=2)(x)
    x = pair(x, 1024)
    for f, skip in zip((512, 256, 128, 64), reversed(skips)):
        up = L.Conv2DTranspose(f, 3, strides=2, padding="same", activation="relu")(x)
        x = L.scse_block(pair(L.concatenate([up, skip]), f))
    out = L.Conv2D(num_classes, 1, padding="same", activation="softmax")(x)
    return Model(inputs=inp, outputs=out, name="UNet_scse")


# --------------------------------------------------------------------------------------------- Res34-UNet
class ResNetFamily:
    """`ResNetFamily(input_shape).run_model('res34')` (predict.py:19-20).  Layer names follow the reference
    (conv1_1.., pool1..4, conv{2-5}_{i}_{1,2}[_BN/_AC], upsame_{1-4})."""

    def __init__(self, input_shape=(512, 512, 3)):
        self.inputs = L.Input(input_shape)
        self.f_size = 64

    def bn_conv_a(self, x, f, name):
        x = L.Conv2D(f, 3, padding="same", name=name, kernel_initializer="he_normal")(x)
        x = L.BatchNormalization(name=f"{name}_BN")(x)
        return L.Activation("relu", name=f"{name}_AC")(x)

    def res_block1(self, x, f, name):
        y = self.bn_conv_a(x, f, f"{name}_1")
        y = self.bn_conv_a(y, f, f"{name}_2")
        return L.Activation("relu", name=f"{name}_AC")(L.add([x, y], name=f"{name}_add"))

    def res34(self, x):
        f = self.f_size
        for i in (1, 2, 3):
            x = self.bn_conv_a(x, f, f"conv1_{i}")
        feats = [x]
        for stage, (mult, reps) in enumerate(((1, 3), (2, 4), (4, 6), (8, 3)), start=2):
            x = L.Conv2D(f * mult, 1, strides=2, padding="same", name=f"pool{stage - 1}")(x)  # "pool" = 1x1 s2 conv
            for i in range(reps):
                x = self.res_block1(x, f * mult, f"conv{stage}_{i}")
            feats.append(x)
        return feats

    @staticmethod
    def _fuse1x1(ts):
        y = L.concatenate(ts)
        return L.Conv2D(y.shape[-1], 1, activation="relu", kernel_initializer="he_normal")(y)

    def low_to_high_feature(self, low, mid, high):
        low1 = L.MaxPool2D()(low)
        low2 = L.MaxPool2D(strides=4)(low)          # 2x2 window, stride 4 (res34.py:153)
        mid1 = L.MaxPool2D()(mid)
        high_out = self._fuse1x1([high, mid1, low2])
        mid_out = self._fuse1x1([mid, low1])
        return mid_out, high_out

    @staticmethod
    def attention_demo(x):
        c = x.shape[-1]
        g = L.GlobalAveragePooling2D()(x)
        g = L.ReLU()(L.BatchNormalization()(L.Dense(c // 2)(g)))
        g = L.Activation("sigmoid")(L.BatchNormalization()(L.Dense(c)(g)))
        return L.multiply([x, L.Reshape((1, 1, c))(g)])

    def upsame_feature(self, low, high, name):
        c = low.shape[-1]
        up = L.Conv2DTranspose(c, 2, strides=(2, 2), activation="relu", padding="same")(high)
        y = L.concatenate([low, up])
        y = L.Conv2D(c, 1, activation="relu", kernel_initializer="he_normal")(y)
        return self.res_block1(y, c, f"upsame_{name}")

    def feature_fusion(self, net):
        c1, c2, c3, c4, c5 = net
        c2, c3 = self.low_to_high_feature(c1, c2, c3)
        c3, c4 = self.low_to_high_feature(c2, c3, c4)
        c1, c2, c3, c4, c5 = [self.attention_demo(t) for t in (c1, c2, c3, c4, c5)]
        up = self.upsame_feature(c4, c5, "4")
        up = self.upsame_feature(c3, up, "3")
        up = self.upsame_feature(c2, up, "2")
        up = self.upsame_feature(c1, up, "1")
        y = L.Conv2D(64, 3, padding="same", activation="relu", kernel_initializer="he_normal")(up)
        return L.Conv2D(2, 3, padding="same", activation="softmax", kernel_initializer="he_normal")(y)

    def run_model(self, name):
        if name != "res34":
            raise ValueError("This network does not exist.")
        out = self.feature_fusion(self.res34(self.inputs))
        return Model(self.inputs, out, name="res34_unet")


# -------------------------------------------------------------------------------------------------- HRNet
def _cbr(x, filters, kernel_size=3, strides=1, activate=True):
    x = L.Conv2D(filters, kernel_size, strides, padding="same")(x)
    x = L.BatchNormalization()(x)
    return L.Activation("relu")(x) if activate else x


def _bottleneck(x, f, stride=1, project=False):
    y = _cbr(x, f // 4, 1, stride)
    y = _cbr(y, f // 4, 3)
    y = _cbr(y, f, 1, activate=False)
    sc = _cbr(x, f, 1, stride, activate=False) if project else x
    return L.Activation("relu")(L.add([y, sc]))


def _basic(x, f):
    y = _cbr(x, f, 3)
    y = _cbr(y, f, 3, activate=False)
    return L.Activation("relu")(L.add([y, x]))


def _branch(x, f):
    for _ in range(4):
        x = _basic(x, f)
    return x


def _up(x, f, s):
    return L.UpSampling2D(size=s)(_cbr(x, f, 1, activate=False))


def _fuse2(b):
    """fuse_block_2 (predict_model/hrnet.py:114-139): each of the three branches receives the other two, brought to its
    resolution by 1x1 conv + nearest up-sampling (coarser) or stride-2 3x3 convs (finer); layer creation order as there."""
    x12, x13 = _up(b[1], 32, 2), _up(b[2], 32, 4)
    g0 = L.add([b[0], x12, x13])
    x21 = _cbr(b[0], 64, 3, 2, activate=False)
    x23 = _up(b[2], 64, 2)
    g1 = L.add([x21, b[1], x23])
    x31 = _cbr(_cbr(b[0], 32, 3, 2), 128, 3, 2, activate=False)
    x32 = _cbr(b[1], 128, 3, 2, activate=False)
    g2 = L.add([x31, x32, b[2]])
    return g0, g1, g2


def HRNet(shape=(512, 512, 3), num_classes=2):
    inp = L.Input(shape=shape)
    x = _cbr(inp, 64, strides=2)
    x = _bottleneck(x, 256, project=True)
    for _ in range(3):
        x = _bottleneck(x, 256)
    # stage 1: two resolutions (1/2, 1/4)
    t = [_cbr(x, 32), _cbr(x, 64, strides=2)]
    b = [_branch(t[0], 32), _branch(t[1], 64)]
    f0 = L.add([b[0], _up(b[1], 32, 2)])
    f1 = L.add([_cbr(b[0], 64, strides=2, activate=False), b[1]])
    # stage 2: three resolutions
    t = [_cbr(f0, 32), _cbr(f1, 64), _cbr(f1, 128, strides=2)]
    b = [_branch(t[0], 32), _branch(t[1], 64), _branch(t[2], 128)]
    g0, g1, g2 = _fuse2(b)
    # stage 3: four resolutions, fused by concatenation at 1/2
    t = [_cbr(g0, 32), _cbr(g1, 64), _cbr(g2, 128), _cbr(g2, 256, strides=2)]
    b = [_branch(t[0], 32), _branch(t[1], 64), _branch(t[2], 128), _branch(t[3], 256)]
    y = L.concatenate([b[0], _up(b[1], 32, 2), _up(b[2], 32, 4), _up(b[3], 32, 8)])
    y = _cbr(L.UpSampling2D(size=2)(y), 64)
    out = L.Conv2D(num_classes, 1, padding="same", activation="softmax")(y)
    return Model(inputs=inp, outputs=out, name="HRNet")
