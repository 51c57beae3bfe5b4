"""`tensorflow`-shaped namespace for the reference's model code (drop-in face of SURVEY.md §8 b-1).

    from building_detection_amd import tfshim as tf            # instead of `import tensorflow as tf`
    from building_detection_amd.tfshim.keras.layers import *   # (via tfshim.install() for unmodified files)

It covers exactly the subset of tf / tf.keras that predict_model/*.py and the model sections of train_model/*.py
use, so the reference's builder text runs unmodified and produces an engine `Model`.  The attention blocks are
spelled in the reference as chains of generic ops (sigmoid -> tf.multiply -> tf.add; RepeatVector -> tf.reshape ->
tf.add -> sigmoid -> tf.multiply -> tf.add; concatenate(axis=-2) -> Softmax(axis=-2) -> Cropping2D -> multiply ->
add).  Instead of materialising those full-tensor intermediates, the shim returns *virtual tensors* that remember
the algebra and, when the closing `tf.add` / `layers.add` arrives, emits the fused node (scSE, BAM, SK fusion).
A virtual tensor that is consumed by anything else is materialised through the generic nodes where one
exists, otherwise a NotImplementedError names the unsupported spelling.
"""
from __future__ import annotations

import sys
import types
from typing import List, Sequence

import numpy as np

from . import layers as L
from .callbacks import Callback as _Callback, backend as _backend
from .graph import KTensor
from .runtime import Model as _Model

newaxis = None
float32, float64, int32, int8 = np.float32, np.float64, np.int32, np.int8


# ============================================================================================ virtual tensors
class _Virtual:
    """Deferred algebra over real KTensors; `.shape` mimics what TensorFlow would report."""
    shape: tuple
    _real = None

    def real(self) -> KTensor:
        """Materialise through generic nodes (once; later consumers share the result)."""
        if self._real is None:
            self._real = self._make()
        return self._real

    def _make(self) -> KTensor:
        raise NotImplementedError(f"{type(self).__name__}: this use of the tensor is not a spelling the reference uses")


class _SigmoidGate(_Virtual):  # sigmoid(logit); logit is [N,H,W,1], [N,1,1,C] or [N,C]
    def __init__(self, logit: KTensor):
        self.logit, self.shape = logit, logit.shape

    def _make(self):
        return L.Activation("sigmoid")(self.logit)


class _Gated(_Virtual):  # x * sigmoid(logit)
    def __init__(self, x: KTensor, gate: _SigmoidGate):
        self.x, self.gate, self.shape = x, gate, x.shape

    @property
    def spatial(self):
        return self.gate.logit.shape[-1] == 1 and len(self.gate.logit.shape) == 4 and self.gate.logit.shape[1:3] == self.x.shape[1:3]

    def _make(self):
        return L.multiply([self.x, self.gate.real()])


class _BcastNC(_Virtual):  # RepeatVector(H*W)(v[N,C]) (+ tf.reshape to [N,H,W,C])
    def __init__(self, src: KTensor, shape):
        self.src, self.shape = src, tuple(shape)


class _GateSum(_Virtual):  # mc[N,C] broadcast + ms[N,H,W,1]
    def __init__(self, mc: KTensor, ms: KTensor, shape):
        self.mc, self.ms, self.shape = mc, ms, tuple(shape)


class _SigGateSum(_Virtual):
    def __init__(self, gs: _GateSum):
        self.gs, self.shape = gs, gs.shape


class _BamProd(_Virtual):  # sigmoid(mc + ms) * x
    def __init__(self, x: KTensor, gs: _GateSum):
        self.x, self.gs, self.shape = x, gs, x.shape


class _BranchStack(_Virtual):  # concatenate([w_i [N,1,1,C]], axis=-2), optionally softmaxed over the branches
    def __init__(self, logits: List[KTensor], softmaxed=False):
        self.logits, self.softmaxed = logits, softmaxed
        n, h, _, c = logits[0].shape
        self.shape = (n, h, len(logits), c)


class _BranchWeight(_Virtual):
    def __init__(self, stack: _BranchStack, i: int):
        self.stack, self.i, self.shape = stack, i, stack.logits[0].shape


class _WeightedBranch(_Virtual):
    def __init__(self, x: KTensor, w: _BranchWeight):
        self.x, self.w, self.shape = x, w, x.shape


def _r(t):
    return t.real() if isinstance(t, _Virtual) else t


# ================================================================================================ tf.* functions
def add(a, b, name=None):
    # scSE: tf.add(sSE, cSE) on the same x  (predict_model/v3plus.py:163-167)
    if isinstance(a, _Gated) and isinstance(b, _Gated) and a.x is b.x and a.spatial != b.spatial:
        s, c = (a, b) if a.spatial else (b, a)
        return L._ScseCombineNode().build(a.x, s.gate.logit, c.gate.logit)
    # BAM: tf.add(c_out, s_out) then ... tf.add(tf.multiply(out, inputs), inputs)  (predict_model/bam.py:57-71)
    for u, v in ((a, b), (b, a)):
        if isinstance(u, _BcastNC) and isinstance(v, KTensor) and v.shape[-1] == 1 and len(u.shape) == 4:
            return _GateSum(u.src, v, u.shape)
        if isinstance(u, _BamProd) and v is u.x:
            return L._BamCombineNode().build(u.x, u.gs.mc, u.gs.ms)
    return L.add([_r(a), _r(b)], name)


def multiply(a, b, name=None):
    for g, x in ((a, b), (b, a)):
        if isinstance(g, _SigmoidGate) and isinstance(x, KTensor):
            return _Gated(x, g)
        if isinstance(g, _SigGateSum) and isinstance(x, KTensor):
            return _BamProd(x, g.gs)
        if isinstance(g, _BranchWeight) and isinstance(x, KTensor):
            return _WeightedBranch(x, g)
    return L.multiply([_r(a), _r(b)], name)


def concat(values, axis=-1, name=None):
    return _concat(list(values), axis, name)


def _concat(values, axis, name=None):
    rank = len(values[0].shape)
    if axis in (-2, rank - 2) and all(isinstance(v, KTensor) and v.shape[1:3] == (1, 1) for v in values):
        return _BranchStack(values)
    return L.concatenate([_r(v) for v in values], axis, name)


def reshape(t, shape, name=None):
    shape = [(-1 if (s is None) else int(s)) for s in shape]
    if isinstance(t, _BcastNC):  # [N,HW,C] -> [N,H,W,C]
        return _BcastNC(t.src, (None,) + tuple(shape[1:]))
    if isinstance(t, _SigmoidGate):
        return _SigmoidGate(L.Reshape(tuple(shape[1:]))(t.logit))
    return L.Reshape(tuple(shape[1:]))(_r(t))


def argmax(a, axis=-1, output_type=None):
    return np.argmax(np.asarray(a), axis=axis)  # lowest index on ties, like tf.argmax


def squeeze(a, axis=None):
    return np.squeeze(np.asarray(a), axis=axis)


def expand_dims(a, axis):
    return np.expand_dims(np.asarray(a), axis)


def cast(a, dtype):
    return np.asarray(a).astype(dtype)


# ============================================================================================== layer wrappers
def _wrap(cls):
    class W(cls):
        def __call__(self, x):
            if isinstance(x, (list, tuple)):
                return super().__call__([_r(t) for t in x])
            return super().__call__(_r(x))
    W.__name__ = W.__qualname__ = cls.__name__
    return W


class _Activation(L.Activation):
    def __call__(self, x):
        if self.act == "sigmoid":
            if isinstance(x, _GateSum):
                return _SigGateSum(x)
            if isinstance(x, KTensor) and (len(x.shape) == 2 or x.shape[-1] == 1 or x.shape[1:3] == (1, 1)):
                return _SigmoidGate(x)  # a gate: fused into its consumer when that is a multiply
        return super().__call__(_r(x))


class _Softmax(L.Softmax):
    def __call__(self, x):
        if isinstance(x, _BranchStack) and self.axis in (-2, 2):
            return _BranchStack(x.logits, softmaxed=True)
        return super().__call__(_r(x))


class _RepeatVector(L.Layer):
    def __init__(self, n, name=None, **kw):
        super().__init__(name)
        self.n = int(n)

    def __call__(self, x):
        x = _r(x)
        assert len(x.shape) == 2
        return _BcastNC(x, (None, self.n, x.shape[-1]))


class _Cropping2D(L.Layer):
    def __init__(self, cropping=((0, 0), (0, 0)), name=None, **kw):
        super().__init__(name)
        self.cropping = cropping

    def __call__(self, x):
        (t, b), (l, r) = self.cropping
        if isinstance(x, _BranchStack) and x.softmaxed and t == 0 and b == 0 and l + r == len(x.logits) - 1:
            return _BranchWeight(x, l)
        raise NotImplementedError("Cropping2D is only used to pick one softmaxed SK branch (v3plus.py:125)")


def _layers_add(xs, name=None):
    xs = list(xs)
    if xs and all(isinstance(t, _WeightedBranch) for t in xs):
        stack = xs[0].w.stack
        if all(t.w.stack is stack for t in xs) and sorted(t.w.i for t in xs) == list(range(len(stack.logits))):
            xs = sorted(xs, key=lambda t: t.w.i)
            return L.sk_fuse([t.x for t in xs], stack.logits)
    if len(xs) == 2:
        return add(xs[0], xs[1], name)
    return L.add([_r(t) for t in xs], name)


def _layers_multiply(xs, name=None):
    return multiply(xs[0], xs[1], name)


class _AddLayer(L.Layer):
    def __call__(self, xs):
        return _layers_add(xs, self._name)


class _MultiplyLayer(L.Layer):
    def __call__(self, xs):
        return _layers_multiply(xs, self._name)


class _ConcatLayer(L.Layer):
    def __init__(self, axis=-1, name=None, **kw):
        super().__init__(name)
        self.axis = axis

    def __call__(self, xs):
        return _concat(list(xs), self.axis, self._name)


class _ModelFactory:
    """tf.keras.Model(inputs=..., outputs=...) / Model(inputs, outputs)."""

    def __call__(self, inputs=None, outputs=None, name=None, **kw):
        return _Model(inputs, _r(outputs), name=name)


def to_categorical(y, num_classes=None, dtype="float32"):
    """keras.utils.to_categorical: integer truncation, one-hot (DeepLabv3plus.py:70)."""
    y = np.array(y, dtype="int")
    shape = y.shape
    if shape and shape[-1] == 1 and len(shape) > 1:
        shape = tuple(shape[:-1])
    y = y.ravel()
    num_classes = num_classes or int(y.max()) + 1
    out = np.zeros((y.shape[0], num_classes), dtype=dtype)
    out[np.arange(y.shape[0]), y] = 1
    return out.reshape(shape + (num_classes,))


# ================================================================================================ module tree
def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


_layer_names = dict(
    Input=L.Input, Conv2D=_wrap(L.Conv2D), SeparableConv2D=_wrap(L.SeparableConv2D),
    Conv2DTranspose=_wrap(L.Conv2DTranspose), Dense=_wrap(L.Dense), BatchNormalization=_wrap(L.BatchNormalization),
    Activation=_Activation, ReLU=_wrap(L.ReLU), Softmax=_Softmax, MaxPooling2D=_wrap(L.MaxPooling2D),
    MaxPool2D=_wrap(L.MaxPooling2D), AveragePooling2D=_wrap(L.AveragePooling2D),
    GlobalAveragePooling2D=_wrap(L.GlobalAveragePooling2D), GlobalAvgPool2D=_wrap(L.GlobalAveragePooling2D),
    UpSampling2D=_wrap(L.UpSampling2D), Reshape=_wrap(L.Reshape), RepeatVector=_RepeatVector, Cropping2D=_Cropping2D,
    Add=_AddLayer, add=_layers_add, Multiply=_MultiplyLayer, multiply=_layers_multiply,
    Concatenate=_ConcatLayer, concatenate=lambda xs, axis=-1, name=None: _concat(list(xs), axis, name))

layers = _module(__name__ + ".keras.layers", __all__=list(_layer_names), **_layer_names)
backend = _module(__name__ + ".keras.backend", epsilon=_backend.epsilon, get_value=_backend.get_value,
                  set_value=_backend.set_value)
callbacks = _module(__name__ + ".keras.callbacks", Callback=_Callback)
models = _module(__name__ + ".keras.models", Model=_ModelFactory())
utils = _module(__name__ + ".keras.utils", to_categorical=to_categorical)
keras = _module(__name__ + ".keras", layers=layers, backend=backend, callbacks=callbacks, models=models, utils=utils,
                Model=_ModelFactory())
config = _module(__name__ + ".config", experimental=_module(
    __name__ + ".config.experimental", list_physical_devices=lambda kind=None: [], set_memory_growth=lambda *a: None))


def install(as_name: str = "tensorflow"):
    """Register the shim in sys.modules under `as_name` so unmodified reference files
    (`import tensorflow as tf; from tensorflow.keras.layers import *; from tensorflow.keras import backend as K`)
    import it.  Returns the names registered (pass them to `uninstall`)."""
    me = sys.modules[__name__]
    table = {as_name: me, as_name + ".keras": keras, as_name + ".keras.layers": layers,
             as_name + ".keras.backend": backend, as_name + ".keras.callbacks": callbacks,
             as_name + ".keras.models": models, as_name + ".keras.utils": utils, as_name + ".config": config}
    for k, v in table.items():
        sys.modules[k] = v
    return list(table)


def uninstall(names: Sequence[str]):
    for k in names:
        sys.modules.pop(k, None)
