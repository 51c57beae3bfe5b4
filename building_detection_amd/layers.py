"""Keras-style layers of the engine: each call adds a Node whose forward/backward launch libsegengine kernels.

The classes mirror the tf.keras.layers calls the reference's builders make (predict_model/*.py): same
constructor arguments, same defaults (padding, strides, initialisers, BatchNormalization momentum/epsilon) and
the same per-layer weight order as `get_weights()`.  On top of those, three fused combine nodes (`ScseCombine`,
`BamCombine`, `SKFuse`) implement the attention blocks in one pass each; `zoo/` uses them, `tfshim` maps the
reference's un-fused spelling onto the generic nodes.
"""
from __future__ import annotations

import os

from typing import List, Optional, Sequence

import numpy as np

from .graph import KTensor, Node
from .ops import conv_out_geometry, same_pad
from . import _lib


def _pair(v):
    if isinstance(v, (tuple, list)):
        assert len(v) == 2 and v[0] == v[1], f"only square kernels/strides occur on this path: {v}"
        return int(v[0])
    return int(v)


def _act_name(a):
    if a is None or a == "linear":
        return None
    if callable(a):
        a = getattr(a, "__name__", str(a))
    if a not in ("relu", "sigmoid", "softmax"):
        raise ValueError(f"activation {a!r} is not used on this path")
    return a


class Layer:
    """Base of the callable layer objects (single use: the reference never shares layers)."""

    def __init__(self, name=None, **_ignored):
        self._name = name
        self._used = False

    def _once(self):
        if self._used:
            raise RuntimeError(f"{type(self).__name__}: layer objects are single-use in this engine")
        self._used = True


def Input(shape=None, batch_size=None, name=None, dtype=None, **_):
    t = KTensor((None,) + tuple(shape), None, name or "input")
    return t


# =============================================================================================== conv nodes
class _ConvNode(Node):
    op = "conv2d"

    def __init__(self, name, filters, k, stride, dilation, padding, activation, use_bias, kinit):
        super().__init__(name)
        self.filters, self.k, self.stride, self.dilation, self.padding = filters, k, stride, dilation, padding
        self.activation, self.use_bias, self.kinit = activation, use_bias, kinit

    def build(self, x: KTensor) -> KTensor:
        _, h, w, cin = x.shape
        ho, wo, _, _ = conv_out_geometry(h, w, self.k, self.k, self.stride, self.dilation, self.padding)
        self.w = self.add_param("kernel", (self.k, self.k, cin, self.filters), self.kinit)
        self.b = self.add_param("bias", (self.filters,), "zeros", kind="bias") if self.use_bias else None
        if self.activation == "softmax" and self.filters != 2:
            raise ValueError("softmax heads on this path have 2 classes")
        return self.connect([x], (None, ho, wo, self.filters))

    def desc(self, rt, x):
        return rt.eng.conv_desc(tuple(x.shape), self.filters, self.k, self.k, self.stride, self.dilation, self.padding)

    # UpSampling2D(2) -> this 3x3 convolution, fused (Model._fuse sets up_src; runtime._Runtime.up2_on decides per runtime):
    # the up-sampling node hands its SOURCE through, this node's kernels read / write it directly (csrc/conv_x6p.h:
    # sub-pixel forward, dgrad with the 2 x 2 sum in its epilogue, filter gradient gathering h >> 1, w >> 1)
    up_src = None
    # BatchNormalization(+ReLU) -> this convolution, the normalisation applied in this layer's loaders (Model._fuse sets bn_src /
    # _BNNode.defer_conv; the BatchNormalization node then hands its RAW input through - saved state "deferred")
    bn_src = None

    def _bn_in(self, rt, training):
        """(gamma, beta, mean, invstd | moving variance, relu, infer, eps) when this call's input is the raw input of bn_src"""
        src = self.bn_src
        if src is None:
            return None
        sv = rt._saved.get(id(src))
        if sv is None or not sv.get("deferred"):
            return None
        if training:
            return (rt.param(src.gamma), rt.param(src.beta), sv["mean"], sv["invstd"], src.relu, False, src.epsilon)
        return (rt.param(src.gamma), rt.param(src.beta), rt.param(src.mm), rt.param(src.mv), src.relu, True, src.epsilon)

    def _up2(self, rt):
        return self.up_src is not None and rt.up2_on(self.up_src)

    def _up_desc(self, rt, x_src):
        n, h, w, c = x_src.shape
        return rt.eng.conv_desc((n, 2 * h, 2 * w, c), self.filters, self.k, self.k, self.stride, self.dilation, self.padding)

    def plane_sites(self, rt, batch):
        """(tag, weight, forward-conv descriptor, dgrad?, phase) of every launch of this layer that reads its kernel through
        the matrix-pipe weight planes: runtime._Runtime.ensure_planes prepares them once per optimiser step."""
        d = rt.eng.conv_desc((batch,) + tuple(self.inputs[0].shape[1:]), self.filters, self.k, self.k, self.stride,
                             self.dilation, self.padding)
        head = self.activation == "softmax"
        sites = [("f", self.w, d, 0, "fwd", head)]
        if self._up2(rt):   # the sub-pixel forward makes its own (summed-tap) planes per launch
            sites = []
        if rt.needs_grad(self.inputs[0]):
            sites.append(("d", self.w, d, 1, "bwd", head))
        return sites

    @property
    def _tag(self):  # the north_star target kernels: ASPP / SK dilated 3x3 (rates 6, 12, 18)
        return "dilated_conv" if (self.k == 3 and self.dilation >= 6) else None

    def forward(self, rt, xs, training):
        (x,) = xs
        b = rt.param(self.b) if self.b else None
        up2 = self._up2(rt)   # x is then the up-sampling's source
        d = self._up_desc(rt, x) if up2 else self.desc(rt, x)
        with rt.eng.timed(self._tag):
            # training, fp32, a long-K multi-tap layer whose kernels read bf16 planes (csrc/conv_x6w.h, the planes-in filter
            # gradient): the activation is split ONCE per step - for every consumer of the tensor (the ASPP input feeds five
            # convolutions) and kept for this layer's filter gradient - instead of once per launch
            xp = rt.act_planes(self, x, d) if (training and not up2) else None
            bn_in = self._bn_in(rt, training)
            if training:
                rt.save(self, bn_in=bn_in)
            if training and getattr(self, "emit_bn_stats", False):
                # the following BatchNormalization takes its statistics from this conv's epilogue
                y, st = rt.eng.conv2d_fwd(x, rt.param(self.w), b, desc=d, want_stats=True,
                                          planes=None if up2 else rt.planes(self, "f"), up2=up2, x_planes=xp, bn_in=bn_in)
                if st is not None:
                    rt.bn_stats[id(y)] = st
                return y
            # the softmax head stays fp32 under bf16 storage (logits, probabilities, loss: SG_HEAD_F32)
            y = rt.eng.conv2d_fwd(x, rt.param(self.w), b, desc=d, relu=self.activation == "relu",
                                  head_f32=self.activation == "softmax", planes=None if up2 else rt.planes(self, "f"), up2=up2,
                                  x_planes=xp, bn_in=bn_in)
        if self.activation == "sigmoid":
            y = rt.eng.act_fwd(y, _lib.SG_ACT_SIGMOID, out=y)
        elif self.activation == "softmax":
            y = rt.eng.softmax2_fwd(y, out=y)
        return y

    def backward(self, rt, xs, y, dy):
        (x,) = xs
        e = rt.eng
        if self.activation == "relu":
            dz = e.act_bwd(y, dy, _lib.SG_ACT_RELU)
        elif self.activation == "sigmoid":
            dz = e.act_bwd(y, dy, _lib.SG_ACT_SIGMOID)
        elif self.activation == "softmax":
            dz = e.softmax2_bwd(y, dy)
        else:
            dz = dy
        up2 = self._up2(rt)
        d = self._up_desc(rt, x) if up2 else self.desc(rt, x)
        with e.timed(self._tag):
            want_b = self.b is not None and not getattr(self, "bias_grad_zero", False)
            # planes of the forward's activation (kept by rt.act_planes) and of dz: the planes-in filter gradient takes both as
            # they are, the planes-in dgrad shares dz's (one split of dz instead of one per launch)
            xp = None if up2 else rt.act_planes(self, x, d, make=False)
            wg_planes = xp is not None and dz.dtype == x.dtype and e.conv2d_wgrad_planes_ok(d)
            dzp = None
            if dz.dtype == x.dtype and not up2 and (wg_planes or (rt.needs_grad(self.inputs[0]) and rt.planes_in_on() and e.conv2d_planes_in(d, True))):
                dzp = e.split_planes(dz)
            # the input gradient first: the chain goes on with it, the filter gradient follows on the side stream beside the
            # bandwidth-bound kernels of the next node (two MFMA kernels side by side only share the matrix pipe)
            dx = None
            if rt.needs_grad(self.inputs[0]):
                # a gradient already collected for this layer's input (it has other consumers: the ASPP branches, a block's
                # shortcut convolution) is added in the dgrad kernel's epilogue where the launch takes the slab kernels
                res = None
                # (not for the dilated ASPP / SK convolutions: they are the roofline kernel set, timed as pure convolutions)
                # or the thin 1x1 kernel (scSE's spatial squeeze, Cout = 1)
                thin = (self.k == 1 and self.stride == 1 and self.filters <= 4 and x.shape[-1] % 4 == 0 and x.shape[-1] >= 16
                        and "SG_CONV_NOTHIN" not in os.environ)
                if (self._tag is None and dz.dtype == x.dtype and not up2
                        and (thin or (rt.plane_kind(self, "d") == 1 and rt.planes(self, "d") is not None))):
                    root = self.inputs[0]
                    while isinstance(root.node, _ActNode) and root.node.fused_away and len(root.consumers) == 1:
                        root = root.node.inputs[0]
                    res = rt.take_pending(root)
                dx = e.conv2d_dgrad(dz, rt.param(self.w), d, out_dtype=x.dtype, planes=rt.planes(self, "d"), res=res, down2=up2,
                                    dy_planes=dzp if res is None else None)
            gw, gb = rt.grad(self.w), (rt.grad(self.b) if want_b else None)
            bn_in = rt.saved(self).get("bn_in") if id(self) in rt._saved else None
            if bn_in is not None:   # x is the raw input of the BatchNormalization in front: the filter gradient normalises it as the forward did
                e.side_run(self._tag, (x, dz) + tuple(bn_in[:4]),
                           lambda: e.conv2d_wgrad(x, dz, d, want_bias=want_b, dw=gw, db=gb, bn_in=bn_in))
            elif wg_planes:
                def wgrad_from_planes():
                    e.conv2d_wgrad_planes(xp, dzp, d, dw=gw)
                    if want_b:
                        e.bias_grad(dz, gb)
                e.side_run(self._tag, (xp, dzp, dz), wgrad_from_planes)
            else:
                e.side_run(self._tag, (x, dz), lambda: e.conv2d_wgrad(x, dz, d, want_bias=want_b, dw=gw, db=gb, x_up2=up2))
        return [dx]

    def flops(self, batch):
        _, ho, wo, co = self.output.shape
        return 2 * batch * ho * wo * co * self.k * self.k * self.inputs[0].shape[-1]


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", dilation_rate=1, activation=None,
                 use_bias=True, kernel_initializer="glorot_uniform", name=None, **kw):
        super().__init__(name)
        self.args = (int(filters), _pair(kernel_size), _pair(strides), _pair(dilation_rate), padding,
                     _act_name(activation), use_bias, kernel_initializer)

    def __call__(self, x):
        self._once()
        return _ConvNode(self._name, *self.args[:2], *self.args[2:]).build(x)


class _SepConvNode(Node):
    op = "separable_conv2d"

    def __init__(self, name, filters, stride, activation):
        super().__init__(name)
        self.filters, self.stride, self.activation = filters, stride, activation
        self.pre_relu = False  # set by the fusion pass when the producer is a single-consumer ReLU
        self.bn_src = None     # set by the fusion pass: the training-mode BatchNormalization(+ReLU) applied in the gather
        self.bnsum_src = None  # set by the fusion pass: the BatchNormalization whose backward sums this layer's dgrad produces

    def build(self, x):
        _, h, w, c = x.shape
        ho, wo, _, _ = conv_out_geometry(h, w, 3, 3, self.stride, 1, "same")
        self.dw = self.add_param("depthwise_kernel", (3, 3, c, 1), "glorot_uniform", kind="depthwise_kernel")
        self.pw = self.add_param("pointwise_kernel", (1, 1, c, self.filters), "glorot_uniform", kind="pointwise_kernel")
        self.b = self.add_param("bias", (self.filters,), "zeros", kind="bias")
        return self.connect([x], (None, ho, wo, self.filters))

    def plane_sites(self, rt, batch):
        _, ho, wo, _ = self.output.shape
        d = rt.eng.conv_desc((batch, ho, wo, self.inputs[0].shape[-1]), self.filters, 1, 1)
        return [("f", self.pw, d, 0, "fwd", False), ("d", self.pw, d, 1, "bwd", False)]

    def forward(self, rt, xs, training):
        (x,) = xs
        e = rt.eng
        bn = None
        if training and self.bn_src is not None:
            sv = rt._saved.get(id(self.bn_src))
            if sv is not None and sv.get("deferred"):  # x is that layer's RAW input: normalise (+ReLU) while gathering
                src = self.bn_src
                bn = (rt.param(src.gamma), rt.param(src.beta), sv["mean"], sv["invstd"], src.relu)
        t = e.dwconv_fwd(x, rt.param(self.dw), self.stride, self.pre_relu, bn=bn)
        if training:
            rt.save(self, t=t, bn=bn)
            if getattr(self, "emit_bn_stats", False):
                y, st = e.conv2d_fwd(t, rt.param(self.pw), rt.param(self.b), want_stats=True, planes=rt.planes(self, "f"))
                if st is not None:
                    rt.bn_stats[id(y)] = st
                return y
        return e.conv2d_fwd(t, rt.param(self.pw), rt.param(self.b), relu=self.activation == "relu", planes=rt.planes(self, "f"))

    def backward(self, rt, xs, y, dy):
        (x,) = xs
        e = rt.eng
        t = rt.saved(self)["t"]
        dpw = e.conv_desc(tuple(t.shape), self.filters, 1, 1)
        want_b = not getattr(self, "bias_grad_zero", False)
        keep = ()
        if isinstance(dy, BnBackwardDeferred):
            # the BatchNormalization behind this layer left its backward apply to this dgrad (y = its raw input): one launch gives
            # the input gradient of the pointwise convolution AND the applied gradient dz the filter gradient reads
            q = dy
            dt, dz = e.conv2d_dgrad_bnb(q.dy, y, rt.param(self.pw), dpw, q.gamma, q.beta, q.mean, q.invstd, q.dgamma, q.dbeta, q.relu,
                                        planes=rt.planes(self, "d"))
        else:
            dz = e.act_bwd(y, dy, _lib.SG_ACT_RELU) if self.activation == "relu" else dy
            dt = e.conv2d_dgrad(dz, rt.param(self.pw), dpw, planes=rt.planes(self, "d"))
        gpw, gb = rt.grad(self.pw), (rt.grad(self.b) if want_b else None)
        e.side_run(None, (t, dz) + keep, lambda: e.conv2d_wgrad(t, dz, dpw, want_b, dw=gpw, db=gb))
        ddw = e.conv_desc(tuple(x.shape), x.shape[-1], 3, 3, self.stride, 1, "same")
        dx = None
        if rt.needs_grad(self.inputs[0]):
            # a gradient already collected for this layer's input (the residual add of the block it opens) is added by the kernel
            # (looked up through ReLU nodes that were absorbed into this layer's gather: they pass their gradient on unchanged)
            root = self.inputs[0]
            while isinstance(root.node, _ActNode) and root.node.fused_away and len(root.consumers) == 1:
                root = root.node.inputs[0]
            res = rt.take_pending(root) if e.dwconv_dgrad_acc_ok(ddw) else None
            # this layer's input is the output of a training-mode BatchNormalization with no other consumer: dx IS that layer's
            # output gradient, and the kernel sums what its backward needs (dgamma, dbeta) while it writes dx - the
            # BatchNormalization then only applies (Model._fuse: bnsum_src / sums_from; sg_dwconv2d_dgrad_bnsums)
            src = self.bnsum_src
            sv = rt._saved.get(id(src)) if src is not None else None
            # (through a residual add - src.defer_add - the gradient must be complete here: nothing left in the sweep's table)
            bn_x = rt.values.get(id(src.inputs[0])) if src is not None else None
            # (the kernel reads the BatchNormalization's raw input as a dense tensor of dt's storage type, pixel stride = C:
            # another dtype or a strided view takes the unfused pair dwconv_dgrad + bn_train_bwd instead of wrong sums)
            if (sv is not None and "mean" in sv and e.dwconv_dgrad_acc_ok(ddw) and dt.dtype == x.dtype
                    and bn_x is not None and bn_x.dtype == dt.dtype and bn_x.is_contiguous() and tuple(bn_x.shape) == tuple(x.shape)
                    and (rt._pending is None or id(root) not in rt._pending)):
                dx = e.dwconv_dgrad_bnsums(dt, rt.param(self.dw), ddw, bn_x, sv["mean"], sv["invstd"],
                                           rt.param(src.gamma), rt.param(src.beta), src.relu, rt.grad(src.gamma),
                                           rt.grad(src.beta), x=x, pre_relu=self.pre_relu, res=res)
                sv["sums_done"] = True
            else:
                dx = e.dwconv_dgrad(dt, rt.param(self.dw), ddw, x=x, pre_relu=self.pre_relu, res=res)
        bn = rt.saved(self).get("bn")
        # (the deferred BatchNormalization's mean / invstd belong to that node's saved state, which the sweep drops right after
        # its backward: the side stream's reader keeps them alive as it keeps x and dt)
        gdw = rt.grad(self.dw)
        e.side_run(None, (x, dt) + (tuple(bn[:4]) if bn else ()),
                   lambda: e.dwconv_wgrad(x, dt, ddw, self.pre_relu, dw=gdw, bn=bn), kind=3)
        return [dx]

    def flops(self, batch):
        _, ho, wo, co = self.output.shape
        c = self.inputs[0].shape[-1]
        return 2 * batch * ho * wo * (9 * c + c * co)


class SeparableConv2D(Layer):
    def __init__(self, filters, kernel_size=3, strides=1, padding="same", activation=None, name=None, **kw):
        super().__init__(name)
        assert _pair(kernel_size) == 3 and padding == "same", "SeparableConv2D on this path is 3x3 'same'"
        self.filters, self.stride, self.activation = int(filters), _pair(strides), _act_name(activation)

    def __call__(self, x):
        self._once()
        return _SepConvNode(self._name, self.filters, self.stride, self.activation).build(x)


class _ConvTNode(Node):
    """Conv2DTranspose(k, strides=2, 'same') = input-gradient of the SAME conv F mapping the 2x grid back."""
    op = "conv2d_transpose"

    def __init__(self, name, filters, k, activation, kinit):
        super().__init__(name)
        self.filters, self.k, self.activation, self.kinit = filters, k, activation, kinit

    def build(self, x):
        _, h, w, cin = x.shape
        # Keras computes the fans of a transposed kernel from its stored shape [kh,kw,out,in] (fan_in = out*rf)
        self.w = self.add_param("kernel", (self.k, self.k, self.filters, cin), self.kinit)
        self.b = self.add_param("bias", (self.filters,), "zeros", kind="bias")
        return self.connect([x], (None, h * 2, w * 2, self.filters))

    def fdesc(self, rt, x):
        n, h, w, cin = x.shape
        return rt.eng.conv_desc((n, 2 * h, 2 * w, self.filters), cin, self.k, self.k, 2, 1, "same")

    def plane_sites(self, rt, batch):
        _, h, w, cin = self.inputs[0].shape
        d = rt.eng.conv_desc((batch, 2 * h, 2 * w, self.filters), cin, self.k, self.k, 2, 1, "same")
        sites = [("d", self.w, d, 1, "fwd", False)]            # the layer's forward is the dgrad of F
        if rt.needs_grad(self.inputs[0]):
            sites.append(("f", self.w, d, 0, "bwd", False))    # its input gradient is F's forward
        return sites

    def forward(self, rt, xs, training):
        (x,) = xs
        return rt.eng.conv2d_dgrad(x, rt.param(self.w), self.fdesc(rt, x), bias=rt.param(self.b),
                                   relu=self.activation == "relu", planes=rt.planes(self, "d"))

    def backward(self, rt, xs, y, dy):
        (x,) = xs
        e = rt.eng
        dz = e.act_bwd(y, dy, _lib.SG_ACT_RELU) if self.activation == "relu" else dy
        d = self.fdesc(rt, x)
        # dw_F = wgrad_F(x_F = dz, dy_F = x); the bias gradient is the column sum of dz
        gw, gb = rt.grad(self.w), rt.grad(self.b)

        def filter_and_bias_gradient():
            e.conv2d_wgrad(dz, x, d, want_bias=False, dw=gw)
            e.bias_grad(dz, gb)
        e.side_run(None, (dz, x), filter_and_bias_gradient)
        dx = e.conv2d_fwd(dz, rt.param(self.w), None, desc=d, planes=rt.planes(self, "f")) if rt.needs_grad(self.inputs[0]) else None
        return [dx]

    def flops(self, batch):
        _, h, w, cin = self.inputs[0].shape
        return 2 * batch * h * w * cin * self.filters * self.k * self.k


class Conv2DTranspose(Layer):
    def __init__(self, filters, kernel_size, strides=1, padding="valid", activation=None,
                 kernel_initializer="glorot_uniform", name=None, **kw):
        super().__init__(name)
        assert _pair(strides) == 2 and padding == "same", "Conv2DTranspose on this path is stride 2 'same'"
        self.filters, self.k, self.activation, self.kinit = int(filters), _pair(kernel_size), _act_name(activation), kernel_initializer

    def __call__(self, x):
        self._once()
        return _ConvTNode(self._name, self.filters, self.k, self.activation, self.kinit).build(x)


class _DenseNode(Node):
    op = "dense"

    def __init__(self, name, units, activation):
        super().__init__(name)
        self.units, self.activation = units, activation

    def build(self, x):
        assert len(x.shape) == 2
        self.w = self.add_param("kernel", (x.shape[-1], self.units), "glorot_uniform")
        self.b = self.add_param("bias", (self.units,), "zeros", kind="bias")
        return self.connect([x], (None, self.units))

    def _desc(self, rt, x):
        return rt.eng.conv_desc((x.shape[0], 1, 1, x.shape[1]), self.units, 1, 1)

    def forward(self, rt, xs, training):
        (x,) = xs
        w = rt.param(self.w).view(1, 1, *self.w.shape)
        y = rt.eng.conv2d_fwd(x, w, rt.param(self.b), desc=self._desc(rt, x), relu=self.activation == "relu")
        y = y.view(x.shape[0], self.units)
        if self.activation == "sigmoid":
            rt.eng.act_fwd(y, _lib.SG_ACT_SIGMOID, out=y)
        return y

    def backward(self, rt, xs, y, dy):
        (x,) = xs
        e = rt.eng
        if self.activation == "relu":
            dy = e.act_bwd(y, dy, _lib.SG_ACT_RELU)
        elif self.activation == "sigmoid":
            dy = e.act_bwd(y, dy, _lib.SG_ACT_SIGMOID)
        d = self._desc(rt, x)
        e.conv2d_wgrad(x, dy, d, True, dw=rt.grad(self.w), db=rt.grad(self.b))
        dx = None
        if rt.needs_grad(self.inputs[0]):
            dx = e.conv2d_dgrad(dy, rt.param(self.w).view(1, 1, *self.w.shape), d).view(x.shape)
        return [dx]

    def flops(self, batch):
        return 2 * batch * self.inputs[0].shape[-1] * self.units


class Dense(Layer):
    def __init__(self, units, activation=None, name=None, **kw):
        super().__init__(name)
        self.units, self.activation = int(units), _act_name(activation)

    def __call__(self, x):
        self._once()
        return _DenseNode(self._name, self.units, self.activation).build(x)


# ======================================================================================== normalisation
class BnBackwardDeferred:
    """What a BatchNormalization hands its producer instead of dx when that producer - a pointwise convolution on the wide
    kernel - evaluates the backward apply in its own dgrad (csrc/conv_pw.h, BNB form; Engine.conv2d_dgrad_bnb): the gradient of
    the layer's output and everything the apply needs.  The column sums dgamma / dbeta are finished (sums_done)."""

    def __init__(self, dy, gamma, beta, mean, invstd, dgamma, dbeta, relu):
        self.dy, self.gamma, self.beta, self.mean, self.invstd = dy, gamma, beta, mean, invstd
        self.dgamma, self.dbeta, self.relu = dgamma, dbeta, relu
        self.dtype, self.shape = dy.dtype, dy.shape

    def tensors(self):
        return (self.dy, self.gamma, self.beta, self.mean, self.invstd, self.dgamma, self.dbeta)


class _BNNode(Node):
    op = "batch_normalization"

    def __init__(self, name, momentum, epsilon):
        super().__init__(name)
        self.momentum, self.epsilon = momentum, epsilon
        self.relu = False  # fused by the optimisation pass when followed by a single-consumer ReLU
        self.defer_to = None  # fused by the optimisation pass: the SeparableConv2D that applies this layer in its gather
        self.defer_add = None  # fused by the optimisation pass: the two-operand Add that applies this layer while it sums
        self.sums_from = None  # fused by the optimisation pass: the SeparableConv2D whose depthwise dgrad sums this layer's dgamma / dbeta
        self.defer_conv = None  # fused by the optimisation pass: the Conv2D whose loaders apply this layer (thin 1x1 / patch kernels)
        self.bnb_to = None  # fused by the optimisation pass: the SeparableConv2D / 1x1 Conv2D whose dgrad evaluates this layer's backward apply

    def build(self, x):
        c = x.shape[-1]
        self.gamma = self.add_param("gamma", (c,), "ones", kind="gamma")
        self.beta = self.add_param("beta", (c,), "zeros", kind="beta")
        self.mm = self.add_param("moving_mean", (c,), "zeros", trainable=False, kind="moving_mean")
        self.mv = self.add_param("moving_variance", (c,), "ones", trainable=False, kind="moving_var")
        return self.connect([x], x.shape)

    def forward(self, rt, xs, training):
        (x,) = xs
        e = rt.eng
        if training:
            st = rt.bn_stats.pop(id(x), None)
            if st is not None:  # statistics already produced by the conv that wrote x
                defer = self.defer_to is not None or self.defer_add is not None or (self.defer_conv is not None and rt.bn_conv_on(self))
                y, mean, invstd = e.bn_train_fwd_from_tiles(x, st[0], st[1], rt.param(self.gamma), rt.param(self.beta),
                                                            rt.param(self.mm), rt.param(self.mv), relu=self.relu,
                                                            momentum=self.momentum, eps=self.epsilon, apply=not defer)
                if defer:  # the consumer (depthwise gather / residual add) normalises: this layer's "output" is its raw input
                    rt.save(self, mean=mean, invstd=invstd, deferred=True)
                    return x
            else:
                y, mean, invstd = e.bn_train_fwd(x, rt.param(self.gamma), rt.param(self.beta), rt.param(self.mm),
                                                 rt.param(self.mv), relu=self.relu, momentum=self.momentum, eps=self.epsilon)
            rt.save(self, mean=mean, invstd=invstd, deferred=False)
            return y
        if self.defer_add is not None and x.shape[-1] % 4 == 0:   # inference: the Add applies the moving statistics
            rt.save(self, deferred=True)
            return x
        if self.defer_conv is not None and rt.bn_conv_on(self):   # inference: the consuming convolution's loader applies them
            rt.save(self, deferred=True)
            return x
        rt.save(self, deferred=False)
        return e.bn_infer(x, rt.param(self.gamma), rt.param(self.beta), rt.param(self.mm), rt.param(self.mv),
                          relu=self.relu, eps=self.epsilon)

    def backward(self, rt, xs, y, dy):
        (x,) = xs
        s = rt.saved(self)
        if s.get("sums_done") and self.bnb_to is not None and rt.bnb_on(self, x):
            # ... and that pass rides in the A path of the producer's pointwise dgrad: hand the pieces over
            return [BnBackwardDeferred(dy, rt.param(self.gamma), rt.param(self.beta), s["mean"], s["invstd"], rt.grad(self.gamma),
                                       rt.grad(self.beta), self.relu)]
        if s.get("sums_done"):   # dgamma / dbeta came out of the consumer's depthwise dgrad (sums_from): only the apply pass is left
            return [rt.eng.bn_train_bwd_apply(x, dy, rt.param(self.gamma), rt.param(self.beta), s["mean"], s["invstd"],
                                              rt.grad(self.gamma), rt.grad(self.beta), relu=self.relu)]
        # beta lets the kernels recompute the fused ReLU's mask from x instead of reading y (two tensor passes less)
        dx, _, _ = rt.eng.bn_train_bwd(x, y, dy, rt.param(self.gamma), s["mean"], s["invstd"], relu=self.relu,
                                       dgamma=rt.grad(self.gamma), dbeta=rt.grad(self.beta), beta=rt.param(self.beta))
        return [dx]


class BatchNormalization(Layer):
    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, name=None, **kw):
        super().__init__(name)
        assert axis in (-1, 3, 1), "BatchNormalization on this path normalises the last axis"
        self.momentum, self.epsilon = momentum, epsilon

    def __call__(self, x):
        self._once()
        return _BNNode(self._name, self.momentum, self.epsilon).build(x)


# ============================================================================================ activations
class _ActNode(Node):
    op = "activation"

    def __init__(self, name, act, axis=-1):
        super().__init__(name)
        self.act, self.axis = act, axis
        self.fused_away = False  # True => identity (absorbed into producer / consumer)

    def build(self, x):
        if self.act == "softmax":
            ax = self.axis if self.axis >= 0 else len(x.shape) + self.axis
            self.sm_axis = ax
            if ax == len(x.shape) - 1:
                assert x.shape[-1] == 2, "last-axis softmax on this path has 2 classes"
            else:
                assert len(x.shape) == 4 and ax == 2 and x.shape[1] == 1, "branch softmax expects [N,1,B,C]"
        return self.connect([x], x.shape)

    def forward(self, rt, xs, training):
        (x,) = xs
        if self.fused_away:
            return x
        e = rt.eng
        if self.act == "relu":
            return e.act_fwd(x, _lib.SG_ACT_RELU)
        if self.act == "sigmoid":
            return e.act_fwd(x, _lib.SG_ACT_SIGMOID)
        if self.sm_axis == len(x.shape) - 1:
            return e.softmax2_fwd(x)
        n, _, b, c = x.shape
        return e.softmax_branch_fwd(x.view(n, b, c)).view(x.shape)

    def backward(self, rt, xs, y, dy):
        if self.fused_away:
            return [rt.shared(dy)]
        e = rt.eng
        if self.act == "relu":
            return [e.act_bwd(y, dy, _lib.SG_ACT_RELU)]
        if self.act == "sigmoid":
            return [e.act_bwd(y, dy, _lib.SG_ACT_SIGMOID)]
        if self.sm_axis == len(y.shape) - 1:
            return [e.softmax2_bwd(y, dy)]
        n, _, b, c = y.shape
        return [e.softmax_branch_bwd(y.view(n, b, c), dy.view(n, b, c)).view(y.shape)]


class Activation(Layer):
    def __init__(self, activation, name=None, **kw):
        super().__init__(name)
        self.act = _act_name(activation)

    def __call__(self, x):
        self._once()
        return _ActNode(self._name, self.act).build(x)


class ReLU(Activation):
    def __init__(self, name=None, **kw):
        super().__init__("relu", name)


class Softmax(Layer):
    def __init__(self, axis=-1, name=None, **kw):
        super().__init__(name)
        self.axis = axis

    def __call__(self, x):
        self._once()
        return _ActNode(self._name, "softmax", self.axis).build(x)


# ================================================================================================ pooling
class _MaxPoolNode(Node):
    op = "max_pooling2d"

    def __init__(self, name, pool, stride, padding):
        super().__init__(name)
        self.pool, self.stride, self.padding = pool, stride, padding

    def build(self, x):
        _, h, w, c = x.shape
        if self.padding == "same":
            ho, wo = same_pad(h, self.pool, self.stride)[0], same_pad(w, self.pool, self.stride)[0]
        else:
            ho, wo = (h - self.pool) // self.stride + 1, (w - self.pool) // self.stride + 1
        return self.connect([x], (None, ho, wo, c))

    def forward(self, rt, xs, training):
        if training and self.pool <= 15:   # the winning cell of every window, one byte each: the backward reads it and dy only
            y, geom, idx = rt.eng.maxpool_fwd(xs[0], self.pool, self.stride, self.padding, want_idx=True)
            rt.save(self, geom=geom, idx=idx)
            return y
        y, geom = rt.eng.maxpool_fwd(xs[0], self.pool, self.stride, self.padding)
        rt.save(self, geom=geom)
        return y

    def backward(self, rt, xs, y, dy):
        sv = rt.saved(self)
        if sv.get("idx") is not None:
            return [rt.eng.maxpool_bwd_idx(dy, sv["idx"], tuple(xs[0].shape), sv["geom"])]
        return [rt.eng.maxpool_bwd(xs[0], y, dy, sv["geom"])]


class MaxPooling2D(Layer):
    def __init__(self, pool_size=2, strides=None, padding="valid", name=None, **kw):
        super().__init__(name)
        self.pool = _pair(pool_size)
        self.stride = self.pool if strides is None else _pair(strides)
        self.padding = padding

    def __call__(self, x):
        self._once()
        return _MaxPoolNode(self._name, self.pool, self.stride, self.padding).build(x)


MaxPool2D = MaxPooling2D


class _AvgPoolNode(Node):
    op = "average_pooling2d"

    def __init__(self, name, pool, global_pool=False):
        super().__init__(name)
        self.pool, self.global_pool = pool, global_pool

    def build(self, x):
        _, h, w, c = x.shape
        if self.global_pool:
            return self.connect([x], (None, c))
        return self.connect([x], (None, h // self.pool, w // self.pool, c))

    def _k(self, x):
        return (x.shape[1], x.shape[2]) if self.global_pool else (self.pool, self.pool)

    def forward(self, rt, xs, training):
        (x,) = xs
        kh, kw = self._k(x)
        y = rt.eng.avgpool_fwd(x, kh, kw)
        return y.view(x.shape[0], x.shape[3]) if self.global_pool else y

    def backward(self, rt, xs, y, dy):
        (x,) = xs
        kh, kw = self._k(x)
        # a gradient buffer the sweep already owns for this input (scSE's combine node, the other ASPP branches): the spread-out
        # dy / (kh kw) is added into it in place instead of being written out and summed by a separate add
        acc = rt.take_pending(self.inputs[0], owned_only=True)
        if acc is not None and acc.dtype == dy.dtype:
            return [rt.eng.avgpool_bwd(dy, tuple(x.shape), kh, kw, out=acc, accumulate=True)]
        if acc is not None:
            rt.put_back(self.inputs[0], acc)
        return [rt.eng.avgpool_bwd(dy, tuple(x.shape), kh, kw)]


class AveragePooling2D(Layer):
    def __init__(self, pool_size=2, strides=None, padding="valid", name=None, **kw):
        super().__init__(name)
        self.pool = _pair(pool_size)
        assert strides is None or _pair(strides) == self.pool

    def __call__(self, x):
        self._once()
        return _AvgPoolNode(self._name, self.pool).build(x)


class GlobalAveragePooling2D(Layer):
    def __call__(self, x):
        self._once()
        return _AvgPoolNode(self._name, 0, True).build(x)


GlobalAvgPool2D = GlobalAveragePooling2D


class _UpNode(Node):
    op = "up_sampling2d"

    def __init__(self, name, size):
        super().__init__(name)
        self.size = size

    def build(self, x):
        _, h, w, c = x.shape
        return self.connect([x], (None, h * self.size, w * self.size, c))

    fused_into = None   # the 3x3 convolution that reads this node's SOURCE directly (Model._fuse; _ConvNode.up_src)

    def forward(self, rt, xs, training):
        if self.fused_into is not None and rt.up2_on(self):
            return xs[0]   # never materialised: the consumer's kernels address the source (h >> 1, w >> 1)
        return rt.eng.upsample_fwd(xs[0], self.size)

    def backward(self, rt, xs, y, dy):
        if self.fused_into is not None and rt.up2_on(self):
            return [rt.shared(dy)]   # the consumer's dgrad already summed the 2 x 2 cells: dy has the source's shape
        return [rt.eng.upsample_bwd(dy, tuple(xs[0].shape), self.size)]


class UpSampling2D(Layer):
    def __init__(self, size=2, interpolation="nearest", name=None, **kw):
        super().__init__(name)
        assert interpolation == "nearest", "the reference only uses nearest up-sampling"
        self.size = _pair(size)

    def __call__(self, x):
        self._once()
        return _UpNode(self._name, self.size).build(x)


# ====================================================================================== structural nodes
class _ReshapeNode(Node):
    op = "reshape"

    def __init__(self, name, target):
        super().__init__(name)
        self.target = tuple(target)

    def build(self, x):
        n_in = int(np.prod(x.shape[1:]))
        tgt = list(self.target)
        if -1 in tgt:
            known = int(np.prod([t for t in tgt if t != -1]))
            tgt[tgt.index(-1)] = n_in // known
        assert int(np.prod(tgt)) == n_in, (x.shape, self.target)
        return self.connect([x], (None,) + tuple(tgt))

    def forward(self, rt, xs, training):
        return xs[0].view(xs[0].shape[0], *self.output.shape[1:])

    def backward(self, rt, xs, y, dy):
        return [rt.shared(dy.view(xs[0].shape))]  # aliases dy: never accumulate into it in place


class Reshape(Layer):
    def __init__(self, target_shape, name=None, **kw):
        super().__init__(name)
        self.target = tuple(target_shape)

    def __call__(self, x):
        self._once()
        return _ReshapeNode(self._name, self.target).build(x)


class _ConcatNode(Node):
    op = "concatenate"

    def build(self, xs):
        base = xs[0].shape[:-1]
        for t in xs:
            assert t.shape[:-1] == base, f"concat shapes differ: {[t.shape for t in xs]}"
        return self.connect(xs, base + (sum(t.shape[-1] for t in xs),))

    def forward(self, rt, xs, training):
        return rt.eng.concat(xs)

    def backward(self, rt, xs, y, dy):
        outs, off = [], 0
        for t, sym in zip(xs, self.inputs):
            c = t.shape[-1]
            if rt.needs_grad(sym):
                g = rt.eng.empty(*t.shape, dtype=dy.dtype)
                rt.eng.copy_channels(dy, off, g, 0, c)
                outs.append(g)
            else:
                outs.append(None)
            off += c
        return outs


def concatenate(xs, axis=-1, name=None):
    assert axis in (-1, len(xs[0].shape) - 1), "channel concat only"
    return _ConcatNode(name).build(list(xs))


class Concatenate(Layer):
    def __init__(self, axis=-1, name=None, **kw):
        super().__init__(name)
        self.axis = axis

    def __call__(self, xs):
        self._once()
        return concatenate(xs, self.axis, self._name)


class _AddNode(Node):
    op = "add"

    def __init__(self, name=None):
        super().__init__(name)
        self.relu = False
        self.bn_src = [None, None]   # fused by the optimisation pass: the BatchNormalization layers this Add applies to its operands

    def build(self, xs):
        for t in xs:
            assert t.shape == xs[0].shape, f"add shapes differ: {[t.shape for t in xs]}"
        return self.connect(xs, xs[0].shape)

    def forward(self, rt, xs, training):
        if len(xs) == 2 and (self.bn_src[0] is not None or self.bn_src[1] is not None):
            bn = [None, None]
            for i, src in enumerate(self.bn_src):
                if src is None:
                    continue
                sv = rt._saved.get(id(src)) or {}
                if sv.get("deferred"):   # that layer handed its RAW input on (it had the statistics / is in inference mode)
                    bn[i] = ((sv["mean"], sv["invstd"]) if training else (rt.param(src.mm), rt.param(src.mv))) + \
                            (rt.param(src.gamma), rt.param(src.beta))
            if bn[0] is not None or bn[1] is not None:
                eps = next(s.epsilon for s in self.bn_src if s is not None)
                ra, rb = (s is not None and s.relu for s in self.bn_src)
                return rt.eng.add2_bn(xs[0], xs[1], bn[0], bn[1], relu=self.relu, infer=not training, eps=eps,
                                      relu_a=ra and bn[0] is not None, relu_b=rb and bn[1] is not None)
        return rt.eng.add_n(xs, relu=self.relu)

    def backward(self, rt, xs, y, dy):
        g = rt.eng.act_bwd(y, dy, _lib.SG_ACT_RELU) if self.relu else dy
        return [rt.shared(g) for _ in xs]


def add(xs, name=None):
    return _AddNode(name).build(list(xs))


class Add(Layer):
    def __call__(self, xs):
        self._once()
        return add(xs, self._name)


class _BcastMulNode(Node):
    """x[N,H,W,C] * g, g = [N,C] / [N,1,1,C] (channel gate) or [N,H,W,1] (spatial gate)."""
    op = "multiply"

    def build(self, x, g):
        if len(g.shape) == 2 or g.shape[1:3] == (1, 1):
            assert g.shape[-1] == x.shape[-1]
            self.mode = 0
        else:
            assert g.shape[-1] == 1 and g.shape[1:3] == x.shape[1:3], (x.shape, g.shape)
            self.mode = 1
        return self.connect([x, g], x.shape)

    def forward(self, rt, xs, training):
        x, g = xs
        return rt.eng.bcast_mul_fwd(x, g, self.mode)

    def backward(self, rt, xs, y, dy):
        x, g = xs
        dx, dg = rt.eng.bcast_mul_bwd(x, g, dy, self.mode)
        return [dx, dg]


def multiply(xs, name=None):
    a, b = xs
    if len(a.shape) < len(b.shape) or (len(a.shape) == 4 and len(b.shape) == 4 and
                                       int(np.prod(a.shape[1:])) < int(np.prod(b.shape[1:]))):
        a, b = b, a
    if a.shape == b.shape:
        raise NotImplementedError("same-shape multiply is not used on the fused path")
    return _BcastMulNode(name).build(a, b)


class _ScseCombineNode(Node):
    """y = x * (sigmoid(s) + sigmoid(c)):  sSE_block + cSE + tf.add of predict_model/v3plus.py:141-167."""
    op = "scse_combine"

    def build(self, x, s, c):
        assert s.shape[-1] == 1 and c.shape[-1] == x.shape[-1]
        return self.connect([x, s, c], x.shape)

    def forward(self, rt, xs, training):
        x, s, c = xs
        return rt.eng.scse_fwd(x, s, c)

    def backward(self, rt, xs, y, dy):
        x, s, c = xs
        dx, ds, dc = rt.eng.scse_bwd(x, s, c, dy)
        return [dx, ds, dc]


class _BamCombineNode(Node):
    """y = x + x * sigmoid(mc + ms):  BAM_attention of predict_model/bam.py:57-71."""
    op = "bam_combine"

    def build(self, x, mc, ms):
        assert ms.shape[-1] == 1 and mc.shape[-1] == x.shape[-1]
        return self.connect([x, mc, ms], x.shape)

    def forward(self, rt, xs, training):
        x, mc, ms = xs
        return rt.eng.bam_fwd(x, mc, ms)

    def backward(self, rt, xs, y, dy):
        x, mc, ms = xs
        dx, dmc, dms = rt.eng.bam_bwd(x, mc, ms, dy)
        return [dx, dmc, dms]


class _SKFuseNode(Node):
    """Selective-kernel fusion: softmax over the B branch logits (Softmax(axis=-2) + Cropping2D) and the
    weighted sum of the branches (multiply + add), predict_model/v3plus.py:120-134."""
    op = "sk_fuse"

    def build(self, branches: Sequence[KTensor], logits: Sequence[KTensor]):
        assert len(branches) == len(logits)
        self.B = len(branches)
        return self.connect(list(branches) + list(logits), branches[0].shape)

    def forward(self, rt, xs, training):
        e = rt.eng
        B = self.B
        br, lg = xs[:B], xs[B:]
        n, c = br[0].shape[0], br[0].shape[-1]
        z = e.empty(B, n * c, dtype=br[0].dtype)
        for i, l in enumerate(lg):
            e.copy_channels(l.view(1, n * c), 0, z[i].view(1, n * c), 0, n * c)
        p = e.softmax_branch_fwd(z.view(1, B, n * c)).view(B, n, c)
        y = e.bcast_mul_fwd(br[0], p[0], 0)
        for i in range(1, B):
            e.bcast_mul_fwd(br[i], p[i], 0, out=y, accumulate=True)
        if training:
            rt.save(self, p=p)
        return y

    def backward(self, rt, xs, y, dy):
        e = rt.eng
        B = self.B
        br = xs[:B]
        p = rt.saved(self)["p"]
        n, c = br[0].shape[0], br[0].shape[-1]
        dp = e.empty(B, n, c, dtype=dy.dtype)
        dbr = []
        for i in range(B):
            dx, dg = e.bcast_mul_bwd(br[i], p[i], dy, 0)
            e.copy_channels(dg.view(1, n * c), 0, dp[i].view(1, n * c), 0, n * c)
            dbr.append(dx)
        dz = e.softmax_branch_bwd(p.view(1, B, n * c), dp.view(1, B, n * c)).view(B, n, c)
        dl = []
        for i in range(B):
            g = e.empty(*xs[B + i].shape, dtype=dz.dtype)
            e.copy_channels(dz[i].view(1, n * c), 0, g.view(1, n * c), 0, n * c)
            dl.append(g)
        return dbr + dl


# ================================================================================== composite attention
def scse_block(x: KTensor) -> KTensor:
    """scSE_block (sSE + cSE, predict_model/v3plus.py:141-167; scse.py:20-46) with the fused combine.
    Parameter order = reference creation order: sSE 1x1 conv, then the two cSE 1x1 convs."""
    c = x.shape[-1]
    s = Conv2D(1, 1, strides=1, padding="same")(x)
    g = GlobalAveragePooling2D()(x)
    g = Reshape((1, 1, c))(g)
    g = Conv2D(c // 16, 1, strides=1, padding="same")(g)
    g = Conv2D(c, 1, strides=1, padding="same")(g)
    return _ScseCombineNode().build(x, s, g)


def bam_block(x: KTensor, rate=16, d=4) -> KTensor:
    """BAM_attention (channel_gate, spatial_gate, combine; predict_model/bam.py:20-71), fused combine."""
    c = x.shape[-1]
    r = c // rate
    a = GlobalAveragePooling2D()(x)
    a = Activation("relu")(BatchNormalization()(Dense(r)(a)))
    a = Activation("relu")(BatchNormalization()(Dense(r)(a)))
    mc = Dense(c)(a)
    s = Activation("relu")(BatchNormalization()(Conv2D(r, 1)(x)))
    s = Activation("relu")(BatchNormalization()(Conv2D(r, 3, dilation_rate=d, padding="same")(s)))
    s = Activation("relu")(BatchNormalization()(Conv2D(r, 3, dilation_rate=d, padding="same")(s)))
    ms = Conv2D(1, 1)(s)
    return _BamCombineNode().build(x, mc, ms)


def sk_fuse(branches, logits) -> KTensor:
    return _SKFuseNode().build(branches, logits)
