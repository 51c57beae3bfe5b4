"""Model: the tf.keras.Model surface the reference's callers use, executed on MI355X.

    model = Model(inputs, outputs)           predict_model/v3plus.py:347
    model.predict(x)                         predict.py:109, train_model/DeepLabv3plus.py:815
    model.compile(optimizer='adam', loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score])   :834-837
    model.fit_generator(generator, steps_per_epoch, epochs, callbacks, validation_data, validation_steps)  :844-849
    model.load_weights / save_weights        predict.py:21-49, DeepLabv3plus.py:780
    model.optimizer.lr (get/set), model.stop_training, model.summary()

Execution model: the graph is static, so a training step is one forward sweep over the node list (every node
output kept), the fused loss kernel, one reverse sweep that hands each node its output gradient and lets it
launch its dgrad / wgrad kernels (weight gradients land directly in one flat fp32 arena), then ONE fused Adam
launch over the whole arena.  Under data parallelism the gradient arena is all-reduced in buckets over RCCL
(dist.py) between the reverse sweep and Adam.  All arithmetic is in libsegengine; torch supplies memory.
"""
from __future__ import annotations

import math
import os
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .graph import KTensor, Node, ParamSpec, collect_nodes, init_array
from . import layers as L
from . import losses as LS

ALIGN = 4  # every parameter starts on a 16-byte boundary inside its arena


class _Shared:
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class LRVariable:
    """`model.optimizer.lr`: what the reference hands to K.get_value / K.set_value (DeepLabv3plus.py:658,736)."""

    def __init__(self, v):
        self.value = float(v)

    def assign(self, v):
        self.value = float(v)

    def numpy(self):
        return np.float32(self.value)

    def __float__(self):
        return self.value

    def __repr__(self):
        return f"<lr {self.value:g}>"


class Optimizer:
    """Keras-2 Adam state holder (`compile(optimizer='adam')`); `lr` is what the LR callbacks set."""

    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self._lr = LRVariable(lr)
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.iterations = 0

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, v):
        self._lr.assign(float(v))

    learning_rate = lr


class History:
    def __init__(self):
        self.history: Dict[str, list] = {}
        self.epoch: List[int] = []


class Model:
    def __init__(self, inputs, outputs, name=None, seed: int = 1103, dtype=None):
        """dtype: "float32" (the reference's precision) or "mixed_bfloat16" / "bfloat16" = bf16 storage of every
        activation with fp32 arithmetic, fp32 master weights, BatchNorm statistics, softmax head, loss and Adam
        (BASELINE config 3); None takes mixed_precision.global_policy()."""
        self.inputs = [inputs] if isinstance(inputs, KTensor) else list(inputs)
        self.outputs = [outputs] if isinstance(outputs, KTensor) else list(outputs)
        assert len(self.inputs) == 1 and len(self.outputs) == 1, "the path's models are single-input single-output"
        self.name = name or "model"
        self.nodes: List[Node] = collect_nodes(self.outputs)
        for i, n in enumerate(self.nodes):
            n.index = i
        self.seed = seed
        self.stop_training = False
        self.optimizer: Optional[Optimizer] = None
        self.loss_kind: Optional[int] = None
        self.metric_names: List[str] = []
        self._rt = None
        from . import mixed_precision
        self.compute_dtype = mixed_precision.resolve(dtype)   # "float32" | "bfloat16"
        self._fuse()
        self._layout_params()
        self.dist = None  # set by dist.DataParallel

    # ------------------------------------------------------------------------------------- graph passes
    def _fuse(self):
        """Peephole fusions that remove full-tensor passes (each is exact, not an approximation):
        BN -> ReLU   => BN kernel applies ReLU (and masks in backward);
        ReLU -> SeparableConv2D => depthwise gather applies ReLU (pre_relu);
        Add -> ReLU  => add_n applies ReLU.
        A ReLU is absorbed only if its output has exactly one consumer (or, for producers, it is the only
        consumer of the producer's output and is not a model output)."""
        outs = {id(t) for t in self.outputs}
        # A bias added right before a training-mode BatchNormalization has an identically zero gradient (BN's input
        # gradient sums to zero over the batch by construction): the column-sum launch is dropped and the slot in the
        # gradient arena stays at its initial 0 (what the sum would return up to fp32 noise of ~1e-10).
        for n in self.nodes:
            if isinstance(n, (L._ConvNode, L._SepConvNode)) and getattr(n, "activation", None) in (None, "linear") \
                    and id(n.output) not in outs:
                cons = n.output.consumers
                if len(cons) == 1 and isinstance(cons[0], L._BNNode) and len(n.output.shape) == 4:
                    n.bias_grad_zero = True
                    n.emit_bn_stats = True  # the conv epilogue hands BN its statistics (sg_conv2d_fwd_stats)
        # UpSampling2D(2) -> Conv2D 3x3 'same' (the decoder's last stage, train_model/DeepLabv3plus.py:476-477): the convolution's
        # kernels read the up-sampling's source directly (sub-pixel forward with 4/9 of the products; csrc/conv_x6p.h) and the
        # 4x tensor and its gradient are never built.  Marked here for every such pair; whether a runtime takes the fused
        # kernels (fp32 storage, geometry, SG_UP2_FUSE != 0) is _Runtime.up2_on.
        for n in self.nodes:
            if isinstance(n, L._UpNode) and n.size == 2 and id(n.output) not in outs and len(n.output.consumers) == 1:
                c = n.output.consumers[0]
                if (isinstance(c, L._ConvNode) and c.k == 3 and c.stride == 1 and c.dilation == 1 and c.padding == "same"
                        and c.activation in (None, "linear", "relu")):
                    n.fused_into, c.up_src = c, n
        for n in self.nodes:
            if not isinstance(n, L._ActNode) or n.act != "relu" or n.fused_away:
                continue
            src = n.inputs[0]
            prod = src.node
            if prod is not None and len(src.consumers) == 1 and id(src) not in outs:
                if isinstance(prod, L._BNNode) and not prod.relu:
                    prod.relu, n.fused_away = True, True
                    continue
                if isinstance(prod, L._AddNode) and not prod.relu:
                    prod.relu, n.fused_away = True, True
                    continue
            cons = n.output.consumers
            if len(cons) == 1 and isinstance(cons[0], L._SepConvNode) and id(n.output) not in outs and not cons[0].pre_relu:
                cons[0].pre_relu, n.fused_away = True, True
        # BatchNormalization (+ fused ReLU) -> SeparableConv2D, training mode: the depthwise gather applies the normalisation
        # to the raw tensor (sg_dwconv2d_fwd_bn / _wgrad_bn), so the normalised tensor is never written or read - two of the
        # layer's seven tensor passes.  Needs the statistics from the producing convolution's epilogue (decided at run
        # time: _BNNode.forward falls back to the materialising form) and the stride-1 run kernels' geometry.
        # Default ON since round 4 (SG_BN_DEFER=0 restores the materialising form; tests/test_models_gpu.py keeps one leg on it).
        # Round 2 measured "no net gain" (the 39 dropped bn_apply launches saved 1.05 ms and the depthwise kernels of that
        # round gave 0.8 ms of it back); re-measured on the round-3 run kernels: 76.59 -> 75.71 and 76.96 -> 75.93 ms per fp32
        # step, bit-identical, peak memory -2.7 GiB (LAB_NOTEBOOK.md 11.2).
        # BatchNormalization (no ReLU) -> two-operand Add: the residual adds of the Xception blocks sum a normalised branch and
        # the shortcut (itself a normalised 1x1 convolution in the entry / exit flow).  The Add applies the normalisation while
        # it sums (sg_add2_bn), the normalised tensor is never written or read: one of the layer's two forward passes
        # (SG_BN_ADD=0 keeps them apart).  The backward is untouched: dy of the Add IS dy of the BatchNormalization, whose
        # backward reads its raw input.
        if os.environ.get("SG_BN_ADD", "1") == "1":
            for n in self.nodes:
                if not isinstance(n, L._AddNode) or len(n.inputs) != 2 or len(n.output.shape) != 4:
                    continue
                epss = set()
                for i, t in enumerate(n.inputs):
                    p_ = t.node
                    through_relu = False   # BN -> ReLU (absorbed into the BN) -> Add: the Add applies both (res34.py's blocks)
                    if (isinstance(p_, L._ActNode) and p_.fused_away and len(t.consumers) == 1 and id(t) not in outs
                            and isinstance(p_.inputs[0].node, L._BNNode) and p_.inputs[0].node.relu
                            and len(p_.inputs[0].consumers) == 1):
                        t, p_, through_relu = p_.inputs[0], p_.inputs[0].node, True
                    if (isinstance(p_, L._BNNode) and p_.relu == through_relu and p_.defer_to is None and p_.defer_add is None
                            and len(t.consumers) == 1 and id(t) not in outs and t.shape[-1] % 4 == 0
                            and n.inputs[0] is not n.inputs[1]):
                        n.bn_src[i], p_.defer_add = p_, n
                        epss.add(p_.epsilon)
                if len(epss) > 1:   # (one eps per launch: two different ones never occur on this path)
                    for i, p_ in enumerate(n.bn_src):
                        if p_ is not None:
                            p_.defer_add, n.bn_src[i] = None, None
        # BatchNormalization (+ fused ReLU) -> SeparableConv2D (stride 1), the layer's only consumer: the input gradient that the
        # depthwise convolution's dgrad writes IS the BatchNormalization's output gradient.  That kernel also sums, per channel,
        # dbeta = sum g and dgamma = sum g * xhat in its epilogue (sg_dwconv2d_dgrad_bnsums: one more read of the layer's raw
        # input), so the reduction pass of the BatchNormalization's backward - two tensor reads of its five passes - is not run
        # (round 4; conv_bn_relu / the Xception blocks, train_model/DeepLabv3plus.py:323-416,424-429; SG_BN_SUMS=0 switches it
        # off).  Decided again at run time (geometry, training mode: _SepConvNode.backward).
        if os.environ.get("SG_BN_SUMS", "1") == "1":
            for n in self.nodes:
                if not isinstance(n, L._BNNode) or len(n.output.shape) != 4 or id(n.output) in outs or n.defer_add is not None:
                    continue
                t = n.output
                while (len(t.consumers) == 1 and isinstance(t.consumers[0], L._ActNode) and t.consumers[0].fused_away
                       and id(t.consumers[0].output) not in outs):
                    t = t.consumers[0].output
                if len(t.consumers) != 1 or not isinstance(t.consumers[0], L._SepConvNode) or id(t) in outs:
                    continue
                sc = t.consumers[0]
                _, h, w, c = t.shape
                if sc.stride == 1 and sc.bnsum_src is None and w % 4 == 0 and c % 4 == 0:
                    n.sums_from, sc.bnsum_src = sc, n
            # The same through a residual Add that applies the layer (defer_add): the Add hands its output gradient on unchanged,
            # so a BatchNormalization in front of it receives the gradient of the Add's OUTPUT - and where that output opens the
            # next Xception block, its complete gradient leaves the depthwise dgrad of the block's first SeparableConv2D (which
            # adds the gradient collected from the block's own residual add: take_pending).  Condition: that SeparableConv2D
            # (with the ReLU absorbed into its gather) is the LAST consumer of the tensor in the backward sweep, i.e. the first
            # in node order.
            for sc in self.nodes:
                if not isinstance(sc, L._SepConvNode) or sc.bnsum_src is not None or sc.stride != 1:
                    continue
                root, first = sc.inputs[0], sc
                while isinstance(root.node, L._ActNode) and root.node.fused_away and len(root.consumers) == 1:
                    root, first = root.node.inputs[0], root.node
                a = root.node
                if not isinstance(a, L._AddNode) or a.relu or id(root) in outs or len(root.shape) != 4:
                    continue
                if any(cn is not first and cn.index < sc.index for cn in root.consumers):
                    continue
                cands = [b for b in a.bn_src if b is not None and b.sums_from is None and b.defer_add is a]
                _, h, w, c = root.shape
                if cands and w % 4 == 0 and c % 4 == 0:
                    cands[0].sums_from, sc.bnsum_src = sc, cands[0]
        if os.environ.get("SG_BN_DEFER", "1") == "1":
            for n in self.nodes:
                if not isinstance(n, L._BNNode) or len(n.output.shape) != 4 or id(n.output) in outs:
                    continue
                t = n.output
                while (len(t.consumers) == 1 and isinstance(t.consumers[0], L._ActNode) and t.consumers[0].fused_away
                       and id(t.consumers[0].output) not in outs):
                    t = t.consumers[0].output
                if len(t.consumers) != 1 or not isinstance(t.consumers[0], L._SepConvNode) or id(t) in outs:
                    continue
                sc = t.consumers[0]
                _, h, w, c = t.shape
                if sc.stride == 1 and not sc.pre_relu and sc.bn_src is None and w % 4 == 0 and c % 4 == 0:
                    n.defer_to, sc.bn_src = sc, n
        self._fuse_bn_conv(outs)
        # SeparableConv2D -> BatchNormalization whose backward column sums come from elsewhere (sums_from): the backward APPLY can
        # ride in the A path of the pointwise dgrad (csrc/conv_pw.h, BNB form; _Runtime.bnb_on decides per runtime and batch)
        for n in self.nodes:
            if isinstance(n, L._BNNode) and n.sums_from is not None and len(n.output.shape) == 4:
                prod = n.inputs[0].node
                if (isinstance(prod, L._SepConvNode) and prod.activation in (None, "linear") and len(n.inputs[0].consumers) == 1
                        and id(n.inputs[0]) not in outs):
                    n.bnb_to = prod

    # (continued in _fuse_bn_conv, called at the end of _fuse)
    def _fuse_bn_conv(self, outs):
        """BatchNormalization(+ReLU) -> Conv2D whose kernels can normalise while they load (round 5): a thin 1x1 convolution (the
        softmax head, the sSE gate) or a 3x3 stride-1 convolution on 32 / 64 channels (the patch kernels).  The decoder's last stage
        `conv_bn_relu(32) -> conv_bn_relu(32) -> Conv2D(2, softmax)` (train_model/DeepLabv3plus.py:477-480) has two such pairs on
        512 x 512 x 32 tensors.  The normalised tensor is then never written or read (forward: one write + one read of the
        tensor; the filter gradient re-normalises in its loader; backward of the BatchNormalization is unchanged - it already
        recomputes the ReLU mask from its raw input)."""
        for n in self.nodes:
            if (not isinstance(n, L._BNNode) or len(n.output.shape) != 4 or id(n.output) in outs
                    or n.defer_to is not None or n.defer_add is not None):
                continue
            t = n.output
            while (len(t.consumers) == 1 and isinstance(t.consumers[0], L._ActNode) and t.consumers[0].fused_away
                   and id(t.consumers[0].output) not in outs):
                t = t.consumers[0].output
            if len(t.consumers) != 1 or id(t) in outs:
                continue
            c = t.consumers[0]
            if not isinstance(c, L._ConvNode) or c.bn_src is not None or c.up_src is not None or c.stride != 1 or c.dilation != 1:
                continue
            cin = t.shape[-1]
            thin = c.k == 1 and c.filters <= 4 and cin % 4 == 0 and cin >= 16
            patch = c.k == 3 and c.padding == "same" and cin in (32, 64)
            if thin or patch:
                n.defer_conv, c.bn_src = c, n

    def _layout_params(self):
        self.params: List[ParamSpec] = [p for n in self.nodes for p in n.params]
        off_t = off_n = 0
        for p in self.params:
            if p.trainable:
                p.offset, off_t = off_t, off_t + (p.size + ALIGN - 1) // ALIGN * ALIGN
            else:
                p.offset, off_n = off_n, off_n + (p.size + ALIGN - 1) // ALIGN * ALIGN
        self._n_train, self._n_frozen = off_t, off_n

    # ---------------------------------------------------------------------------------------- inspection
    @property
    def layers(self):
        return self.nodes

    @property
    def trainable_weights(self):
        return [p for p in self.params if p.trainable]

    @property
    def non_trainable_weights(self):
        return [p for p in self.params if not p.trainable]

    def count_params(self):
        return sum(p.size for p in self.params)

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"')
        print_fn(f"{'Layer (type)':<44}{'Output Shape':<26}{'Param #':>10}")
        for n in self.nodes:
            print_fn(f"{n.name + ' (' + n.op + ')':<44}{str(n.output.shape):<26}{sum(p.size for p in n.params):>10}")
        tr = sum(p.size for p in self.params if p.trainable)
        print_fn(f"Total params: {self.count_params():,}")
        print_fn(f"Trainable params: {tr:,}")
        print_fn(f"Non-trainable params: {self.count_params() - tr:,}")

    def flops(self, batch=1) -> int:
        """Nominal forward FLOPs (2*MACs of conv / convT / dense), SURVEY.md §8d convention."""
        return sum(n.flops(batch) for n in self.nodes)

    # ------------------------------------------------------------------------------------------- runtime
    def _runtime(self):
        if self._rt is None:
            self._rt = _Runtime(self)
        return self._rt

    def get_weights(self) -> List[np.ndarray]:
        rt = self._runtime()
        return [rt.param(p).detach().cpu().numpy().copy() for p in self.params]

    def set_weights(self, weights: Sequence[np.ndarray]):
        import torch
        rt = self._runtime()
        if len(weights) != len(self.params):
            raise ValueError(f"set_weights: expected {len(self.params)} arrays, got {len(weights)}")
        for p, w in zip(self.params, weights):
            w = np.asarray(w, dtype=np.float32)
            if tuple(w.shape) != p.shape:
                raise ValueError(f"set_weights: {p.name} expects {p.shape}, got {w.shape}")
            rt.param(p).copy_(torch.from_numpy(np.ascontiguousarray(w)))
        rt.weights_changed()

    def get_gradients(self) -> List[np.ndarray]:
        """Trainable-weight gradients of the last train step / backward, in trainable_weights order."""
        rt = self._runtime()
        return [rt.grad(p).detach().cpu().numpy().copy() for p in self.params if p.trainable]

    def save_weights(self, path):
        """Under data parallelism the BatchNorm moving statistics of the replicas are averaged first and only rank 0
        writes the file; every rank must call it (it is a collective then)."""
        from .weights_io import save_weights
        if self.dist is None:
            save_weights(self, path)
            return
        self.dist.sync_moving_stats(self._runtime())
        err = None
        if self.dist.rank == 0:
            try:
                save_weights(self, path)
            except Exception as e:  # the other ranks are about to enter the collective below: meet them there first
                err = e
        failed = self.dist.any_failed(err is not None)  # collective OR of the outcome; doubles as the barrier
        if err is not None:
            raise err
        if failed:
            raise RuntimeError(f"save_weights({path!r}) failed on rank 0")

    def load_weights(self, path):
        from .weights_io import load_weights
        load_weights(self, path)

    # ------------------------------------------------------------------------------------------ compile
    def compile(self, optimizer="adam", loss=None, metrics=None, jit_compile=False, **kw):
        """jit_compile (tf.keras's name for "compile the train step"): capture the whole training step - forward, loss,
        metric counts, backward, Adam - into one hipGraph per input shape and replay it (GraphedTrainStep).  Same
        kernels, same order: bit-identical to the eager step; ~1800 launches per step become one."""
        if isinstance(optimizer, str):
            if optimizer.lower() != "adam":
                raise ValueError("the reference compiles with optimizer='adam' (DeepLabv3plus.py:835)")
            optimizer = Optimizer()
        self.optimizer = optimizer
        self.loss_kind = LS.resolve_loss(loss)
        self.metric_names = [LS.resolve_metric(m) for m in (metrics or [])]
        self.jit_compile = bool(jit_compile)
        self._train_graphs = {}
        self._train_graph_seen = {}

    # ------------------------------------------------------------------------------------------ predict
    def predict(self, x, batch_size=32, verbose=0, **kw):
        """numpy [N,H,W,3] (any float dtype; predict.py feeds float64) -> numpy float32 probabilities."""
        import torch
        rt = self._runtime()
        x = np.asarray(x)
        outs = []
        for i in range(0, x.shape[0], batch_size):
            xh = torch.from_numpy(np.ascontiguousarray(x[i:i + batch_size], dtype=np.float32))
            # serialised per device (buildAPI.py:78,111 calls predict from Flask request threads on shared models):
            # the value table of the sweep and the engine's scratch buffer are not re-entrant
            with rt.eng.lock:
                outs.append(rt.forward(xh.to(rt.eng.device), training=False).cpu().numpy())
                rt.release()
        return np.concatenate(outs, 0)

    def _last_use(self):
        lu = getattr(self, "_lu", None)
        if lu is None:
            lu = {}
            for i, n in enumerate(self.nodes):
                for t in n.inputs:
                    lu[id(t)] = i
            lu.pop(id(self.outputs[0]), None)
            self._lu = lu
        return lu

    def capture_predict(self, batch: int):
        """hipGraph-captured inference forward for a fixed batch (BASELINE config 5): returns a callable
        `f(x_dev[batch,H,W,3]) -> probs_dev` that replays ONE graph launch instead of several hundred kernel
        launches.  Every libsegengine entry point is capture-safe (no allocation / synchronisation inside)."""
        return GraphedPredict(self, batch)

    def predict_device(self, x_dev):
        """Device tensor in, device tensor out (used by the tile pipeline and the benchmark)."""
        rt = self._runtime()
        with rt.eng.lock:
            y = rt.forward(x_dev, training=False)
            rt.release()
        return y

    def __call__(self, x, training=False):
        return self.predict_device(x) if not training else self._runtime().forward(x, True)

    # ----------------------------------------------------------------------------------------- training
    def train_on_batch(self, x, y, return_device_scalars=False):
        """One optimisation step; x [N,H,W,3], y [N,H,W,4|2] as numpy or device tensors.
        Returns dict(loss=..., PA=..., ...) of python floats (forces one host sync) unless
        `return_device_scalars`."""
        import torch
        if self.optimizer is None or self.loss_kind is None:
            raise RuntimeError("compile() the model before training")
        rt = self._runtime()
        with rt.eng.lock:
            xd, yd = rt.to_device(x), rt.to_device(y)
            if getattr(self, "jit_compile", False):
                key = (tuple(xd.shape), tuple(yd.shape), xd.dtype, yd.dtype)
                g = self._train_graphs.get(key)
                if g is None:
                    # two eager steps per shape first: they size the scratch buffer and the weight-plane arena, run every
                    # kernel's one-time attribute calls and warm the allocator - and they are real training steps
                    seen = self._train_graph_seen.get(key, 0)
                    self._train_graph_seen[key] = seen + 1
                    if seen >= 2:
                        g = self._train_graphs[key] = GraphedTrainStep(self, xd, yd)
                if g is not None:
                    loss, counts = g.run(xd, yd)
                    return (loss, counts) if return_device_scalars else self._logs(loss, counts)
            p = rt.forward(xd, training=True)
            loss = rt.eng.loss_fwd(self.loss_kind, p, yd)
            counts = rt.eng.confusion_counts(p, yd) if self.metric_names else None
            dp = rt.eng.loss_bwd(self.loss_kind, p, yd, 1.0)
            rt.backward(dp)
            grad_scale = 1.0
            if self.dist is not None:
                grad_scale = self.dist.allreduce_grads(rt)
                loss, counts = self.dist.reduce_step_scalars(loss, counts)  # logs describe the GLOBAL batch
            opt = self.optimizer
            opt.iterations += 1
            t = opt.iterations
            lr_t = float(opt.lr) * math.sqrt(1.0 - opt.beta_2 ** t) / (1.0 - opt.beta_1 ** t)
            rt.eng.adam_step(rt.w_train, rt.adam_m, rt.adam_v, rt.g_train, lr_t, opt.beta_1, opt.beta_2, opt.epsilon,
                             grad_scale)
            rt.weights_changed()
            rt.release()
        if return_device_scalars:
            return loss, counts
        return self._logs(loss, counts)

    def test_on_batch(self, x, y):
        rt = self._runtime()
        with rt.eng.lock:
            xd, yd = rt.to_device(x), rt.to_device(y)
            p = rt.forward(xd, training=False)
            loss = rt.eng.loss_fwd(self.loss_kind, p, yd)
            counts = rt.eng.confusion_counts(p, yd) if self.metric_names else None
            if self.dist is not None:  # val_* logs describe the GLOBAL validation batch on every rank, as the training
                loss, counts = self.dist.reduce_step_scalars(loss, counts)  # logs do: callbacks then decide alike
            rt.release()
        return self._logs(loss, counts)

    def _logs(self, loss, counts):
        logs = {"loss": float(loss.item())}
        if counts is not None:
            tp, tn, fp, fn = [int(v) for v in counts.cpu().tolist()]
            m = LS.metrics_from_counts(tp, tn, fp, fn)
            for name in self.metric_names:
                logs[name] = m[name]
        return logs

    def fit_generator(self, generator, steps_per_epoch=None, epochs=1, verbose=1, callbacks=None,
                      validation_data=None, validation_steps=None, initial_epoch=0, **kw):
        """Keras-2 `fit_generator` loop (train_model/DeepLabv3plus.py:844-849): per-batch callbacks, epoch
        logs are the MEAN of the per-batch values, validation in inference mode."""
        callbacks = list(callbacks or [])
        hist = History()
        for cb in callbacks:
            cb.set_model(self)
        self.stop_training = False
        for cb in callbacks:
            cb.on_train_begin({})
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                cb.on_epoch_begin(epoch, {})
            sums: Dict[str, float] = {}
            t0 = time.time()
            for step in range(steps_per_epoch):
                for cb in callbacks:
                    cb.on_batch_begin(step, {})
                x, y = next(generator)
                logs = self.train_on_batch(x, y)
                for k, v in logs.items():
                    sums[k] = sums.get(k, 0.0) + v
                for cb in callbacks:
                    cb.on_batch_end(step, dict(logs))
            ep_logs = {k: v / max(steps_per_epoch, 1) for k, v in sums.items()}
            if validation_data is not None:
                vs: Dict[str, float] = {}
                nval = validation_steps or 1
                for _ in range(nval):
                    x, y = next(validation_data)
                    for k, v in self.test_on_batch(x, y).items():
                        vs[k] = vs.get(k, 0.0) + v
                ep_logs.update({"val_" + k: v / nval for k, v in vs.items()})
            if verbose:
                msg = " - ".join(f"{k}: {v:.4f}" for k, v in ep_logs.items())
                print(f"Epoch {epoch + 1}/{epochs} - {time.time() - t0:.1f}s - {msg}")
            hist.epoch.append(epoch)
            for k, v in ep_logs.items():
                hist.history.setdefault(k, []).append(v)
            for cb in callbacks:
                cb.on_epoch_end(epoch, ep_logs)
            if self.stop_training:
                break
        for cb in callbacks:
            cb.on_train_end({})
        self.history = hist
        return hist

    def fit(self, x=None, y=None, batch_size=None, epochs=1, steps_per_epoch=None, **kw):
        if hasattr(x, "__next__"):
            return self.fit_generator(x, steps_per_epoch=steps_per_epoch, epochs=epochs, **kw)
        x, y = np.asarray(x), np.asarray(y)
        bs = batch_size or 32
        steps = steps_per_epoch or max(x.shape[0] // bs, 1)

        def gen():
            while True:
                for i in range(steps):
                    yield x[i * bs:(i + 1) * bs], y[i * bs:(i + 1) * bs]
        return self.fit_generator(gen(), steps_per_epoch=steps, epochs=epochs, **kw)


class _CaptureGuard:
    """Around a stream capture: Python's cyclic garbage collector must not run inside it.  A collection can finalise an
    unrelated hipGraph (e.g. the captured step of a model dropped earlier, kept alive until now by a reference cycle), and
    destroying a graph while a stream is capturing is a HIP error that ends the process (hipErrorStreamCaptureUnsupported
    thrown from a destructor).  Collect first, then keep the collector off until the capture has ended."""

    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        import gc
        if self._was:
            gc.enable()
        return False


class GraphedPredict:
    """Static-shape inference forward captured into a hipGraph (stream capture of the engine's launches)."""

    def __init__(self, model: Model, batch: int):
        import torch
        rt = model._runtime()
        self.torch, self.rt = torch, rt
        shape = (batch,) + tuple(model.inputs[0].shape[1:])
        eng = rt.eng
        with eng.lock:
            self.x = eng.zeros(*shape)
            peak_before, eng._ws_peak = eng._ws_peak, 0
            for _ in range(2):  # warm-up outside capture: first-launch attribute calls, and the sizing pass of the scratch
                rt.forward(self.x, training=False)
                rt.release()
            torch.cuda.synchronize(eng.device)
            # The graph's kernel nodes carry the scratch pointer: give this graph a buffer of its own (sized by the
            # warm-up pass), never the engine's shared one, which a later eager call or another model may re-grow.
            self.ws = torch.empty(max(eng._ws_peak, 256) + 256, dtype=torch.uint8, device=eng.device)
            # the engine-wide peak stays monotone: a training-step capture that follows sizes its buffer by it
            eng._ws_peak = max(eng._ws_peak, peak_before)
            self.graph = torch.cuda.CUDAGraph()
            with eng.private_ws(self.ws), _CaptureGuard():
                with torch.cuda.graph(self.graph):
                    self.y = rt.forward(self.x, training=False)
            rt.values = {}
            # the planes this graph reads: kept alive and refreshed by the graph's owner, whatever the runtime's current
            # planes are by then (a later call with another batch size re-keys and re-allocates them)
            self._planes = (rt._planes_arena, rt._planes_jobs, rt._planes_launch)
            self._planes_version = rt._w_version

    def __call__(self, x_dev):
        """Replays the graph on x_dev.  The returned tensor is the graph's static output buffer: consume (or clone) it
        before the next call."""
        with self.rt.eng.lock:
            if self._planes_version != self.rt._w_version and self._planes[0] is not None:  # weights replaced since
                self.rt.prepare_into(*self._planes)
                self._planes_version = self.rt._w_version
            self.x.copy_(x_dev, non_blocking=True)
            self.graph.replay()
        return self.y


class _Lane:
    """The filter-gradient calls deferred while one segment of a training step is captured (ops.Engine.side_run)."""

    def __init__(self):
        self.thunks, self.keep, self.blocks = [], [], 0

    def defer(self, fn, tensors):
        self.thunks.append(fn)
        self.keep.extend(tensors)
        self.blocks += 1

    def take(self):
        out = (self.thunks, self.keep)
        self.thunks, self.keep, self.blocks = [], [], 0
        return out


class GraphedTrainStep:
    """One optimisation step of a compiled model - forward, loss, confusion counts, backward, Adam - captured into
    hipGraphs for one (x, y) shape (Model.compile(jit_compile=True)).  The step's inputs are two static device buffers, its
    outputs (loss, counts) two more; Adam's bias-corrected learning rate, the only number that changes from replay to
    replay, is read by the kernel from a one-float device buffer written before each replay (sg_adam_step_lr).  The weight
    planes are rebuilt inside the graph (the capture starts with them marked stale), BatchNorm's moving statistics and the
    optimiser moments are updated in place as in the eager step.

    The capture is CUT into a chain of graph SEGMENTS, plus one graph for Adam:
      * under data parallelism (model.dist set) wherever the backward sweep completes a gradient bucket
        (dist.BucketReducer's schedule): between two segments the host hands the finished bucket to the transport EXACTLY as
        the eager step does (an eager sg_comm_allreduce_sum / all_reduce on the communication stream), so RCCL never runs
        inside a capture and the all-reduce of bucket k still overlaps the backward of segment k+1;
      * with lanes (SG_JIT_LANES, default on) also every SG_JIT_LANE_BLOCKS filter gradients: a segment's filter-gradient
        calls are NOT recorded into its main graph M_k but collected (Engine.side_run -> _Lane) and recorded, right after
        M_k, into a side graph W_k of their own.  The replay launches M_0, W_0 || M_1, W_1 || M_2 ...: W_k goes to the
        second stream behind an event after M_k and runs beside M_(k+1) - the eager step's overlap of the MFMA-bound filter
        gradients with the bandwidth-bound kernels of the chain, for two graph launches and three event
        calls per segment instead of ~1800 kernel launches.  (ONE forked graph replays correctly too, but hipGraphLaunch
        then queues its nodes from the host one by one: 56 ms.)  Memory: the operands W_k reads live in the capture's pool;
        they are held until the capture of M_(k+2) begins, and the replay makes M_(k+2) wait for W_k.
    Same kernels on the same operands as the eager step: bit-identical results."""

    def __init__(self, model: "Model", xd, yd):
        import torch
        rt = model._runtime()
        eng = rt.eng
        self.torch, self.model, self.rt = torch, model, rt
        opt = model.optimizer
        dist = model.dist
        self.x, self.y = torch.empty_like(xd), torch.empty_like(yd)
        self.lr = eng.zeros(4)
        lanes = os.environ.get("SG_JIT_LANES", "1") != "0" and eng._side_on
        lane_blocks = max(1, int(os.environ.get("SG_JIT_LANE_BLOCKS", "24")))
        # (the eager sizing steps ran the filter gradients on the side stream with its own scratch; without lanes they run
        # inline here, on the main scratch)
        self.ws = torch.empty(max(eng._ws_peak, eng._ws2_peak, 256) + 256, dtype=torch.uint8, device=eng.device)
        self.ws2 = torch.empty(max(eng._ws2_peak, 256) + 256, dtype=torch.uint8, device=eng.device) if lanes else None
        rt.release()
        # The job table of the weight planes is (re)built HERE, outside the capture, for this batch and mode: a forward with
        # another batch size since the eager warm-up steps (a validation batch, a predict()) has re-keyed the runtime's
        # planes, and the rebuild allocates the plane arena and copies the table from pageable host memory - neither may
        # happen inside a stream capture.  What the capture then records of ensure_planes is the one sg_prepare_planes launch.
        rt.ensure_planes(int(xd.shape[0]), True)
        rt.weights_changed()  # the graph must contain the plane preparation: every replay follows an optimiser step
        self._planes_key = rt._planes_key
        torch.cuda.synchronize(eng.device)
        self.segments = []        # [main graph, [(start, end) arena ranges complete after it], side graph | None, kept operands]
        segmented = dist is not None or lanes
        pool = torch.cuda.graph_pool_handle() if segmented else None
        cuts = _SegmentCuts(dist.reducer.buckets) if dist is not None else None
        lane = _Lane() if lanes else None
        state = {"ctx": None, "graph": None}
        # every capture is "thread_local" (ADVICE r4): a HIP call from ANOTHER thread during the capture - a DataLoader's
        # pin-memory thread, a Flask predict thread waiting on the Engine lock - must not invalidate it.  The side graphs W_k
        # replay one after the other on the side stream, so they share ONE private pool of their own (never the main graphs':
        # W_k runs beside M_(k+1)) instead of one pool each.
        side_pool = torch.cuda.graph_pool_handle() if lanes else None
        mode = "thread_local"

        def begin():
            if len(self.segments) >= 2:   # the replay makes this segment wait for W_(k-2): its operands may be reused from here on
                self.segments[-2][3] = None
            g = torch.cuda.CUDAGraph()
            kw = {"pool": pool} if pool is not None else {}
            ctx = torch.cuda.graph(g, capture_error_mode=mode, **kw)
            ctx.__enter__()
            state["ctx"], state["graph"] = ctx, g

        def end(ready):
            state["ctx"].__exit__(None, None, None)
            main_graph = state["graph"]
            state["ctx"] = state["graph"] = None
            side_graph, keep = None, None
            if lane is not None and lane.thunks:
                thunks, keep = lane.take()
                side_graph = torch.cuda.CUDAGraph()
                eng.lane = None          # (the calls below ARE the deferred ones)
                eng._in_side = True      # their scratch is the side buffer
                try:
                    with torch.cuda.graph(side_graph, pool=side_pool, capture_error_mode=mode):
                        for fn in thunks:
                            fn()
                finally:
                    eng._in_side = False
                    eng.lane = lane
            self.segments.append([main_graph, ready, side_graph, keep])

        def fires(index):
            return cuts is not None and cuts.next < len(cuts.buckets) and cuts.buckets[cuts.next][2] >= index

        def node_done(index):   # the backward sweep has finished node `index`: cut when that completes buckets / fills the lane
            ready = cuts.pop_ready(index) if cuts is not None else []
            if ready or (lane is not None and lane.blocks >= lane_blocks):
                end(ready)
                if index > 0:   # node 0 ends the sweep: nothing is left to capture behind it
                    begin()

        saved_hook, saved_fires = rt.on_node_done, rt.node_done_fires
        with eng.private_ws(self.ws, self.ws2), _CaptureGuard():
            try:
                rt.on_node_done = node_done if segmented else None
                rt.node_done_fires = fires if segmented else None
                eng.lane = lane
                begin()
                p = rt.forward(self.x, training=True)
                self.loss = eng.loss_fwd(model.loss_kind, p, self.y)
                self.counts = eng.confusion_counts(p, self.y) if model.metric_names else None
                dp = eng.loss_bwd(model.loss_kind, p, self.y, 1.0)
                rt.backward(dp)
                if not segmented:  # Adam rides in the same graph
                    eng.adam_step(rt.w_train, rt.adam_m, rt.adam_v, rt.g_train, 0.0, opt.beta_1, opt.beta_2, opt.epsilon, 1.0,
                                  lr_dev=self.lr)
                    end([])
                else:
                    if state["ctx"] is not None:
                        end(cuts.pop_ready(-1) if cuts is not None else [])
                    assert cuts is None or cuts.done(), "a gradient bucket was never handed over"
                    eng.lane = None
                    begin()  # its own graph: it runs after the LAST all-reduce / side graph has joined the compute stream
                    eng.adam_step(rt.w_train, rt.adam_m, rt.adam_v, rt.g_train, 0.0, opt.beta_1, opt.beta_2, opt.epsilon,
                                  1.0 / (dist.world if dist is not None else 1), lr_dev=self.lr)
                    end([])
            finally:
                eng.lane = None
                if state["ctx"] is not None:   # an error inside a capture: leave capture mode before re-raising
                    try:
                        state["ctx"].__exit__(None, None, None)
                    except Exception:
                        pass
                rt.on_node_done, rt.node_done_fires = saved_hook, saved_fires
        for seg in self.segments:
            seg[3] = None   # nothing is allocated in the pool from here on
        self.segmented = segmented
        self.graph = self.segments[0][0]
        self.side_graphs = sum(1 for seg in self.segments if seg[2] is not None)
        self._lanes = _StreamLanes(eng) if self.side_graphs else None
        self._loss_out = self._counts_out = None
        assert rt._planes_key == self._planes_key, "the weight-plane job table was rebuilt inside the capture"
        self._planes = (rt._planes_arena, rt._planes_jobs)  # kept alive: the graph's nodes carry their addresses
        rt.release()
        rt.weights_changed()

    def run(self, xd, yd):
        """Replays the step on (xd, yd).  Returns the graph's static loss / counts buffers: read them before the next step."""
        opt = self.model.optimizer
        dist = self.model.dist
        self.x.copy_(xd, non_blocking=True)
        self.y.copy_(yd, non_blocking=True)
        opt.iterations += 1
        t = opt.iterations
        lr_t = float(opt.lr) * math.sqrt(1.0 - opt.beta_2 ** t) / (1.0 - opt.beta_1 ** t)
        self.lr.fill_(lr_t)
        if not self.segmented:
            self.graph.replay()
            self.rt.weights_changed()
            return self.loss, self.counts
        run_segments([(g.replay, ready, None if w is None else w.replay) for g, ready, w, _ in self.segments[:-1]],
                     self.rt.g_train, dist.tp if dist is not None else None, self._lanes)
        out = (self.loss, self.counts)
        if dist is not None:
            # loss / counts of the GLOBAL batch: reduced in ordinary buffers of this object (the graphs' own outputs live in
            # the capture's private pool and are rewritten by the next replay)
            if self._loss_out is None:
                self._loss_out = self.torch.empty_like(self.loss)
                self._counts_out = None if self.counts is None else self.torch.empty_like(self.counts)
            self._loss_out.copy_(self.loss)
            if self.counts is not None:
                self._counts_out.copy_(self.counts)
            out = dist.reduce_step_scalars(self._loss_out, self._counts_out)
        self.segments[-1][0].replay()   # Adam on the summed gradients (1 / world folded into the kernel)
        self.rt.weights_changed()
        return out


class _StreamLanes:
    """The two HIP streams of a replayed step with side graphs: `main` is torch's current stream, `side` the engine's second
    stream.  run_segments drives it; the CPU tests pass an object with the same four methods that runs everything in line."""

    def __init__(self, eng):
        import torch
        self.torch, self.eng = torch, eng
        if eng._side_stream is None:
            eng._side_stream = torch.cuda.Stream(device=eng.device)
        self.side = eng._side_stream
        self._free = []

    def _event(self):
        return self._free.pop() if self._free else self.torch.cuda.Event()

    def fork(self):
        """The side stream waits for everything queued on the main stream so far."""
        ev = self._event()
        ev.record(self.torch.cuda.current_stream(self.eng.device))
        self.side.wait_event(ev)
        self._free.append(ev)

    def on_side(self):
        """Context: launches inside go to the side stream."""
        return self.torch.cuda.stream(self.side)

    def mark_side(self):
        """An event behind everything queued on the side stream so far."""
        ev = self._event()
        ev.record(self.side)
        return ev

    def wait_on_main(self, ev):
        """The main stream waits for `ev` (mark_side); the event returns to the pool."""
        if ev is not None:
            self.torch.cuda.current_stream(self.eng.device).wait_event(ev)
            self._free.append(ev)


class _SegmentCuts:
    """Where a captured data-parallel step is cut: `buckets` are dist.plan_buckets' (start, end, ready_node_index) in firing
    order; pop_ready(i) returns the arena ranges that are complete once the backward sweep (descending node index) has
    finished node i - the same rule as dist.BucketReducer.node_done, so the captured step hands buckets over at exactly
    the points the eager step does."""

    def __init__(self, buckets):
        self.buckets = sorted(buckets, key=lambda b: -b[2])
        self.next = 0

    def pop_ready(self, node_index):
        out = []
        while self.next < len(self.buckets) and self.buckets[self.next][2] >= node_index:
            s, e, _ = self.buckets[self.next]
            out.append((s, e))
            self.next += 1
        return out

    def done(self):
        return self.next == len(self.buckets)


def run_segments(segments, arena, transport, lanes=None):
    """The replay loop of a segmented step: `segments` = [(launch, [(start, end), ...], side_launch | None)] (two-element
    entries: no side graph).

    launch() replays the segment's main graph M_k (stream-ordered, returns at once).  side_launch() replays its filter
    gradients W_k: on the side stream (`lanes`: _StreamLanes) behind everything queued on the main stream so far, i.e.
    beside M_(k+1); the main stream waits for W_(k-2) before M_k (W_(k-2)'s operands may be overwritten from M_k on: the
    capture held them that long).  Each arena range a segment completed then goes to the transport's asynchronous
    sum-all-reduce - with lanes ALWAYS issued from the side stream, behind M_k (a fork) and behind every side graph launched so
    far: the range holds gradients that W_k or the side graph of an earlier segment writes - which runs beside the next segment; at the end the main stream joins the side stream
    and the transport before whatever follows (Adam).  Device-agnostic: the CPU tests drive it over gloo with callables that
    write gradients and `lanes` = an in-line stand-in (or None: side launches run in line)."""
    import contextlib
    marks = []
    for k, seg in enumerate(segments):
        launch, ready = seg[0], seg[1]
        side_launch = seg[2] if len(seg) > 2 else None
        if lanes is not None and k >= 2:
            lanes.wait_on_main(marks[k - 2])
            marks[k - 2] = None
        launch()
        ctx = contextlib.nullcontext()
        mark = None
        if side_launch is not None:
            if lanes is not None:
                lanes.fork()
                with lanes.on_side():
                    side_launch()
                mark = lanes.mark_side()
                ctx = lanes.on_side()
            else:
                side_launch()
        marks.append(mark)
        if transport is not None and ready:
            if lanes is not None and side_launch is None:
                # a bucket may hold gradients that the side graph of an EARLIER segment writes (a bucket spans several
                # segments when the lanes cut between two hand-overs) and that graph may still be running: the hand-over
                # always goes through the side stream, behind this segment's main graph and every side graph so far
                lanes.fork()
                ctx = lanes.on_side()
            with ctx:
                for s, e in ready:
                    transport.allreduce_async(arena[s:e])
    if lanes is not None:
        for ev in marks:
            lanes.wait_on_main(ev)
    if transport is not None:
        transport.join()


class _Runtime:
    """Device state of one model: arenas, saved activations, gradient bookkeeping."""

    def __init__(self, model: Model):
        import torch
        from .ops import get_engine
        self.torch = torch
        self.model = model
        dev = int(os.environ.get("LOCAL_RANK", "0")) if torch.cuda.is_available() and torch.cuda.device_count() > 1 else 0
        self.eng = get_engine(dev)  # makes `dev` torch's current device (Engine.__init__)
        e = self.eng
        self._up2: Dict[tuple, bool] = {}   # up_sampling2d node -> fused with its convolution on this runtime (up2_on)
        self._bn_conv: Dict[tuple, bool] = {}        # (batch_normalization node, arithmetic) -> applied by its consumer convolution
        self._bnb: Dict[tuple, bool] = {}            # (batch_normalization node, batch, arithmetic) -> backward apply in the producer's dgrad
        self._act_planes: Dict[int, tuple] = {}      # id(symbolic tensor) -> (fp32 activation, its bf16 planes) of this step
        self._act_planes_use: Dict[int, bool] = {}   # id(conv node) -> its forward reads planes (geometry)
        self.w_train = e.zeros(max(model._n_train, ALIGN))
        self.g_train = e.zeros(max(model._n_train, ALIGN))
        self.adam_m = e.zeros(max(model._n_train, ALIGN))
        self.adam_v = e.zeros(max(model._n_train, ALIGN))
        self.w_frozen = e.zeros(max(model._n_frozen, ALIGN))
        rng = np.random.default_rng(model.seed)
        host_t = np.zeros(max(model._n_train, ALIGN), np.float32)
        host_n = np.zeros(max(model._n_frozen, ALIGN), np.float32)
        for p in model.params:
            a = init_array(p, rng).reshape(-1)
            (host_t if p.trainable else host_n)[p.offset:p.offset + p.size] = a
        self.w_train.copy_(torch.from_numpy(host_t))
        self.w_frozen.copy_(torch.from_numpy(host_n))
        self._pviews: Dict[int, object] = {}
        self._gviews: Dict[int, object] = {}
        self.values: Dict[int, object] = {}
        self._saved: Dict[int, dict] = {}
        self.bn_stats: Dict[int, tuple] = {}  # id(conv output tensor) -> (per-tile statistics, tiles)
        self.on_node_done = None
        self._pending = None          # the sweep's gradient table while a node's backward runs (take_pending)
        self.node_done_fires = None   # optional predicate: will on_node_done(index) hand gradients over (all-reduce / cut)?
        # weight planes of the matrix-pipe convolutions, prepared once per optimiser step (ensure_planes)
        self._planes_key = None
        self._planes_ptr: Dict[tuple, int] = {}
        self._planes_kind: Dict[tuple, int] = {}
        self._planes_arena = None
        self._planes_jobs = None
        self._planes_launch = (0, 0)
        self._planes_dirty = True
        self._w_version = 0   # bumped by weights_changed(): captured graphs compare it with the version their planes hold
        self._use_planes = os.environ.get("SG_PREPARED_PLANES", "1") != "0"

    # -- parameters ---------------------------------------------------------------------------------------
    def param(self, p: ParamSpec):
        v = self._pviews.get(id(p))
        if v is None:
            arena = self.w_train if p.trainable else self.w_frozen
            v = arena[p.offset:p.offset + p.size].view(p.shape)
            self._pviews[id(p)] = v
        return v

    def grad(self, p: ParamSpec):
        v = self._gviews.get(id(p))
        if v is None:
            v = self.g_train[p.offset:p.offset + p.size].view(p.shape)
            self._gviews[id(p)] = v
        return v

    def to_device(self, a):
        torch = self.torch
        if isinstance(a, torch.Tensor):
            return a if a.is_cuda and a.dtype == torch.float32 else a.to(self.eng.device, torch.float32)
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.eng.device)

    # -- prepared weight planes -----------------------------------------------------------------------------
    def weights_changed(self):
        """Call after writing the weight arena (Adam, set_weights, broadcast): the bf16 operand planes the convolution
        kernels read are re-derived from it before the next forward."""
        self._planes_dirty = True
        self._w_version += 1

    def prepare_into(self, arena, jobs, launch):
        """Rebuild the weight planes of a SNAPSHOT (arena, job table, (jobs, blocks)) taken at a hipGraph capture: the graph's
        kernel nodes carry that arena's addresses whatever batch size / mode the runtime's current planes describe."""
        import ctypes as C
        if launch[0]:
            _lib.check(self.eng.lib.sg_prepare_planes(self.eng.h, self.eng.stream, C.c_void_p(self.w_train.data_ptr()),
                                                      C.c_void_p(arena.data_ptr()), C.c_void_p(jobs.data_ptr()), launch[0],
                                                      launch[1]), "sg_prepare_planes")

    def bnb_on(self, bn_node, x) -> bool:
        """Does THIS runtime leave BatchNormalization `bn_node`'s backward apply to the dgrad of the pointwise convolution that
        produced its input (Model._fuse: bnb_to)?  fp32 storage, a launch the wide pointwise kernel takes at this batch (asked of
        the library), SG_BN_PW=1.  OFF by default: built, bit-identical to the unfused pair, and a LOSS in the step - the 56
        bn_bwd_apply launches it removes (47 us each in the profile) ran beside the side stream's filter gradients for free,
        while the 25 us it adds to every dgrad sit on the matrix pipe, which is what bounds the step (71.13 / 71.43 ms without,
        71.58 / 71.69 with, alternating runs on one box: gpurun_out/r5x; stand-alone 123 us against 22.5 + 96.8)."""
        key = (id(bn_node), int(x.shape[0]), int(self.eng.lib.sg_get_conv_x6()))
        got = self._bnb.get(key)
        if got is None:
            p = bn_node.bnb_to
            got = False
            if p is not None and os.environ.get("SG_BN_PW", "0") == "1" and self.model.compute_dtype == "float32" and x.dtype == self.torch.float32:
                n, h, w, c = x.shape
                cin = p.inputs[0].shape[-1]
                got = bool(self.eng.conv2d_dgrad_bnb_ok(self.eng.conv_desc((n, h, w, cin), c, 1, 1, 1, 1, "same")))
            self._bnb[key] = got
        return got

    def bn_conv_on(self, bn_node) -> bool:
        """Does THIS runtime apply BatchNormalization `bn_node` in the loaders of its consumer convolution (Model._fuse: defer_conv /
        bn_src)?  fp32 storage, a launch geometry the thin 1x1 / patch kernels cover (asked of the library), SG_BN_CONV != 0."""
        key = (id(bn_node), int(self.eng.lib.sg_get_conv_x6()))
        got = self._bn_conv.get(key)
        if got is None:
            c = bn_node.defer_conv
            got = False
            if c is not None and os.environ.get("SG_BN_CONV", "1") != "0" and self.model.compute_dtype == "float32":
                _, h, w, cin = c.inputs[0].shape
                d = self.eng.conv_desc((1, h, w, cin), c.filters, c.k, c.k, c.stride, c.dilation, c.padding)
                got = bool(self.eng.conv2d_bn_in_ok(d))
            self._bn_conv[key] = got
        return got

    def planes_in_on(self) -> bool:
        """fp32 storage with the six-pass arithmetic, SG_ACT_PLANES != 0: activation planes are made once per tensor and step."""
        return (self.model.compute_dtype == "float32" and os.environ.get("SG_ACT_PLANES", "1") != "0"
                and int(self.eng.lib.sg_get_conv_x6()) == 1)

    def act_planes(self, node, x, d, make=True):
        """The three bf16 planes of activation `x` (Engine.split_planes), made at most once per tensor and training step - every
        consumer whose forward reads planes (csrc/conv_x6w.h) shares them and its filter gradient takes them again (kept until
        the step's backward sweep ends: release()).  None when `node`'s launches have no use for them."""
        if not self.planes_in_on():
            return None
        key = id(node.inputs[0])
        got = self._act_planes.get(key)
        if got is not None and got[0] is x:
            return got[1]
        if not make:
            return None
        use = self._act_planes_use.get(id(node))
        if use is None:   # decided once per node from its geometry
            use = bool(self.eng.conv2d_planes_in(d, False))
            self._act_planes_use[id(node)] = use
        if not use:
            return None
        pl = self.eng.split_planes(x)
        self._act_planes[key] = (x, pl)
        return pl

    def up2_on(self, up_node) -> bool:
        """Does THIS runtime run the pair `up_node` -> 3x3 convolution (Model._fuse: fused_into / up_src) on the fused kernels?
        fp32 storage, the geometry csrc/conv_x6p.h covers (asked of the library), SG_UP2_FUSE != 0.  Decided once per pair:
        the up-sampling node (identity then) and the convolution must agree."""
        key = (id(up_node), int(self.eng.lib.sg_get_conv_x6()))   # (the arithmetic switch may change between steps: tests do)
        got = self._up2.get(key)
        if got is None:
            c = up_node.fused_into
            got = False
            if c is not None and os.environ.get("SG_UP2_FUSE", "1") != "0" and self.model.compute_dtype == "float32":
                _, h, w, cin = c.inputs[0].shape
                d = self.eng.conv_desc((1, h, w, cin), c.filters, c.k, c.k, c.stride, c.dilation, c.padding)
                got = bool(self.eng.conv2d_up2_ok(d))
            self._up2[key] = got
        return got

    def plane_kind(self, node, tag) -> int:
        """Which kernel family `node`'s launch `tag` takes (sg_conv2d_planes_job's kind; 0: none of the prepared-plane kernels)."""
        return self._planes_kind.get((id(node), tag), 0) if self._planes_ptr else 0

    def planes(self, node, tag):
        """Device address of the prepared planes of `node`'s launch `tag` ("f" / "d"), or None (the launch then converts its
        kernel itself, in the workspace)."""
        return self._planes_ptr.get((id(node), tag))

    def ensure_planes(self, batch: int, training: bool):
        """(Re)build the job table when the batch, the storage dtype or the arithmetic mode changed; convert every kernel of
        the model in ONE launch when the weights changed since the last conversion (sg_prepare_planes)."""
        if not self._use_planes:
            return
        import ctypes as C
        e, m, torch = self.eng, self.model, self.torch
        bwd = training or m.optimizer is not None
        key = (int(batch), m.compute_dtype, e.lib.sg_get_conv_x6(), bwd)
        if key != self._planes_key:
            dt = _lib.SG_BF16 if m.compute_dtype == "bfloat16" else _lib.SG_F32
            jobs, ptr_of, kinds, off, blocks = [], {}, {}, 0, 0
            for n in m.nodes:
                sites = getattr(n, "plane_sites", None)
                if sites is None:
                    continue
                for tag, wspec, d, dgrad, phase, head in sites(self, int(batch)):
                    if (phase == "bwd" and not bwd) or not wspec.trainable:
                        continue
                    job, nbytes = _lib.PlanesJob(), C.c_size_t(0)
                    _lib.check(e.lib.sg_conv2d_planes_job(e.h, dt | (_lib.SG_HEAD_F32 if (head and dt == _lib.SG_BF16) else 0),
                                                          C.byref(d), dgrad, C.byref(job), C.byref(nbytes)), "sg_conv2d_planes_job")
                    if job.kind == 0:
                        continue
                    job.w_off, job.out_off, job.block0 = wspec.offset, off, blocks
                    ptr_of[(id(n), tag)] = off
                    kinds[(id(n), tag)] = int(job.kind)
                    off += (nbytes.value + 255) // 256 * 256
                    blocks += job.nblocks
                    jobs.append(job)
            self._planes_key = key
            self._planes_kind = kinds   # 1: slab kernels (conv_x6 / conv_b16), 2: patch kernel, 3: wide pointwise kernel
            self._planes_ptr = {}
            self._planes_arena = self._planes_jobs = None
            self._planes_launch = (len(jobs), blocks)
            if jobs:
                self._planes_arena = torch.empty(off, dtype=torch.uint8, device=e.device)
                arr = (_lib.PlanesJob * len(jobs))(*jobs)
                host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                self._planes_jobs = host.to(e.device)
                base = self._planes_arena.data_ptr()
                self._planes_ptr = {k: base + o for k, o in ptr_of.items()}
            self._planes_dirty = True
        if self._planes_dirty and self._planes_launch[0]:
            _lib.check(e.lib.sg_prepare_planes(e.h, e.stream, C.c_void_p(self.w_train.data_ptr()),
                                               C.c_void_p(self._planes_arena.data_ptr()), C.c_void_p(self._planes_jobs.data_ptr()),
                                               self._planes_launch[0], self._planes_launch[1]), "sg_prepare_planes")
        self._planes_dirty = False

    # -- tape ---------------------------------------------------------------------------------------------
    def save(self, node, **kw):
        self._saved.setdefault(id(node), {}).update(kw)

    def saved(self, node):
        return self._saved[id(node)]

    def needs_grad(self, sym: KTensor) -> bool:
        return sym.node is not None

    def take_pending(self, sym: KTensor, owned_only: bool = False):
        """Inside a node's backward: the gradient collected so far for tensor `sym` (the node's input, or the tensor its input
        is an identity of), REMOVED from the sweep's table; None if there is none.  The node must return an input gradient
        that includes it - the table then receives the complete gradient and no separate add runs.  owned_only: only a
        buffer the sweep may write in place (not a gradient shared with another tensor) - for nodes that accumulate into it."""
        if self._pending is None or os.environ.get("SG_GRAD_ACC", "1") != "1":
            return None
        cur = self._pending.get(id(sym))
        if cur is None or (owned_only and not cur[1]):
            return None
        del self._pending[id(sym)]
        return cur[0]

    def put_back(self, sym: KTensor, t):
        """Undo an owned take_pending the node could not use."""
        self._pending[id(sym)] = [t, True]

    def shared(self, t):
        return _Shared(t)

    def release(self):
        self.values.clear()
        self._saved.clear()
        self.bn_stats.clear()
        self._act_planes.clear()

    def forward(self, x, training: bool):
        m = self.model
        exp = m.inputs[0].shape[1:]
        if tuple(x.shape[1:]) != tuple(exp):
            raise ValueError(f"input shape {tuple(x.shape)} does not match the model's {(None,) + tuple(exp)}")
        x = x.contiguous()
        self.ensure_planes(x.shape[0], training)
        if m.compute_dtype == "bfloat16":  # activations are bf16 from the first layer on (sg_cast, round to nearest even)
            x = self.eng.cast(x, self.torch.bfloat16)
        self.values = {id(m.inputs[0]): x}
        self._act_planes.clear()   # (a step's activation planes: made in this sweep, read again by its filter gradients)
        if training:
            for n in m.nodes:
                xs = [self.values[id(t)] for t in n.inputs]
                self.values[id(n.output)] = n.forward(self, xs, training)
            return self._as_f32(self.values[id(m.outputs[0])])
        # inference: an activation is dropped as soon as its last consumer has run (the allocator - or the
        # hipGraph's private pool under capture - reuses the block), so peak memory is the live set, not the sum
        last_use = m._last_use()
        for i, n in enumerate(m.nodes):
            xs = [self.values[id(t)] for t in n.inputs]
            self.values[id(n.output)] = n.forward(self, xs, False)
            for t in n.inputs:
                if last_use.get(id(t)) == i:
                    self.values.pop(id(t), None)
            del xs
        return self._as_f32(self.values[id(m.outputs[0])])

    def _as_f32(self, y):
        """The model's output leaves the engine as fp32 (the softmax head already is; a bf16 output of a head-less graph
        is widened)."""
        return y if y.dtype == self.torch.float32 else self.eng.cast(y, self.torch.float32)

    def backward(self, dout):
        """Reverse sweep.  Gradients of a tensor with several consumers are summed with sg_add_n; a gradient
        handed out as `_Shared` (pass-through of an Add) is never written in place."""
        m = self.model
        yout = self.values.get(id(m.outputs[0]))
        if yout is not None and yout.dtype != dout.dtype:  # a head-less bf16 graph: the fp32 loss gradient enters in bf16
            dout = self.eng.cast(dout, yout.dtype)
        grads: Dict[int, list] = {id(m.outputs[0]): [dout, True]}
        e = self.eng
        hook = self.on_node_done
        for n in reversed(m.nodes):
            slot = grads.pop(id(n.output), None)
            if slot is None:  # output does not influence the loss
                if hook is not None:
                    if self.node_done_fires is None or self.node_done_fires(n.index):
                        e.join_side()
                    hook(n.index)
                continue
            dy = slot[0]
            xs = [self.values[id(t)] for t in n.inputs]
            y = self.values[id(n.output)]
            self._pending = grads
            dxs = n.backward(self, xs, y, dy)
            self._pending = None
            for sym, g in zip(n.inputs, dxs):
                if g is None or sym.node is None:
                    continue
                fresh = True
                if isinstance(g, _Shared):
                    g, fresh = g.t, False
                elif g is dy:
                    fresh = False
                cur = grads.get(id(sym))
                if cur is None:
                    grads[id(sym)] = [g, fresh]

                elif cur[1]:
                    e.add_n([cur[0], g], out=cur[0])
                elif fresh:
                    e.add_n([cur[0], g], out=g)
                    grads[id(sym)] = [g, True]
                else:
                    grads[id(sym)] = [e.add_n([cur[0], g]), True]
            # this node's saved state and output gradient are dead now
            self._saved.pop(id(n), None)
            if hook is not None:
                if self.node_done_fires is None or self.node_done_fires(n.index):
                    e.join_side()  # a bucket's all-reduce reads filter gradients the side stream may still be writing
                hook(n.index)  # data-parallel: launches the all-reduce of every gradient bucket now complete
        self._act_planes.clear()   # (the side stream's launches hold their own references)
        e.join_side()  # Adam, get_gradients() and the loss-scaling checks read the gradients on the main stream
