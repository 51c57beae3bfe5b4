"""Symbolic graph of the engine: tensors, nodes and the parameter arenas.

A model is a static list of nodes in creation (= topological) order, exactly what the reference's functional
tf.keras builders produce (predict_model/*.py).  Building the graph needs no GPU; `runtime.py` executes it.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


class KTensor:
    """Symbolic tensor: static shape with a free batch dimension (None), produced by `node`."""

    __slots__ = ("shape", "node", "consumers", "name")

    def __init__(self, shape, node=None, name=None):
        self.shape = tuple(shape)
        self.node = node
        self.consumers: List["Node"] = []
        self.name = name

    @property
    def channels(self):
        return self.shape[-1]

    def __repr__(self):
        return f"KTensor{self.shape}<{self.node.name if self.node else 'input'}>"


class ParamSpec:
    __slots__ = ("name", "shape", "init", "trainable", "kind", "offset", "size", "fan")

    def __init__(self, name, shape, init, trainable=True, kind="kernel", fan=None):
        self.name, self.shape, self.init, self.trainable, self.kind = name, tuple(shape), init, trainable, kind
        self.size = int(np.prod(shape))
        self.offset = -1
        self.fan = fan


class Node:
    """One layer application.  Subclasses (layers.py) implement shape inference and fwd/bwd launchers."""

    op = "node"
    _counter: Dict[str, int] = {}
    _serial_counter = 0

    def __init__(self, name: Optional[str] = None):
        idx = Node._counter.get(self.op, 0)
        Node._counter[self.op] = idx + 1
        self.name = name or (self.op if idx == 0 else f"{self.op}_{idx}")
        Node._serial_counter += 1
        self._serial = Node._serial_counter
        self.inputs: List[KTensor] = []
        self.output: Optional[KTensor] = None
        self.params: List[ParamSpec] = []
        self.index = -1

    # graph construction ---------------------------------------------------------------------------------
    def connect(self, inputs: Sequence[KTensor], out_shape) -> KTensor:
        self.inputs = list(inputs)
        for t in self.inputs:
            t.consumers.append(self)
        self.output = KTensor(out_shape, self)
        return self.output

    def add_param(self, suffix, shape, init, trainable=True, kind="kernel", fan=None) -> ParamSpec:
        p = ParamSpec(f"{self.name}/{suffix}", shape, init, trainable, kind, fan)
        self.params.append(p)
        return p

    # execution (overridden) -----------------------------------------------------------------------------
    def forward(self, rt, xs, training):  # -> output tensor ; may stash state in rt.saved[self]
        raise NotImplementedError

    def backward(self, rt, xs, y, dy):  # -> list of input grads (None where not needed)
        raise NotImplementedError

    def flops(self, batch: int) -> int:
        """Nominal forward MACs*2 (SURVEY §8d convention); 0 for bandwidth ops."""
        return 0


def reset_names():
    Node._counter.clear()


# ------------------------------------------------------------------------------------------- initialisers
def fans(shape) -> Tuple[int, int]:
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def init_array(spec: ParamSpec, rng: np.random.Generator) -> np.ndarray:
    """Keras default initialisers restated with the engine's own seeded RNG."""
    shape = spec.shape
    if spec.init == "zeros":
        return np.zeros(shape, np.float32)
    if spec.init == "ones":
        return np.ones(shape, np.float32)
    fan_in, fan_out = spec.fan if spec.fan else fans(shape)
    if spec.init == "glorot_uniform":
        limit = math.sqrt(6.0 / (fan_in + fan_out))
        return rng.uniform(-limit, limit, size=shape).astype(np.float32)
    if spec.init == "he_normal":
        std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
        out = rng.normal(0.0, std, size=shape)
        bad = np.abs(out) > 2 * std
        while bad.any():  # truncated normal by resampling, as TF does
            out[bad] = rng.normal(0.0, std, size=int(bad.sum()))
            bad = np.abs(out) > 2 * std
        return out.astype(np.float32)
    raise ValueError(f"unknown initializer {spec.init!r}")


def collect_nodes(outputs: Sequence[KTensor]) -> List[Node]:
    """All nodes the outputs depend on, in creation order (creation order is topological)."""
    seen, order = set(), []
    stack = [t.node for t in outputs if t.node is not None]
    while stack:
        n = stack.pop()
        if id(n) in seen:
            continue
        seen.add(id(n))
        order.append(n)
        for t in n.inputs:
            if t.node is not None:
                stack.append(t.node)
    order.sort(key=lambda n: n._serial)
    return order
