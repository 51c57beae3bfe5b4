#!/usr/bin/env python3
"""Per-node time census of one training step (HIP events around every node's forward and backward), grouped
by (op, output shape).  For the bandwidth-bound nodes it also prints the effective GB/s against the minimum
traffic (inputs + output once for forward; dy + inputs + dx for backward), which tells which kernels are far
from the HBM roofline.  Use: python scripts/node_census.py [model] [batch] [size]."""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd import zoo  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402
from building_detection_amd.losses import edge_focal_loss  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "v3plus"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
size = int(sys.argv[3]) if len(sys.argv) > 3 else 512
if os.environ.get("DTYPE", "f32") == "bf16":   # DTYPE=bf16: the mixed_bfloat16 policy (bf16 storage of the activations)
    from building_detection_amd import mixed_precision as MP  # noqa: E402
    MP.set_global_policy("mixed_bfloat16")
model = zoo.BUILDERS[name]((size, size, 3))
model.compile(optimizer="adam", loss=edge_focal_loss, metrics=[])
x, y = synthetic_batch(N, size, size, seed=1)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
for _ in range(2):
    model.train_on_batch(xd, yd)
rt = model._runtime()

fwd_ev, bwd_ev = {}, {}
for n in model.nodes:
    of, ob = n.forward, n.backward

    def f(rt_, xs, training, _of=of, _n=n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = _of(rt_, xs, training)
        b.record()
        fwd_ev.setdefault(_n.index, []).append((a, b))
        return r

    def g(rt_, xs, y_, dy, _ob=ob, _n=n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = _ob(rt_, xs, y_, dy)
        b.record()
        bwd_ev.setdefault(_n.index, []).append((a, b))
        return r

    n.forward, n.backward = f, g

REP = 3
for _ in range(REP):
    model.train_on_batch(xd, yd)
torch.cuda.synchronize()


def ms(evs):
    return sum(a.elapsed_time(b) for a, b in evs) / REP


groups = OrderedDict()
for n in model.nodes:
    shp = tuple(n.output.shape[1:])
    ins = tuple(tuple(t.shape[1:]) for t in n.inputs)
    key = (n.op, ins, shp)
    g_ = groups.setdefault(key, [0, 0.0, 0.0])
    g_[0] += 1
    g_[1] += ms(fwd_ev.get(n.index, []))
    g_[2] += ms(bwd_ev.get(n.index, []))


def numel(s):
    r = 1
    for d in s:
        r *= d
    return r


tf = sum(v[1] for v in groups.values())
tb = sum(v[2] for v in groups.values())
print(f"{name} bs{N} {size}: node forward {tf:.1f} ms, backward {tb:.1f} ms per step (event-bracketed; excludes loss/Adam)")
by_op = OrderedDict()
for (op, ins, shp), (c, a, b) in groups.items():
    o = by_op.setdefault(op, [0, 0.0, 0.0])
    o[0] += c
    o[1] += a
    o[2] += b
print("-- by op")
for op, (c, a, b) in sorted(by_op.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{op:24s} x{c:4d}  fwd {a:7.2f} ms  bwd {b:7.2f} ms")
print("-- top groups (GB/s = minimum traffic / time; meaningful for the bandwidth-bound ops only)")
rows = sorted(groups.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:60]
for (op, ins, shp), (c, a, b) in rows:
    bi = sum(numel(s) for s in ins) * 4 * N
    bo = numel(shp) * 4 * N
    fbytes, bbytes = bi + bo, bo + 2 * bi
    gf = fbytes * c / a / 1e6 if a > 0 else 0
    gb = bbytes * c / b / 1e6 if b > 0 else 0
    print(f"{op:22s} {str(ins[0] if ins else ''):20s}->{str(shp):20s} x{c:3d} fwd {a:6.2f} ms {gf:6.0f} GB/s | bwd {b:6.2f} ms {gb:6.0f} GB/s")
