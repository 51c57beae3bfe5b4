#!/usr/bin/env python3
"""Which node makes an inference result depend on the batch it travels in?  Runs the graph node by node on a
batch of 16 tiles and on tiles 4..8 alone and prints every node whose output differs (first = culprit)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd import zoo  # noqa: E402
from building_detection_amd.data import synthetic_batch  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "v3plus"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
model = zoo.BUILDERS[name]((size, size, 3))
rt = model._runtime()
x, _ = synthetic_batch(16, size, size, seed=1103)
xd = torch.from_numpy(x).cuda()


def run(xin):
    vals = {id(model.inputs[0]): xin.contiguous()}
    for n in model.nodes:
        vals[id(n.output)] = n.forward(rt, [vals[id(t)] for t in n.inputs], False)
    return vals


a = run(xd)
a = {k: (v[4:8].clone() if torch.is_tensor(v) and v.shape[0] == 16 else v) for k, v in a.items()}
torch.cuda.empty_cache()
b = run(xd[4:8])
bad = 0
for n in model.nodes:
    va, vb = a[id(n.output)], b[id(n.output)]
    if not torch.is_tensor(va) or va.shape != vb.shape:
        continue
    if not torch.equal(va, vb):
        dirty_in = any(torch.is_tensor(a[id(t)]) and a[id(t)].shape == b[id(t)].shape and not torch.equal(a[id(t)], b[id(t)])
                       for t in n.inputs)
        print(f"{n.index:4d} {n.op:22s} {n.name:40s} max diff {(va - vb).abs().max().item():.3e}  inputs {'differ' if dirty_in else 'IDENTICAL  <-- source'}")
        bad += 1
        if bad > 40:
            break
print("nodes that differ:", bad)
