#!/usr/bin/env python3
"""BASELINE config 5: 5-model ensemble inference on 1024x1024 tiles, batch 8, hipGraph-captured forward.
Reports per-model and ensemble tiles/s for eager launches vs one graph replay per model, checks that the graph
replay is bit-identical to the eager forward, and runs the 3-of-5 vote on the argmax masks."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd import zoo  # noqa: E402
from building_detection_amd.ops import get_engine  # noqa: E402

size = int(os.environ.get("SIZE", "1024"))
batch = int(os.environ.get("BATCH", "8"))
iters = int(os.environ.get("ITERS", "3"))
DTYPE = os.environ.get("DTYPE", "f32")  # "bf16": activations stored as bf16 (mixed_bfloat16 policy)
if DTYPE == "bf16":
    from building_detection_amd import mixed_precision
    mixed_precision.set_global_policy("mixed_bfloat16")
eng = get_engine(0)
g = torch.Generator().manual_seed(1103)
x = (torch.randint(0, 256, (batch, size, size, 3), generator=g).float() / 127.5 - 1).cuda()
fwd_gflop = {"res34": 1996.22, "hrnet": 749.91, "v3plus": 808.47, "scse": 1627.64, "bam": 607.05}  # per 1024^2 tile, BASELINE.md


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


out = {"config": f"5-model ensemble inference {size}x{size} bs={batch} {'bf16 storage' if DTYPE == 'bf16' else 'fp32'}", "models": {}}
masks = []
tot_e = tot_g = 0.0
for name in ("res34", "hrnet", "v3plus", "scse", "bam"):
    m = zoo.BUILDERS[name]((size, size, 3))
    y_eager = m.predict_device(x).clone()
    t_e = timeit(lambda: m.predict_device(x))
    gp = m.capture_predict(batch)
    y_graph = gp(x)
    same = bool(torch.equal(y_graph, y_eager))
    t_g = timeit(lambda: gp(x))
    masks.append(((y_graph[..., 1] > y_graph[..., 0]).to(torch.uint8) * 255).contiguous())
    fl = fwd_gflop[name] * batch * (size / 1024.0) ** 2 / 1e3
    out["models"][name] = {"eager_ms": round(t_e * 1e3, 2), "graph_ms": round(t_g * 1e3, 2), "graph_equals_eager": same,
                           "tflops_graph": round(fl / t_g, 1), "launch_nodes": len(m.nodes)}
    tot_e += t_e
    tot_g += t_g
    print(name, out["models"][name], flush=True)
    del m, gp, y_eager, y_graph
    torch.cuda.empty_cache()
vote = eng.vote_ge([mk.view(-1) for mk in masks], 3)
out["ensemble_tiles_per_s_eager"] = round(batch / tot_e, 3)
out["ensemble_tiles_per_s_graph"] = round(batch / tot_g, 3)
out["vote_positive_fraction"] = round(float((vote == 255).float().mean().item()), 4)
out["peak_mem_GiB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
print(json.dumps(out))
