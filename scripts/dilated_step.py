#!/usr/bin/env python3
"""The roofline kernel set exactly as the round-5 training step launches it: the six dilated 3x3 convolutions of DeepLabv3+
(3 ASPP 2048 -> 256 on ONE shared input, 3 SK 256 -> 256; 32 x 32 maps, batch 16), forward + dgrad + filter gradient, with the
activation planes made once per tensor (Engine.split_planes: the ASPP input once for its three branches, every layer's dy once for
its dgrad and its planes-in filter gradient) - what layers._ConvNode does through _Runtime.act_planes.  scripts/dilated_bench.py
launches every convolution on its own (one split per launch); this script is what the PMC passes of round 5 profile
(scripts/gpu_ci.sh <tag> pmc5): ITERS=1 runs the set exactly twice (warm-up + 1), the parsers halve the sums.
Prints the set's time from HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
N = int(os.environ.get("BATCH", "16"))
iters = int(os.environ.get("ITERS", "5"))
g = torch.Generator(device="cpu").manual_seed(0)
c5 = (torch.rand(N, 32, 32, 2048, generator=g) * 2 - 1).cuda()
xs = (torch.rand(N, 32, 32, 256, generator=g) * 2 - 1).cuda()
layers = []
for cin, x in ((2048, c5), (256, xs)):
    for dil in (6, 12, 18):
        w = ((torch.rand(3, 3, cin, 256, generator=g) * 2 - 1) * 0.02).cuda()
        d = e.conv_desc(tuple(x.shape), 256, 3, 3, 1, dil, "same")
        layers.append(dict(x=x, w=w, b=torch.zeros(256).cuda(), d=d, dy=(torch.rand(N, 32, 32, 256, generator=g) * 2 - 1).cuda(),
                           y=e.empty(N, 32, 32, 256), dx=e.empty(*x.shape), dw=e.empty(3, 3, cin, 256),
                           planes_f=e.conv2d_planes_in(d, False), planes_w=e.conv2d_wgrad_planes_ok(d) and e.conv2d_planes_in(d, False)))


def step():
    planes = {}
    for L in layers:   # forward: one split per TENSOR
        xp = None
        if L["planes_f"]:
            xp = planes.get(id(L["x"]))
            if xp is None:
                xp = planes[id(L["x"])] = e.split_planes(L["x"])
        L["xp"] = xp
        e.conv2d_fwd(L["x"], L["w"], L["b"], desc=L["d"], out=L["y"], want_stats=True, x_planes=xp)
    for L in reversed(layers):   # backward: dy's planes serve dgrad and filter gradient
        dzp = e.split_planes(L["dy"]) if (L["planes_w"] or e.conv2d_planes_in(L["d"], True)) else None
        e.conv2d_dgrad(L["dy"], L["w"], L["d"], out=L["dx"], dy_planes=dzp)
        if L["planes_w"]:
            e.conv2d_wgrad_planes(L["xp"], dzp, L["d"], dw=L["dw"])
        else:
            e.conv2d_wgrad(L["x"], L["dy"], L["d"], want_bias=False, dw=L["dw"])


step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters):
    step()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / iters
tf = 97.84e-3 * N
print(f"dilated set as the step runs it: {ms:.3f} ms per step = {tf / ms * 1e3:.1f} TFLOP/s nominal = {tf / ms * 1e3 / (2500 / 6):.3f} of the bf16 pipe / 6")
