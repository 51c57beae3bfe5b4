#!/usr/bin/env python3
"""Phase durations inside conv_x6p_kernel (SG_X6P_ABLATE=4 makes workgroup 0..63 write, per tile, the shader-clock
deltas top->barrier, barrier->loads landed, split+LDS write, K loop, exchange+epilogue into y instead of the result)."""
import os
import sys

import torch

os.environ["SG_X6P_ABLATE"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from building_detection_amd.ops import get_engine  # noqa: E402

e = get_engine(0)
g = torch.Generator(device="cpu").manual_seed(0)
for h, cin, cout in [(512, 64, 64), (512, 32, 32), (512, 64, 32)]:
    x = (torch.rand(16, h, h, cin, generator=g) * 2 - 1).cuda()
    w = ((torch.rand(3, 3, cin, cout, generator=g) * 2 - 1) * 0.05).cuda()
    d = e.conv_desc(tuple(x.shape), cout, 3, 3, 1, 1, "same")
    y = e.conv2d_fwd(x, w, None, desc=d)
    y.zero_()
    torch.cuda.synchronize()
    e.conv2d_fwd(x, w, None, desc=d, out=y)
    torch.cuda.synchronize()
    t = y.view(-1)[: 64 * 16 * 12].view(64, 16, 12).cpu()
    names = ["top->barrier", "loads landed", "split+write+bar", "K loop", "exchange+epilogue"]
    print(f"{h}x{h} {cin}->{cout}: shader-clock cycles per phase (mean over 64 workgroups), tiles 0, 1, 2, 8, 15")
    for k, nm in enumerate(names):
        print(f"  {nm:18s}", " ".join(f"{t[:, i, k].mean().item():9.0f}" for i in (0, 1, 2, 8, 15)))
    # workgroups that share a CU (same XCC / SE / SH / CU id): K-loop intervals of tiles 4..6, relative to the first
    by_cu = {}
    for b in range(64):
        by_cu.setdefault(int(t[b, 4, 6]), []).append(b)
    shown = 0
    for cu, bs in by_cu.items():
        if len(bs) >= 2 and shown < 3:
            base = min(int(t[b, 4, 5]) for b in bs)
            print(f"  CU key {cu:#x}: " + " | ".join(f"WG{b}: " + " ".join(f"[{(int(t[b, i, 5]) - base) % (1 << 24)}..{(int(t[b, i, 7]) - base) % (1 << 24)}]" for i in (4, 5, 6)) for b in bs))
            shown += 1
    print(f"  CUs seen: {len(by_cu)} for 64 workgroups")
    clk = (t[:, 1:, :5].sum(-1) / t[:, 1:, 8].clamp(min=1)).median().item() * 100.0  # MHz
    print(f"  shader clock inside the kernel: {clk:.0f} MHz (cycles per tile / 10 ns ticks per tile, median)")
    tot = t[:, 1:, :5].sum(-1).mean().item()
    print(f"  per tile (steady) {tot:.0f} cycles; start stamps of WG0..3 tile 1: {[int(t[b, 1, 5]) for b in range(4)]}")
