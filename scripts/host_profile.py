#!/usr/bin/env python3
"""cProfile of the Python host side of five eager DeepLabv3+ 512x512 bs16 training steps (after three warm-up steps): where the
60 ms of enqueue per step go.  Read with care: a launch call's time includes blocking on a full HIP queue when the host is ahead
of the GPU (add_n / BatchNormalization calls show ~100 us each for that reason, not for their own work); the pure interpreter
overhead of a step is ~20 ms.  Use: python scripts/host_profile.py"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import torch
from building_detection_amd import zoo
from building_detection_amd.data import synthetic_batch
from building_detection_amd.losses import edge_focal_loss, PA, IoU, MIoU, F1_score
m = zoo.BUILDERS["v3plus"]((512, 512, 3), 2, aspp_pool=32)
m.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score])
x, y = synthetic_batch(16, 512, 512, seed=1)
xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
for _ in range(3):
    m.train_on_batch(xd, yd, return_device_scalars=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.train_on_batch(xd, yd, return_device_scalars=True)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
