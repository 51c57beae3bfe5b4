import os, sys, torch
sys.path.insert(0, "/root/repo")
from building_detection_amd.ops import get_engine
e = get_engine(0)
g = torch.Generator(device="cpu").manual_seed(0)
BF = torch.bfloat16
def timed(fn, iters=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for name, h, cin, cout, k, dil in [("pw728", 32, 728, 728, 1, 1), ("aspp18", 32, 2048, 256, 3, 18), ("dec304", 128, 304, 256, 3, 1)]:
    x = (torch.rand(16, h, h, cin, generator=g) * 2 - 1).cuda().to(BF)
    w = ((torch.rand(k, k, cin, cout, generator=g) * 2 - 1) * 0.02).cuda()
    d = e.conv_desc(tuple(x.shape), cout, k, k, 1, dil, "same")
    y = e.conv2d_fwd(x, w, None, desc=d)
    t = timed(lambda: e.conv2d_fwd(x, w, None, desc=d, out=y))
    print(f"{name}: {t:8.1f} us", flush=True)
