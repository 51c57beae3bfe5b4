"""BASELINE.json configs at their FULL sizes.

config 1  Res34-UNet 256x256x3 bs=2, fwd+bwd: compared directly with the CPU oracle (it is small enough).
config 2  DeepLabv3+ 512x512 bs=16 fp32 and config 4 (SCSE-UNet, DeepLab-BAM 512x512 bs=16): the oracle needs
          minutes per step at that size, so the full-size run is pinned by (a) the oracle on ONE 512x512 tile
          (inference is per-tile independent: BatchNorm uses moving statistics), (b) size-independent
          properties: bit-exact batch-slice invariance of inference, bit-exact run-to-run determinism of a
          training step (no float atomics anywhere on the path), and a directional-derivative check of the
          whole backward pass: (L(w + h d) - L(w - h d)) / 2h  ~=  <grad, d>.
config 5  (5-model ensemble 1024x1024 bs=8, hipGraph) is exercised by scripts/bench_infer.py and, at test size,
          by tests/test_pipeline_gpu.py::test_hipgraph_capture_is_bit_identical_to_eager.
"""
import numpy as np
import pytest
import torch

from oracle import models as M

pytestmark = pytest.mark.gpu


def _compiled(model):
    from building_detection_amd.losses import edge_focal_loss, PA, IoU, MIoU, F1_score
    model.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score])
    return model


def test_config1_res34_256_bs2_train_step_vs_oracle(engine):
    from building_detection_amd import zoo
    from building_detection_amd.data import synthetic_batch
    model = _compiled(zoo.ResNetFamily((256, 256, 3)).run_model("res34"))
    x, y = synthetic_batch(2, 256, 256, seed=1103)
    ws0 = model.get_weights()
    logs = model.train_on_batch(x, y)
    gg = model.get_gradients()
    P = M.Params(weights=ws0)
    p = M.res34_unet(P, torch.from_numpy(x), training=True)
    loss = M.loss_fn("edge_focal_loss", torch.from_numpy(y), p)
    loss.backward()
    assert abs(logs["loss"] - loss.item()) <= 2e-5 * abs(loss.item()), (logs["loss"], loss.item())
    gc = [t.grad.numpy() for t in P.trainable_tensors()]
    num = sum(float(np.square(a.astype(np.float64) - b).sum()) for a, b in zip(gg, gc))
    den = sum(float(np.square(b.astype(np.float64)).sum()) for b in gc)
    rel = (num / den) ** 0.5
    print(f"config 1: loss gpu {logs['loss']:.6f} cpu {loss.item():.6f}; global rel-L2 gradient difference {rel:.2e}")
    assert rel <= 2e-2  # fp32 ReLU-flip noise (see test_models_gpu.py); a wrong term would be O(1)
    cm = M.metrics_from_counts(*M.confusion(torch.from_numpy(y), p.detach()))
    assert abs(logs["MIoU"] - cm["MIoU"]) <= 2e-3


FULL = [("v3plus", "deeplab_v3plus"), ("scse", "scse_unet"), ("bam", "deeplab_v3plus_bam")]


@pytest.mark.parametrize("name,oracle_fn", FULL, ids=[f[0] for f in FULL])
def test_full_size_512_bs16(engine, name, oracle_fn):
    from building_detection_amd import zoo
    from building_detection_amd.data import synthetic_batch
    torch.cuda.empty_cache()
    model = _compiled(zoo.BUILDERS[name]((512, 512, 3)))
    x, y = synthetic_batch(16, 512, 512, seed=1103)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    rt = model._runtime()

    # (a) one tile against the oracle, (b) batch-slice invariance, bit exact
    p16 = model.predict_device(xd).clone()
    p4 = model.predict_device(xd[4:8].contiguous())
    assert torch.equal(p16[4:8], p4), "inference result of a tile depends on its batch neighbours"
    P = M.Params(weights=model.get_weights())
    with torch.no_grad():
        pc = getattr(M, oracle_fn)(P, torch.from_numpy(x[5:6]), training=False).numpy()
    err = float(np.abs(p16[5].cpu().numpy() - pc[0]).max())
    print(f"{name} 512x512: max |p_gpu - p_cpu| on one full-size tile = {err:.2e}")
    assert err <= 1e-3
    del p16, p4

    # (c) determinism of a full training step: same weights, same batch -> bit-identical loss and gradients
    w0 = rt.w_train.clone()
    f0 = rt.w_frozen.clone()

    def step():
        rt.w_train.copy_(w0)
        rt.weights_changed()
        rt.w_frozen.copy_(f0)
        rt.adam_m.zero_()
        rt.adam_v.zero_()
        model.optimizer.iterations = 0
        loss, _ = model.train_on_batch(xd, yd, return_device_scalars=True)
        return float(loss.item()), rt.g_train.clone()

    l1, g1 = step()
    l2, g2 = step()
    assert l1 == l2 and torch.equal(g1, g2), "training step is not run-to-run deterministic"

    # (d) directional derivative of the loss along a random direction d (normalised per tensor scale)
    # random magnitudes relative to each weight, every component signed along its gradient: <g, d> is then a sum
    # of positive terms, far above the fp32 resolution of the loss (a direction of random signs can cancel to a
    # derivative that the finite difference cannot resolve)
    gen = torch.Generator(device="cpu").manual_seed(7)
    d = torch.randn(w0.numel(), generator=gen).abs().cuda()
    d *= (w0.abs() + 1e-3) * torch.sign(g1)
    gd = float((g1.double() * d.double()).sum().item())

    def loss_at(w):
        rt.w_train.copy_(w)
        rt.weights_changed()
        rt.w_frozen.copy_(f0)
        pr = rt.forward(xd, training=True)
        val = float(rt.eng.loss_fwd(model.loss_kind, pr, yd).item())
        rt.release()
        return val

    # Central differences at three step sizes.  The network is piecewise smooth: ReLU / max-pool kinks crossed
    # inside the step leave an error proportional to h (third-order curvature only h^2), so the last two steps
    # are extrapolated linearly to h = 0 and THAT is compared with <g, d>; the steps stay large enough for the
    # loss difference to sit far above fp32 resolution.
    hs = (4e-4, 1e-4, 2.5e-5)
    fds = [(loss_at(w0 + h * d) - loss_at(w0 - h * d)) / (2 * h) for h in hs]
    rt.w_train.copy_(w0)
    rt.weights_changed()
    rt.w_frozen.copy_(f0)
    fd0 = fds[2] + (fds[2] - fds[1]) * hs[2] / (hs[1] - hs[2])
    print(f"{name} 512x512 bs16: loss {l1:.6f}; directional derivative fd(h=4e-4, 1e-4, 2.5e-5) = "
          f"{fds[0]:.5e}, {fds[1]:.5e}, {fds[2]:.5e} -> h=0: {fd0:.5e} vs <g,d> {gd:.5e}")
    assert abs(fd0 - gd) <= 0.05 * max(abs(gd), abs(fd0)) + 1e-6, (fds, fd0, gd)
    del g1, g2, d, w0, f0
    torch.cuda.empty_cache()


# Convolution kernels at the BASELINE sizes through identities that hold for any size: forward, dgrad and wgrad are the
# three faces of one trilinear form, <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(x, dy)> (exact in real arithmetic;
# fp32 kernels + fp64 inner products agree to ~1e-6).  One case per kernel family of the DeepLabv3+ step.
ADJ = [
    ("aspp_d12_2048_256", 16, 32, 2048, 256, 3, 12),   # the north_star dilated conv: conv_x6_kernel / wgrad_x6_kernel
    ("middle_pw_728", 16, 32, 728, 728, 1, 1),        # the 48 pointwise GEMMs of the middle flow
    ("decoder_512_64_64", 16, 512, 64, 64, 3, 1),     # LDS-patch kernel (forward + dgrad), x6 wgrad with tiles spanning taps
    ("decoder_512_64_32", 8, 512, 64, 32, 3, 1),      # patch kernel with the four-way K split (N = 32)
]


@pytest.mark.parametrize("case", ADJ, ids=[c[0] for c in ADJ])
def test_conv_trilinear_identities_full_size(engine, case):
    _, n, hw, cin, cout, k, dil = case
    torch.cuda.empty_cache()
    g = torch.Generator(device="cuda").manual_seed(hw + cin)
    x = torch.rand(n, hw, hw, cin, generator=g, device="cuda") * 2 - 1
    w = (torch.rand(k, k, cin, cout, generator=g, device="cuda") * 2 - 1) * (1.0 / np.sqrt(k * k * cin))
    d = engine.conv_desc(tuple(x.shape), cout, k, k, 1, dil, "same")
    y = engine.conv2d_fwd(x, w, None, desc=d)
    dy = torch.rand(*y.shape, generator=g, device="cuda") * 2 - 1
    dx = engine.conv2d_dgrad(dy, w, d)
    dw, _ = engine.conv2d_wgrad(x, dy, d, want_bias=False)

    def dot(a, b):  # fp64 inner product in chunks (the tensors are up to 1 GiB)
        a, b = a.reshape(-1), b.reshape(-1)
        s = 0.0
        for i in range(0, a.numel(), 1 << 26):
            s += float((a[i:i + (1 << 26)].double() * b[i:i + (1 << 26)].double()).sum().item())
        return s

    t_y, t_x, t_w = dot(y, dy), dot(x, dx), dot(w, dw)
    scale = float(y.double().norm().item() * dy.double().norm().item())
    print(f"{case[0]}: <y,dy> {t_y:.9e}  <x,dx> {t_x:.9e}  <w,dw> {t_w:.9e}  (|y||dy| = {scale:.3e})")
    assert abs(t_y - t_x) <= 2e-6 * scale and abs(t_y - t_w) <= 2e-6 * scale
    # and run-to-run bit-identity of all three (fixed-order split-K and K-class sums, no float atomics)
    assert torch.equal(engine.conv2d_fwd(x, w, None, desc=d), y)
    assert torch.equal(engine.conv2d_dgrad(dy, w, d), dx)
    assert torch.equal(engine.conv2d_wgrad(x, dy, d, want_bias=False)[0], dw)


C5 = [("res34", "res34_unet"), ("hrnet", "hrnet"), ("v3plus", "deeplab_v3plus"), ("scse", "scse_unet"),
      ("bam", "deeplab_v3plus_bam")]


def test_config5_ensemble_1024_bs8_hipgraph(engine):
    """BASELINE configs[4] at its workload: the five predict_model builders (predict.py:17-54 order) at 1024x1024,
    batch 8, forward captured into a hipGraph.  Per model: graph replay == eager launches bit for bit on two different
    batches while the earlier models' graphs stay alive; ONE 1024x1024 tile against oracle/models.py (<= 1e-3 on the
    probabilities, argmax equal wherever the oracle's class margin exceeds 1e-6 - the excused near-ties are counted
    and printed); then the 3-of-5 vote of the five masks (model_fuse.py:315-323) against oracle/pipeline.vote_ref,
    exact.  At 1024 the ASPP AveragePooling2D(32) yields a 2x2 map, not a global pool."""
    from building_detection_amd import zoo
    from oracle import pipeline as OP
    size, batch = 1024, 8
    torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(1103)
    xa = (torch.randint(0, 256, (batch, size, size, 3), generator=g).float() / 127.5 - 1)
    xb = torch.flip(xa, dims=[0, 2]).contiguous()
    xa_d, xb_d = xa.cuda(), xb.cuda()
    graphs, masks_gpu, masks_cpu = [], [], []
    for name, oracle_fn in C5:
        m = zoo.BUILDERS[name]((size, size, 3))
        ea = m.predict_device(xa_d).clone()
        eb = m.predict_device(xb_d).clone()
        gp = m.capture_predict(batch)
        graphs.append(gp)  # kept alive: config 5 holds the five captured models together
        assert torch.equal(gp(xa_d), ea), f"{name}: graph replay differs from the eager forward"
        assert torch.equal(gp(xb_d), eb), f"{name}: graph replay differs on a second batch"
        P = M.Params(weights=m.get_weights())
        with torch.no_grad():
            pc = getattr(M, oracle_fn)(P, xa[3:4], training=False).numpy()[0]
        pg = ea[3].cpu().numpy()
        err = float(np.abs(pg - pc).max())
        margin = np.abs(pc[..., 1] - pc[..., 0])
        mg, mc = pg[..., 1] > pg[..., 0], pc[..., 1] > pc[..., 0]
        strict = margin > 1e-6
        excused = int((mg != mc)[~strict].sum())
        print(f"config 5 {name}: max|p_gpu-p_cpu| {err:.2e}; near-ties (margin <= 1e-6): {int((~strict).sum())}, "
              f"of which argmax differs: {excused}")
        assert err <= 1e-3
        assert np.array_equal(mg[strict], mc[strict]), f"{name}: argmax differs outside the near-tie margin"
        masks_gpu.append(((ea[3, ..., 1] > ea[3, ..., 0]).to(torch.uint8) * 255).contiguous())
        masks_cpu.append(mc.astype(np.uint8) * 255)
        del ea, eb, m
    # the earlier graphs still replay correctly after the later models allocated, captured and ran
    first = graphs[0](xa_d).clone()
    assert torch.equal(graphs[0](xa_d), first)
    vote_gpu = engine.vote_ge([mk.view(-1) for mk in masks_gpu], 3).view(size, size).cpu().numpy()
    np.testing.assert_array_equal(vote_gpu, OP.vote_ref([mk.cpu().numpy() for mk in masks_gpu], 3))
    agree = float((vote_gpu == OP.vote_ref(masks_cpu, 3)).mean())
    print(f"config 5 vote: {float((vote_gpu == 255).mean()):.4f} positive; agreement with the all-oracle vote {agree:.6f}")
    assert agree >= 1 - 1e-5  # only the excused near-tie pixels can differ
    assert torch.cuda.max_memory_allocated() < 120 * 2 ** 30
