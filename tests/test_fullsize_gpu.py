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


def _diag_groups():
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("diag_groups", os.path.join(os.path.dirname(os.path.dirname(__file__)), "scripts",
                                                                               "diag_bf16_groups.py"))
    dg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dg)
    return dg


def _param_groups(model, name):
    """trainable parameter -> group label: DeepLabv3+'s blocks (scripts/diag_bf16_groups.py), else quarters of the list."""
    tr = [p for p in model.params if p.trainable]
    if name == "v3plus":
        return _diag_groups().groups_of(model)
    return ["q%d" % (1 + 4 * i // len(tr)) for i in range(len(tr))]


def _group_directional_derivatives(model, grp, d, g, w0, f0, xd, yd, hs=(3e-5, 1.5e-5)):
    """Per parameter group: central differences of the training-mode loss along `d` restricted to the group (two steps,
    extrapolated linearly to h = 0) over <g, d_group>.  1.0 = that group's backward is the gradient of the forward."""
    rt = model._runtime()
    tr = [p for p in model.params if p.trainable]

    def loss_at(w):
        rt.w_train.copy_(w)
        rt.weights_changed()
        rt.w_frozen.copy_(f0)
        pr = rt.forward(xd, training=True)
        val = float(rt.eng.loss_fwd(model.loss_kind, pr, yd).item())
        rt.release()
        return val

    out = {}
    for gname in dict.fromkeys(grp):
        mask = torch.zeros_like(w0)
        for p_, q in zip(tr, grp):
            if q == gname:
                mask[p_.offset:p_.offset + p_.size] = 1.0
        dgm = d * mask
        gd = float((g.double() * dgm.double()).sum().item())
        f = [(loss_at(w0 + h * dgm) - loss_at(w0 - h * dgm)) / (2 * h) for h in hs]
        f0_ = f[1] + (f[1] - f[0]) * hs[1] / (hs[0] - hs[1])
        out[gname] = f0_ / gd if gd != 0.0 else float("nan")
    rt.w_train.copy_(w0)
    rt.weights_changed()
    rt.w_frozen.copy_(f0)
    return out


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
    # (e) the same per parameter group, at steps small enough to resolve it (round 3): every block's backward is the
    # gradient of the forward on its own - measured 0.9965 ... 1.0006 on DeepLabv3+ (profiles/r03_diag_fd_512.txt)
    ratios = _group_directional_derivatives(model, _param_groups(model, name), d, g1, w0, f0, xd, yd)
    print(f"{name} 512x512 bs16: fd / <g,d> per parameter group: " + "  ".join(f"{k} {v:.4f}" for k, v in ratios.items()))
    for k, v in ratios.items():
        assert abs(v - 1.0) <= 0.02, (name, k, v)
    del g1, g2, d, w0, f0
    torch.cuda.empty_cache()


def test_config3_deeplab_bf16_512_bs16(engine):
    """BASELINE configs[2] at its per-GPU workload: DeepLabv3+ 512x512 bs 16 with mixed_bfloat16 storage (the 8-GPU part of
    config 3 is the data-parallel exchange, tests/test_dist_gpu.py / test_host_cpu.py).  As for the fp32 configs:
      (a) ONE full-size tile against oracle/models.py (fp32, inference mode) and the whole batch against the fp32 ENGINE,
          within the bf16 tolerance contract of DESIGN.md section 8 - mean |dp| <= 5e-3, argmax flips <= 1 % of the pixels
          (the north_star bar "1e-3, bit-exact argmax" is an fp32 statement; the numbers measured here are printed);
      (b) batch-slice invariance of inference, bit exact;
      (c) run-to-run determinism of the training step, bit exact (loss and the whole fp32 gradient arena);
      (d) a SIGN / SCALE SANITY CHECK of the whole backward pass, NOT a gradient check: central differences of the bf16 loss
          along a direction signed with the gradient against <g, d>, over all parameters (asserted within [0.5, 1.1]: a
          backward pass with the wrong sign, a lost factor of 2 or a dead parameter group moves it out) and per parameter
          group (reported; a majority inside a wide band).  It cannot resolve more: the bf16 loss is a staircase in the
          weights (a weight moves its bf16 plane only when it crosses a rounding boundary) and at random init a third of the
          encoder gradient's signs are noise, so a group's quotient scatters by +-30 % and changes sign under changes of
          ROUNDING alone (lab notebook 10.3).  What guards the bf16 backward pass is (e) below and the exact block
          chains of tests/test_block_chains_gpu.py (every block incl. the Xception middle flow, cosine >= 0.99 vs fp64);
      (e) per-layer-group agreement of the bf16 gradient with the fp32 engine's at THIS workload on weights after 50 fp32 Adam
          steps, where the comparison is conditioned well enough to fail (VERDICT r2 next #1b); see the comment there."""
    from building_detection_amd import zoo, mixed_precision as MP
    from building_detection_amd.data import synthetic_batch
    torch.cuda.empty_cache()
    MP.set_global_policy("mixed_bfloat16")
    try:
        model = _compiled(zoo.Xception_DeepLabV3_Plus((512, 512, 3)))
    finally:
        MP.set_global_policy("float32")
    assert model.compute_dtype == "bfloat16"
    m32 = _compiled(zoo.Xception_DeepLabV3_Plus((512, 512, 3)))
    ws = model.get_weights()
    m32.set_weights(ws)
    x, y = synthetic_batch(16, 512, 512, seed=1103)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    rt = model._runtime()

    # (a), (b)
    p16 = model.predict_device(xd).clone()
    p4 = model.predict_device(xd[4:8].contiguous())
    assert p16.dtype == torch.float32 and torch.equal(p16[4:8], p4), "bf16 inference result of a tile depends on its batch neighbours"
    P = M.Params(weights=ws)
    with torch.no_grad():
        pc = M.deeplab_v3plus(P, torch.from_numpy(x[5:6]), training=False).numpy()[0]
    pg = p16[5].cpu().numpy()
    dmean, dmax = float(np.abs(pg - pc).mean()), float(np.abs(pg - pc).max())
    flips = float(((pg[..., 1] > pg[..., 0]) != (pc[..., 1] > pc[..., 0])).mean())
    pf = m32.predict_device(xd)
    emean = float((p16 - pf).abs().mean().item())
    eflips = float(((p16[..., 1] > p16[..., 0]) != (pf[..., 1] > pf[..., 0])).float().mean().item())
    print(f"config 3, 512x512 bf16: one tile vs the oracle: mean|dp| {dmean:.2e} max|dp| {dmax:.2e} argmax flips {flips:.2e}; "
          f"16 tiles vs the fp32 engine: mean|dp| {emean:.2e} flips {eflips:.2e}")
    assert dmean <= 5e-3 and flips <= 1e-2 and emean <= 5e-3 and eflips <= 1e-2
    del p16, p4, pf

    # (c)
    w0, f0 = rt.w_train.clone(), rt.w_frozen.clone()

    def step(mdl):
        r = mdl._runtime()
        r.w_train.copy_(w0)
        r.weights_changed()
        r.w_frozen.copy_(f0)
        r.adam_m.zero_()
        r.adam_v.zero_()
        mdl.optimizer.iterations = 0
        loss, _ = mdl.train_on_batch(xd, yd, return_device_scalars=True)
        return float(loss.item()), r.g_train.clone()

    l1, g1 = step(model)
    l2, g2 = step(model)
    assert l1 == l2 and torch.equal(g1, g2), "the bf16 training step is not run-to-run deterministic"
    del g2

    # (d) sign / scale sanity check (see the docstring: not a gradient check).  Per parameter group, as for fp32 (same direction, same steps: the fp32 loss has so much curvature along a direction
    # signed with the gradient that only steps <= 3e-5 resolve it, and the bf16 forward answers such steps without bias -
    # 64 M weights cross their bf16 rounding boundaries in proportion; profiles/r03_diag_fd_512.txt).  What the ratio shows
    # in bf16 is how well <g, d> - d is signed with the bf16 gradient's own signs - predicts the real change of the loss:
    # 1 where the gradient is accurate (decoder), lower where the bf16 gradient is noisy at random init (encoder: (e) below).
    gen = torch.Generator(device="cpu").manual_seed(7)
    d = torch.randn(w0.numel(), generator=gen).abs().cuda()
    d *= (w0.abs() + 1e-3) * torch.sign(g1)
    ratios = _group_directional_derivatives(model, _param_groups(model, "v3plus"), d, g1, w0, f0, xd, yd)
    n_tr = sum(1 for p_ in model.params if p_.trainable)
    whole = _group_directional_derivatives(model, ["all"] * n_tr, d, g1, w0, f0, xd, yd)["all"]
    print(f"config 3 bf16 512x512 bs16: fd / <g,d> over all parameters {whole:.4f}; per group: " +
          "  ".join(f"{k} {v:.4f}" for k, v in ratios.items()))
    # Measured (profiles/r03_diag_fd_512.txt and the first run of this test): all parameters 0.66; groups 0.54 ... 1.31.  The
    # bf16 loss is a staircase at these step sizes (a weight moves its bf16 plane only when it crosses a rounding boundary):
    # over 64 M parameters the crossings average out, over a group of 1-2 M they leave +-30 % of scatter, and where the bf16
    # gradient is noisy (random init, (e)) <g, d> overstates the real slope.  So this part only excludes a wrong sign or
    # scale of a whole group; the discriminating gradient checks are (e) and tests/test_block_chains_gpu.py.
    # Round 3, later: the per-group quotient turned out to be noise for the smaller groups - a change of ROUNDING alone (the
    # residual add applying its operand's BatchNormalization without storing the normalised bf16 tensor; the depthwise dgrad
    # adding a collected gradient before the one rounding) moves 'aspp' from 0.54 to -0.75, 'neck' from 1.04 to -1.86 or
    # 'entry' from 1.01 to -0.24, while the fp32 step of the same builds is bit-identical and all parameters together stay at
    # 0.58 - 0.69.  Asserted therefore: the whole, and a majority of the groups inside the band.
    assert 0.5 <= whole <= 1.1, whole
    inside = [k for k, v in ratios.items() if GROUP_FD[k][0] <= v <= GROUP_FD[k][1]]
    assert len(inside) >= 4, ratios
    del g1, d, w0, f0
    torch.cuda.empty_cache()

    # (e) per-layer-group agreement with the fp32 engine.  At random init the comparison is ill-conditioned whatever the
    # batch (scripts/diag_bf16_groups.py, profiles/r03_diag_bf16_groups.txt: at 512x512 bs16 the encoder cosine is 0.51 -
    # and two CORRECT fp32 evaluations, x6 vs native MFMA, already differ by 1.2e-2 there against 8.6e-4 in the decoder: the
    # net amplifies rounding noise ~15x from the decoder back to the encoder).  After 50 fp32 Adam steps the same net is
    # conditioned well enough to discriminate: every group's bf16 gradient must then align with the fp32 one (GROUP_COS), and
    # the ratio (bf16 error) / (fp32 noise floor = native-MFMA vs x6 error) must be flat across the groups - a group whose
    # bf16 backward had an error of its own would stand out of that profile, whatever the amplification upstream of it.
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("diag_groups", os.path.join(os.path.dirname(os.path.dirname(__file__)), "scripts",
                                                                               "diag_bf16_groups.py"))
    dg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dg)
    for s_ in range(50):
        xb, yb = synthetic_batch(16, 512, 512, seed=500 + s_ % 4)
        m32.train_on_batch(torch.from_numpy(xb).cuda(), torch.from_numpy(yb).cuda(), return_device_scalars=True)
    wt = m32.get_weights()
    grp = dg.groups_of(m32)
    xg, yg = synthetic_batch(16, 512, 512, seed=11)
    xgd, ygd = torch.from_numpy(xg).cuda(), torch.from_numpy(yg).cuda()
    l32 = float(m32.train_on_batch(xgd, ygd, return_device_scalars=True)[0].item())
    g32 = m32.get_gradients()
    del m32
    torch.cuda.empty_cache()
    model.set_weights(wt)
    l16 = float(model.train_on_batch(xgd, ygd, return_device_scalars=True)[0].item())
    g16 = model.get_gradients()
    prev = engine.lib.sg_set_conv_x6(0)
    try:
        mn = _compiled(zoo.Xception_DeepLabV3_Plus((512, 512, 3)))
        mn.set_weights(wt)
        mn.train_on_batch(xgd, ygd, return_device_scalars=True)
        gn = mn.get_gradients()
    finally:
        engine.lib.sg_set_conv_x6(prev)
    del mn
    r16 = dg.compare("config 3 after 50 fp32 steps, bf16        vs fp32", g16, g32, grp)
    rn = dg.compare("config 3 after 50 fp32 steps, native MFMA vs fp32", gn, g32, grp)
    assert abs(l16 - l32) <= 3e-2 * abs(l32), (l16, l32)
    for k, (c, r) in r16.items():
        assert c >= GROUP_COS[k], (k, c, r)
    ratio = {k: r16[k][1] / rn[k][1] for k in r16}
    print("config 3: (bf16 error) / (fp32 noise floor) per group: " + "  ".join(f"{k} {v:.0f}" for k, v in ratio.items()))
    assert max(ratio.values()) <= 8.0 * min(ratio.values()), ratio
    torch.cuda.empty_cache()


# lowest per-group cosine accepted between the bf16 and the fp32 gradient at 512x512 bs16 after 50 fp32 steps.  Measured
# (profiles/r03_diag_bf16_groups.txt): entry 0.845, middle 0.864, exit 0.968, sk 0.971, aspp 0.977, neck 0.997, decoder
# 1.0000 (at random init: 0.51 / 0.51 / 0.62 / 0.66 / 0.68 / 0.76 / 0.998); the kernels are deterministic, so the
# head-room only has to cover later changes of summation order.
GROUP_COS = {"entry": 0.75, "middle": 0.78, "exit": 0.94, "sk": 0.94, "aspp": 0.95, "neck": 0.99, "decoder": 0.9995}
# accepted range of (finite difference of the bf16 loss) / <g_bf16, d> per group at RANDOM INIT (part (d)); fp32 gives 1.00 in
# every group.  First measured run: entry 1.01, middle 0.62, exit 1.00, sk 1.05, aspp 0.54, neck 1.04, decoder 1.31.
GROUP_FD = {k: (0.3, 1.7) for k in ("entry", "middle", "exit", "sk", "aspp", "neck", "decoder")}


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
