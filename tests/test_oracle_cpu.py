"""Pins the oracle without the reference (which cannot run here and holds no goldens — parity unpinned):
two independent CPU restatements must agree (torch-functional oracle/tfops.py vs plain C oracle/conv_ref.c),
TF 'same' padding rules are checked on their documented corner cases, autograd gradients are checked against
fp64 central finite differences, and the five graphs reproduce the structural known-answers,
including the reference's only recorded number: 22,910,272 backbone parameters (train_model/res34.py:305)."""
import numpy as np
import pytest
import torch

from oracle import models as M
from oracle import ref_c
from oracle import tfops as T


def rnd(shape, seed):
    return np.random.default_rng(seed).uniform(-1, 1, size=shape).astype(np.float32)


def test_same_pad_rules():
    # 3x3 s1 -> (1,1); 3x3 dilated d -> (d,d); 3x3 s2 even -> (0,1); 1x1 s2 -> (0,0)
    assert T.same_pad(32, 3, 1) == (32, 1, 1)
    assert T.same_pad(32, 3, 1, 18) == (32, 18, 18)
    assert T.same_pad(512, 3, 2) == (256, 0, 1)
    assert T.same_pad(512, 1, 2) == (256, 0, 0)
    assert T.same_pad(17, 3, 2) == (9, 1, 1)     # odd size: symmetric
    assert T.same_pad(512, 2, 2) == (256, 0, 0)  # convT k=2 forward conv: no pad
    assert T.same_pad(256, 3, 2) == (128, 0, 1)  # maxpool 3x3 s2 'same'


@pytest.mark.parametrize("n,h,w,cin,cout,k,stride,dil", [
    (2, 9, 11, 5, 7, 3, 1, 1), (1, 12, 12, 4, 6, 3, 1, 6), (2, 10, 8, 3, 4, 3, 2, 1), (2, 9, 7, 6, 5, 3, 2, 1),
    (1, 8, 8, 8, 3, 1, 2, 1), (1, 6, 6, 4, 4, 3, 1, 18), (2, 5, 5, 3, 2, 1, 1, 1)])
def test_conv_torch_vs_c(n, h, w, cin, cout, k, stride, dil):
    x, wt, b = rnd((n, h, w, cin), 1), rnd((k, k, cin, cout), 2), rnd((cout,), 3)
    xt, wtt, bt = [torch.tensor(a, requires_grad=True) for a in (x, wt, b)]
    y = T.conv2d(xt, wtt, bt, stride, dil, "same")
    yc = ref_c.conv2d_fwd(x, wt, b, stride, dil)
    np.testing.assert_allclose(y.detach().numpy(), yc, rtol=1e-5, atol=1e-5)
    dy = rnd(tuple(y.shape), 4)
    y.backward(torch.tensor(dy))
    np.testing.assert_allclose(xt.grad.numpy(), ref_c.conv2d_dgrad(dy, wt, x.shape, stride, dil), rtol=1e-5, atol=1e-5)
    dw, db = ref_c.conv2d_wgrad(x, dy, wt.shape, stride, dil)
    np.testing.assert_allclose(wtt.grad.numpy(), dw, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(bt.grad.numpy(), db, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("stride", [1, 2])
def test_depthwise_and_separable_vs_c(stride):
    x, dw, pw, b = rnd((2, 9, 10, 6), 5), rnd((3, 3, 6, 1), 6), rnd((1, 1, 6, 4), 7), rnd((4,), 8)
    yd = T.depthwise_conv2d(torch.tensor(x), torch.tensor(dw), stride).numpy()
    ydc = ref_c.dwconv2d_fwd(x, dw[..., 0], stride)
    np.testing.assert_allclose(yd, ydc, rtol=1e-5, atol=1e-5)
    ys = T.separable_conv2d(torch.tensor(x), torch.tensor(dw), torch.tensor(pw), torch.tensor(b), stride).numpy()
    np.testing.assert_allclose(ys, ref_c.conv2d_fwd(ydc, pw, b, 1, 1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("k", [2, 3])
def test_conv_transpose_vs_c_and_adjoint(k):
    x, w, b = rnd((2, 5, 6, 4), 9), rnd((k, k, 3, 4), 10), rnd((3,), 11)
    y = T.conv2d_transpose(torch.tensor(x), torch.tensor(w), torch.tensor(b), 2, "same").numpy()
    assert y.shape == (2, 10, 12, 3)
    np.testing.assert_allclose(y, ref_c.conv2d_transpose(x, w, b, 2), rtol=1e-5, atol=1e-5)
    # definition: convT (without bias) is the adjoint of the stride-2 SAME conv with the same kernel
    u = rnd((2, 10, 12, 3), 12)
    lhs = float((ref_c.conv2d_transpose(x, w, None, 2) * u).sum())
    rhs = float((x * ref_c.conv2d_fwd(u, w, None, 2, 1)).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


@pytest.mark.parametrize("k,stride,same", [(3, 2, True), (2, 2, False), (2, 4, False)])
def test_maxpool_vs_c(k, stride, same):
    x = rnd((2, 12, 16, 5), 13)
    y = T.max_pool(torch.tensor(x), k, stride, "same" if same else "valid").numpy()
    np.testing.assert_array_equal(y, ref_c.maxpool_fwd(x, k, stride, same))


def test_batchnorm_semantics():
    x = torch.tensor(rnd((4, 3, 3, 5), 14)) * 2 + 1
    g, b = torch.tensor(rnd((5,), 15)) + 1.5, torch.tensor(rnd((5,), 16))
    mm, mv = torch.zeros(5), torch.ones(5)
    y, nm, nv = T.batch_norm(x, g, b, mm, mv, training=True)
    xf = x.reshape(-1, 5).double()
    mean, var = xf.mean(0), xf.var(0, unbiased=False)
    ref = ((xf - mean) / torch.sqrt(var + 1e-3)).float().reshape(x.shape) * g + b
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)
    n = xf.shape[0]
    np.testing.assert_allclose(nm.numpy(), (0.01 * mean).float().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(nv.numpy(), (0.99 + 0.01 * var * n / (n - 1)).float().numpy(), rtol=1e-5)  # fused: unbiased
    x2 = x[:, 0, 0, :]
    _, _, nv2 = T.batch_norm(x2, g, b, mm, mv, training=True)
    np.testing.assert_allclose(nv2.numpy(), (0.99 + 0.01 * x2.double().var(0, unbiased=False)).float().numpy(), rtol=1e-5)
    yi, _, _ = T.batch_norm(x, g, b, nm, nv, training=False)
    np.testing.assert_allclose(yi.numpy(), ((x - nm) / torch.sqrt(nv + 1e-3) * g + b).numpy(), rtol=1e-5, atol=1e-6)


def test_upsample_and_pools():
    x = torch.arange(2 * 2 * 3 * 1, dtype=torch.float32).reshape(2, 2, 3, 1)
    u = T.upsample_nearest(x, 2)
    assert u.shape == (2, 4, 6, 1) and u[0, 1, 3, 0] == x[0, 0, 1, 0] and u[1, 3, 5, 0] == x[1, 1, 2, 0]
    a = T.avg_pool(u, 2)
    np.testing.assert_allclose(a.numpy(), x.numpy())
    np.testing.assert_allclose(T.global_avg_pool(x).numpy(), x.mean((1, 2)).numpy())


@pytest.mark.parametrize("kind", ["binary_crossentropy", "focal_loss", "edge_focal_loss"])
def test_loss_gradient_matches_appendix_c(kind):
    """Analytic dL/dz of SURVEY App. C vs autograd in fp64."""
    g = torch.Generator().manual_seed(3)
    z = torch.randn(2, 4, 4, 2, generator=g, dtype=torch.float64, requires_grad=True)
    m = (torch.rand(2, 4, 4, generator=g) > 0.5).double()
    yt = torch.stack([1 - m, m, 1 + (torch.rand(2, 4, 4, generator=g) > 0.5).double(),
                      1 + (torch.rand(2, 4, 4, generator=g) > 0.5).double()], -1)
    p = torch.softmax(z, -1)
    M.loss_fn(kind, yt, p).backward()
    eps, Mn = 1e-7, 2 * 4 * 4
    pd = p.detach()
    if kind == "binary_crossentropy":
        a, gp = yt[..., :2], None
        gpc = -(1.0 / Mn) * a / (pd + eps)
    else:
        alpha = torch.tensor([0.5, 0.5] if kind == "focal_loss" else [0.35, 0.65], dtype=torch.float64)
        wgt = yt[..., 2:] if kind == "edge_focal_loss" else 1.0
        a = alpha * wgt * yt[..., :2]
        gpc = (1.0 / Mn) * a * (2 * (1 - pd) * torch.log(pd + eps) - (1 - pd) ** 2 / (pd + eps))
    dz = pd * (gpc - (gpc * pd).sum(-1, keepdim=True))
    np.testing.assert_allclose(z.grad.numpy(), dz.numpy(), rtol=1e-9, atol=1e-12)


def test_finite_difference_gradients_small_graph():
    """fp64 central differences through conv -> BN(train) -> relu -> sepconv -> convT -> scSE -> softmax loss."""
    torch.manual_seed(0)
    P = M.Params(seed=5, dtype=torch.float64)
    x = torch.rand(2, 8, 8, 3, dtype=torch.float64) * 2 - 1
    yt = torch.zeros(2, 16, 16, 4, dtype=torch.float64)
    yt[..., 0] = 1
    yt[:, 4:9, 3:12, 0], yt[:, 4:9, 3:12, 1] = 0, 1
    yt[..., 2:] = 1 + (torch.rand(2, 16, 16, 2) > 0.7).double()

    def fwd():
        n = M.Net(P, True)
        h = n.conv_bn_relu(x, 8, 3, dilation=2)
        h = n.bn(n.sepconv(torch.relu(h), 16))
        h = n.convT(h, 16, 3)
        h = n.scse(h)
        return M.loss_fn("edge_focal_loss", yt, torch.softmax(n.conv(h, 2, 1), -1))

    loss = fwd()
    tr = P.trainable_tensors()
    grads = torch.autograd.grad(loss, tr)
    rng = np.random.default_rng(0)
    for t, g in zip(tr, grads):
        flat = t.detach().view(-1)
        for idx in rng.choice(flat.numel(), size=min(3, flat.numel()), replace=False):
            old = flat[idx].item()
            h = 1e-5
            flat[idx] = old + h
            lp = fwd().item()
            flat[idx] = old - h
            lm = fwd().item()
            flat[idx] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - g.reshape(-1)[idx].item()) <= 1e-6 + 1e-4 * abs(fd), (tuple(t.shape), idx, fd, g.reshape(-1)[idx].item())


EXPECTED = {"v3plus": (64509482, 106192), "bam": (62863400, 105770), "scse": (34558914, 0),
            "res34": (38519778, 25536), "hrnet": (9588226, 19584)}


@pytest.mark.parametrize("name", list(EXPECTED))
def test_structural_known_answers(name):
    P = M.Params()
    kw = {"aspp_pool": 4} if name in ("v3plus", "bam") else {}
    with torch.no_grad():
        y = M.BUILDERS[name](P, torch.zeros(1, 64, 64, 3), **kw)
    assert tuple(y.shape) == (1, 64, 64, 2)
    assert (P.count(True), P.count(False)) == EXPECTED[name]


def test_res34_backbone_reference_known_answer():
    P = M.Params()
    with torch.no_grad():
        M.res34_unet(P, torch.zeros(1, 32, 32, 3), backbone_only=True)
    assert P.count(True) == 22910272  # train_model/res34.py:305 "Trainable params: 22,910,272"


# ---- golden fixtures (tests/golden/, generated by make_golden.py from this oracle; see its docstring) -------------
def _load_generator():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "make_golden.py")
    spec = importlib.util.spec_from_file_location("make_golden", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, os.path.dirname(path)


def test_golden_ops_reproduced():
    """The committed op-level vectors (TF 'same' asymmetry at stride 2, Conv2DTranspose, fused/non-fused BN moving
    variance, pools, losses, confusion counts, Keras Adam, the LR schedule) are reproduced by the oracle exactly."""
    gen, here = _load_generator()
    stored = np.load(f"{here}/ops.npz")
    fresh = gen.ops_fixture()
    assert set(stored.files) == set(fresh.keys())
    for k in stored.files:
        np.testing.assert_allclose(np.asarray(fresh[k], dtype=np.float64), stored[k].astype(np.float64), rtol=1e-6, atol=1e-7,
                                   err_msg=k)


@pytest.mark.parametrize("name,fn,size,kw", [("scse", "scse_unet", 32, {}), ("hrnet", "hrnet", 32, {})],
                         ids=["scse", "hrnet"])
def test_golden_models_reproduced(name, fn, size, kw):
    """Two of the five model fixtures are re-derived on CPU here (all five on the GPU box against the engine):
    seeded weights, predict() probabilities, training loss and per-tensor gradient norms."""
    gen, here = _load_generator()
    stored = np.load(f"{here}/model_{name}.npz")
    fresh = gen.model_fixture(name, fn, size, kw)
    assert int(stored["n_tensors"]) == int(fresh["n_tensors"])
    np.testing.assert_allclose(fresh["w_abs_sum"], stored["w_abs_sum"], rtol=1e-9)
    np.testing.assert_allclose(fresh["probs"], stored["probs"], atol=2e-6)
    np.testing.assert_allclose(fresh["train_loss"], stored["train_loss"], rtol=1e-9)
    np.testing.assert_allclose(fresh["grad_norms"], stored["grad_norms"], rtol=1e-7, atol=1e-12)
