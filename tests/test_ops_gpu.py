"""HIP kernel parity (through the C ABI) against the CPU oracle (oracle/tfops.py) on seeded inputs.

Tolerance: the north_star bar is 1e-3 absolute on fp32 outputs; these op tests hold each kernel to a much
tighter bound, max|gpu-cpu| <= RTOL * max|cpu| with RTOL = 2e-5 (fp32 accumulation-order noise only), so
errors cannot hide inside the end-to-end budget.  Integer outputs (confusion counts, argmax, vote) are exact.
"""
import zlib

import numpy as np
import pytest
import torch

from oracle import tfops as T

pytestmark = pytest.mark.gpu

RTOL = 2e-5


def close(got, ref, rtol=RTOL, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.2e})"


def rnd(gen, *shape, lo=-1.0, hi=1.0):
    return (torch.rand(*shape, generator=gen) * (hi - lo) + lo).float()


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, dil, tag
    (2, 32, 32, 2048, 256, 3, 1, 6, "aspp_d6"),
    (1, 32, 32, 2048, 256, 3, 1, 12, "aspp_d12"),
    (1, 32, 32, 2048, 256, 3, 1, 18, "aspp_d18"),
    (2, 32, 32, 256, 256, 3, 1, 6, "sk_d6"),
    (2, 16, 16, 728, 728, 1, 1, 1, "pw_728"),
    (2, 24, 20, 3, 32, 3, 2, 1, "stem_s2"),
    (2, 17, 19, 64, 64, 3, 2, 1, "odd_s2"),
    (2, 16, 16, 256, 128, 1, 2, 1, "short_1x1_s2"),
    (1, 16, 16, 45, 45, 3, 1, 4, "bam_d4_c45"),
    (2, 16, 16, 64, 1, 1, 1, 1, "sse_cout1"),
    (2, 16, 16, 32, 2, 1, 1, 1, "head_cout2"),
    (2, 12, 12, 64, 2, 3, 1, 1, "res34_head"),
    (3, 8, 8, 512, 96, 3, 1, 1, "cout96"),
    (16, 1, 1, 256, 16, 1, 1, 1, "gap_1x1"),
    # thin 1x1 convs (Cout <= 4) take the streaming kernels; odd pixel counts, non-power-of-two chunk counts
    (3, 13, 11, 64, 1, 1, 1, 1, "thin_sse_ragged"),
    (2, 9, 7, 48, 2, 1, 1, 1, "thin_c48_cout2"),
    (1, 8, 8, 256, 3, 1, 1, 1, "thin_c256_cout3"),
    (2, 6, 5, 728, 4, 1, 1, 1, "thin_c728_cout4"),
    # aligned-slab wgrad path (W % 32 == 0): pointwise with ragged channel tile, 3x3 with Cin % 128 == 0
    (1, 32, 32, 200, 136, 1, 1, 1, "wgrad_aligned_pw_ragged"),
    (1, 32, 64, 128, 64, 3, 1, 2, "wgrad_aligned_3x3_d2"),
    # x6 wgrad with a 128-row tile spanning several taps (Cin 32 / 64), ragged last tile (K = 9*48 = 432)
    (1, 32, 32, 32, 32, 3, 1, 1, "x6_wgrad_c32"),
    (2, 32, 64, 64, 48, 3, 1, 1, "x6_wgrad_c64_cout48"),
    (1, 64, 32, 48, 80, 3, 1, 3, "x6_wgrad_c48_d3"),
    # round 5: the patch kernel walks 128 / 256 reduction channels per tap in chunks of 64 (forward: Cin, dgrad: Cout)
    (2, 16, 32, 128, 64, 3, 1, 1, "patch_chunks_128_to_64"),
    (1, 16, 16, 256, 128, 3, 1, 1, "patch_chunks_256_to_128"),
    (2, 8, 16, 64, 128, 3, 1, 1, "patch_chunks_dgrad_128"),
    (1, 24, 16, 128, 32, 3, 1, 1, "patch_chunks_128_to_32"),
    (2, 16, 16, 256, 32, 3, 1, 1, "patch_chunks_256_to_32"),
    (1, 8, 32, 512, 64, 3, 1, 1, "patch_chunks_512_to_64"),
    # LDS-patch x6 kernel (3x3 s1 d1, 32 / 64 reduction channels, H % 8 == 0, W % 16 == 0): every (C, BN) form forward
    # and dgrad, K split 4 / 2 / 1 ways, image borders on all sides, several column tiles
    (2, 16, 32, 32, 32, 3, 1, 1, "patch_c32_n32"),
    (2, 24, 48, 64, 64, 3, 1, 1, "patch_c64_n64"),
    (1, 16, 16, 64, 32, 3, 1, 1, "patch_c64_n32"),
    (3, 8, 16, 32, 64, 3, 1, 1, "patch_c32_n64_one_tile"),
    (1, 8, 32, 64, 128, 3, 1, 1, "patch_c64_n128"),
    (2, 16, 16, 32, 256, 3, 1, 1, "patch_c32_n256"),
    # patch-form wgrad (conv_x6wp.h: 3x3 s1 d1, Cin and Cout in {32, 64}, H % 4 == 0, W % 16 == 0; the cases above with
    # such shapes take it too): more tiles than workgroups (576 > 512: the persistent walk and its prefetch), H % 8 != 0
    # (forward on the im2col kernel, wgrad on the patch one), every k-class split
    (3, 96, 128, 64, 32, 3, 1, 1, "wpatch_c64_n32_576_tiles"),
    (2, 20, 48, 32, 32, 3, 1, 1, "wpatch_c32_n32_h20"),
    (5, 12, 16, 32, 64, 3, 1, 1, "wpatch_c32_n64"),
    (1, 44, 80, 64, 64, 3, 1, 1, "wpatch_c64_n64"),
    # stride-2 filter gradients on the bf16-pipe slab kernel (OW % 32 == 0): 3x3 with TF's asymmetric "same" padding, the
    # 1x1 shortcut, an odd input height (the last input row is never read)
    (2, 64, 64, 64, 128, 3, 2, 1, "x6_wgrad_s2_3x3"),
    (2, 64, 64, 128, 256, 1, 2, 1, "x6_wgrad_s2_1x1"),
    (1, 63, 64, 32, 48, 3, 2, 1, "x6_wgrad_s2_odd_h"),
    # RGB stems (Cin = 3: the scalar any-shape kernels): the models' own stems are 3x3 (stride 2: DeepLab / HRNet; stride 1:
    # the U-Nets, predict_model/res34.py:50-52, scse.py); the 7x7 case is a shape no model has (K = 147, large-kernel path)
    (2, 64, 64, 3, 32, 3, 2, 1, "stem_s2_w32"),
    (1, 64, 64, 3, 64, 7, 2, 1, "stem7_s2_w32"),
    (2, 32, 32, 3, 64, 3, 1, 1, "stem_s1_w32"),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[-1] for c in CONV_CASES])
def test_conv2d_fwd_dgrad_wgrad(engine, case):
    n, h, w, cin, cout, k, stride, dil, tag = case
    g = torch.Generator().manual_seed(zlib.crc32(tag.encode()) % (2 ** 31))
    x = rnd(g, n, h, w, cin)
    wt = rnd(g, k, k, cin, cout) * (1.0 / np.sqrt(k * k * cin))
    b = rnd(g, cout)
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = T.conv2d(xr, wr, br, stride, dil, "same")
    dy = rnd(g, *yr.shape)
    yr.backward(dy)

    xd, wd, bd, dyd = x.cuda(), wt.cuda(), b.cuda(), dy.cuda()
    y = engine.conv2d_fwd(xd, wd, bd, stride, dil, "same")
    close(y, yr, what=f"{tag} fwd")
    y2 = engine.conv2d_fwd(xd, wd, bd, stride, dil, "same", relu=True)
    close(y2, torch.relu(yr), what=f"{tag} fwd+relu")
    d = engine.conv_desc(x.shape, cout, k, k, stride, dil, "same")
    dx = engine.conv2d_dgrad(dyd, wd, d)
    close(dx, xr.grad, what=f"{tag} dgrad")
    dw, db = engine.conv2d_wgrad(xd, dyd, d)
    close(dw, wr.grad, what=f"{tag} wgrad")
    close(db, br.grad, what=f"{tag} bias grad")


@pytest.mark.parametrize("hw", [(12, 12), (16, 32)], ids=["im2col", "patch"])
def test_conv2d_into_concat_slice(engine, hw):
    """y_ld / x_ld: a conv writing into (and reading from) a channel slice of a wider buffer."""
    g = torch.Generator().manual_seed(7)
    h, w = hw
    x = rnd(g, 2, h, w, 64)
    wt = rnd(g, 3, 3, 32, 64) * 0.1
    # input = channels [16,48) of x ; output = channels [64,128) of a 160-wide buffer
    xs = x[..., 16:48].contiguous()
    yr = T.conv2d(xs, wt, None, 1, 1, "same")
    xd = x.cuda()
    buf = torch.zeros(2, h, w, 160, device="cuda")
    d = engine.conv_desc((2, h, w, 32), 64, 3, 3, 1, 1, "same", x_ld=64, y_ld=160)
    engine.conv2d_fwd(xd.view(-1)[16:], wt.cuda(), None, out=buf.view(-1)[64:], desc=d)
    close(buf[..., 64:128], yr, what="slice conv")
    assert buf[..., :64].abs().max().item() == 0 and buf[..., 128:].abs().max().item() == 0
    # dgrad through the same slices: dy = channels [64,128) of the wide buffer, dx into channels [16,48) of a 64-wide one
    dy = rnd(g, 2, h, w, 64)
    xr = xs.clone().requires_grad_()
    T.conv2d(xr, wt, None, 1, 1, "same").backward(dy)
    buf[..., 64:128] = dy.cuda()
    dxb = torch.zeros(2, h, w, 64, device="cuda")
    engine.conv2d_dgrad(buf.view(-1)[64:], wt.cuda(), d, out=dxb.view(-1)[16:])
    close(dxb[..., 16:48], xr.grad, what="slice dgrad")
    assert dxb[..., :16].abs().max().item() == 0 and dxb[..., 48:].abs().max().item() == 0
    # wgrad from the same two slices (x_ld = 64, y_ld = 160; (16, 32): the patch-form kernel)
    wr = wt.clone().requires_grad_()
    T.conv2d(xs, wr, None, 1, 1, "same").backward(dy)
    dw, _ = engine.conv2d_wgrad(xd.view(-1)[16:], buf.view(-1)[64:], d, want_bias=False)
    close(dw, wr.grad, what="slice wgrad")


@pytest.mark.parametrize("k,tag,hw", [(3, "convT3", (9, 11)), (2, "convT2", (9, 11)), (3, "convT3_w32", (16, 32)),
                                      (2, "convT2_w32", (8, 32)), (3, "convT3_class_pure_tiles", (32, 64))])
def test_conv2d_transpose(engine, k, tag, hw):
    """(…_w32: the kernel gradient is a stride-2 filter gradient with OW % 32 == 0, i.e. the bf16-pipe slab kernel)
    The forward is a stride-2 dgrad: rows in parity-class order (IgemmParams::perm2) - 9 x 11 maps have 128-row tiles that
    straddle the classes, 32 x 64 maps tiles of one class each (taps of the other parities dropped per tile)."""
    g = torch.Generator().manual_seed(11 + k)
    n, (h, w), cin, cout = 2, hw, 64, 32
    x = rnd(g, n, h, w, cin)
    wt = rnd(g, k, k, cout, cin) * 0.1
    b = rnd(g, cout)
    xr, wr, br = x.clone().requires_grad_(), wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = T.conv2d_transpose(xr, wr, br, 2, "same")
    dy = rnd(g, *yr.shape)
    yr.backward(dy)
    # forward conv F whose dgrad is this transpose: input [n,2h,2w,cout] -> [n,h,w,cin], kernel HWIO = wt
    d = engine.conv_desc((n, 2 * h, 2 * w, cout), cin, k, k, 2, 1, "same")
    assert (d.Ho, d.Wo) == (h, w)
    y = engine.conv2d_dgrad(x.cuda(), wt.cuda(), d, bias=b.cuda())
    close(y, yr, what=f"{tag} fwd")
    yrelu = engine.conv2d_dgrad(x.cuda(), wt.cuda(), d, bias=b.cuda(), relu=True)
    close(yrelu, torch.relu(yr), what=f"{tag} fwd relu")
    dx = engine.conv2d_fwd(dy.cuda(), wt.cuda(), None, desc=d)
    close(dx, xr.grad, what=f"{tag} dx")
    dw, _ = engine.conv2d_wgrad(dy.cuda(), x.cuda(), d, want_bias=False)
    close(dw, wr.grad, what=f"{tag} dw")


@pytest.mark.parametrize("c,stride,pre_relu", [(728, 1, True), (128, 2, False), (45, 1, True), (256, 2, True)])
def test_depthwise(engine, c, stride, pre_relu):
    g = torch.Generator().manual_seed(c + stride)
    n, h, w = 2, 14, 18
    x = rnd(g, n, h, w, c)
    wt = rnd(g, 3, 3, c, 1)
    xr, wr = x.clone().requires_grad_(), wt.clone().requires_grad_()
    yr = T.depthwise_conv2d(torch.relu(xr) if pre_relu else xr, wr, stride)
    dy = rnd(g, *yr.shape)
    yr.backward(dy)
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    y = engine.dwconv_fwd(xd, wd, stride, pre_relu)
    close(y, yr, what="dw fwd")
    d = engine.conv_desc(x.shape, c, 3, 3, stride, 1, "same")
    dx = engine.dwconv_dgrad(dyd, wd, d, x=xd, pre_relu=pre_relu)
    close(dx, xr.grad, what="dw dgrad")
    dw = engine.dwconv_wgrad(xd, dyd, d, pre_relu)
    close(dw, wr.grad, what="dw wgrad")


@pytest.mark.parametrize("c,h,w,pre_relu", [(728, 12, 16, True), (64, 9, 20, False), (128, 32, 32, True), (4, 5, 4, False),
                                            (8, 20, 8, False), (16, 136, 8, True), (36, 4, 4, True), (68, 16, 4, False)])
def test_depthwise_stride1_register_window_path(engine, c, h, w, pre_relu):
    """W % 4 == 0, stride 1: the run kernels (4 outputs per thread) for forward, dgrad (flipped taps + mask), wgrad; with
    H % 4 == 0 the filter gradient takes the column-strip kernel (bands of 8 / 16 rows, a short last band, one-strip-wide maps)."""
    g = torch.Generator().manual_seed(c + h + w)
    x = rnd(g, 3, h, w, c)
    wt = rnd(g, 3, 3, c, 1)
    xr, wr = x.clone().requires_grad_(), wt.clone().requires_grad_()
    yr = T.depthwise_conv2d(torch.relu(xr) if pre_relu else xr, wr, 1)
    dy = rnd(g, *yr.shape)
    yr.backward(dy)
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    close(engine.dwconv_fwd(xd, wd, 1, pre_relu), yr, what="dw run fwd")
    d = engine.conv_desc(x.shape, c, 3, 3, 1, 1, "same")
    close(engine.dwconv_dgrad(dyd, wd, d, x=xd, pre_relu=pre_relu), xr.grad, what="dw run dgrad")
    close(engine.dwconv_wgrad(xd, dyd, d, pre_relu), wr.grad, what="dw run wgrad")


@pytest.mark.parametrize("shape,relu", [((4, 16, 16, 728), True), ((2, 9, 7, 45), False), ((16, 24), True),
                                         ((2, 64, 64, 64), True)])
def test_batchnorm(engine, shape, relu):
    g = torch.Generator().manual_seed(sum(shape))
    c = shape[-1]
    x = rnd(g, *shape) * 2 + 0.7
    gamma, beta = rnd(g, c) + 1.5, rnd(g, c)
    mm, mv = rnd(g, c), rnd(g, c, lo=0.5, hi=2.0)
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    yr, nm, nv = T.batch_norm(xr, gr, br, mm, mv, training=True)
    if relu:
        yr = torch.relu(yr)
    dy = rnd(g, *shape)
    yr.backward(dy)
    mmd, mvd = mm.cuda(), mv.cuda()
    y, mean, invstd = engine.bn_train_fwd(x.cuda(), gamma.cuda(), beta.cuda(), mmd, mvd, relu=relu)
    close(y, yr, what="bn fwd")
    close(mmd, nm, what="moving mean")
    close(mvd, nv, what="moving var")
    dx, dg, db = engine.bn_train_bwd(x.cuda(), y, dy.cuda(), gamma.cuda(), mean, invstd, relu=relu)
    close(dx, xr.grad, rtol=1e-4, what="bn dx")
    close(dg, gr.grad, rtol=1e-4, what="bn dgamma")
    close(db, br.grad, rtol=1e-4, what="bn dbeta")
    # the same backward with the ReLU mask recomputed from x (beta given) must be bit-identical to the y-mask form
    dx2, dg2, db2 = engine.bn_train_bwd(x.cuda(), None if relu else y, dy.cuda(), gamma.cuda(), mean, invstd, relu=relu,
                                        beta=beta.cuda())
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    yi, _, _ = T.batch_norm(x, gamma, beta, mm, mv, training=False)
    yg = engine.bn_infer(x.cuda(), gamma.cuda(), beta.cuda(), mm.cuda(), mv.cuda(), relu=relu)
    close(yg, torch.relu(yi) if relu else yi, what="bn infer")


def test_activations_add_concat(engine):
    g = torch.Generator().manual_seed(3)
    x = rnd(g, 2, 7, 9, 36) * 3
    for act, f in ((0, torch.relu), (1, torch.sigmoid)):
        xr = x.clone().requires_grad_()
        yr = f(xr)
        dy = rnd(g, *x.shape)
        yr.backward(dy)
        y = engine.act_fwd(x.cuda(), act)
        close(y, yr, what=f"act{act}")
        dx = engine.act_bwd(y, dy.cuda(), act)
        close(dx, xr.grad, what=f"act{act} bwd")
    xs = [rnd(g, 2, 5, 5, 13) for _ in range(5)]
    close(engine.add_n([t.cuda() for t in xs]), sum(xs), what="add_n")
    close(engine.add_n([t.cuda() for t in xs[:2]], relu=True), torch.relu(xs[0] + xs[1]), what="add relu")
    parts = [rnd(g, 2, 5, 5, c) for c in (8, 20, 3)]
    close(engine.concat([t.cuda() for t in parts]), torch.cat(parts, -1), what="concat")


def test_softmaxes(engine):
    g = torch.Generator().manual_seed(5)
    z = rnd(g, 2, 9, 9, 2) * 6
    zr = z.clone().requires_grad_()
    pr = torch.softmax(zr, -1)
    dp = rnd(g, *z.shape)
    pr.backward(dp)
    p = engine.softmax2_fwd(z.cuda())
    close(p, pr, what="softmax2")
    close(engine.softmax2_bwd(p, dp.cuda()), zr.grad, what="softmax2 bwd")
    zb = rnd(g, 3, 5, 256) * 4
    zbr = zb.clone().requires_grad_()
    pbr = torch.softmax(zbr, 1)
    dpb = rnd(g, *zb.shape)
    pbr.backward(dpb)
    pb = engine.softmax_branch_fwd(zb.cuda())
    close(pb, pbr, what="softmax branch")
    close(engine.softmax_branch_bwd(pb, dpb.cuda()), zbr.grad, what="softmax branch bwd")


@pytest.mark.parametrize("c", [64, 45, 728])
def test_gates(engine, c):
    g = torch.Generator().manual_seed(c)
    n, h, w = 3, 10, 12
    x = rnd(g, n, h, w, c)
    dy = rnd(g, n, h, w, c)
    # channel gate
    gc = rnd(g, n, c)
    xr, gr = x.clone().requires_grad_(), gc.clone().requires_grad_()
    yr = xr * gr.view(n, 1, 1, c)
    yr.backward(dy)
    close(engine.bcast_mul_fwd(x.cuda(), gc.cuda(), 0), yr, what="cgate fwd")
    dx, dg = engine.bcast_mul_bwd(x.cuda(), gc.cuda(), dy.cuda(), 0)
    close(dx, xr.grad, what="cgate dx")
    close(dg, gr.grad, rtol=1e-4, what="cgate dg")
    # spatial gate
    gs = rnd(g, n, h, w, 1)
    xr, gr = x.clone().requires_grad_(), gs.clone().requires_grad_()
    yr = xr * gr
    yr.backward(dy)
    close(engine.bcast_mul_fwd(x.cuda(), gs.cuda(), 1), yr, what="sgate fwd")
    dx, dg = engine.bcast_mul_bwd(x.cuda(), gs.cuda(), dy.cuda(), 1)
    close(dx, xr.grad, what="sgate dx")
    close(dg, gr.grad, rtol=1e-4, what="sgate dg")
    # scSE
    s, cl = rnd(g, n, h, w, 1) * 3, rnd(g, n, c) * 3
    xr, sr, cr = x.clone().requires_grad_(), s.clone().requires_grad_(), cl.clone().requires_grad_()
    yr = xr * torch.sigmoid(sr) + xr * torch.sigmoid(cr).view(n, 1, 1, c)
    yr.backward(dy)
    close(engine.scse_fwd(x.cuda(), s.cuda(), cl.cuda()), yr, what="scse fwd")
    dx, ds, dc = engine.scse_bwd(x.cuda(), s.cuda(), cl.cuda(), dy.cuda())
    close(dx, xr.grad, what="scse dx")
    close(ds, sr.grad, rtol=1e-4, what="scse ds")
    close(dc, cr.grad, rtol=1e-4, what="scse dc")
    # BAM
    mc, ms = rnd(g, n, c) * 2, rnd(g, n, h, w, 1) * 2
    xr, mcr, msr = x.clone().requires_grad_(), mc.clone().requires_grad_(), ms.clone().requires_grad_()
    gate = torch.sigmoid(mcr.view(n, 1, 1, c) + msr)
    yr = gate * xr + xr
    yr.backward(dy)
    close(engine.bam_fwd(x.cuda(), mc.cuda(), ms.cuda()), yr, what="bam fwd")
    dx, dmc, dms = engine.bam_bwd(x.cuda(), mc.cuda(), ms.cuda(), dy.cuda())
    close(dx, xr.grad, what="bam dx")
    close(dmc, mcr.grad, rtol=1e-4, what="bam dmc")
    close(dms, msr.grad, rtol=1e-4, what="bam dms")


@pytest.mark.parametrize("k,stride,padding", [(3, 2, "same"), (2, 2, "valid"), (2, 4, "valid")])
def test_maxpool(engine, k, stride, padding):
    g = torch.Generator().manual_seed(k * 10 + stride)
    x = rnd(g, 2, 16, 20, 24)
    xr = x.clone().requires_grad_()
    yr = T.max_pool(xr, k, stride, padding)
    dy = rnd(g, *yr.shape)
    yr.backward(dy)
    y, geom = engine.maxpool_fwd(x.cuda(), k, stride, padding)
    close(y, yr, what="maxpool fwd")
    dx = engine.maxpool_bwd(x.cuda(), y, dy.cuda(), geom)
    close(dx, xr.grad, what="maxpool bwd")


@pytest.mark.parametrize("k,stride,padding,c", [(3, 2, "same", 24), (2, 2, "valid", 24), (2, 4, "valid", 8), (3, 2, "same", 5)])
def test_maxpool_training_form_routes_by_the_recorded_cell(engine, k, stride, padding, c):
    """sg_maxpool_fwd_idx / sg_maxpool_bwd_idx: same y, and a dx BIT-IDENTICAL to the recomputing backward - also with ties
    (a quantised input: many equal maxima per window, the first in scan order must win) and windows of -inf."""
    g = torch.Generator().manual_seed(k + stride + c)
    x = torch.round(rnd(g, 2, 15, 18, c) * 3) / 3          # few distinct values: ties in most windows
    x[0, :4, :4] = -float("inf")
    xd = x.cuda()
    y0, geom = engine.maxpool_fwd(xd, k, stride, padding)
    y1, geom1, idx = engine.maxpool_fwd(xd, k, stride, padding, want_idx=True)
    assert geom == geom1 and torch.equal(y0, y1)
    assert int(idx.max()) < k * k
    dy = rnd(g, *y0.shape).cuda()
    dx0 = engine.maxpool_bwd(xd, y0, dy, geom)
    dx1 = engine.maxpool_bwd_idx(dy, idx, tuple(x.shape), geom)
    assert torch.equal(dx0, dx1)
    # every window's dy lands exactly once
    assert abs(float(dx1.double().sum()) - float(dy.double().sum())) <= 1e-3
    # directly against the oracle: autograd of T.max_pool in fp64 on the same quantised input - ties in most
    # windows, routed to the first maximum in scan order - without the all(-inf) corner, whose gradient the oracle drops
    xq = (torch.round(rnd(g, 2, 15, 18, c) * 3) / 3)
    xr = xq.double().requires_grad_()
    yr = T.max_pool(xr, k, stride, padding)
    dyq = rnd(g, *yr.shape)
    yr.backward(dyq.double())
    yq, geomq, idxq = engine.maxpool_fwd(xq.cuda(), k, stride, padding, want_idx=True)
    assert torch.equal(yq.cpu().double(), yr.detach())
    dxq = engine.maxpool_bwd_idx(dyq.cuda(), idxq, tuple(xq.shape), geomq)
    close(dxq, xr.grad, rtol=1e-6, what="maxpool_bwd_idx vs oracle")


def test_maxpool_odd_same(engine):
    g = torch.Generator().manual_seed(99)
    x = rnd(g, 1, 15, 17, 8)
    yr = T.max_pool(x, 3, 2, "same")
    y, _ = engine.maxpool_fwd(x.cuda(), 3, 2, "same")
    close(y, yr, what="maxpool odd")


@pytest.mark.parametrize("shape,kh", [((2, 32, 32, 2048), 32), ((2, 64, 64, 64), 64), ((2, 64, 64, 256), 32),
                                       ((3, 12, 12, 45), 12)])
def test_avgpool_gap(engine, shape, kh):
    g = torch.Generator().manual_seed(shape[-1])
    x = rnd(g, *shape)
    xr = x.clone().requires_grad_()
    yr = T.avg_pool(xr, kh)
    dy = rnd(g, *yr.shape)
    yr.backward(dy)
    y = engine.avgpool_fwd(x.cuda(), kh, kh)
    close(y, yr, what="avgpool fwd")
    dx = engine.avgpool_bwd(dy.cuda(), shape, kh, kh)
    close(dx, xr.grad, what="avgpool bwd")


@pytest.mark.parametrize("s", [2, 4, 32])
def test_upsample(engine, s):
    g = torch.Generator().manual_seed(s)
    x = rnd(g, 2, 3, 5, 24)
    xr = x.clone().requires_grad_()
    yr = T.upsample_nearest(xr, s)
    dy = rnd(g, *yr.shape)
    yr.backward(dy)
    close(engine.upsample_fwd(x.cuda(), s), yr, what="up fwd")
    close(engine.upsample_bwd(dy.cuda(), x.shape, s), xr.grad, what="up bwd")


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 96, 3, 1, 2), (2, 32, 32, 64, 128, 1, 2, 1), (1, 16, 16, 256, 40, 3, 1, 1),
                                  (2, 16, 16, 64, 1, 1, 1, 1)],
                         ids=["dilated3x3", "stride2_1x1_parity_rows", "ragged_columns", "thin_1x1_to_1"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_dgrad_adds_a_collected_gradient(engine, case, dtype):
    """sg_conv2d_dgrad_acc: dx = dgrad(dy) + res in the epilogue of the slab kernels (conv_x6_kernel / conv_b16_kernel, also
    with the stride-2 parity-class row order).  fp32: the bits of the dgrad followed by add_n, res may be the output itself;
    bf16: within one rounding of it."""
    n, h, w, cin, cout, k, stride, dil = case
    g = torch.Generator().manual_seed(cin + cout + k)
    x_shape = (n, h, w, cin)
    wt = (rnd(g, k, k, cin, cout) * (1.0 / np.sqrt(k * k * cin))).cuda()
    d = engine.conv_desc(x_shape, cout, k, k, stride, dil, "same")
    dy = rnd(g, n, d.Ho, d.Wo, cout).cuda().to(dtype)
    res = rnd(g, *x_shape).cuda().to(dtype)
    plain = engine.conv2d_dgrad(dy, wt, d)
    ref = engine.add_n([res, plain])
    got = engine.conv2d_dgrad(dy, wt, d, res=res)
    # directly against the oracle: fp64 autograd of the forward convolution, plus res
    xr = torch.zeros(*x_shape, dtype=torch.float64, requires_grad=True)
    T.conv2d(xr, wt.cpu().double(), None, stride, dil, "same").backward(dy.float().cpu().double())
    want = xr.grad + res.float().cpu().double()
    close(got.float(), want, rtol=RTOL if dtype == torch.float32 else 2 ** -7, what="conv2d_dgrad_acc vs oracle")
    if dtype == torch.float32:
        assert torch.equal(got, ref)
        buf = res.clone()
        engine.conv2d_dgrad(dy, wt, d, res=buf, out=buf)
        assert torch.equal(buf, ref)
    else:
        exact = res.float() + plain.float()
        assert float((got.float() - exact).abs().max()) <= 2 ** -7 * float(exact.abs().max())


@pytest.mark.parametrize("case,kind_wanted,same_kernel", [
    # the decoder's 64^2 x 512 -> 256: 256 reduction channels in the dgrad, but the planes-in kernel's shape.  Without `res` the launch
    # takes conv_x6w_kernel, with it conv_x6_kernel (the planes-in kernel has no such epilogue; both read kind-1 planes): same
    # products, another order of summation over K = 2304
    ((1, 64, 64, 512, 256, 3, 1), 1, False),
    ((2, 16, 32, 64, 96, 3, 2), 1, True),       # dilated, 96 reduction channels: the slab kernel with and without `res`
    ((2, 16, 32, 64, 128, 3, 1), 2, None),      # 128 reduction channels in the dgrad: the patch kernel in two chunks
    ((2, 16, 32, 64, 64, 3, 1), 2, None),       # one chunk
    ((6, 32, 32, 728, 728, 1, 1), 3, None),     # the wide pointwise kernel (6144 rows)
], ids=["long_k_256ch", "slab_dilated", "patch_two_chunks", "patch_one_chunk", "wide_pointwise"])
def test_dgrad_kernel_choice_does_not_depend_on_a_collected_gradient(engine, case, kind_wanted, same_kernel):
    """include/segengine.h, sg_conv2d_dgrad_acc: a launch adds `res` exactly when sg_conv2d_planes_job (which knows no `res`) says
    kind 1 for it - the weight planes of a step are laid out from that answer, so the launch must come to the same one with `res` as
    without.  Round 5: x6p_ok asked x6w_plan, which declines a launch that carries `res`; the decoder's 64^2 x 512 -> 256 dgrad then took
    the patch kernel on planes laid out for the slab kernel's family and refused (loudly) in the training step - no op test saw it
    (scripts/regress_kernel_choice.sh rebuilds with that form: `long_k_256ch` then fails with the step's rc=-3, profiles/r05_regress_kernel_choice.txt).
    Kind 1: accepted, and dgrad + res - bit for bit where the launch takes the same kernel with and without `res`, within the
    rounding of two fp32 summation orders where it does not.  Kinds 2 / 3: SG_EUNSUPPORTED and nothing launched (dx untouched)."""
    import ctypes as C
    from building_detection_amd import _lib
    n, h, w, cin, cout, k, dil = case
    g = torch.Generator().manual_seed(7 * cin + cout + k)
    wt = (rnd(g, k, k, cin, cout) * (1.0 / np.sqrt(k * k * cin))).cuda()
    d = engine.conv_desc((n, h, w, cin), cout, k, k, 1, dil, "same")
    job, nbytes = _lib.PlanesJob(), C.c_size_t(0)
    _lib.check(engine.lib.sg_conv2d_planes_job(engine.h, _lib.SG_F32, C.byref(d), 1, C.byref(job), C.byref(nbytes)), "sg_conv2d_planes_job")
    assert int(job.kind) == kind_wanted, f"this case no longer covers kind {kind_wanted} (planes job says {job.kind})"
    dy = rnd(g, n, d.Ho, d.Wo, cout).cuda()
    res = rnd(g, n, h, w, cin).cuda()
    plain = engine.conv2d_dgrad(dy, wt, d)
    if kind_wanted == 1:
        got = engine.conv2d_dgrad(dy, wt, d, res=res)
        ref = engine.add_n([res, plain])
        if same_kernel:
            assert torch.equal(got, ref)
        else:   # |dx| ~ 0.6 here; two correct fp32 sums of 2304 x6 products differ by a few ulp of the largest partial sum
            assert float((got - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
            assert not torch.equal(got, ref), "same bits: this case no longer crosses from the planes-in to the slab kernel"
    else:
        buf = torch.full_like(res, 123.0)
        with pytest.raises(_lib.SgError, match=r"rc=-3"):
            engine.conv2d_dgrad(dy, wt, d, res=res, out=buf)
        torch.cuda.synchronize()
        assert bool((buf == 123.0).all()), "a refused launch wrote into dx"


@pytest.mark.parametrize("pre_relu", [False, True])
def test_depthwise_dgrad_adds_a_collected_gradient(engine, pre_relu):
    """sg_dwconv2d_dgrad_acc: dx = dgrad(dy) [masked by x > 0] + res inside the kernel - the bits of the dgrad followed by
    add_n; res may be the output buffer itself."""
    g = torch.Generator().manual_seed(5 + pre_relu)
    n, h, w, c = 2, 12, 16, 728
    x, dy, res = rnd(g, n, h, w, c).cuda(), rnd(g, n, h, w, c).cuda(), rnd(g, n, h, w, c).cuda()
    wt = rnd(g, 3, 3, c, 1).cuda()
    d = engine.conv_desc((n, h, w, c), c, 3, 3, 1, 1, "same")
    assert engine.dwconv_dgrad_acc_ok(d)
    ref = engine.add_n([res, engine.dwconv_dgrad(dy, wt, d, x=x, pre_relu=pre_relu)])
    got = engine.dwconv_dgrad(dy, wt, d, x=x, pre_relu=pre_relu, res=res)
    assert torch.equal(got, ref)
    # directly against the oracle: fp64 autograd of depthwise(relu?(x)), plus res
    xr = x.cpu().double().requires_grad_()
    T.depthwise_conv2d(torch.relu(xr) if pre_relu else xr, wt.cpu().double(), 1, "same").backward(dy.cpu().double())
    close(got, xr.grad + res.cpu().double(), what="dwconv2d_dgrad_acc vs oracle")
    buf = res.clone()
    engine.dwconv_dgrad(dy, wt, d, x=x, pre_relu=pre_relu, res=buf, out=buf)
    assert torch.equal(buf, ref)
    assert not engine.dwconv_dgrad_acc_ok(engine.conv_desc((n, h, w, c), c, 3, 3, 2, 1, "same"))


@pytest.mark.parametrize("case", [("aspp_fwd_two_shares", 2, 32, 32, 2048, 256, 6), ("aspp_like_d18_b3", 3, 32, 32, 1024, 256, 18),
                                  ("whole_k_two_column_tiles", 1, 64, 64, 256, 512, 2), ("plain_3x3_decoder_256", 1, 64, 64, 256, 256, 1),
                                  ("plain_3x3_512_two_shares_odd_stage_count", 2, 32, 32, 480, 192, 1)], ids=lambda c: c[0])
def test_planes_in_x6_kernel(engine, case):
    """conv_x6w.h (round 4): the long-K multi-tap fp32 convolutions (dilated or not: SG_X6_WIDE=2) with the activation split once
    into bf16 planes and both operands by LDS-DMA.  Forward (two K shares for the ASPP shape: statistics from the reduction's registers; whole K for a
    64 x 64 map: statistics from the accumulators), dgrad and the per-128-row-tile BatchNormalization statistics against
    the fp64 oracle at the x6 kernels' tolerance; an image's result does not depend on its batch and repeats run to run, bit
    for bit."""
    name, N, H, W, Cin, Cout, dil = case
    g = torch.Generator().manual_seed(Cin + Cout + dil)
    x = rnd(g, N + 1, H, W, Cin)
    w = rnd(g, 3, 3, Cin, Cout) * (1.0 / np.sqrt(9 * Cin))
    b = rnd(g, Cout) * 0.1
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    d = engine.conv_desc(tuple(x.shape), Cout, 3, 3, 1, dil, "same")
    y, st = engine.conv2d_fwd(xd, wd, bd, desc=d, want_stats=True)
    assert st is not None
    stats, tiles = st
    assert tiles == (N + 1) * H * W // 128
    xr = x.double().requires_grad_()
    yr = T.conv2d(xr, w.double(), b.double(), 1, dil, "same")
    close(y, yr.detach(), what=f"{name} fwd")
    tr = yr.detach().reshape(tiles, 128, Cout)
    sv = stats.view(tiles, 2, Cout).double().cpu()
    assert float((sv[:, 0] - tr.sum(1)).abs().max()) <= 2e-5 * float(tr.abs().sum(1).max())
    q_ref = ((tr - tr.mean(1, keepdim=True)) ** 2).sum(1)
    assert float((sv[:, 1] - q_ref).abs().max()) <= 1e-4 * float(q_ref.max())
    dy = rnd(g, *yr.shape)
    yr.backward(dy.double())
    dx = engine.conv2d_dgrad(dy.cuda(), wd, d)
    close(dx, xr.grad, what=f"{name} dgrad")
    y2, _ = engine.conv2d_fwd(xd, wd, bd, desc=d, want_stats=True)
    assert torch.equal(y, y2) and torch.equal(dx, engine.conv2d_dgrad(dy.cuda(), wd, d))
    d1 = engine.conv_desc((1, H, W, Cin), Cout, 3, 3, 1, dil, "same")
    y1 = engine.conv2d_fwd(xd[N:N + 1].contiguous(), wd, bd, desc=d1)
    assert torch.equal(y1[0], y[N]), "an image's result depends on its batch"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bn_relu,pre_relu,with_res,shape", [(True, False, False, (2, 12, 16, 728)), (False, True, True, (3, 8, 8, 128)),
                                                          (False, False, False, (2, 6, 20, 36)), (True, False, True, (1, 64, 32, 64))])
def test_depthwise_dgrad_sums_the_batchnorm_backward(engine, bn_relu, pre_relu, with_res, shape, dtype):
    """sg_dwconv2d_dgrad_bnsums + sg_bn_train_bwd_apply (round 4): z = BatchNormalization[+ReLU](y) in training mode, t =
    depthwise3x3([relu](z)).  Given dt, the depthwise dgrad writes dz (+ res) AND the BatchNormalization's dgamma / dbeta; the
    apply pass then gives dy.  Against fp64 autograd of the oracle's layers (tfops.batch_norm / depthwise_conv2d) and against the
    unfused engine path (sg_dwconv2d_dgrad -> sg_bn_train_bwd: same values up to the order of the column sums).  Shapes: the
    middle flow's, a pre-activation ReLU in the gather with a collected gradient riding along, a ragged channel count
    (36: idle lanes in the channel dimension), a large map whose partial rows hit the cap of the finalize launch."""
    n, h, w, c = shape
    g = torch.Generator().manual_seed(c + h)
    y = rnd(g, n, h, w, c) * 2
    gamma, beta = torch.rand(c, generator=g) + 0.5, rnd(g, c) * 0.3
    wt = rnd(g, 3, 3, c, 1)
    dt = rnd(g, n, h, w, c)
    res = rnd(g, n, h, w, c) if with_res else None
    if dtype == torch.bfloat16:   # the engine sees bf16 tensors: the reference starts from the same rounded values
        y, dt = y.bfloat16().float(), dt.bfloat16().float()
        res = res.bfloat16().float() if with_res else None
    # oracle, fp64
    yr, gr, br = y.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    z, _, _ = T.batch_norm(yr, gr, br, torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64), True)
    z = torch.relu(z) if bn_relu else z
    z.retain_grad()
    t = T.depthwise_conv2d(torch.relu(z) if pre_relu else z, wt.double(), 1, "same")
    extra = (z * res.double()).sum() if with_res else 0.0      # a second consumer of z whose gradient is `res`
    (((t * dt.double()).sum()) + extra).backward()
    # engine
    yd, dtd = y.cuda().to(dtype), dt.cuda().to(dtype)
    gd, bd, wd = gamma.cuda(), beta.cuda(), wt.cuda()
    zd, mean, invstd = engine.bn_train_fwd(yd, gd, bd, torch.zeros(c).cuda(), torch.ones(c).cuda(), relu=bn_relu)
    d = engine.conv_desc((n, h, w, c), c, 3, 3, 1, 1, "same")
    assert engine.dwconv_dgrad_acc_ok(d)
    resd = res.cuda().to(dtype) if with_res else None
    dgam, dbet = torch.full((c,), 7.0).cuda(), torch.full((c,), 7.0).cuda()
    dz = engine.dwconv_dgrad_bnsums(dtd, wd, d, yd, mean, invstd, gd, bd, bn_relu, dgam, dbet, x=zd, pre_relu=pre_relu, res=resd)
    dy = engine.bn_train_bwd_apply(yd, dz, gd, bd, mean, invstd, dgam, dbet, relu=bn_relu)
    # the unfused engine path
    dz0 = engine.dwconv_dgrad(dtd, wd, d, x=zd, pre_relu=pre_relu, res=resd)
    dy0, dg0, db0 = engine.bn_train_bwd(yd, zd, dz0, gd, mean, invstd, relu=bn_relu, beta=bd)
    assert torch.equal(dz, dz0), "the gradient tensor itself must not change"
    tol = RTOL if dtype == torch.float32 else 2 ** -7
    # (bf16 storage: the fused sums see the gradient before its rounding to bf16, the unfused reduction reads it rounded)
    close(dgam, dg0, rtol=1e-5 if dtype == torch.float32 else 2 ** -7, what="dgamma vs the unfused path")
    close(dbet, db0, rtol=1e-5 if dtype == torch.float32 else 2 ** -7, what="dbeta vs the unfused path")
    close(dy.float(), dy0.float(), rtol=1e-5 if dtype == torch.float32 else 2 ** -7, what="dy vs the unfused path")
    close(dgam, gr.grad, rtol=tol, what="dgamma vs oracle")
    close(dbet, br.grad, rtol=tol, what="dbeta vs oracle")
    close(dy.float(), yr.grad, rtol=tol, what="dy vs oracle")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("which", ["a", "b", "ab"])
def test_add2_bn_equals_batchnorm_then_add(engine, which, dtype):
    """sg_add2_bn: the residual add applies the BatchNormalization of its operand(s) while it sums.  fp32 storage: the bits of
    bn (training statistics / inference moving statistics) followed by add_n(+ReLU); bf16 storage skips the rounding of the
    normalised tensor, so it is compared with the fp32 result at bf16 resolution."""
    g = torch.Generator().manual_seed(len(which) + (0 if dtype == torch.float32 else 7))
    n, h, w, c = 3, 6, 10, 728
    a, b = rnd(g, n, h, w, c).cuda(), rnd(g, n, h, w, c).cuda()
    par = [tuple(t.cuda() for t in (rnd(g, c), torch.rand(c, generator=g) + 0.5, torch.rand(c, generator=g) + 0.5, rnd(g, c)))
           for _ in range(2)]   # (mean, invstd or variance, gamma, beta)
    for infer in (False, True):
        for relu in (False, True):
            def norm(x, p):
                mean, iv, gamma, beta = p
                if infer:
                    return engine.bn_infer(x, gamma, beta, mean, iv, eps=1e-3)
                y = torch.empty_like(x)
                from building_detection_amd.ops import _ptr, _dt, check
                check(engine.lib.sg_bn_apply(engine.h, engine.stream, _dt(x), x.numel() // c, c, _ptr(x), _ptr(gamma), _ptr(beta),
                                             _ptr(mean), _ptr(iv), _ptr(y), 0), "sg_bn_apply")
                return y
            ref = engine.add_n([norm(a, par[0]) if "a" in which else a, norm(b, par[1]) if "b" in which else b], relu=relu)
            ad, bd = a.to(dtype), b.to(dtype)
            got = engine.add2_bn(ad, bd, par[0] if "a" in which else None, par[1] if "b" in which else None, relu=relu,
                                 infer=infer, eps=1e-3)
            if dtype == torch.float32 and which == "ab":   # an operand's own fused ReLU (res34.py: activation before the add)
                refr = engine.add_n([torch.relu(norm(a, par[0])), norm(b, par[1])], relu=relu)
                gotr = engine.add2_bn(a, b, par[0], par[1], relu=relu, infer=infer, eps=1e-3, relu_a=True)
                assert torch.equal(gotr, refr), ("relu_a", infer, relu)
            # directly against the oracle's arithmetic, fp64: f(x) = (x - mean) * invstd * gamma + beta with
            # invstd given (training) or rsqrt(variance + eps) (inference)
            def norm64(x, p):
                mean, iv, gamma, beta = [t.cpu().double() for t in p]
                inv = torch.rsqrt(iv + 1e-3) if infer else iv
                return (x.float().cpu().double() - mean) * inv * gamma + beta
            want = (norm64(ad, par[0]) if "a" in which else ad.float().cpu().double()) + \
                   (norm64(bd, par[1]) if "b" in which else bd.float().cpu().double())
            want = torch.relu(want) if relu else want
            close(got.float(), want, rtol=RTOL if dtype == torch.float32 else 2 ** -7, what=f"add2_bn vs oracle {which} {infer} {relu}")
            if dtype == torch.float32:
                assert torch.equal(got, ref), (which, infer, relu)
            else:
                ref16 = engine.add_n([norm(ad.float(), par[0]) if "a" in which else ad.float(),
                                      norm(bd.float(), par[1]) if "b" in which else bd.float()], relu=relu)
                assert float((got.float() - ref16).abs().max()) <= 2 ** -7 * float(ref16.abs().max())


def test_upsample_bwd_large_window_with_pixel_stride_and_accumulate(engine):
    """The ASPP image-pooling case (1x1 -> HxW): the window kernel, rectangular windows, dy read out of a wider buffer
    (channel slice of a concat gradient), added to an existing dx."""
    g = torch.Generator().manual_seed(11)
    n, h, w, c, sh, sw, ld = 3, 1, 2, 40, 8, 16, 56
    wide = rnd(g, n, h * sh, w * sw, ld)
    dx0 = rnd(g, n, h, w, c)
    dy = wide[..., 8:8 + c]
    ref = dx0 + dy.reshape(n, h, sh, w, sw, c).sum(dim=(2, 4))
    wd = wide.cuda()
    out = dx0.cuda().clone()
    engine.upsample_bwd(wd.view(-1)[8:], (n, h, w, c), sh, sw, out=out, accumulate=True, dy_ld=ld)
    close(out, ref, what="up bwd window")


def _ref_loss(kind, y_true, p):
    eps = 1e-7
    y = y_true[..., :2]
    if kind == 0:
        l = y * torch.log(p + eps)
    elif kind == 1:
        l = torch.tensor([0.5, 0.5]) * y * (1 - p) * (1 - p) * torch.log(p + eps)
    else:
        l = torch.tensor([0.35, 0.65]) * y_true[..., 2:] * y * (1 - p) * (1 - p) * torch.log(p + eps)
    return -(l[..., 0] + l[..., 1]).mean()


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_loss_and_metrics(engine, kind):
    g = torch.Generator().manual_seed(kind)
    n, h, w = 2, 24, 24
    z = rnd(g, n, h, w, 2) * 4
    p = torch.softmax(z, -1)
    m = (torch.rand(n, h, w, generator=g) > 0.6).float()
    yt = torch.stack([1 - m, m, 1 + (torch.rand(n, h, w, generator=g) > 0.8).float(),
                      1 + (torch.rand(n, h, w, generator=g) > 0.8).float()], -1)
    pr = p.clone().requires_grad_()
    lr = _ref_loss(kind, yt, pr)
    lr.backward()
    pd, yd = p.cuda(), yt.cuda()
    close(engine.loss_fwd(kind, pd, yd), lr.reshape(1), what="loss")
    close(engine.loss_bwd(kind, pd, yd), pr.grad, what="loss bwd")
    cnt = engine.confusion_counts(pd, yd).cpu()
    pred = (p[..., 1] > p[..., 0]).long()
    tru = m.long()
    exp = torch.tensor([(pred * tru).sum(), ((1 - pred) * (1 - tru)).sum(), (pred * (1 - tru)).sum(),
                        ((1 - pred) * tru).sum()])
    assert torch.equal(cnt, exp)


def test_adam(engine):
    g = torch.Generator().manual_seed(1)
    n = 100003
    w, m, v, gr = rnd(g, n), rnd(g, n) * 0.1, rnd(g, n, lo=0, hi=0.01), rnd(g, n)
    wd, md, vd = w.cuda(), m.cuda(), v.cuda()
    t, lr, b1, b2, eps = 7, 3e-4, 0.9, 0.999, 1e-7
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    engine.adam_step(wd, md, vd, gr.cuda(), lr_t)
    m2 = b1 * m + (1 - b1) * gr
    v2 = b2 * v + (1 - b2) * gr * gr
    w2 = w - lr_t * m2 / (v2.sqrt() + eps)
    close(md, m2, what="adam m")
    close(vd, v2, what="adam v")
    close(wd, w2, what="adam w")


def test_edge_labels_match_generator(engine):
    """sg_edge_labels (separable row / column min-max through LDS) vs the ORACLE's restatement of train_data_gen's label
    channels (oracle/input_pipeline.py:label_channels, DeepLabv3plus.py:70-100): bit-exact, on rectangles touching every
    border, on tiles that are not multiples of the kernel's 16 x 64 workgroup tile, on grey (non-binary) label values, for
    the reference's 5 iterations and other radii (incl. the plain window-walk kernel behind radius > 8)."""
    from building_detection_amd.data import synthetic_batch
    from oracle import input_pipeline as OIP
    _, y = synthetic_batch(3, 96, 80, seed=4)
    labs = np.ascontiguousarray(y[..., 1])
    labs[0, :7, :9] = 1; labs[0, -4:, -11:] = 1; labs[1, 40:, :3] = 1          # all four borders
    labs[2, 10:20, 10:20] = np.float32(128 / 255)   # anti-aliased values: background for to_categorical, no weight-2 band
    labs[2, 50, 60] = np.float32(254 / 255)
    got = engine.edge_labels(torch.from_numpy(labs).cuda()).cpu().numpy()
    for k in range(3):
        assert np.array_equal(got[k].astype(np.float64), OIP.label_channels(labs[k])), k
    assert (got[2, 10:20, 10:20, 1] == 0).all() and (got[2, 10:20, 10:20, 0] == 1).all()
    rng = np.random.default_rng(9)
    for (h, w, it) in [(17, 65, 5), (1, 1, 5), (33, 130, 2), (16, 64, 8), (40, 70, 0), (45, 50, 11)]:
        lab = (rng.random((2, h, w)) > 0.6).astype(np.float32)
        lab[0, h // 2:, : w // 2] = 1.0
        g = engine.edge_labels(torch.from_numpy(lab).cuda(), iterations=it).cpu().numpy()
        for k in range(2):
            assert np.array_equal(g[k].astype(np.float64), OIP.label_channels(lab[k], it)), (h, w, it, k)
    # the full-size case of the generator: 512 x 512 (bs 16 in one launch), against the oracle on two of the images
    _, yf = synthetic_batch(16, 512, 512, seed=12)
    lf = np.ascontiguousarray(yf[..., 1])
    gf = engine.edge_labels(torch.from_numpy(lf).cuda()).cpu().numpy()
    for k in (0, 15):
        assert np.array_equal(gf[k].astype(np.float64), OIP.label_channels(lf[k])), k


@pytest.mark.parametrize("shape", [(2, 300, 400, 3), (1, 1024, 1024, 3), (3, 700, 333), (2, 512, 512, 3), (1, 64, 48, 3),
                                   (2, 513, 511), (1, 2048, 1536, 3)])
def test_resize_linear_u8_matches_the_oracles_cv_resize(engine, shape):
    """sg_resize_linear_u8 = cv.resize(img, (512, 512)) in OpenCV's fixed-point arithmetic (decode_img / decode_lbel of
    DeepLabv3plus.py:35,45 for tiles that are not 512 x 512), bit-exact against oracle/input_pipeline.py:resize_linear_u8:
    up- and down-scaling, non-square, one and three channels, the exact-2x INTER_AREA substitution, the identity."""
    from oracle import input_pipeline as OIP
    rng = np.random.default_rng(sum(shape))
    a = rng.integers(0, 256, size=shape, dtype=np.uint8)
    got = engine.resize_linear_u8(torch.from_numpy(a).cuda(), 512, 512).cpu().numpy()
    for k in range(shape[0]):
        ref = OIP.resize_linear_u8(a[k])
        assert got[k].shape == ref.shape and np.array_equal(got[k], ref), (shape, k, int(np.abs(got[k].astype(int) - ref).max()))
    small = engine.resize_linear_u8(torch.from_numpy(a[:1]).cuda(), 37, 91).cpu().numpy()[0]   # another target size
    assert np.array_equal(small, OIP.resize_linear_u8(a[0], (91, 37)))


def test_inference_tail(engine):
    g = torch.Generator().manual_seed(2)
    p = torch.softmax(rnd(g, 1, 16, 16, 2) * 3, -1)
    p[0, 0, 0] = torch.tensor([0.5, 0.5])  # tie -> class 0
    canvas = torch.zeros(40, 40, dtype=torch.int8, device="cuda")
    engine.argmax_accumulate(p.cuda(), canvas, 5, 7)
    engine.argmax_accumulate(p.cuda(), canvas, 10, 7)
    ref = np.zeros((40, 40), np.int8)
    mask = (p[0, ..., 1] > p[0, ..., 0]).numpy().astype(np.int8)
    ref[5:21, 7:23] += mask
    ref[10:26, 7:23] += mask
    assert np.array_equal(canvas.cpu().numpy(), ref)
    assert canvas[5, 7].item() == 0
    masks = [(torch.rand(50, 60, generator=g) > 0.5).to(torch.uint8) * 255 for _ in range(5)]
    out = engine.vote_ge([m.cuda() for m in masks], 3).cpu()
    exp = ((sum((m // 255).int() for m in masks) >= 3).to(torch.uint8) * 255)
    assert torch.equal(out, exp)


def test_golden_ops(engine):
    """The HIP ops against the committed op-level fixture tests/golden/ops.npz (made by tests/golden/make_golden.py
    from the CPU oracle): the TF-semantics corner cases of SURVEY App. B, each through the C ABI."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ops.npz"))
    c = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).cuda()  # noqa: E731
    x, w, b = c("conv_x"), c("conv_w"), c("conv_b")
    close(engine.conv2d_fwd(x, w, b, 2, 1, "same"), torch.from_numpy(g["conv_s2_even"]), what="3x3 s2 even size: pad (0,1)")
    close(engine.conv2d_fwd(x, w, b, 1, 2, "same"), torch.from_numpy(g["conv_d2"]), what="3x3 dilation 2")
    close(engine.conv2d_fwd(x, w[1:2, 1:2].contiguous(), None, 2, 1, "same"), torch.from_numpy(g["conv_1x1_s2"]), what="1x1 s2")
    dw, pw = c("sep_dw"), c("sep_pw")
    for stride, key in ((1, "sep_y"), (2, "sep_y_s2")):
        t = engine.dwconv_fwd(x, dw, stride)
        close(engine.conv2d_fwd(t, pw, b[:5].contiguous(), 1, 1, "same"), torch.from_numpy(g[key]), what=f"SeparableConv2D s{stride}")
    n, h, wd, cin = x.shape
    for k, wk, yk, bias in ((3, "convT_w3", "convT_k3", b[:5].contiguous()), (2, "convT_w2", "convT_k2", None)):
        d = engine.conv_desc((n, 2 * h, 2 * wd, 5), cin, k, k, 2, 1, "same")
        close(engine.conv2d_dgrad(x, c(wk), d, bias=bias), torch.from_numpy(g[yk]), what=f"Conv2DTranspose k{k} s2")
    gam, bet = c("bn_gamma"), c("bn_beta")
    mm, mv = torch.zeros(4).cuda(), torch.ones(4).cuda()
    y, _, _ = engine.bn_train_fwd(x, gam, bet, mm, mv)
    close(y, torch.from_numpy(g["bn_train_y"]), what="BN train y")
    close(mm, torch.from_numpy(g["bn_new_mean"]), what="BN moving mean")
    close(mv, torch.from_numpy(g["bn_new_var"]), what="BN moving var (4-D: unbiased)")
    mm2, mv2 = torch.zeros(4).cuda(), torch.ones(4).cuda()
    y2, _, _ = engine.bn_train_fwd(c("bn2_x"), gam, bet, mm2, mv2)
    close(y2, torch.from_numpy(g["bn2_train_y"]), what="BN 2-D train y")
    close(mv2, torch.from_numpy(g["bn2_new_var"]), what="BN moving var (2-D: biased)")
    close(engine.bn_infer(x, gam, bet, c("bn_imean"), c("bn_ivar")), torch.from_numpy(g["bn_infer_y"]), what="BN inference")
    close(engine.maxpool_fwd(x, 3, 2, "same")[0], torch.from_numpy(g["maxpool_3s2_same"]), what="maxpool 3x3 s2 same")
    close(engine.maxpool_fwd(x, 2, 4, "valid")[0], torch.from_numpy(g["maxpool_2s4"]), what="maxpool 2x2 s4")
    close(engine.maxpool_fwd(x, 2, 2, "valid")[0], torch.from_numpy(g["maxpool_2s2"]), what="maxpool 2x2 s2")
    close(engine.avgpool_fwd(x, 4, 4), torch.from_numpy(g["avgpool_4"]), what="avgpool 4")
    close(engine.upsample_fwd(x, 2), torch.from_numpy(g["up_2"]), what="upsample x2")
    yt, yp = c("loss_y_true").float(), c("loss_y_pred")
    for kind in range(3):
        got = float(engine.loss_fwd(kind, yp, yt).item())
        assert abs(got - float(g["loss_values"][kind])) <= 2e-6 * abs(float(g["loss_values"][kind])), (kind, got)
    assert engine.confusion_counts(yp, yt).cpu().tolist() == g["confusion"].tolist()
    p, m, v = c("adam_p0").clone(), torch.zeros(16).cuda(), torch.zeros(16).cuda()
    for step in (1, 2, 3):
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        engine.adam_step(p, m, v, c("adam_g0") * step, lr_t)
    close(p, torch.from_numpy(g["adam_p3"]), what="Adam params after 3 steps")
    close(m, torch.from_numpy(g["adam_m3"]), what="Adam m")
    close(v, torch.from_numpy(g["adam_v3"]), what="Adam v")


X6_CASES = [
    # n, h, w, cin, cout, k, dil
    (2, 32, 32, 1024, 256, 3, 6),    # ASPP-like: K = 9216
    (4, 32, 32, 728, 728, 1, 1),     # middle-flow pointwise, ragged K and N
    (1, 64, 64, 64, 32, 3, 1),       # entry-flow: tile spanning two taps in wgrad
]


@pytest.mark.parametrize("case", X6_CASES, ids=["aspp", "pw728", "c64"])
def test_x6_at_least_as_accurate_as_native_fp32_mfma(engine, case):
    """Both convolution paths on the same inputs against an fp64 reference: the six-pass bf16 split ("x6") must be
    no less accurate than the native fp32 MFMA kernels it replaces (and both within the 2e-5 of the op tests)."""
    n, h, w, cin, cout, k, dil = case
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = rnd(g, n, h, w, cin)
    x = x * torch.exp(2 * rnd(g, n, h, w, cin))              # activations with a few orders of magnitude of range
    wt = rnd(g, k, k, cin, cout) * (1.0 / np.sqrt(k * k * cin))
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    yr = T.conv2d(xr, wr, None, 1, dil, "same")
    dy = rnd(g, *yr.shape)
    yr.backward(dy.double())
    refs = {"fwd": yr.detach(), "dgrad": xr.grad, "wgrad": wr.grad}
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    d = engine.conv_desc(x.shape, cout, k, k, 1, dil, "same")
    errs = {}
    prev = engine.set_conv_x6(True)
    try:
        for on in (True, False):
            engine.set_conv_x6(on)
            got = {"fwd": engine.conv2d_fwd(xd, wd, None, 1, dil, "same"), "dgrad": engine.conv2d_dgrad(dyd, wd, d),
                   "wgrad": engine.conv2d_wgrad(xd, dyd, d, want_bias=False)[0]}
            for key, val in got.items():
                ref = refs[key]
                errs[(on, key)] = float((val.cpu().double() - ref).abs().max() / ref.abs().max())
    finally:
        engine.set_conv_x6(prev)
    for key in ("fwd", "dgrad", "wgrad"):
        ex, en = errs[(True, key)], errs[(False, key)]
        print(f"{key}: max rel err x6 {ex:.2e}  native fp32 MFMA {en:.2e}")
        assert ex <= 2e-5 and en <= 2e-5
        assert ex <= 1.5 * en + 2e-8, (key, ex, en)


@pytest.mark.parametrize("shape", [(2, 32, 32, 256, 728, 1), (3, 20, 32, 64, 96, 3), (1, 32, 32, 128, 40, 3),
                                   (2, 24, 32, 64, 32, 3), (1, 16, 48, 32, 64, 3), (2, 16, 32, 128, 64, 3), (1, 16, 16, 256, 128, 3)],
                         ids=["pw_728", "ragged_rows", "cout40", "patch_c64_n32", "patch_c32_n64", "patch_chunks_128_64", "patch_chunks_256_128"])
def test_conv_epilogue_bn_statistics(engine, shape):
    """SG_EPI bn_stats: the per-tile (sum, centred sum of squares) a convolution leaves for the following
    BatchNormalization give the same normalised output, saved statistics and moving statistics as BN's own pass."""
    n, h, w, cin, cout, k = shape
    g = torch.Generator().manual_seed(cout)
    x = rnd(g, n, h, w, cin).cuda()
    wt = (rnd(g, k, k, cin, cout) * (1.0 / np.sqrt(k * k * cin))).cuda()
    b = (rnd(g, cout) * 3).cuda()                      # a bias that shifts the channel means away from zero
    gam, bet = (rnd(g, cout) + 1.5).cuda(), rnd(g, cout).cuda()
    y, st = engine.conv2d_fwd(x, wt, b, 1, 1, "same", want_stats=True)
    assert st is not None, "this shape is on the x6 path and must produce statistics"
    mm1, mv1 = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    mm2, mv2 = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    z1, mean1, inv1 = engine.bn_train_fwd(y, gam, bet, mm1, mv1, relu=True)
    z2, mean2, inv2 = engine.bn_train_fwd_from_tiles(y, st[0], st[1], gam, bet, mm2, mv2, relu=True)
    close(mean2, mean1, rtol=2e-6, what="mean")
    close(inv2, inv1, rtol=2e-6, what="inv-std")
    close(mm2, mm1, rtol=2e-6, what="moving mean")
    close(mv2, mv1, rtol=2e-6, what="moving variance")
    close(z2, z1, rtol=5e-6, what="normalised output")
    # and against the oracle in float64
    yr = T.conv2d(x.cpu().double(), wt.cpu().double(), b.cpu().double(), 1, 1, "same")
    zr, _, _ = T.batch_norm(yr, gam.cpu().double(), bet.cpu().double(), torch.zeros(cout).double(), torch.ones(cout).double(), True)
    close(z2, torch.relu(zr), rtol=2e-5, what="conv -> BN -> ReLU vs fp64 oracle")


@pytest.mark.parametrize("shape", [(2, 16, 32), (3, 8, 16), (1, 24, 48)], ids=["two_tiles_per_image", "one_tile", "ragged_grid_3x3"])
def test_upsampling_fused_into_the_3x3_convolution(engine, shape):
    """UpSampling2D(2) -> Conv2D(32, 3, 'same') on 64 channels (train_model/DeepLabv3plus.py:476-477) on the fused kernels
    (csrc/conv_x6p.h, round 5): the sub-pixel forward (SG_PRO_UP2: four phases from the 2 x 2 source pixels each sees, the
    kernel's taps summed beforehand, statistics for the following BatchNormalization), the dgrad that adds the 2 x 2 cells in its
    epilogue (SG_EPI_DOWN2) and the filter gradient that gathers the source at (h >> 1, w >> 1) (SG_X_UP2).  Against the fp64
    oracle AND against the unfused pair of engine launches: forward within rounding (the summed taps round differently), both
    gradients BIT-identical (same products, same order of additions)."""
    n, hs, ws_ = shape   # source map; the convolution runs on (2 hs) x (2 ws)
    cin, cout = 64, 32
    g = torch.Generator().manual_seed(hs * 131 + ws_)
    x = rnd(g, n, hs, ws_, cin)
    wt = rnd(g, 3, 3, cin, cout) * (1.0 / np.sqrt(9 * cin))
    b = rnd(g, cout) * 3
    dy = rnd(g, n, 2 * hs, 2 * ws_, cout)
    xd, wd, bd, dyd = x.cuda(), wt.cuda(), b.cuda(), dy.cuda()
    d = engine.conv_desc((n, 2 * hs, 2 * ws_, cin), cout, 3, 3, 1, 1, "same")
    assert engine.conv2d_up2_ok(d), "this geometry is the fused kernels' own"
    # fp64 oracle through the materialised up-sampling
    xr, wr, br = x.double().requires_grad_(), wt.double().requires_grad_(), b.double().requires_grad_()
    yr = T.conv2d(T.upsample_nearest(xr, 2), wr, br, 1, 1, "same")
    yr.backward(dy.double())
    # the unfused engine pair
    up = engine.upsample_fwd(xd, 2)
    y_un = engine.conv2d_fwd(up, wd, bd, desc=d)
    dx_un = engine.upsample_bwd(engine.conv2d_dgrad(dyd, wd, d), tuple(x.shape), 2)
    dw_un, db_un = engine.conv2d_wgrad(up, dyd, d)
    # fused
    y, st = engine.conv2d_fwd(xd, wd, bd, desc=d, want_stats=True, up2=True)
    assert st is not None and st[1] == n * 4 * hs * ws_ // 128
    close(y, yr, what="sub-pixel forward vs fp64")
    close(y, y_un, rtol=1e-5, what="sub-pixel forward vs the unfused pair")
    e64 = (y.double().cpu() - yr.detach()).abs().max().item(), (y_un.double().cpu() - yr.detach()).abs().max().item()
    # (the summed taps are rounded to fp32 once before the three-way split: one more 2^-24 on the kernel, measured 8e-7 against
    # the unfused pair's 4.5e-7 of |y|_max ~ 4 on the first shape - both far inside the 2e-5 bar above)
    assert e64[0] <= 3 * e64[1] + 5e-7, f"sub-pixel form {e64[0]:.2e} from fp64, unfused {e64[1]:.2e}"
    dx = engine.conv2d_dgrad(dyd, wd, d, down2=True)
    assert tuple(dx.shape) == tuple(x.shape)
    close(dx, xr.grad, what="dgrad with the 2x2 sum vs fp64")
    assert torch.equal(dx, dx_un), "dgrad + up-sampling backward fused: not the bits of the unfused pair"
    dw, db = engine.conv2d_wgrad(xd, dyd, d, x_up2=True)
    close(dw, wr.grad, what="filter gradient from the source vs fp64")
    assert torch.equal(dw, dw_un) and torch.equal(db, db_un), "filter gradient gathered from the source: not the unfused bits"
    # the statistics the epilogue leaves (one tile per phase and source tile) = BatchNormalization's own pass over y
    gam, bet = (rnd(g, cout) + 1.5).cuda(), rnd(g, cout).cuda()
    mm1, mv1 = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    mm2, mv2 = torch.zeros(cout).cuda(), torch.ones(cout).cuda()
    z1, mean1, inv1 = engine.bn_train_fwd(y, gam, bet, mm1, mv1, relu=True)
    z2, mean2, inv2 = engine.bn_train_fwd_from_tiles(y, st[0], st[1], gam, bet, mm2, mv2, relu=True)
    close(mean2, mean1, rtol=2e-6, what="mean")
    close(inv2, inv1, rtol=2e-6, what="inv-std")
    close(mv2, mv1, rtol=2e-6, what="moving variance")
    close(z2, z1, rtol=5e-6, what="normalised output")
    # ReLU in the epilogue (a Conv2D(activation='relu') behind an up-sampling), no bias
    close(engine.conv2d_fwd(xd, wd, None, desc=d, relu=True, up2=True), torch.relu(T.conv2d(T.upsample_nearest(x.double(), 2), wt.double(), None, 1, 1, "same")),
          what="sub-pixel forward with ReLU")
    # batch-slice invariance, bit for bit: an image's result does not depend on the batch it travels in
    if n > 1:
        d1 = engine.conv_desc((1, 2 * hs, 2 * ws_, cin), cout, 3, 3, 1, 1, "same")
        assert torch.equal(engine.conv2d_fwd(xd[1:2].contiguous(), wd, bd, desc=d1, up2=True)[0], y[1])
    # a geometry the fused kernels do not cover is refused loudly, never computed some other way
    from building_detection_amd._lib import SgError
    d_bad = engine.conv_desc((n, 2 * hs, 2 * ws_, cin), cout, 3, 3, 1, 2, "same")
    assert not engine.conv2d_up2_ok(d_bad)
    with pytest.raises(SgError):
        engine.conv2d_fwd(xd, wd, bd, desc=d_bad, up2=True)


@pytest.mark.parametrize("case", [(2, 32, 32, 2048, 256, 3, 6), (3, 32, 32, 1024, 256, 3, 18), (1, 64, 64, 512, 256, 3, 1), (2, 32, 64, 256, 136, 3, 2),
                                  (2, 32, 32, 264, 40, 3, 1), (2, 32, 32, 1280, 256, 1, 1)],
                         ids=["aspp_d6", "aspp_like_d18_b3", "decoder_512", "w64_cout136_d2", "ragged_c264_cout40", "pointwise_1280"])
def test_activation_planes_handed_in(engine, case):
    """Round 5: sg_split_planes + sg_conv2d_fwd_stats_ap / _dgrad_ap / sg_conv2d_wgrad_planes.  The long-K fp32 convolutions read
    their activation as three bf16 planes (conv_x6w.h) and split it in every launch; with the planes handed in - made once per
    tensor and step - forward and dgrad skip that split and the filter gradient (wgrad_x6_kernel<.., PIN>) takes BOTH operands as
    planes instead of splitting them on the VALU.  The planes ARE the exact split the kernels make themselves, so every result
    must have the bits of the launch without them; the planes themselves must add up to the tensor exactly."""
    n, h, w, cin, cout, k, dil = case
    g = torch.Generator().manual_seed(cin + dil)
    x = rnd(g, n, h, w, cin).cuda()
    wt = (rnd(g, k, k, cin, cout) * (1.0 / np.sqrt(k * k * cin))).cuda()
    b = rnd(g, cout).cuda()
    dy = rnd(g, n, h, w, cout).cuda()
    d = engine.conv_desc(tuple(x.shape), cout, k, k, 1, dil, "same")
    xp, dyp = engine.split_planes(x), engine.split_planes(dy)
    assert tuple(xp.shape) == (3, n * h * w, cin) and xp.dtype == torch.int16
    parts = (xp.view(torch.bfloat16).double().sum(0)).view(n, h, w, cin)   # a1 + a2 + a3, exact in fp64
    assert torch.equal(parts, x.double()), "the three planes do not add up to the tensor"
    y0, st0 = engine.conv2d_fwd(x, wt, b, desc=d, want_stats=True)
    y1, st1 = engine.conv2d_fwd(x, wt, b, desc=d, want_stats=True, x_planes=xp)
    assert torch.equal(y0, y1) and (st0 is None) == (st1 is None)
    if st0 is not None:
        assert st0[1] == st1[1] and torch.equal(st0[0], st1[0])
    assert torch.equal(engine.conv2d_fwd(x, wt, None, desc=d, relu=True), engine.conv2d_fwd(x, wt, None, desc=d, relu=True, x_planes=xp))
    assert torch.equal(engine.conv2d_dgrad(dy, wt, d), engine.conv2d_dgrad(dy, wt, d, dy_planes=dyp))
    dw0, _ = engine.conv2d_wgrad(x, dy, d, want_bias=False)
    if engine.conv2d_wgrad_planes_ok(d):
        dw1 = engine.conv2d_wgrad_planes(xp, dyp, d)
        assert torch.equal(dw0, dw1), "planes-in filter gradient: not the bits of the fp32-operand kernel"
    else:
        assert k == 1, "only the wide pointwise layer of these cases belongs to another kernel family"
    xr, wr = x.cpu().double().requires_grad_(), wt.cpu().double().requires_grad_()
    T.conv2d(xr, wr, b.cpu().double(), 1, dil, "same").backward(dy.cpu().double())
    close(dw0, wr.grad, what="filter gradient vs fp64")
    # which launches read planes: the long-K multi-tap ones (the query the runtime asks before it makes planes)
    assert engine.conv2d_planes_in(d, False) == (k == 3 and k * k * cin >= 2048 and cout >= 192)


@pytest.mark.parametrize("case", [(2, 16, 32, 32, 32, 3), (2, 24, 32, 64, 32, 3), (3, 8, 16, 64, 64, 3), (1, 16, 16, 32, 64, 3), (2, 13, 11, 32, 2, 1),
                                  (2, 16, 16, 64, 1, 1)],
                         ids=["patch_32_32", "patch_64_32", "patch_64_64", "patch_32_64", "thin_head_ragged", "thin_sse"])
@pytest.mark.parametrize("infer", [False, True], ids=["train_stats", "moving_stats"])
def test_batchnorm_applied_in_the_convolution_loaders(engine, case, infer):
    """Round 5: BatchNormalization(+ReLU) -> Conv2D with the normalisation applied while the convolution's kernels load their
    input (sg_conv2d_fwd_stats_bn / sg_conv2d_wgrad_bn: the patch kernels conv_x6p.h / conv_x6wp.h and the thin 1x1 kernels).  The
    loaders evaluate bn_apply's own expression on every pixel inside the image, so forward (with its statistics epilogue) and the
    filter gradient must have the BITS of the unfused pair of launches - the zero padding included, which is padding of the
    NORMALISED tensor and must not be normalised."""
    n, h, w, cin, cout, k = case
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    x = (rnd(g, n, h, w, cin) * 2 + 0.3).cuda()
    wt = (rnd(g, k, k, cin, cout) * (1.0 / np.sqrt(k * k * cin))).cuda()
    b = rnd(g, cout).cuda()
    dy = rnd(g, n, h, w, cout).cuda()
    gam, bet = (rnd(g, cin) + 1.5).cuda(), rnd(g, cin).cuda()
    d = engine.conv_desc(tuple(x.shape), cout, k, k, 1, 1, "same")
    assert engine.conv2d_bn_in_ok(d)
    eps = 1e-3
    if infer:
        mean, var = rnd(g, cin).cuda(), (rnd(g, cin) + 1.5).cuda()
        xn = engine.bn_infer(x, gam, bet, mean, var, relu=True, eps=eps)
        bn = (gam, bet, mean, var, True, True, eps)
    else:
        xn, mean, inv = engine.bn_train_fwd(x, gam, bet, torch.zeros(cin).cuda(), torch.ones(cin).cuda(), relu=True, eps=eps)
        bn = (gam, bet, mean, inv, True, False, eps)
    patch = k == 3
    y0 = engine.conv2d_fwd(xn, wt, b, desc=d, want_stats=patch)
    y1 = engine.conv2d_fwd(x, wt, b, desc=d, want_stats=patch, bn_in=bn)
    if patch:
        (y0, st0), (y1, st1) = y0, y1
        assert st0 is not None and st1 is not None and st0[1] == st1[1] and torch.equal(st0[0], st1[0])
    assert torch.equal(y0, y1), f"forward: max diff {float((y0 - y1).abs().max())}"
    dw0, db0 = engine.conv2d_wgrad(xn, dy, d)
    dw1, db1 = engine.conv2d_wgrad(x, dy, d, bn_in=bn)
    assert torch.equal(dw0, dw1) and torch.equal(db0, db1), f"filter gradient: max diff {float((dw0 - dw1).abs().max())}"
    # without the ReLU too (a BatchNormalization feeding a convolution directly)
    bn2 = bn[:4] + (False,) + bn[5:]
    xn2 = engine.bn_infer(x, gam, bet, bn[2], bn[3], relu=False, eps=eps) if infer else \
        engine.bn_train_fwd(x, gam, bet, torch.zeros(cin).cuda(), torch.ones(cin).cuda(), relu=False, eps=eps)[0]
    assert torch.equal(engine.conv2d_fwd(xn2, wt, None, desc=d), engine.conv2d_fwd(x, wt, None, desc=d, bn_in=bn2))
    # fp64 oracle of the pair
    xr = T.batch_norm(x.cpu().double(), gam.cpu().double(), bet.cpu().double(), bn[2].cpu().double(), bn[3].cpu().double(), False)[0] if infer else None
    if xr is not None:
        close(y1, T.conv2d(torch.relu(xr), wt.cpu().double(), b.cpu().double(), 1, 1, "same"), what="BN(infer) -> ReLU -> conv vs fp64")
    # a launch outside the covered kernels is refused, not computed without the normalisation
    from building_detection_amd._lib import SgError
    d_bad = engine.conv_desc(tuple(x.shape), cout, k, k, 1, 2 if k == 3 else 1, "same") if k == 3 else engine.conv_desc((n, h, w, cin), 8, 1, 1, 1, 1, "same")
    assert not engine.conv2d_bn_in_ok(d_bad)
    with pytest.raises(SgError):
        engine.conv2d_fwd(x, (rnd(g, k, k, cin, 8 if k == 1 else cout)).cuda(), None, desc=d_bad, bn_in=bn)


@pytest.mark.parametrize("case", [(6, 32, 32, 728, 728, True), (6, 32, 32, 728, 728, False), (7, 32, 32, 728, 1016, True), (3, 64, 64, 1024, 728, True)],
                         ids=["middle_flow", "no_relu", "ragged_k_1016", "tiles_256_wide"])
def test_batchnorm_backward_apply_in_the_pointwise_dgrad(engine, case):
    """Round 5 (csrc/conv_pw.h, BNB form; sg_conv2d_dgrad_bnb): SeparableConv2D -> BatchNormalization, backward - the layer's
    backward APPLY evaluated in the A path of the pointwise dgrad, the applied gradient also stored for the filter gradient.  Both
    outputs must have the BITS of sg_bn_train_bwd_apply + sg_conv2d_dgrad (the transformed value is fenced before the x6 split:
    left to the compiler, the split's residual was an fma of the unrounded product and the input gradient differed in the last
    place).  Off in the step by default (a loss there: _Runtime.bnb_on), kept correct."""
    n, h, w_, cin, cout, relu = case
    g = torch.Generator().manual_seed(cin + cout)
    t = rnd(g, n, h, w_, cin).cuda()
    wt = (rnd(g, 1, 1, cin, cout) * (1.0 / np.sqrt(cin))).cuda()
    d = engine.conv_desc(tuple(t.shape), cout, 1, 1, 1, 1, "same")
    assert engine.conv2d_dgrad_bnb_ok(d), "a launch of the wide pointwise kernel"
    y = engine.conv2d_fwd(t, wt, None, desc=d)
    gam, bet = (rnd(g, cout) + 1.5).cuda(), (rnd(g, cout) * 0.3).cuda()
    z, mean, inv = engine.bn_train_fwd(y, gam, bet, torch.zeros(cout).cuda(), torch.ones(cout).cuda(), relu=relu)
    dyb = rnd(g, n, h, w_, cout).cuda()
    dz_ref, dgam, dbet = engine.bn_train_bwd(y, z, dyb, gam, mean, inv, relu=relu, beta=bet)
    dx_ref = engine.conv2d_dgrad(dz_ref, wt, d)
    dx, dz = engine.conv2d_dgrad_bnb(dyb, y, wt, d, gam, bet, mean, inv, dgam, dbet, relu)
    assert torch.equal(dz, dz_ref), f"applied gradient: max diff {float((dz - dz_ref).abs().max())}"
    assert torch.equal(dx, dx_ref), f"input gradient: max diff {float((dx - dx_ref).abs().max())}"
    # fp64: BatchNormalization backward through autograd, then the pointwise dgrad
    yr = y.cpu().double().requires_grad_()
    zr = T.batch_norm(yr, gam.cpu().double(), bet.cpu().double(), torch.zeros(cout).double(), torch.ones(cout).double(), True)[0]
    (torch.relu(zr) if relu else zr).backward(dyb.cpu().double())
    close(dz, yr.grad, rtol=2e-5, what="applied gradient vs fp64 autograd")
    from building_detection_amd._lib import SgError
    small = engine.conv_desc((1, 32, 32, cin), cout, 1, 1, 1, 1, "same")   # 1024 rows: the narrow kernels' launch
    assert not engine.conv2d_dgrad_bnb_ok(small)
    with pytest.raises(SgError):
        engine.conv2d_dgrad_bnb(dyb[:1].contiguous(), y[:1].contiguous(), wt, small, gam, bet, mean, inv, dgam, dbet, relu)


def test_sub_batch_paths_of_oversized_tensors(engine):
    """activations beyond 2 GiB run as sub-batches of whole images (forward, dgrad) / as chunks with one reduce
    (wgrad).  SG_CONV_MAX_BYTES (read once per process) lowers that limit, so a child process runs a 5-image convolution
    with a 2-image limit - uneven last chunk - against the unlimited result of this process: bit-identical forward / dgrad
    (every image takes the same kernel), wgrad within fp32 summation-order noise (the split differs)."""
    import os
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from building_detection_amd.ops import get_engine
e = get_engine(0)
g = torch.Generator().manual_seed(5)
out = {}
for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
  for shp, cout, dil in (("a", 96, 2), ("p", 32, 1)):   # "p": the patch-form kernels (forward / dgrad and wgrad)
    x = (torch.randn(5, 32, 32, 64, generator=g)).cuda().to(dt)
    w = (torch.randn(3, 3, 64, cout, generator=g) * 0.05).cuda()
    b = torch.randn(cout, generator=g).cuda()
    d = e.conv_desc(tuple(x.shape), cout, 3, 3, 1, dil, "same")
    y = e.conv2d_fwd(x, w, b, desc=d)
    dy = torch.randn(*y.shape, generator=g).cuda().to(dt)
    dx = e.conv2d_dgrad(dy, w, d)
    dw, db = e.conv2d_wgrad(x, dy, d)
    for k, v in (("y", y), ("dx", dx), ("dw", dw), ("db", db)):
        out[f"{tag}{shp}_{k}"] = v.float().cpu().numpy()
# the fp32 softmax head of a bf16 model that is not a 1x1 convolution (Res34-UNet: 3x3, 64 -> 2): bf16 in, fp32 out
x = torch.randn(5, 32, 32, 64, generator=g).cuda().to(torch.bfloat16)
w = (torch.randn(3, 3, 64, 2, generator=g) * 0.05).cuda()
b = torch.randn(2, generator=g).cuda()
d = e.conv_desc(tuple(x.shape), 2, 3, 3, 1, 1, "same")
y = e.conv2d_fwd(x, w, b, desc=d, head_f32=True)
assert y.dtype == torch.float32
dyh = torch.randn(*y.shape, generator=g).cuda()
dx = e.conv2d_dgrad(dyh, w, d, out_dtype=torch.bfloat16)
assert dx.dtype == torch.bfloat16
out["head_y"], out["head_dx"] = y.cpu().numpy(), dx.float().cpu().numpy()
np.savez(sys.argv[2], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for name, limit in (("whole", None), ("chunked", str(2 * 32 * 32 * 64 * 4 + 1000))):   # two fp32 input images fit
            env = dict(os.environ)
            if limit:
                env["SG_CONV_MAX_BYTES"] = limit
            path = os.path.join(td, name + ".npz")
            subprocess.run([sys.executable, "-c", code, root, path], check=True, env=env, timeout=300)
            res[name] = dict(np.load(path))
    assert np.array_equal(res["whole"]["head_y"], res["chunked"]["head_y"])
    assert np.array_equal(res["whole"]["head_dx"], res["chunked"]["head_dx"])
    for tag in ("f32a", "f32p", "bf16a", "bf16p"):
        assert np.array_equal(res["whole"][f"{tag}_y"], res["chunked"][f"{tag}_y"])
        assert np.array_equal(res["whole"][f"{tag}_dx"], res["chunked"][f"{tag}_dx"])
        for k in ("dw", "db"):
            a, b = res["whole"][f"{tag}_{k}"], res["chunked"][f"{tag}_{k}"]
            assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max(), (tag, k)


def test_fused_second_stage_of_the_reductions_is_bit_identical(engine):
    """sg_reduce.h: the second stage (fp64 sums of the partial rows + the Op's finalize) runs inside the reduce kernel when one
    workgroup covers a column block (default), as a separate launch (SG_SEG_FUSED=0) or in the last-arriving workgroup
    (SG_SEG_FUSED=2, arrival counters behind an agent-scope release; SG_SEG_FUSED=3, the same with write-through partial rows
    and no release - the sc1 hand-off of MI355X_MICROARCH.md).  Same lanes, same order of additions: the four forms
    must agree to the bit on BatchNorm backward, the depthwise filter gradient (both kernels), pooling and a bias gradient -
    all of them reductions over several row slabs per column block at the first shape."""
    import os
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from building_detection_amd.ops import get_engine
e = get_engine(0)
g = torch.Generator().manual_seed(11)
out = {}
for tag, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
    for n, h, c in ((4, 32, 728), (2, 64, 96), (3, 8, 40)):
        x = torch.randn(n, h, h, c, generator=g).cuda().to(dt)
        dy = torch.randn(n, h, h, c, generator=g).cuda().to(dt)
        gamma, beta = torch.rand(c, generator=g).cuda() + 0.5, torch.randn(c, generator=g).cuda()
        y, mean, invstd = e.bn_train_fwd(x, gamma, beta, torch.zeros(c).cuda(), torch.ones(c).cuda(), relu=True)
        dx, dg, db = e.bn_train_bwd(x, y, dy, gamma, mean, invstd, relu=True, beta=beta)
        wd = torch.randn(3, 3, c, generator=g).cuda()
        d = e.conv_desc(tuple(x.shape), c, 3, 3, 1, 1, "same")
        dwg = e.dwconv_wgrad(x, dy, d, True)
        ds = e.conv_desc(tuple(x.shape), c, 3, 3, 2, 1, "same")
        dys = torch.randn(n, ds.Ho, ds.Wo, c, generator=g).cuda().to(dt)
        dwg2 = e.dwconv_wgrad(x, dys, ds, False)
        gap = e.avgpool_fwd(x, h, h)   # GlobalAveragePooling2D: one window per image
        bias = e.empty(c)
        e.bias_grad(dy, bias)
        for k, v in (("mean", mean), ("invstd", invstd), ("dx", dx), ("dg", dg), ("db", db), ("dwg", dwg), ("dwg2", dwg2),
                     ("gap", gap), ("bias", bias)):
            out[f"{tag}_{h}_{c}_{k}"] = v.float().cpu().numpy()
np.savez(sys.argv[2], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    with tempfile.TemporaryDirectory() as td:
        for mode in ("0", "1", "2", "3"):
            env = dict(os.environ)
            env["SG_SEG_FUSED"] = mode
            path = os.path.join(td, mode + ".npz")
            subprocess.run([sys.executable, "-c", code, root, path], check=True, env=env, timeout=300)
            res[mode] = dict(np.load(path))
    assert len(res["0"]) == 2 * 3 * 9
    for k, v in res["0"].items():
        assert np.isfinite(v).all(), k
        assert np.array_equal(v, res["1"][k]), ("separate launch vs fused (S == 1)", k)
        assert np.array_equal(v, res["2"][k]), ("separate launch vs arrival counters", k)
        assert np.array_equal(v, res["3"][k]), ("separate launch vs arrival counters with write-through partial rows", k)
