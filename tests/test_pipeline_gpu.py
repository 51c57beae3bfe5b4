"""predict.py tile loop + model_fuse.py vote on the engine vs their CPU restatement (oracle/pipeline.py)."""
import numpy as np
import pytest
import torch

from oracle import pipeline as OP
from oracle import tfops as T

pytestmark = pytest.mark.gpu


def tiny_model():
    """A 512x512 two-layer segmentation net: enough to drive the pipeline cheaply on both sides."""
    from building_detection_amd import layers as L
    from building_detection_amd.runtime import Model
    inp = L.Input((512, 512, 3))
    x = L.Conv2D(8, 3, padding="same", activation="relu")(inp)
    x = L.Conv2D(8, 3, padding="same", dilation_rate=3, activation="relu")(x)
    out = L.Conv2D(2, 1, activation="softmax")(x)
    return Model(inp, out, name="tiny", seed=4)


def oracle_predict_fn(ws):
    w = [torch.tensor(a, dtype=torch.float64) for a in ws]

    def fn(tile):
        with torch.no_grad():
            x = torch.from_numpy(np.asarray(tile, dtype=np.float64))
            x = torch.relu(T.conv2d(x, w[0], w[1]))
            x = torch.relu(T.conv2d(x, w[2], w[3], dilation=3))
            return torch.softmax(T.conv2d(x, w[4], w[5]), -1).numpy()
    return fn


@pytest.mark.parametrize("h,w", [(512, 512), (700, 640), (400, 300), (600, 1000)])
def test_detection_matches_reference_loop(engine, h, w):
    from building_detection_amd import pipeline as PL
    model = tiny_model()
    ws = model.get_weights()
    rng = np.random.default_rng(h * 7 + w)
    ws[5] = np.array([0.0, 0.02], np.float32)  # bias the head so both classes occur
    model.set_weights(ws)
    img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    # smooth blobs so the mask is not pure noise
    img[h // 4:h // 2, w // 5:w // 2] //= 3
    got = PL.detection(img, None, model, batch=3)
    # the restatement's predict_fn also records where ITS class margin is below 1e-6 (an fp32 evaluation cannot resolve
    # such a pixel): the masks must agree exactly everywhere else; the near-tie pixels are counted and printed
    base_fn = oracle_predict_fn(ws)
    (ch, cw), origins = PL.tile_origins(h, w, True)
    near = np.zeros((ch, cw), bool)
    it = iter(origins)

    def fn(tile):
        p = base_fn(tile)
        i, j = next(it)  # detection_ref visits the tiles in tile_origins order (predict.py:105-106)
        near[i:i + 512, j:j + 512] |= np.abs(p[0, ..., 1] - p[0, ..., 0]) <= 1e-6
        return p

    ref = OP.detection_ref(img, fn)
    assert got.shape == ref.shape == (h, w) and got.dtype == np.uint8
    diff = got != ref
    strict_bad = int((diff & ~near[:h, :w]).sum())
    print(f"{h}x{w}: {int(near[:h, :w].sum())} near-tie pixels, {int((diff & near[:h, :w]).sum())} of them differ; "
          f"{strict_bad} differ elsewhere")
    assert strict_bad == 0, f"{strict_bad} of {h * w} mask pixels differ outside the 1e-6 tie margin"
    assert set(np.unique(got)) <= {0, 255}
    if w > h + 360:  # the reference's column loop never reaches the right-hand tiles (predict.py:106)
        assert got[:, 872:].max() == 0
        fixed = PL.detection(img, None, model, batch=4, reference_jloop=False)
        assert fixed[:, 872:].max() == 255


def test_portrait_image_raises_like_reference(engine):
    from building_detection_amd import pipeline as PL
    with pytest.raises(ValueError, match="predict.py:106"):
        PL.tile_origins(1300, 500, reference_jloop=True)
    (ch, cw), origins = PL.tile_origins(1300, 500, reference_jloop=False)
    assert (ch, cw) == (1592, 512) and len(origins) == 4


@pytest.mark.parametrize("name", ["hrnet", "v3plus"])
def test_hipgraph_capture_is_bit_identical_to_eager(engine, name):
    """BASELINE config 5: the hipGraph-captured forward must replay exactly the eager launches."""
    from building_detection_amd import zoo
    kw = {"aspp_pool": 4} if name == "v3plus" else {}
    m = zoo.BUILDERS[name]((64, 64, 3), 2, **kw) if kw else zoo.BUILDERS[name]((64, 64, 3))
    g = torch.Generator().manual_seed(9)
    x1 = (torch.rand(3, 64, 64, 3, generator=g) * 2 - 1).cuda()
    x2 = (torch.rand(3, 64, 64, 3, generator=g) * 2 - 1).cuda()
    e1, e2 = m.predict_device(x1).clone(), m.predict_device(x2).clone()
    gp = m.capture_predict(3)
    assert torch.equal(gp(x1), e1)
    assert torch.equal(gp(x2), e2)
    assert torch.equal(gp(x1), e1)  # replay is repeatable
    assert not torch.equal(e1, e2)


def test_vote(engine):
    from building_detection_amd import pipeline as PL
    rng = np.random.default_rng(0)
    masks = [(rng.random((300, 280)) > 0.5).astype(np.uint8) * 255 for _ in range(5)]
    np.testing.assert_array_equal(PL.vote(masks, 3), OP.vote_ref(masks, 3))


def test_file_generator_feeds_fit_generator(engine, tmp_path):
    """train_data_gen / val_data_gen over files on disk drive fit_generator exactly as the reference's
    training script does (DeepLabv3plus.py:840-849); the GPU-built label channels equal the host-built ones."""
    from PIL import Image
    from building_detection_amd import input_pipeline as IP, zoo
    from building_detection_amd.losses import edge_focal_loss, PA, IoU, MIoU, F1_score
    rng = np.random.default_rng(3)
    imgs, labs = [], []
    for i in range(2):
        lab = np.zeros((512, 512), np.uint8)
        lab[60 + 100 * i:200 + 100 * i, 40:300] = 255
        lab[:9, 480:] = 255
        pi, pl = tmp_path / f"i{i}.png", tmp_path / f"l{i}.png"
        Image.fromarray(rng.integers(0, 256, size=(512, 512, 3), dtype=np.uint8)).save(pi)
        Image.fromarray(lab).save(pl)
        imgs.append(str(pi)); labs.append(str(pl))
    from oracle import input_pipeline as OIP
    xh, yh = next(OIP.data_gen(list(imgs), list(labs), 2))     # the oracle's restatement of train_data_gen
    xd, yd = next(IP.train_data_gen(list(imgs), list(labs), 2, engine=engine))
    assert np.array_equal(xh, xd) and yd.dtype == np.float64 and np.array_equal(yh, yd)
    model = zoo.Xception_DeepLabV3_Plus_bam((512, 512, 3), 2)
    model.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score])
    hist = model.fit_generator(IP.train_data_gen(list(imgs), list(labs), 1), steps_per_epoch=2, epochs=1, verbose=0,
                               validation_data=IP.val_data_gen(list(imgs), list(labs), 1), validation_steps=1)
    logs = hist.history
    for k in ("loss", "PA", "IoU", "MIoU", "F1_score", "val_loss", "val_PA"):
        assert k in logs and np.isfinite(logs[k][-1]), (k, logs)


def test_two_live_graphs_keep_their_own_workspace(engine):
    """a hipGraph bakes the scratch pointer into its kernel nodes.  Two captured models with different
    workspace needs stay alive together, the engine's shared scratch is re-grown (freed) in between by a bigger eager
    call, and interleaved replays must still equal the eager results bit for bit."""
    from building_detection_amd import zoo
    small = zoo.BUILDERS["hrnet"]((64, 64, 3))
    big = zoo.BUILDERS["v3plus"]((128, 128, 3), 2, aspp_pool=8)
    g = torch.Generator().manual_seed(21)
    xs = (torch.rand(2, 64, 64, 3, generator=g) * 2 - 1).cuda()
    xb = (torch.rand(2, 128, 128, 3, generator=g) * 2 - 1).cuda()
    es, eb = small.predict_device(xs).clone(), big.predict_device(xb).clone()
    gs = small.capture_predict(2)
    gb = big.capture_predict(2)
    assert gs.ws.data_ptr() != gb.ws.data_ptr() != engine._ws.data_ptr()
    # force the shared scratch to be replaced, then scribble over whatever the allocator hands out next
    engine.ws(engine._ws.numel() * 2 + (64 << 20))
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]
    for _ in range(2):
        assert torch.equal(gs(xs), es)
        assert torch.equal(gb(xb), eb)
    del junk
    assert torch.equal(small.predict_device(xs), es)  # eager path still fine on the re-grown shared scratch
    assert torch.equal(gb(xb), eb)


def test_predict_is_serialised_across_threads(engine):
    """SURVEY 8(b-1) threading note (buildAPI.py:78,111: Flask request threads share the models): concurrent
    predict() calls on ONE model, and on two models sharing the device, return exactly the single-threaded result."""
    import threading
    from building_detection_amd import zoo
    m1 = zoo.BUILDERS["hrnet"]((64, 64, 3))
    m2 = zoo.BUILDERS["v3plus"]((64, 64, 3), 2, aspp_pool=4)
    rng = np.random.default_rng(5)
    xs = [rng.uniform(-1, 1, size=(2, 64, 64, 3)) for _ in range(4)]
    ref1 = [m1.predict(x) for x in xs]
    ref2 = [m2.predict(x) for x in xs]
    errs = []

    def worker(model, refs, order):
        try:
            for _ in range(6):
                for i in order:
                    got = model.predict(xs[i])
                    if not np.array_equal(got, refs[i]):
                        errs.append(f"{model.name}: input {i} differs")
        except Exception as e:  # a clobbered value table shows up as KeyError / shape errors
            errs.append(repr(e))

    ts = [threading.Thread(target=worker, args=(m1, ref1, [0, 1, 2, 3])),
          threading.Thread(target=worker, args=(m1, ref1, [3, 2, 1, 0])),
          threading.Thread(target=worker, args=(m2, ref2, [1, 3, 0, 2]))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:5]


@pytest.mark.parametrize("size", [512, 384])
def test_device_data_gen_is_bit_identical_to_the_oracle_generator(engine, tmp_path, size):
    """SURVEY 8f-1, the non-synchronous feed: files decoded by worker threads ahead of the consumer, uint8 pixels over PCIe,
    resize (tiles that are not 512 x 512), normalisation and label channels on the device - the values of the ORACLE's
    train_data_gen restatement (oracle/input_pipeline.py, DeepLabv3plus.py:32-107) bit for bit, in the same (sorted,
    cycled) order, and fit_generator consumes the device tensors directly."""
    from PIL import Image
    from building_detection_amd import input_pipeline as IP, zoo
    from building_detection_amd.losses import edge_focal_loss, PA, IoU, MIoU, F1_score
    from oracle import input_pipeline as OIP
    rng = np.random.default_rng(8)
    imgs, labs = [], []
    q = size / 512.0
    for i in range(3):
        lab = np.zeros((size, size), np.uint8)
        lab[int((40 + 90 * i) * q):int((180 + 90 * i) * q), int(60 * q):int((260 + 40 * i) * q)] = 255
        lab[size - 12:, :30] = 255
        lab[int(100 * q), int(400 * q)] = 128          # a grey value that is NOT building (to_categorical truncation)
        pi, pl = tmp_path / f"i{i}.png", tmp_path / f"l{i}.png"
        Image.fromarray(rng.integers(0, 256, size=(size, size, 3), dtype=np.uint8)).save(pi)
        Image.fromarray(lab).save(pl)
        imgs.append(str(pi)); labs.append(str(pl))
    host = OIP.data_gen(list(imgs), list(labs), 2)
    dev = IP.device_data_gen(list(imgs), list(labs), 2, engine, depth=2, workers=3)
    for _ in range(4):   # more than one cycle of the three files
        xh, yh = next(host)
        xd, yd = next(dev)
        assert xd.is_cuda and xd.dtype == torch.float32 and yd.dtype == torch.float32
        assert tuple(xd.shape) == (2, 512, 512, 3) and tuple(yd.shape) == (2, 512, 512, 4)
        assert np.array_equal(xd.cpu().numpy(), xh)
        assert np.array_equal(yd.cpu().numpy().astype(np.float64), yh)
    if size == 512:
        model = zoo.HRNet((512, 512, 3))
        model.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score])
        hist = model.fit_generator(dev, steps_per_epoch=2, epochs=1, verbose=0)
        assert np.isfinite(hist.history["loss"][-1])
    dev.close()
