"""Host-side logic that needs no GPU: graph construction and structural known-answers of the engine's
builders, the C-ABI library (loads, exports every symbol include/segengine.h declares), LR schedule,
callbacks, metrics arithmetic, synthetic data, fusion pass, gradient bucketing and a world_size-2 gloo run of
the data-parallel reducer."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EXPECTED = {  # trainable, non-trainable, forward GFLOP per 512x512 tile
    "v3plus": (64509482, 106192, 202.12), "bam": (62863400, 105770, 151.76), "scse": (34558914, 0, 406.91),
    "res34": (38519778, 25536, 499.06), "hrnet": (9588226, 19584, 187.48)}


@pytest.mark.parametrize("name", list(EXPECTED))
def test_engine_graphs_match_known_answers_and_oracle_order(name):
    from building_detection_amd import zoo
    from oracle import models as M
    m = zoo.BUILDERS[name]((512, 512, 3))
    tr = sum(p.size for p in m.params if p.trainable)
    nt = sum(p.size for p in m.params if not p.trainable)
    assert (tr, nt) == EXPECTED[name][:2]
    assert m.outputs[0].shape == (None, 512, 512, 2)
    assert abs(m.flops(1) / 1e9 - EXPECTED[name][2]) < 0.01
    P = M.Params()
    kw = {"aspp_pool": 4} if name in ("v3plus", "bam") else {}
    with torch.no_grad():
        M.BUILDERS[name](P, torch.zeros(1, 64, 64, 3), **kw)
    assert [tuple(t.shape) for t in P.tensors] == [p.shape for p in m.params]
    assert P.kinds == [p.kind for p in m.params]


def test_dilated_subset_flops_match_survey():
    """3 ASPP + 3 SK dilated convs: 32.61 GFLOP fwd per tile, x3 = 97.84."""
    from building_detection_amd import zoo, layers as L
    m = zoo.Xception_DeepLabV3_Plus()
    dil = [n for n in m.nodes if isinstance(n, L._ConvNode) and n._tag]
    assert len(dil) == 6
    f = sum(n.flops(1) for n in dil)
    assert abs(f / 1e9 - 32.61) < 0.01 and abs(3 * f / 1e9 - 97.84) < 0.02


def test_fusion_pass_counts():
    from building_detection_amd import zoo, layers as L
    m = zoo.Xception_DeepLabV3_Plus()
    seps = [n for n in m.nodes if isinstance(n, L._SepConvNode)]
    assert len(seps) == 62
    # every ReLU is absorbed into a BN, an Add or a SeparableConv gather
    assert all(n.fused_away for n in m.nodes if isinstance(n, L._ActNode) and n.act == "relu")
    assert sum(n.pre_relu for n in seps) > 0
    r = zoo.ResNetFamily().run_model("res34")
    assert any(isinstance(n, L._AddNode) and n.relu for n in r.nodes)
    with pytest.raises(ValueError, match="This network does not exist."):
        zoo.ResNetFamily().run_model("res18")


def test_reference_layer_names_res34():
    from building_detection_amd import zoo
    names = {n.name for n in zoo.ResNetFamily((64, 64, 3)).run_model("res34").nodes}
    for want in ("conv1_1", "conv1_1_BN", "pool1", "pool4", "conv2_0_1", "conv5_2_2_BN", "conv3_3_add", "upsame_1_1"):
        assert want in names, want


def test_abi_library_exports_every_declared_symbol():
    from building_detection_amd import _lib
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "segengine.h")).read()
    declared = set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.exported_symbols())
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/segengine.h but not exported"
    # ... and the other direction: every sg_* symbol the built library defines is declared in the header
    # (whatever its return type: sg_mask_split_words returns int64_t, the *_ws_bytes family size_t)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], stdout=subprocess.PIPE, check=True).stdout.decode()
    exported = {l.split()[-1] for l in nm.splitlines() if l.split() and l.split()[-1].startswith("sg_")}
    assert exported, "nm found no sg_* symbols"
    assert exported - declared == set(), f"exported by libsegengine.so but not declared in include/segengine.h: {sorted(exported - declared)}"
    assert declared - exported == set(), f"declared but not defined by the library: {sorted(declared - exported)}"
    assert _lib.load().sg_abi_version() == 1
    assert ctypes.sizeof(_lib.ConvDesc) == 15 * 4


def test_no_cpu_fallback():
    """Without a GPU the product refuses to compute (it must never route through the oracle or the CPU)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from building_detection_amd import zoo, _lib
    m = zoo.HRNet((64, 64, 3))
    with pytest.raises(_lib.SgError):
        m.predict(np.zeros((1, 64, 64, 3), np.float32))
    src = "".join(open(os.path.join(ROOT, "building_detection_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "building_detection_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_lr_schedule_matches_reference_formula():
    from building_detection_amd.callbacks import cosine_decay_with_warmup as f
    from oracle.models import cosine_decay_with_warmup as g
    steps, total, warm = 592, 30 * 592, 3 * 592  # 4736 // 8 steps per epoch (data_enhancement.py:14, bs 8)
    for s in (0, 1, warm - 1, warm, warm + 1, total // 2, total - 1, total):
        assert f(s, 1e-3, total, 1e-5, warm, 0) == g(s, 1e-3, total, 1e-5, warm, 0)
    assert f(0, 1e-3, total, 1e-5, warm) == 1e-5
    assert abs(f(warm, 1e-3, total, 1e-5, warm) - 1e-3) < 1e-18
    assert f(total, 1e-3, total, 1e-5, warm) < 1e-18


def test_callbacks_drive_optimizer_lr(tmp_path):
    from building_detection_amd.callbacks import (WarmUpCosineDecayScheduler, MY_EarlyStoppingAtMinLoss, backend as K,
                                                  cosine_decay_with_warmup)
    from building_detection_amd.runtime import Optimizer

    class FakeModel:
        def __init__(self):
            self.optimizer = Optimizer()
            self.saved = []

        def save_weights(self, p):
            self.saved.append(p)

    fm = FakeModel()
    cb = WarmUpCosineDecayScheduler(1e-3, total_steps=100, warmup_learning_rate=1e-5, warmup_steps=10)
    cb.set_model(fm)
    for step in range(15):
        cb.on_batch_begin(step)
        assert K.get_value(fm.optimizer.lr) == cosine_decay_with_warmup(step, 1e-3, 100, 1e-5, 10)
        cb.on_batch_end(step)
    assert len(cb.learning_rates) == 15
    K.set_value(fm.optimizer.lr, 0.5)  # the reference's spelling (DeepLabv3plus.py:736)
    assert float(fm.optimizer.lr) == 0.5
    es = MY_EarlyStoppingAtMinLoss(6, directory=str(tmp_path / "weights1"))
    es.set_model(fm)
    es.on_train_begin()
    es.on_epoch_end(0, {"val_PA": 0.9})
    assert fm.saved[0].endswith("epoch_1_weights.h5") and es.all_acc == [0.9]


def test_metrics_float32_arithmetic():
    from building_detection_amd.losses import metrics_from_counts, resolve_loss, resolve_metric, edge_focal_loss
    from oracle.models import metrics_from_counts as ref
    for c in ((10, 20, 3, 4), (0, 100, 0, 0), (1234567, 7654321, 1111, 2222)):
        a, b = metrics_from_counts(*c), ref(*c)
        for k in a:
            assert abs(a[k] - b[k]) < 1e-7
    assert metrics_from_counts(0, 5, 0, 0)["IoU"] == 0.0
    assert resolve_loss(edge_focal_loss) == 2 and resolve_loss("focal_loss") == 1

    def binary_crossentropy(y_true, y_pred):  # a reference-style callable is recognised by name
        pass
    import warnings
    from building_detection_amd.losses import ForeignCallableWarning, MIoU
    with pytest.warns(ForeignCallableWarning, match="selects its fused kernel by NAME"):   # ... but no longer silently
        assert resolve_loss(binary_crossentropy) == 0

    def F1_score(y_true, y_pred):
        pass
    with pytest.warns(ForeignCallableWarning, match="530-623"):
        assert resolve_metric(F1_score) == "F1_score"
    with warnings.catch_warnings():   # the engine's own objects and plain names stay quiet
        warnings.simplefilter("error")
        assert resolve_loss(edge_focal_loss) == 2 and resolve_loss("edge_focal_loss") == 2
        assert resolve_metric(MIoU) == "MIoU"
    assert resolve_metric("MIoU") == "MIoU"
    with pytest.raises(ValueError):
        resolve_loss("mse")


def test_synthetic_batch_label_channels():
    from building_detection_amd.data import synthetic_batch, edge_weight_channels
    x, y = synthetic_batch(2, 64, 64, seed=1)
    assert x.shape == (2, 64, 64, 3) and y.shape == (2, 64, 64, 4) and x.dtype == np.float32
    assert x.min() >= -1 and x.max() <= 1
    np.testing.assert_array_equal(y[..., 0] + y[..., 1], 1)
    assert set(np.unique(y[..., 2:])) <= {1.0, 2.0}
    m = np.zeros((32, 32), np.float32)
    m[8:24, 8:24] = 1
    f_edge, p_edge = edge_weight_channels(m)
    # 5 erosions of a 16x16 square leave its 6x6 core: the inner 5-px rim has weight 2 (p_edge), the outer 5-px ring f_edge
    assert p_edge[8, 8] == 2 and p_edge[12, 12] == 2 and p_edge[13, 13] == 1 and p_edge[0, 0] == 1
    assert f_edge[7, 7] == 2 and f_edge[3, 3] == 2 and f_edge[2, 2] == 1 and f_edge[10, 10] == 1


def test_bucket_planner_covers_arena():
    from building_detection_amd.dist import plan_buckets
    from building_detection_amd import zoo
    m = zoo.Xception_DeepLabV3_Plus()
    ranges = [(n.index, p.offset, (p.size + 3) // 4 * 4) for n in m.nodes for p in n.params if p.trainable]
    total = m._n_train
    b = plan_buckets(ranges, total, 12 << 20)
    assert b[0][0] == 0 and b[-1][1] == total
    for (s0, e0, _), (s1, e1, _) in zip(b, b[1:]):
        assert e0 == s1 and e1 > s1
    assert 4 <= len(b) <= 8
    # a bucket becomes ready no later than the node owning its first parameter
    for s, e, ready in b:
        first = min(ni for ni, off, sz in ranges if s <= off < e)
        assert ready == first


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from building_detection_amd.dist import BucketReducer, plan_buckets
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    g = torch.Generator().manual_seed(100 + rank)
    arena = torch.rand(n, generator=g)
    mine = arena.clone()
    ranges = [(i, i * 100, 100) for i in range(10)]  # 10 "layers" of 100 elements
    red = BucketReducer(arena, plan_buckets(ranges, n, 250))
    launched = []
    for node in reversed(range(10)):  # backward sweep
        before = red.next
        fires = red.will_fire(node)   # (the runtime joins its side stream exactly where this says a bucket leaves)
        red.node_done(node)
        launched.append(red.next - before)
        assert fires == (red.next > before), (node, fires, before, red.next)
    red.finish()
    others = [torch.rand(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    ok = torch.allclose(arena, sum(others), atol=1e-6)
    q.put((rank, bool(ok), launched, float((arena - mine).abs().max())))
    dist.destroy_process_group()


def test_data_parallel_reducer_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, launched, delta in res:
        assert ok, f"rank {rank}: all-reduced arena is not the sum of the per-rank arenas"
        # buckets [900,1000) [600,900) [300,600) [0,300) fire as the sweep passes nodes 9, 6, 3, 0 - not all at the end
        assert launched == [1, 0, 0, 1, 0, 0, 1, 0, 0, 1]
        assert delta > 0


class _RecordingTransport:
    """Stands in for RCCL: records which arena range was handed over, and when."""
    name, world, rank = "rec", 1, 0

    def __init__(self):
        self.calls = []

    def allreduce_async(self, t):
        self.calls.append((t.data_ptr(), t.numel()))

    def join(self):
        pass


@pytest.mark.parametrize("name", ["v3plus", "bam", "res34"])
def test_buckets_fire_only_when_complete_on_the_real_graph(name):
    """on the real (fused) graphs, replay the backward sweep of runtime._Runtime.backward - node n
    writes the gradients of ITS parameters, then the hook fires - and check that when a bucket is handed to the
    transport every parameter inside it has been written, that nothing is handed over twice, and that the buckets
    cover the arena."""
    from building_detection_amd import zoo
    from building_detection_amd.dist import BucketReducer, param_ranges, plan_buckets
    m = zoo.BUILDERS[name]((64, 64, 3), 2, aspp_pool=4) if name in ("v3plus", "bam") else zoo.BUILDERS[name]((64, 64, 3))
    total = max(m._n_train, 4)
    arena = torch.zeros(total)
    ranges = param_ranges(m)
    buckets = plan_buckets(ranges, total, 1 << 18)
    assert len(buckets) >= 4
    tp = _RecordingTransport()
    red = BucketReducer(arena, buckets, tp)
    written = torch.zeros(total, dtype=torch.bool)
    base = arena.data_ptr()
    seen = 0
    for n in reversed(m.nodes):
        for p in n.params:
            if p.trainable:
                written[p.offset:p.offset + (p.size + 3) // 4 * 4] = True
        red.node_done(n.index)
        for ptr, cnt in tp.calls[seen:]:
            s = (ptr - base) // 4
            assert bool(written[s:s + cnt].all()), f"bucket [{s},{s + cnt}) fired at node {n.index} before its gradients were all written"
        seen = len(tp.calls)
    red.finish()
    spans = sorted(((ptr - base) // 4, cnt) for ptr, cnt in tp.calls)
    pos = 0
    for s, cnt in spans:
        assert s == pos
        pos += cnt
    assert pos == total
    # the overlap is real: the first bucket leaves long before the sweep ends
    assert red.fired[0][0] > len(m.nodes) // 2


def _dp_graph_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from building_detection_amd import zoo
    from building_detection_amd.dist import BucketReducer, param_ranges, plan_buckets
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = zoo.Xception_DeepLabV3_Plus((64, 64, 3), 2, aspp_pool=4)
    total = m._n_train
    arena = torch.zeros(total)
    red = BucketReducer(arena, plan_buckets(param_ranges(m), total, 1 << 20))

    def grad_of(p, r):  # what rank r's backward writes for parameter p
        return torch.rand(p.size, generator=torch.Generator().manual_seed(p.offset * 7 + r))

    for n in reversed(m.nodes):  # the sweep of runtime._Runtime.backward: write, then the hook
        for p in n.params:
            if p.trainable:
                arena[p.offset:p.offset + p.size] = grad_of(p, rank)
        red.node_done(n.index)
    red.finish()
    ok = True
    for p in m.params:
        if p.trainable:
            want = sum(grad_of(p, r) for r in range(world))
            ok = ok and torch.allclose(arena[p.offset:p.offset + p.size], want, atol=1e-6)
    q.put((rank, bool(ok), len(red.buckets)))
    dist.destroy_process_group()


def test_data_parallel_real_graph_sweep_gloo_world2():
    """World-size-2 gloo run of the bucketed reducer over the DeepLabv3+ arena with the real node order: a bucket fired
    before its last gradient was written would miss that rank-specific value in the sum."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_graph_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, nb in res:
        assert ok, f"rank {rank}: reduced arena != sum of the per-rank gradients"
        assert nb >= 4


def _fallback_worker(rank, world, port, q, fail_rank):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from building_detection_amd import _lib, dist as D
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))

    def no_rccl(lib, _l):
        raise _lib.SgError("sg_comm_unique_id: librccl.so could not be loaded (test)")

    def fake_id(lib, _l):   # a rank whose RCCL is fine (never used: the readiness exchange fails first)
        return bytes(_l.SG_COMM_ID_BYTES)
    D._unique_id = no_rccl if rank == fail_rank else fake_id   # every rank probes its own library: rank 0 by drawing the id,
    D._probe = no_rccl if rank == fail_rank else (lambda lib, _l: 21800)   # the others with sg_comm_probe (no bootstrap root)
    # comm="sg": every rank raises the SAME error, after the same collective, and nobody has entered sg_comm_init ...
    try:
        D.make_transport("sg", 0)
        same_error = False
    except _lib.SgError as e:
        same_error = f"rank(s) {fail_rank}:" in str(e) and "librccl" in str(e)
    # ... so they are still in step: the next collective pairs up
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    # comm="sg_or_torch": all ranks fall back together and the fallback transport carries a reduction
    tp = D.make_transport("sg_or_torch", 0, fallback_backend="gloo")
    g = torch.full((5,), float(rank + 1))
    tp.allreduce_async(g)
    tp.join()
    q.put((rank, same_error, float(t.item()), tp.name, g.tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [0, 1])
def test_sg_comm_failure_on_one_rank_keeps_the_ranks_in_step_gloo_world2(fail_rank):
    """ADVICE r2 (dist.py:95) / r3 (dist.py:177): when ONE rank cannot load RCCL (no librccl to dlopen - the case the
    sg_or_torch fallback exists for), be it rank 0 (which draws the id) or any other, no rank may be left waiting in a
    collective the failed rank never enters - neither the id exchange nor ncclCommInitRank.  Every rank probes its own
    library, the outcome and rank 0's id travel in one all_gather_object, every rank raises the same error before anybody
    calls sg_comm_init, and all of them meet in the next collective."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + 7 * fail_rank
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q, fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=120) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    for rank, same_error, s, name, g in sorted(res):
        assert same_error, f"rank {rank} did not see rank {fail_rank}'s sg_comm_unique_id failure"
        assert s == 3.0
        assert name == "torch" and g == [3.0] * 5


def _comm_init_worker(rank, world, port, q, mode):
    """rank 1's sg_comm_init fails at once (an RCCL error inside ncclCommInitRank); rank 0's never returns (it waits for a peer
    that has gone).  mode 'raise': report what was raised; mode 'exit': do what bench.py does - os._exit(13)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), SG_COMM_INIT_TIMEOUT="2")
    import datetime
    import time
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from building_detection_amd import _lib, dist as D
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))

    class FakeLib:   # the four entry points SgTransport.__init__ reaches before it owns a communicator
        def sg_comm_init(self, uid, r, w, dev, out):
            if r == 1:
                return _lib.SG_ECOMM if hasattr(_lib, "SG_ECOMM") else -4
            time.sleep(600)   # rank 0: inside ncclCommInitRank for ever
            return 0

        def sg_last_error(self):
            return b"ncclCommInitRank: RCCL error 2 (unhandled system error) [test]"

        def sg_comm_destroy(self, h):
            return 0
    D._unique_id = lambda lib, _l: bytes(_l.SG_COMM_ID_BYTES)
    D._probe = lambda lib, _l: 21800
    _lib.load = lambda: FakeLib()
    t0 = time.time()
    try:
        D.make_transport("sg_or_torch", 0, fallback_backend="gloo")   # the fallback must NOT swallow this
        out = ("no error", False, "")
    except D.CommInitError as e:
        out = ("CommInitError", e.stuck, str(e))
    dt = time.time() - t0
    if mode == "exit":
        os._exit(13 if out[0] == "CommInitError" else 0)
    q.put((rank, out, dt))
    q.close()
    q.join_thread()   # the queue's feeder thread has flushed; os._exit would cut it off
    os._exit(0)   # rank 0's worker thread is still "inside RCCL": the process ends without joining it, as bench.py does


def test_a_failure_inside_comm_init_on_one_rank_ends_every_rank_gloo_world2():
    """a rank that fails INSIDE ncclCommInitRank used to leave the others in that collective for
    ever.  sg_comm_init now runs against a deadline on every rank (SG_COMM_INIT_TIMEOUT), the outcomes are exchanged on the
    host, and all ranks raise CommInitError together - the rank that failed at once waits in the exchange until the stuck
    rank's deadline brings it along.  The stuck rank knows it is stuck (its process must end, never be re-used)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_comm_init_worker, args=(r, 2, port, q, "raise")) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=120) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    for rank, (kind, stuck, msg), dt in res:
        assert kind == "CommInitError", (rank, kind)
        assert "rank 0: sg_comm_init did not return within 2 s" in msg and "rank 1: sg_comm_init failed (-4)" in msg and "[test]" in msg, msg
        assert stuck == (rank == 0)
        assert dt < 30, f"rank {rank} took {dt:.1f} s to give up"
    # ... and as processes (what bench.py does with the error): both END, both non-zero, nobody has to be killed
    procs = [ctx.Process(target=_comm_init_worker, args=(r, 2, port + 1, q, "exit")) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=60)
    codes = [p.exitcode for p in procs]
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert codes == [13, 13], codes


_RANK_BODY = """
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(rank)
mode = sys.argv[1]
print(f"chatter from rank {rank}")          # every rank's own stdout: only rank 0's reaches the job's stdout
if mode == "ok":
    if rank == 0:
        print(json.dumps({"metric": "host-only", "n_gpus": world, "args": sys.argv[1:]}))
    sys.exit(0)
if mode == "one_dies":                       # rank 1 fails; rank 2 would wait in a collective for ever; rank 0 finishes
    if rank == 1:
        sys.exit(3)
    if rank == 2:
        time.sleep(600)
    print(json.dumps({"metric": "host-only", "n_gpus": world}))
    sys.exit(0)
"""


def test_spawn_ranks_relays_rank0_propagates_the_worst_code_and_ends_survivors(tmp_path):
    """bench.spawn_ranks (`python bench.py --gpus N` without a launcher) with a host-only rank body:
    the environment of one job on 127.0.0.1, rank 0's stdout relayed as the job's stdout and the other ranks' kept off it,
    the worst exit code propagated, and a rank left waiting for one that died terminated after the grace period."""
    import subprocess
    import time
    body = tmp_path / "rank_body.py"
    body.write_text(_RANK_BODY)
    drv = tmp_path / "drv.py"
    drv.write_text(f"import sys\nsys.path.insert(0, {ROOT!r})\nimport bench\n"
                   f"raise SystemExit(bench.spawn_ranks(int(sys.argv[1]), script={str(body)!r}, argv=sys.argv[2:], grace_s=2.0))\n")
    r = subprocess.run([sys.executable, str(drv), "3", "ok", "--steps", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0, r.stderr.decode()[-400:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    js = [json.loads(l) for l in lines if l.startswith("{")]
    assert js == [{"metric": "host-only", "n_gpus": 3, "args": ["ok", "--steps", "2"]}]
    assert lines.count("chatter from rank 0") == 1 and not any("rank 1" in l or "rank 2" in l for l in lines)
    assert "chatter from rank 1" in r.stderr.decode() and "chatter from rank 2" in r.stderr.decode()
    t0 = time.time()
    r = subprocess.run([sys.executable, str(drv), "3", "one_dies"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    dt = time.time() - t0
    assert r.returncode == 3, (r.returncode, r.stderr.decode()[-400:])
    assert dt < 30, f"the job took {dt:.1f} s to end"
    err = r.stderr.decode()
    assert "rank 1 exited with code 3" in err and "terminated 1 rank(s)" in err
    assert [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")] == [{"metric": "host-only", "n_gpus": 3}]


def _segments_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from building_detection_amd import zoo
    from building_detection_amd.dist import BucketReducer, TorchTransport, param_ranges, plan_buckets
    from building_detection_amd.runtime import _SegmentCuts, run_segments
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = zoo.Xception_DeepLabV3_Plus((64, 64, 3), 2, aspp_pool=4)
    total = m._n_train
    buckets = plan_buckets(param_ranges(m), total, 1 << 20)

    def grad_of(p, r):
        return torch.rand(p.size, generator=torch.Generator().manual_seed(p.offset * 7 + r))

    # "capture": walk the backward sweep once, cutting where the eager reducer would fire a bucket; a segment remembers which
    # nodes' gradient writes it contains (on the GPU: the kernels captured between two cuts)
    cuts = _SegmentCuts(buckets)
    segs, cur = [], []
    for n in reversed(m.nodes):
        cur.append(n)
        ready = cuts.pop_ready(n.index)
        if ready:
            segs.append((cur, ready))
            cur = []
    segs.append((cur, cuts.pop_ready(-1)))
    assert cuts.done()
    # the eager schedule, for comparison: same hand-over points
    eager = BucketReducer(torch.zeros(total), buckets, transport=type("T", (), {"allreduce_async": lambda s, t: None, "join": lambda s: None})())
    for n in reversed(m.nodes):
        eager.node_done(n.index)
    eager.finish()
    fired_eager = [s for _, s in eager.fired]
    fired_seg = [s for _, ready in segs for s, _ in ready]

    arena = torch.zeros(total)

    def launcher(nodes):
        def launch():
            for n in nodes:
                for p in n.params:
                    if p.trainable:
                        arena[p.offset:p.offset + p.size] = grad_of(p, rank)
        return launch

    ok = True
    for step in range(2):   # replayed twice: the second replay overwrites the summed arena with fresh per-rank gradients
        run_segments([(launcher(nodes), ready) for nodes, ready in segs], arena, TorchTransport())
        for p in m.params:
            if p.trainable:
                want = sum(grad_of(p, r) for r in range(world))
                ok = ok and torch.allclose(arena[p.offset:p.offset + p.size], want, atol=1e-6)
    # --- the same step with side graphs (round 4: GraphedTrainStep's lanes): the filter gradients of a segment - here the
    # gradients of its convolution kernels - are written by a separate side launch W_k, replayed behind M_k on the second
    # stream.  An in-line stand-in for the two streams records the order of the calls.
    from building_detection_amd import layers as L
    log = []

    class Lanes:
        def __init__(self):
            self.in_side = False
        def fork(self):
            log.append(("fork",))
        def on_side(self):
            lanes = self
            class Ctx:
                def __enter__(self_):
                    lanes.in_side = True
                def __exit__(self_, *a):
                    lanes.in_side = False
            return Ctx()
        def mark_side(self):
            log.append(("mark", len([1 for e in log if e[0] == "mark"])))
            return log[-1]
        def wait_on_main(self, ev):
            if ev is not None:
                log.append(("wait", ev[1]))

    lanes = Lanes()

    class Tp(TorchTransport):
        def allreduce_async(self, t):
            log.append(("allreduce", lanes.in_side))
            super().allreduce_async(t)

    def is_side(n):
        return isinstance(n, (L._ConvNode, L._SepConvNode))

    def main_launcher(k, nodes):
        def launch():
            log.append(("main", k))
            for n in nodes:
                for i, p in enumerate(n.params):
                    if p.trainable and not (is_side(n) and i == 0):
                        arena[p.offset:p.offset + p.size] = grad_of(p, rank)
        return launch

    def side_launcher(k, nodes):
        side = [n for n in nodes if is_side(n)]
        if not side:
            return None
        def launch():
            log.append(("side", k, lanes.in_side))
            for n in side:
                p = n.params[0]
                if p.trainable:
                    arena[p.offset:p.offset + p.size] = grad_of(p, rank)
        return launch

    arena.zero_()
    run_segments([(main_launcher(k, nodes), ready, side_launcher(k, nodes)) for k, (nodes, ready) in enumerate(segs)],
                 arena, Tp(), lanes)
    ok2 = True
    for p in m.params:
        if p.trainable:
            want = sum(grad_of(p, r) for r in range(world))
            ok2 = ok2 and torch.allclose(arena[p.offset:p.offset + p.size], want, atol=1e-6)
    # order: W_k on the side lane right after M_k; M_k (k >= 2) behind a wait for W_(k-2) where that exists; a segment with a
    # side graph hands its buckets over from the side lane; every mark is waited for exactly once
    order_ok = True
    has_side = [side_launcher(k, nodes) is not None for k, (nodes, _) in enumerate(segs)]
    pos = {e: i for i, e in enumerate(log) if e[0] in ("main",)}
    marks_of = {}
    nmark = 0
    for k, hs in enumerate(has_side):
        if hs:
            marks_of[k] = nmark
            nmark += 1
    for k in range(len(segs)):
        i_main = log.index(("main", k))
        if has_side[k]:
            order_ok = order_ok and log[i_main + 1] == ("fork",) and log[i_main + 2] == ("side", k, True)
        elif segs[k][1]:   # no side graph of its own, but buckets to hand over: from the side lane, behind a fork
            order_ok = order_ok and log[i_main + 1] == ("fork",) and log[i_main + 2] == ("allreduce", True)
        if k >= 2 and has_side[k - 2]:
            order_ok = order_ok and ("wait", marks_of[k - 2]) in log[:i_main]
    waits = [e[1] for e in log if e[0] == "wait"]
    order_ok = order_ok and sorted(waits) == list(range(nmark))
    ar_sides = [e[1] for e in log if e[0] == "allreduce"]
    want_sides = [True for k, (_, ready) in enumerate(segs) for _ in ready]   # every hand-over goes through the side lane
    order_ok = order_ok and ar_sides == want_sides
    q.put((rank, bool(ok and ok2 and order_ok), len(segs), fired_eager == fired_seg, len(buckets)))
    dist.destroy_process_group()


def test_segmented_data_parallel_step_gloo_world2():
    """The replay loop of a hipGraph-captured data-parallel step (runtime.GraphedTrainStep under DataParallel): segments cut
    where dist.BucketReducer would fire, each finished bucket all-reduced right after its segment was launched.  World-2 gloo
    run over the DeepLabv3+ arena with the real node order: the hand-over points equal the eager step's, every bucket is
    complete when it leaves (a bucket cut too early would miss a rank-specific gradient in the sum), twice in a row."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_segments_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, nseg, same_points, nb in res:
        assert ok, f"rank {rank}: the arena after the segmented step is not the sum of the per-rank gradients"
        assert same_points, "segments hand buckets over at other points than the eager reducer"
        assert nb >= 4 and 2 <= nseg <= nb + 1


def test_depthwise_dgrad_takes_over_the_batchnorm_backward_sums(monkeypatch):
    """Model._fuse (round 4): a 4-D BatchNormalization whose output gradient is written by the depthwise dgrad of ONE stride-1
    SeparableConv2D has its backward column sums produced by that kernel (sg_dwconv2d_dgrad_bnsums; layers._BNNode.sums_from /
    _SepConvNode.bnsum_src): (a) the layer's only consumer is that SeparableConv2D (through the ReLU absorbed into either of
    them): 39 layers of DeepLabv3+, 32 in the middle flow and 7 in the entry / exit flows; (b) the layer is applied by a residual
    Add (defer_add) whose output opens the next Xception block, and that block's first SeparableConv2D is the last consumer of
    the tensor in the backward sweep: the inputs of the 16 middle-flow blocks and of the exit flow.  None in the U-Nets (their BatchNormalization layers
    feed ordinary convolutions); SG_BN_SUMS=0 switches the pass off."""
    sys.path.insert(0, ROOT)
    from building_detection_amd import zoo, layers as L

    def count(m):
        pairs = [(n, n.sums_from) for n in m.nodes if isinstance(n, L._BNNode) and n.sums_from is not None]
        direct = through_add = 0
        for bn, sc in pairs:
            assert isinstance(sc, L._SepConvNode) and sc.bnsum_src is bn and sc.stride == 1
            t = bn.output
            while t.consumers and isinstance(t.consumers[0], L._ActNode) and t.consumers[0].fused_away and len(t.consumers) == 1:
                t = t.consumers[0].output
            if bn.defer_add is None:
                assert t.consumers == [sc]
                direct += 1
            else:
                add = bn.defer_add
                assert add in t.consumers and bn in add.bn_src and not add.relu
                r = add.output
                firsts = [c for c in r.consumers if c.index <= sc.index]
                assert len(firsts) == 1   # the SeparableConv2D (or the ReLU absorbed into it) comes first in node order
                through_add += 1
        assert sum(1 for n in m.nodes if isinstance(n, L._SepConvNode) and n.bnsum_src is not None) == len(pairs)
        return direct, through_add

    monkeypatch.delenv("SG_BN_SUMS", raising=False)
    assert count(zoo.BUILDERS["v3plus"]((128, 128, 3), 2, aspp_pool=8)) == (39, 17)
    assert count(zoo.BUILDERS["res34"]((64, 64, 3))) == (0, 0)
    monkeypatch.setenv("SG_BN_SUMS", "0")
    assert count(zoo.BUILDERS["v3plus"]((128, 128, 3), 2, aspp_pool=8)) == (0, 0)


def test_residual_adds_take_over_their_batchnorm_layers(monkeypatch):
    """Model._fuse: a BatchNormalization (no fused ReLU) whose only consumer is a two-operand Add is applied BY that Add
    (sg_add2_bn; layers._AddNode.bn_src / _BNNode.defer_add).  Graph-level check on the CPU: the Xception blocks of DeepLabv3+
    (23 layers: 16 middle-flow branches + the branch and the 1x1 shortcut of the entry / exit blocks + ...), HRNet's basic
    blocks, Res34-UNet's blocks through the ReLU they apply before the add; SG_BN_ADD=0 switches the pass off; every deferred
    layer is consumed by exactly the Add that lists it."""
    sys.path.insert(0, ROOT)
    from building_detection_amd import zoo, layers as L

    def count(m):
        adds = [n for n in m.nodes if isinstance(n, L._AddNode)]
        pairs = [(n, s) for n in adds for s in n.bn_src if s is not None]
        for n, s in pairs:
            cons = s.output.consumers
            if s.relu:   # BN -> ReLU (absorbed into the BN, an identity node now) -> Add
                assert len(cons) == 1 and isinstance(cons[0], L._ActNode) and cons[0].fused_away and cons[0].output.consumers == [n]
            else:
                assert cons == [n]
            assert s.defer_add is n and len(n.inputs) == 2
        assert sum(1 for b in m.nodes if isinstance(b, L._BNNode) and b.defer_add is not None) == len(pairs)
        return len(pairs)

    monkeypatch.delenv("SG_BN_ADD", raising=False)
    assert count(zoo.BUILDERS["v3plus"]((128, 128, 3), 2, aspp_pool=8)) == 23
    assert count(zoo.BUILDERS["hrnet"]((64, 64, 3))) == 42
    assert count(zoo.BUILDERS["res34"]((64, 64, 3))) == 20   # its blocks activate before they add: the Add applies BN and ReLU
    monkeypatch.setenv("SG_BN_ADD", "0")
    assert count(zoo.BUILDERS["v3plus"]((128, 128, 3), 2, aspp_pool=8)) == 0


def test_bench_bf16_leg_folds_the_child_line(monkeypatch):
    """bench.py (round 4): BASELINE configs[2]'s per-GPU workload rides in the fp32 line as config.bf16_leg - a CHILD process
    `bench.py --dtype bf16` after and outside the fp32 timed region (started, never exec'ed).  Host logic only: the command the
    parent builds, what it keeps of the child's JSON line, and that a failing child leaves an error entry instead of an
    exception (the fp32 line must not depend on it)."""
    import json
    import subprocess
    import types
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    line = {"metric": "m", "value": 505.0, "ms_per_step": 31.7, "steps": 10, "warmup": 3, "dtype": "bf16", "dtype_note": "n",
            "roofline": {"frac": 0.33, "achieved": 840.0, "peak": 2500.0, "unit": "TFLOP/s", "ms_per_step": 1.86,
                         "family": {"frac": 0.2, "achieved": 490.0, "ms_per_step": 19.7}},
            "config": {"train_step": "eager launches", "train_step_choice": None, "final_loss": 0.02, "host_enqueue_ms_per_step": 23.0,
                       "peak_device_memory_gib": 13.5}}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return types.SimpleNamespace(returncode=0, stdout=("noise\n" + json.dumps(line) + "\n").encode(), stderr=b"")
    monkeypatch.setattr(subprocess, "run", fake_run)
    args = types.SimpleNamespace(batch=16, size=512, model="v3plus")
    leg = bench.bf16_leg(args)
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1].endswith("bench.py")
    for flag in ("--dtype", "bf16", "--no-cpu-baseline", "--no-bf16-leg", "--steps", "10", "--warmup", "3"):
        assert flag in cmd, (flag, cmd)
    assert leg["ms_per_step"] == 31.7 and leg["tiles_per_s"] == 505.0 and leg["dtype"] == "bf16"
    assert leg["roofline"]["frac"] == 0.33 and leg["family"]["frac"] == 0.2 and leg["steps"] == 10

    def failing_run(cmd, **kw):
        return types.SimpleNamespace(returncode=3, stdout=b"", stderr=b"boom")
    monkeypatch.setattr(subprocess, "run", failing_run)
    assert "error" in bench.bf16_leg(args)
