"""Data-parallel step on the GPU: RCCL through libsegengine's sg_comm_* and through torch.distributed.

World size 1 runs on the one-GPU box (the collective degenerates to a copy, but the whole path - id exchange,
communicator, comm stream, bucket hooks, scalar reductions, rank-0 checkpoint - executes).  The world-size-2 case needs
two GPUs and is skipped otherwise; it checks the averaged gradient against the two single-rank gradients."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    from building_detection_amd import zoo
    from building_detection_amd.losses import edge_focal_loss, PA, IoU, MIoU, F1_score
    m = zoo.Xception_DeepLabV3_Plus((64, 64, 3), 2, aspp_pool=4)
    m.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score])
    return m


def _worker(rank, world, port, comm, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from building_detection_amd.data import synthetic_batch
    from building_detection_amd.dist import DataParallel
    torch.cuda.set_device(rank)
    backend = "nccl" if comm == "torch" else "gloo"
    kw = {"device_id": torch.device("cuda", rank)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        shards = [synthetic_batch(2, 64, 64, seed=40 + r) for r in range(world)]
        # single-rank references first (no DP attached): gradient arena of every shard, from the same weights
        ref = _build()
        rt = ref._runtime()
        w0 = ref.get_weights()
        grads, losses, counts = [], [], []
        for x, y in shards:
            xd, yd = rt.to_device(x), rt.to_device(y)
            p = rt.forward(xd, training=True)
            losses.append(float(rt.eng.loss_fwd(ref.loss_kind, p, yd).item()))
            counts.append(rt.eng.confusion_counts(p, yd).cpu().numpy())
            rt.backward(rt.eng.loss_bwd(ref.loss_kind, p, yd, 1.0))
            grads.append(rt.g_train.clone())
            rt.release()
        # the data-parallel model: same weights, own shard
        m = _build()
        m.set_weights(w0)
        dp = DataParallel(m, bucket_mb=8.0, comm=comm)
        assert dp.tp.name == comm and dp.world == world
        mrt = m._runtime()
        x, y = shards[rank]
        logs = m.train_on_batch(x, y)
        gsum = mrt.g_train.clone()  # Adam does not touch the gradient arena
        want = grads[0].clone()
        for g in grads[1:]:
            want += g
        torch.cuda.synchronize()
        err = float((gsum - want).abs().max() / want.abs().max())
        # the step itself: Adam on the MEAN gradient == single-process Adam fed the mean gradient
        chk = _build()
        chk.set_weights(w0)
        crt = chk._runtime()
        crt.g_train.copy_(want)
        crt.eng.adam_step(crt.w_train, crt.adam_m, crt.adam_v, crt.g_train, 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9),
                          grad_scale=1.0 / world)
        torch.cuda.synchronize()
        werr = float((crt.w_train - mrt.w_train).abs().max())
        tot = np.sum(counts, axis=0)
        from building_detection_amd.losses import metrics_from_counts
        mets = metrics_from_counts(*[int(v) for v in tot])
        path = os.path.join(tmp, "ckpt.h5")
        m.save_weights(path)  # collective: moving statistics averaged, rank 0 writes
        q.put(dict(rank=rank, err=err, werr=werr, loss=logs["loss"], loss_want=float(np.mean(losses)),
                   miou=logs["MIoU"], miou_want=mets["MIoU"], wrote=os.path.exists(path), buckets=len(dp.buckets)))
    finally:
        dist.destroy_process_group()


def _run(world, comm, tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, comm, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=400))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in res:
        assert r["err"] < 1e-6, r           # sum order inside RCCL is the only freedom
        assert r["werr"] < 1e-7, r
        assert abs(r["loss"] - r["loss_want"]) < 1e-6 * max(1.0, abs(r["loss_want"])), r
        assert abs(r["miou"] - r["miou_want"]) < 1e-6, r
        assert r["wrote"] and r["buckets"] >= 2, r
    return res


@pytest.mark.parametrize("comm", ["sg", "torch"])
def test_data_parallel_step_world1(engine, comm, tmp_path):
    _run(1, comm, tmp_path)


@pytest.mark.parametrize("comm", ["sg", "torch"])
def test_data_parallel_step_world2(engine, comm, tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run(2, comm, tmp_path)


def _jit_worker(rank, world, port, comm, policy, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from building_detection_amd import mixed_precision as MP, zoo
    from building_detection_amd.data import synthetic_batch
    from building_detection_amd.dist import DataParallel
    from building_detection_amd.losses import edge_focal_loss, PA, IoU
    from building_detection_amd.runtime import GraphedTrainStep
    torch.cuda.set_device(rank)
    backend = "nccl" if comm == "torch" else "gloo"
    kw = {"device_id": torch.device("cuda", rank)} if backend == "nccl" else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        MP.set_global_policy(policy)
        try:
            ma = zoo.Xception_DeepLabV3_Plus((64, 64, 3), 2, aspp_pool=4)
            mb = zoo.Xception_DeepLabV3_Plus((64, 64, 3), 2, aspp_pool=4)
        finally:
            MP.set_global_policy("float32")
        mb.set_weights(ma.get_weights())
        ma.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU])
        mb.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU], jit_compile=True)
        da = DataParallel(ma, bucket_mb=8.0, comm=comm)
        db = DataParallel(mb, bucket_mb=8.0, comm=comm)
        same = True
        xv, yv = synthetic_batch(3, 64, 64, seed=99 + rank)
        for i in range(7):
            x, y = synthetic_batch(2, 64, 64, seed=40 + 10 * rank + i)
            la, lb = ma.train_on_batch(x, y), mb.train_on_batch(x, y)
            same = same and la == lb
            if i == 3:   # another batch size in between (re-keys the runtime's weight planes), and a new learning rate
                same = same and ma.test_on_batch(xv, yv) == mb.test_on_batch(xv, yv)
                ma.optimizer.lr = mb.optimizer.lr = 3e-4
        g = next(iter(mb._train_graphs.values()))
        weights_equal = all(np.array_equal(a, b) for a, b in zip(ma.get_weights(), mb.get_weights()))
        q.put(dict(rank=rank, same_logs=same, weights_equal=weights_equal, graphed=isinstance(g, GraphedTrainStep),
                   segments=len(g.segments), buckets=len(db.buckets), eager_graphs=len(getattr(ma, "_train_graphs", {}) or {}),
                   side_graphs=g.side_graphs, handover_segments=sum(1 for seg in g.segments if seg[1]),
                   lanes=os.environ.get("SG_JIT_LANES", "1") != "0" and os.environ.get("SG_SIDE_WGRAD", "1") != "0"))
    finally:
        dist.destroy_process_group()


def _run_jit(world, comm, policy):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + os.getpid() % 2000
    procs = [ctx.Process(target=_jit_worker, args=(r, world, port, comm, policy, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=400))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for r in res:
        assert r["graphed"] and r["eager_graphs"] == 0, r
        assert r["same_logs"] and r["weights_equal"], r
        # one cut per point of the sweep that completes buckets (<= one per bucket), a tail segment when the sweep goes on
        # behind the last cut, and the Adam graph; with lanes (round 4: the filter gradients of a segment in a side graph of
        # their own, replayed on the second stream) further cuts wherever a segment has collected SG_JIT_LANE_BLOCKS of them
        assert r["buckets"] >= 2 and 1 <= r["handover_segments"] <= r["buckets"] and r["segments"] >= 3, r
        if r["lanes"]:   # (with 24 buckets the bucket cuts come before a segment has collected 24 filter gradients)
            assert r["side_graphs"] >= 2 and r["segments"] >= r["handover_segments"] + 1, r
        else:
            assert r["side_graphs"] == 0 and r["segments"] <= r["buckets"] + 2, r


@pytest.mark.parametrize("comm,policy", [("sg", "float32"), ("torch", "float32"), ("sg", "mixed_bfloat16")])
def test_captured_data_parallel_step_is_bit_identical_to_the_eager_one_world1(engine, comm, policy):
    """compile(jit_compile=True) under DataParallel.  The step is captured as hipGraph segments
    (cut where the backward sweep completes a gradient bucket, and - lanes - where a segment has collected enough filter
    gradients, which are captured into side graphs replayed on the second stream) plus one graph for Adam; the bucket all-reduces
    (sg_comm_allreduce_sum / torch.distributed nccl) are issued eagerly between the segment launches, the loss and the
    confusion counts are reduced as in the eager step.  Seven steps against the eager data-parallel model on the same
    weights and batches: identical logs at every step and identical weights at the end, to the bit, with a validation batch
    of another size and a learning-rate change in between."""
    _run_jit(1, comm, policy)


@pytest.mark.parametrize("comm", ["sg", "torch"])
def test_captured_data_parallel_step_world2(engine, comm):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_jit(2, comm, "float32")
