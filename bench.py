#!/usr/bin/env python3
"""Headline benchmark: 512x512 tiles/s for one full DeepLabv3+ training step (forward + edge_focal_loss +
metrics + backward + Adam [+ gradient all-reduce]) in fp32, batch 16 per GPU, on N MI355X (BASELINE.json
configs[1]).  Synthetic tiles, random-init weights, inputs resident in HBM before timing.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     the north_star target kernel set = the six dilated 3x3 convs (3 ASPP + 3 SK branches) fwd +
               dgrad + wgrad: algorithmic FLOPs (nominal 2*M*N*K, SURVEY §8d: 97.84 GFLOP/tile) divided by
               their summed device time, measured with HIP events on the launch stream inside the timed
               steps; peak = 2500/6 TFLOP/s on the x6 path (six bf16 MFMA passes per fp32 product), 157.3 with
               SG_CONV_X6=0.  roofline.family: the same figure over EVERY GEMM-convolution launch of the step
               (all of the model's conv FLOPs / their summed device time), from two extra steps after the timed
               region so that its ~600 event pairs do not sit inside the headline number.
  cpu_baseline the CPU oracle (restatement of the TF2 path; TF itself is unavailable) timed on this box's
               host cores for the same step at bs 2: median of 5 steps after 2 warm-ups.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC: without this RCCL's peer mappings fail in hipIpcGetMemHandle.  Exported on
# the image already; set here too so that a launcher with a scrubbed environment still gets working ranks (read at HSA start-up,
# i.e. before torch touches the GPU)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FP32_MFMA_PEAK_TFLOPS = 157.3
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
LABEL = {"v3plus": "DeepLabv3+ (v3plus.py)", "bam": "DeepLabv3+ BAM (bam.py)", "scse": "SCSE-UNet (scse.py)",
         "res34": "Res34-UNet (res34.py)", "hrnet": "HRNet (hrnet.py)"}
DILATED_GFLOP_PER_TILE = 97.84  # fwd + dgrad + wgrad of the 6 dilated convs, SURVEY.md §8d


def cpu_baseline(threads, batch, size, steps, model=None, warmup=2):
    """the CPU oracle's training step (fwd + loss + bwd + Adam) on the host cores - `warmup` untimed
    steps, then the MEDIAN of `steps` (>= 5) timed ones; `nproc`, the CPUs this process may run on and the threads used are
    stated in the record.  With `model` (the engine's DeepLabv3+, after its timed steps) the same weights and tiles are also
    run through both in inference mode: the metric's "mIoU vs TF2 CPU" leg, with the CPU oracle standing in for TF2 (absent)."""
    import statistics
    import torch
    from oracle import models as M
    from building_detection_amd.data import synthetic_batch

    def timed_steps(fn, xs, ys, threads_, warm, reps):
        """-> (Params, per-step seconds of the `reps` steps after `warm` untimed ones)"""
        torch.set_num_threads(threads_)
        Pq = M.Params(seed=1103)
        ts = []
        mq = vq = None
        for it in range(warm + reps):
            t0 = time.time()
            lq = M.loss_fn("edge_focal_loss", ys, fn(Pq, xs, training=True))
            trq = Pq.trainable_tensors()
            for t in trq:
                t.grad = None
            lq.backward()
            if mq is None:   # the first iteration creates the parameters
                mq, vq = [torch.zeros_like(t) for t in trq], [torch.zeros_like(t) for t in trq]
            M.adam_step(trq, [t.grad for t in trq], mq, vq, t=it + 1, lr=1e-3)
            if it >= warm:
                ts.append(time.time() - t0)
        torch.set_num_threads(threads)
        return Pq, ts

    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    x, y = synthetic_batch(batch, size, size, seed=1103)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    P, times = timed_steps(lambda Pq, x_, training: M.deeplab_v3plus(Pq, x_, training=training), xt, yt, threads, warmup, steps)
    med = statistics.median(times)
    out = {"value": round(batch / med, 4), "unit": "tiles/s", "cores": threads, "kind": "port",
           "nproc": os.cpu_count(), "cpus_allowed": affinity, "threads": threads,
           "step_seconds": [round(t, 3) for t in times],
           "bs16_step_seconds_extrapolated_linearly": round(med * 16 / batch, 2),
           "sample": f"CPU restatement of the TF2 path (TF unavailable; oracle/, torch CPU ops) DeepLabv3+ {size}x{size} bs={batch}: "
                     f"median of {steps} full steps (fwd+loss+bwd+Adam) after {warmup} warm-ups, {med:.2f} s/step on {threads} threads"}
    # BASELINE.md section 3 extras, bounded: (a) the same step on ONE thread (one 256x256 tile: 1/8 of the bs-2 512x512
    # sample's work; one timed step after one warm-up - five would take a minute), (b) BASELINE configs[0]: Res34-UNet
    # 256x256 bs 2 on all threads, median of 5 after 2, (c) the dilated set alone
    try:
        xs1, ys1 = synthetic_batch(1, 256, 256, seed=1103)
        _, t1 = timed_steps(lambda Pq, x_, training: M.deeplab_v3plus(Pq, x_, training=training, aspp_pool=16), torch.from_numpy(xs1),
                            torch.from_numpy(ys1), 1, 1, 1)
        out["one_thread"] = {"value": round(0.25 / t1[0], 4), "unit": "512x512-tile equivalents/s", "cores": 1,
                             "sample": f"DeepLabv3+ one 256x256 tile (a quarter of a 512x512 tile's work), one full step after one warm-up, {t1[0]:.2f} s"}
        xs2, ys2 = synthetic_batch(2, 256, 256, seed=1103)
        _, t2 = timed_steps(lambda Pq, x_, training: M.res34_unet(Pq, x_, training=training), torch.from_numpy(xs2), torch.from_numpy(ys2),
                            threads, 2, 5)
        m2 = statistics.median(t2)
        out["config1_res34_256_bs2"] = {"value": round(2 / m2, 4), "unit": "256x256 tiles/s", "cores": threads,
                                        "sample": f"BASELINE configs[0]: Res34-UNet 256x256 bs=2 full step, median of 5 after 2 warm-ups, {m2:.2f} s/step"}
        # (c) the roofline kernel set on the CPU: the six dilated 3x3 convolutions (3 x 2048 -> 256, 3 x 256 -> 256 at 32 x 32,
        # rates 6 / 12 / 18), forward + both gradients, bs 2, through the oracle's conv2d; median of 3 after one warm-up
        from oracle import tfops as T
        gq = torch.Generator().manual_seed(3)
        tt = 0.0
        for cin in (2048, 256):
            for rate in (6, 12, 18):
                xq = torch.randn(2, 32, 32, cin, generator=gq).requires_grad_()
                wq = (torch.randn(3, 3, cin, 256, generator=gq) * 0.02).requires_grad_()
                reps = []
                for rep in range(4):
                    xq.grad = wq.grad = None
                    t0 = time.time()
                    T.conv2d(xq, wq, None, 1, rate, "same").sum().backward()
                    reps.append(time.time() - t0)
                tt += statistics.median(reps[1:])
        out["dilated_set"] = {"value": round(DILATED_GFLOP_PER_TILE * 2 / 1e3 / tt, 4), "unit": "TFLOP/s (nominal)", "cores": threads,
                              "sample": f"the six dilated 3x3 convolutions fwd + dgrad + wgrad, bs=2 (bs 16 = 8x), median of 3 after 1, {tt:.2f} s"}
    except Exception as e:  # extras never take the main figure down
        out["extras_error"] = repr(e)
    if model is not None:
        import numpy as np
        with torch.no_grad():
            p_cpu = M.deeplab_v3plus(P, xt, training=False)
        model.set_weights([t.detach().numpy() for t in P.tensors])
        p_gpu = model.predict(x)
        cm_c = M.metrics_from_counts(*M.confusion(yt, p_cpu))
        cm_g = M.metrics_from_counts(*M.confusion(yt, torch.from_numpy(p_gpu)))
        pc = p_cpu.numpy()
        out["parity"] = {"what": "predict() of the engine vs the CPU oracle, same weights (after the oracle's training steps) "
                                 f"and the same {batch} tiles",
                         "max_abs_prob_diff": float(np.abs(p_gpu - pc).max()),
                         "argmax_mismatch_pixels": int((p_gpu.argmax(-1) != pc.argmax(-1)).sum()),
                         "MIoU_gpu": round(cm_g["MIoU"], 6), "MIoU_cpu_oracle": round(cm_c["MIoU"], 6)}
    return out


def bf16_leg(args):
    """BASELINE configs[2] on one GPU: `bench.py --dtype bf16` (same model, batch and tile size; 3 warm-up + 10 timed steps, its
    own family / dilated-set brackets) as a CHILD process; returns the figures the driver's record should carry."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--dtype", "bf16", "--steps", "10", "--warmup", "3",
           "--batch", str(args.batch), "--size", str(args.size), "--model", args.model, "--no-cpu-baseline", "--no-bf16-leg"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": f"exit {r.returncode}: {r.stderr.decode()[-400:]}"}
        j = json.loads(line[-1])
        rf = j.get("roofline", {})
        return {"ms_per_step": j["ms_per_step"], "tiles_per_s": j["value"], "steps": j["steps"], "warmup": j["warmup"],
                "dtype": j["dtype"], "dtype_note": j.get("dtype_note"),
                "roofline": {"frac": rf.get("frac"), "achieved": rf.get("achieved"), "peak": rf.get("peak"), "unit": rf.get("unit"),
                             "ms_per_step": rf.get("ms_per_step")},
                "family": {"frac": (rf.get("family") or {}).get("frac"), "achieved": (rf.get("family") or {}).get("achieved"),
                           "ms_per_step": (rf.get("family") or {}).get("ms_per_step")},
                "train_step": j["config"].get("train_step"), "train_step_choice": j["config"].get("train_step_choice"),
                "final_loss": j["config"].get("final_loss"), "host_enqueue_ms_per_step": j["config"].get("host_enqueue_ms_per_step"),
                "peak_device_memory_gib": j["config"].get("peak_device_memory_gib"),
                "note": "child process `bench.py --dtype bf16`, after and outside the fp32 timed region"}
    except Exception as e:   # the leg must never take the fp32 line down
        return {"error": repr(e)}


def spawn_ranks(n: int, script: str = None, argv=None, grace_s: float = None) -> int:
    """Start `n` copies of `script` (default: this file, with this process's arguments) as ranks 0..n-1 of one job (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, rendezvous on 127.0.0.1), wait for them, return the worst exit
    code.  Rank 0's stdout (the JSON line) is passed through; the other ranks' stdout goes to stderr.  A rank that ends with
    a non-zero code leaves the others inside a collective it will never join: they get `grace_s` seconds (SG_SPAWN_GRACE,
    default 20) to notice and exit by themselves, then exactly the children started here are terminated (killed if they
    ignore that) - the job never hangs, and its exit code is the first failure's.  The parent never touches the GPU."""
    import socket
    import subprocess
    script = os.path.abspath(__file__) if script is None else script
    argv = sys.argv[1:] if argv is None else list(argv)
    grace_s = float(os.environ.get("SG_SPAWN_GRACE", "20")) if grace_s is None else grace_s
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env,
                                      stdout=_JSON_FD if r == 0 else sys.stderr))  # rank 0 gets the REAL stdout
    rc, deadline = 0, None
    try:
        while True:
            codes = [p.poll() for p in procs]
            for r, c in enumerate(codes):
                if c not in (None, 0) and deadline is None:
                    rc = abs(c) or 1
                    deadline = time.time() + grace_s
                    print(f"[bench] rank {r} exited with code {c}; the other ranks have {grace_s:.0f} s to follow", file=sys.stderr)
            if all(c is not None for c in codes):
                break
            if deadline is not None and time.time() > deadline:
                break
            time.sleep(0.05)
        for c in (p.poll() for p in procs):
            if c not in (None, 0):
                rc = max(rc, abs(c))
    finally:
        left = [p for p in procs if p.poll() is None]
        for p in left:  # a rank that died leaves the others in a collective: end exactly the children we started
            p.terminate()
        for p in left:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        if left:
            rc = max(rc, 1)
            print(f"[bench] terminated {len(left)} rank(s) that were still running", file=sys.stderr)
    return rc


# The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner from
# ncclCommInitRank): file descriptor 1 is pointed at stderr for the whole run and the JSON line goes to a saved copy of the
# real stdout.
_JSON_FD = 1


def _claim_stdout():
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)


def main():
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="tiles per GPU (BASELINE config: 16)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--model", default="v3plus")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = BASELINE configs[1] (the reference's precision, default); bf16 = configs[2]: bf16 activation "
                         "storage, one-pass bf16 MFMA products with fp32 accumulation, fp32 master weights / BN / loss / Adam")
    ap.add_argument("--jit", dest="jit", action="store_true", default=None,
                    help="Model.compile(jit_compile=True): the training step replayed as hipGraphs - ONE graph on a single "
                         "GPU, one segment per gradient bucket under data parallelism with the RCCL all-reduces issued "
                         "eagerly between them (runtime.GraphedTrainStep).  The roofline figures then come from two extra "
                         "EAGER steps after the timed region, because launches inside a replay cannot be bracketed with "
                         "events.  Default: on for --gpus > 1 (eight Python hosts queueing ~1800 launches per step each leave "
                         "no slack); on one GPU the warm-up tries both forms and the timed region runs the faster one "
                         "(config.train_step_choice)")
    ap.add_argument("--no-jit", dest="jit", action="store_false")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bf16-leg", action="store_true",
                    help="skip config.bf16_leg (BASELINE configs[2]'s per-GPU workload: the same model and batch with bf16 storage, "
                         "3 warm-up + 10 steps in a child process after the fp32 timed region; single-GPU fp32 DeepLabv3+ runs only)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--comm", default=os.environ.get("SG_BENCH_COMM", "sg"), choices=["sg", "torch"],
                    help="gradient all-reduce transport: sg = RCCL through libsegengine's sg_comm_* (C ABI), torch = "
                         "torch.distributed 'nccl' (also RCCL)")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the RCCL data-parallel path even with one rank (rehearsal on a 1-GPU box)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU).  The parent
        # never touches the GPU (no HIP call before or after the spawn), it only relays rank 0's JSON line.
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one process per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.comm == "torch":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:  # the process group is only the rendezvous of the 128-byte RCCL id and the host-side barrier
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from building_detection_amd import zoo
    from building_detection_amd.data import synthetic_batch
    from building_detection_amd.losses import edge_focal_loss, PA, IoU, MIoU, F1_score
    from building_detection_amd.ops import get_engine

    eng = get_engine(local_rank)
    if args.dtype == "bf16":
        from building_detection_amd import mixed_precision
        mixed_precision.set_global_policy("mixed_bfloat16")
    if args.model in ("v3plus", "bam"):
        model = zoo.BUILDERS[args.model]((args.size, args.size, 3), 2, aspp_pool=args.size // 16)
    else:
        model = zoo.BUILDERS[args.model]((args.size, args.size, 3))
    # One GPU, neither --jit nor --no-jit: the step is captured during the warm-up AND tried both ways (three untimed steps
    # each, median); the timed region runs the faster form.  Eager launches overlap the filter gradients on a second stream (DESIGN
    # 10.9) and are 2 - 5 % faster while the Python host keeps ahead of the GPU (60 ms of enqueue per 77 ms step); on a box whose
    # CPUs are busy with other tenants' work the same step was measured host-bound at 134 ms, where the replay (0.5 ms of host
    # time per step) does not care.
    auto = args.jit is None and world == 1
    jit = True if auto else (bool(args.jit) if args.jit is not None else (world > 1))
    model.compile(optimizer="adam", loss=edge_focal_loss, metrics=[PA, IoU, MIoU, F1_score], jit_compile=jit)
    if jit and args.warmup < 3:
        args.warmup = 3   # two eager steps per shape, the third call captures (and replays) the graph
    if dist is not None:
        from building_detection_amd.dist import DataParallel
        from building_detection_amd.dist import CommInitError
        try:
            dp = DataParallel(model, comm="sg_or_torch" if args.comm == "sg" else "torch")
        except CommInitError as e:
            # ncclCommInitRank failed or timed out on some rank; EVERY rank is here (dist.SgTransport).  A thread of this
            # process may still sit inside RCCL: no clean-up, no re-exec (this process has touched the GPU) - end it with a
            # non-zero code; the launcher (spawn_ranks / torchrun) reports the job as failed and fresh processes retry.
            print(f"[bench] rank {rank}: {e}", file=sys.stderr)
            sys.stderr.flush()
            os._exit(13)
        args.comm = dp.tp.name   # what actually carries the gradients ("sg" may have fallen back to "torch")

    # rank r takes tiles [16 r, 16 r + 16) of the global synthetic batch (weak scaling)
    x, y = synthetic_batch(args.batch, args.size, args.size, seed=1103 + rank)
    xd = torch.from_numpy(x).cuda()
    yd = torch.from_numpy(y).cuda()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        try:
            model.train_on_batch(xd, yd, return_device_scalars=True)
        except Exception as e:   # the capture (third call) failed: this rank goes on eagerly - the eager and the replayed
            if not (jit and i >= 2):   # step issue the SAME sequence of collectives, so the ranks stay in step
                raise
            print(f"[bench] rank {rank}: hipGraph capture of the train step failed ({e!r}); running eager launches", file=sys.stderr)
            jit = model.jit_compile = False
            model.train_on_batch(xd, yd, return_device_scalars=True)
    sync()
    choice = None
    if auto and jit:
        def trial(n=3):   # median of n individually timed steps: one stalled step (a busy neighbour) does not decide
            ts = []
            for _ in range(n):
                sync()
                t = time.perf_counter()
                model.train_on_batch(xd, yd, return_device_scalars=True)
                sync()
                ts.append((time.perf_counter() - t) * 1e3)
            return sorted(ts)[n // 2]
        t_replay = trial()
        model.jit_compile = False
        t_eager = trial()
        jit = model.jit_compile = not (t_eager < 0.995 * t_replay)   # a tie goes to the replay: it cannot become host-bound
        choice = {"untimed_trial_ms_per_step": {"eager_two_streams": round(t_eager, 3), "hipgraph_replay": round(t_replay, 3)},
                  "chosen": "hipgraph_replay" if jit else "eager_two_streams"}
        sync()
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = model.train_on_batch(xd, yd, return_device_scalars=True)
    t_enq = time.perf_counter() - t0   # the host has queued every launch of the timed steps (the device may still be running)
    sync()
    dt = time.perf_counter() - t0
    prof = eng.profile_end()
    peak_mem = torch.cuda.max_memory_allocated(local_rank)
    # outside the timed region: two more steps with EVERY GEMM-convolution launch bracketed (the whole kernel family)
    # (every rank runs them - the data-parallel step all-reduces - but only rank 0 brackets its launches)
    fam = {}
    if jit:
        model.jit_compile = False   # the two bracketed steps below run eagerly
        # ... after one eager step that is neither timed nor bracketed: the replays ran on private buffers, and the first eager
        # step behind them re-grows the shared workspaces - bracketed, it read 54 - 59 ms for the family instead of 52.5 and 6.3
        # instead of 6.15 ms for the dilated set (gpurun_out/r5L; the default one-GPU line tries both forms first and never
        # showed it, but every N > 1 line runs the replay)
        model.train_on_batch(xd, yd, return_device_scalars=True)
    if rank == 0:
        eng.profile_begin(all_convs=True)
    for _ in range(2):
        model.train_on_batch(xd, yd, return_device_scalars=True)
    if rank == 0:
        fam = eng.profile_end()
    sync()
    if jit and rank == 0:   # the dilated set of the two eager steps stands in for the (unbracketable) replays
        prof = {"dilated_conv": fam.get("dilated_conv", 0.0) * args.steps / 2.0,
                "dilated_conv_launches": fam.get("dilated_conv_launches", 0) * args.steps // 2}
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    tiles_per_s = world * args.batch * args.steps / dt
    step_tflop = 3 * model.flops(args.batch) / 1e12  # fwd + dgrad + wgrad, nominal

    if rank == 0:
        dil_ms = prof.get("dilated_conv", 0.0) / max(args.steps, 1)
        dil_tflop = DILATED_GFLOP_PER_TILE * args.batch / 1e3 * (args.size / 512.0) ** 2
        achieved = dil_tflop / (dil_ms / 1e3) if dil_ms > 0 else None
        # The fp32 products of these convolutions run as six bf16 MFMA passes over an exact 3-way bf16 split of the
        # fp32 operands, fp32 accumulation (csrc/conv_x6.h; accuracy >= the fp32 MFMA, profiles/r01_exp_bf16x6.txt).
        # The pipe that bounds them is therefore the bf16 one: 2500 TFLOP/s dense / 6 passes per fp32 FLOP.  With
        # SG_CONV_X6=0 they run on the fp32 MFMA (157.3 TFLOP/s) instead.
        b16 = args.dtype == "bf16"
        x6 = os.environ.get("SG_CONV_X6", "1") != "0" or b16
        peak = BF16_MFMA_PEAK_TFLOPS if b16 else (BF16_MFMA_PEAK_TFLOPS / 6.0 if x6 else FP32_MFMA_PEAK_TFLOPS)
        peak_note = ("bf16 dense MFMA peak (one v_mfma_f32_32x32x16_bf16 pass per product, fp32 accumulation)" if b16 else
                     ("bf16 dense MFMA peak 2500 TFLOP/s / 6 MFMA passes per fp32 product (x6 path: fp32 in, fp32 accumulate, "
                      "exact 3-way bf16 split); the native fp32 MFMA peak is 157.3 TFLOP/s") if x6 else "fp32 MFMA 32x32x2 dense peak")
        # fabric-side bytes of the same kernel set per step, from the committed PMC passes (separate rocprofv3
        # --pmc runs of scripts/dilated_bench.py at this very configuration; scripts/pmc_traffic.py)
        traffic, traffic_note = None, "PMC passes exist for the 512x512 bs16 fp32 configuration only"
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        tj = next((os.path.join(pdir, n) for n in ("r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json") if os.path.exists(os.path.join(pdir, n))), None)
        if args.batch == 16 and args.size == 512 and tj and not b16:
            with open(tj) as f:
                tr = json.load(f)
            traffic = int(tr["set_bytes_per_step"])
            traffic_note = ("NOT measured in this run: a committed constant from the separate rocprofv3 --pmc passes of %s "
                            "(FETCH_SIZE pass x2 for gfx950's 64-B tally + WRITE_SIZE pass, L2-miss side, Infinity-Cache hits "
                            "included) over the same kernel set at this configuration, launched as the step launches it (round 5: "
                            "scripts/dilated_step.py - activation planes made once per tensor, planes-in filter gradients); "
                            "algorithmic bytes of the set = %d (x %.2f)"
                            % (os.path.basename(tj), int(tr["algorithmic_bytes_per_step"]), traffic / tr["algorithmic_bytes_per_step"]))
        # Executed share of the nominal FLOPs: the kernels skip whole padding taps (exact: the skipped products are x0), so
        # the matrix pipe executes fewer bf16 MFMA operations than 6 x nominal.  Counted by SQ_INSTS_VALU_MFMA_MOPS_BF16
        # (x512 FLOP) in a separate rocprofv3 --pmc pass over this kernel set (scripts/pmc_mfma.py); it depends on the
        # shapes only, not on the run.
        exec_ratio, exec_note = None, "no PMC pass for this configuration"
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
        pj = next((q for q in (os.path.join(pdir, n) for n in ("r05_pmc_mfma.json", "r04_pmc_mfma.json", "r03_pmc_mfma.json", "r01_pmc_mfma.json")) if os.path.exists(q)),
                  os.path.join(pdir, "r01_pmc_mfma.json"))
        if args.batch == 16 and args.size == 512 and x6 and os.path.exists(pj):
            with open(pj) as f:
                rec = json.load(f)
            if "bf16_mfma_flops_per_step" in rec:   # round 5: scripts/pmc_set.py over scripts/dilated_step.py, already per step
                executed = rec["bf16_mfma_flops_per_step"] / 6.0  # fp32-equivalent
            else:
                k = rec["kernels"]
                # the profiled script launches every forward 3x, dgrad / wgrad 2x (warm-up + ITERS=1): 30 and 12 launches
                executed = (k["x6_fwd_dgrad"]["bf16_mfma_flops"] * 12 / k["x6_fwd_dgrad"]["launches"]
                            + k["x6_wgrad"]["bf16_mfma_flops"] * 6 / k["x6_wgrad"]["launches"]) / 6.0  # fp32-equivalent
            exec_ratio = executed / (dil_tflop * 1e12)
            exec_note = ("NOT measured in this run: a committed constant from a separate counter pass (SQ_INSTS_VALU_MFMA_MOPS_BF16 x "
                         "512, profiles/%s; it depends on the layer shapes only - rounds 1, 3 and 4 count the same operations) "
                         "/ 6 / nominal FLOPs: padding taps of the dilated convs are skipped, not multiplied" % os.path.basename(pj))
        out = {
            "metric": f"{args.size}x{args.size} tiles/sec fwd+bwd {LABEL.get(args.model, args.model)} (full train step: fwd+loss+bwd+Adam)",
            "value": round(tiles_per_s, 3), "unit": "tiles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if b16 else "f32", "data": "synthetic",
            "dtype_note": ("bf16 activation storage, conv products in one bf16 MFMA pass with fp32 accumulation; fp32 master "
                           "weights, BatchNorm statistics, softmax head, loss, weight gradients and Adam (BASELINE configs[2])") if b16
                          else ("fp32 tensors, fp32 accumulation; conv products as 6 bf16 MFMA passes over an exact 3-way split" if x6 else "fp32 MFMA"),
            "config": {"workload": f"{LABEL.get(args.model, args.model)} {args.size}x{args.size} bs={args.batch}/GPU {'bf16' if b16 else 'fp32'}, "
                                   f"train step, {'dp%d' % world if world > 1 else 'single GPU'}",
                       "comm": None if dist is None else ("RCCL via sg_comm_* (libsegengine C ABI)" if args.comm == "sg"
                                                          else "RCCL via torch.distributed nccl"),
                       "global_batch": world * args.batch, "model_flops_per_step_tflop": round(step_tflop, 3),
                       "step_achieved_tflops": round(step_tflop / (ms_per_step / 1e3), 2),
                       "final_loss": float(loss.item()),
                       "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 2),
                       "train_step_choice": choice,
                       "train_step": (("one hipGraph replay per step (compile(jit_compile=True))" if dist is None else
                                       "hipGraph segments per gradient bucket, eager RCCL all-reduces between them "
                                       "(compile(jit_compile=True) under DataParallel)") if jit else "eager launches")},
            "roofline": {"bound": "mfma", "achieved": None if achieved is None else round(achieved, 2),
                         "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": None if achieved is None else round(achieved / peak, 4),
                         "peak_note": peak_note,
                         "frac_vs_fp32_mfma_peak": None if achieved is None else round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                         "executed_share_of_nominal": None if exec_ratio is None else round(exec_ratio, 4),
                         "frac_executed": None if (achieved is None or exec_ratio is None) else round(achieved * exec_ratio / peak, 4),
                         "frac_executed_note": exec_note,
                         "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": (("conv_b16w_kernel / conv_b16_kernel / wgrad_x6_kernel<NPL=1,bf16>" if b16
                                     else "conv_x6w_kernel (ASPP forward + dgrad, activation planes once per tensor: x6w_split_kernel) / conv_x6_kernel (SK) / "
                                          "wgrad_x6_kernel<.., PIN> (ASPP filter gradients from the planes) / wgrad_x6_kernel (SK)") if x6
                                    else "igemm_conv_kernel / igemm_wgrad_kernel") + " on the 6 dilated 3x3 convs (fwd+dgrad+wgrad)",
                         "ms_per_step": round(dil_ms, 3), "launches_per_step": prof.get("dilated_conv_launches", 0) // max(args.steps, 1)},
        }
        fam_ms = fam.get("gemm_conv", 0.0) / 2
        if fam_ms > 0:  # every convolution / pointwise / transposed-convolution GEMM launch of the step, same peak
            out["roofline"]["family"] = {
                "kernel": "all GEMM convolution launches of the step (pw_wide / conv_x6 / conv_x6w / conv_x6p / wgrad_x6 / wgrad_pw_wide / "
                          "thin kernels + their plane-split and split-K reduce helpers), measured in 2 extra steps after the timed region",
                "achieved": round(step_tflop / (fam_ms / 1e3), 2), "frac": round(step_tflop / (fam_ms / 1e3) / peak, 4),
                "ms_per_step": round(fam_ms, 3), "launches_per_step": fam.get("gemm_conv_launches", 0) // 2}
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            try:
                out["cpu_baseline"] = cpu_baseline(threads, 2, args.size, 5, model if args.model == "v3plus" else None)
                if b16 and "parity" in out["cpu_baseline"]:
                    out["cpu_baseline"]["parity"]["what"] += " (bf16 engine vs the fp32 oracle: the tolerance contract of DESIGN.md section 8, not the 1e-3 fp32 bar)"
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        out["config"]["peak_device_memory_gib"] = round(peak_mem / 2.0 ** 30, 2)
        out["config"]["side_stream_filter_gradients"] = os.environ.get("SG_SIDE_WGRAD", "1") != "0"
        if world == 1 and not b16 and args.model == "v3plus" and not args.no_bf16_leg:
            # BASELINE configs[2]'s per-GPU workload (bf16 storage), AFTER and OUTSIDE the fp32 timed region: a child process
            # (started, not exec'ed: this process has used the GPU) runs the same bench with --dtype bf16 and its line is
            # folded into config.bf16_leg.  The fp32 model's memory is released first.
            del model
            torch.cuda.empty_cache()
            out["config"]["bf16_leg"] = bf16_leg(args)
        sys.stdout.flush()
        os.write(_JSON_FD, (json.dumps(out) + "\n").encode())   # the ONE line on the real stdout
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
