"""Host-side launchers: torch CUDA tensors in, libsegengine kernels on the current HIP stream, tensors out.

torch is used here only for device memory and the stream handle; all arithmetic happens in the hand-written
HIP kernels behind the C ABI (include/segengine.h).  Activations are NHWC float32 contiguous; kernels use the
tf.keras `get_weights()` layouts (HWIO, depthwise [kh,kw,C,1], Conv2DTranspose [kh,kw,Cout,Cin], Dense
[in,out]).  No function here has a CPU path: a CPU tensor raises.
"""
from __future__ import annotations

import collections
import contextlib
import ctypes as C
import os
import threading
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, SG_F32, SG_BF16, SG_HEAD_F32, check


def same_pad(in_size: int, k: int, stride: int, dilation: int = 1) -> Tuple[int, int, int]:
    """tf `padding='same'` -> (out, pad_before, pad_after); the smaller half goes before."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + (k - 1) * dilation + 1 - in_size, 0)
    before = total // 2
    return out, before, total - before


def conv_out_geometry(h: int, w: int, kh: int, kw: int, stride: int, dilation: int, padding: str):
    if padding == "same":
        ho, pt, _ = same_pad(h, kh, stride, dilation)
        wo, pl, _ = same_pad(w, kw, stride, dilation)
    elif padding == "valid":
        ho = (h - (kh - 1) * dilation - 1) // stride + 1
        wo = (w - (kw - 1) * dilation - 1) // stride + 1
        pt = pl = 0
    else:
        raise ValueError(f"padding={padding!r}")
    return ho, wo, pt, pl


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise _lib.SgError(f"{name}: expected a CUDA (HIP) tensor - this engine has no CPU path")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.SgError(f"{name}: expected float32 or bfloat16 storage, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.SgError(f"{name}: tensor must be contiguous")


def _chk32(t: torch.Tensor, name: str):
    _chk(t, name)
    if t.dtype != torch.float32:
        raise _lib.SgError(f"{name}: this tensor is fp32 in every storage mode (weights, statistics, head), got {t.dtype}")


def _dt(t: torch.Tensor) -> int:
    """ABI dtype of an activation tensor: SG_BF16 = bf16 storage with fp32 arithmetic inside the kernels."""
    return SG_BF16 if t.dtype == torch.bfloat16 else SG_F32


_TRACE_MARK = os.environ.get("SG_TRACE_MARK", "0") == "1"
_MARK_TAGS = {"dilated_conv": 0, "gemm_conv": 1}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)


class _NoTimer:
    """`with eng.timed(tag)` while no profile is open: nothing to do, nothing to allocate."""

    __slots__ = ()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_TIMER = _NoTimer()


class _Timed:
    """`with eng.timed(tag):` brackets the launches inside with two HIP events on the current stream while a
    profile is open (eng.profile_begin()); otherwise it is free."""

    __slots__ = ("eng", "tag", "a")

    def __init__(self, eng, tag):
        self.eng, self.tag, self.a = eng, tag, None

    def __enter__(self):
        if getattr(self.eng, "_prof", None) is not None and self.tag:
            self.eng.join_side()   # a bracketed section runs alone: no filter gradient of an earlier layer beside it
            if _TRACE_MARK and self.tag in _MARK_TAGS:  # named marker kernels for a rocprofv3 kernel trace
                self.eng.lib.sg_trace_mark(self.eng.h, self.eng.stream, _MARK_TAGS[self.tag], 0)
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.eng.device))
        return self

    def __exit__(self, *exc):
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record(torch.cuda.current_stream(self.eng.device))
            self.eng._prof.setdefault(self.tag, []).append((self.a, b))
            if _TRACE_MARK and self.tag in _MARK_TAGS:
                self.eng.lib.sg_trace_mark(self.eng.h, self.eng.stream, _MARK_TAGS[self.tag], 1)
        return False


class Engine:
    """Per-device launcher state: the `sg_ctx`, one reusable scratch buffer and the stream to launch on."""

    def __init__(self, device: int = 0):
        if not torch.cuda.is_available():
            raise _lib.SgError("no HIP device visible; building_detection_amd requires an MI355X (gfx950) GPU")
        self.lib = _lib.load()
        # every launch below goes to torch's CURRENT stream of `device`; make that device current so that torch's
        # allocations, the stream handle and the sg_ctx agree (one process per GPU: LOCAL_RANK picks the device)
        torch.cuda.set_device(device)
        self.ctx = _lib.Context(device)
        self.h = self.ctx.handle
        self.device = torch.device("cuda", device)
        self._dev_index = int(device)
        self._prof = None       # open profile: tag -> [(event, event)] (profile_begin / profile_end)
        self._prof_all = False
        self._ws = torch.empty(1 << 20, dtype=torch.uint8, device=self.device)
        self._ws_peak = 0          # largest workspace request seen (GraphedPredict sizes its private buffer by it)
        self._ws_pinned = False    # True while a private workspace is installed: growth is an error, not a realloc
        # One forward / training step at a time per device: the scratch buffer above and a model's value table are
        # shared mutable state, and the reference's Flask front end calls predict() from request threads
        # (buildAPI.py:78,111).  Re-entrant so that predict() may call predict_device().
        self.lock = threading.RLock()
        # Filter gradients beside the input-gradient chain: conv2d_wgrad / dwconv_wgrad launches go to a second
        # stream behind an event, so that these MFMA-bound kernels overlap the bandwidth-bound BatchNormalization / depthwise /
        # add kernels of the chain.  SG_SIDE_WGRAD=0 keeps everything on one stream.
        self._side_mode = int(os.environ.get("SG_SIDE_WGRAD", "1"))   # 1: all filter gradients, 2: GEMM ones only, 3: depthwise only
        self._side_on = self._side_mode > 0
        self._side_stream = None
        self.side_launches = 0   # blocks that went to the side stream (tests)
        self._side_dirty = False
        # Operands of the side stream's launches.  They live in the main stream's pool and must not be handed out again while
        # the side stream may still read them: groups of blocks are closed by an event on the side stream ((event, tensors),
        # oldest first) and dropped as soon as that event has completed (polled at the next side block) - not only at the
        # join at the end of the sweep, which would keep every layer's output gradient alive for the whole backward pass.
        self._side_keep = []         # tensors of the open group (no event recorded yet)
        self._side_groups = collections.deque()
        self._side_free_events = []
        self._side_group_blocks = 0
        self._side_kept = 0          # bytes of operands held (an operand listed twice counts twice: an upper bound)
        self._side_blocks_since_join = 0
        self._side_keep_bound = int(float(os.environ.get("SG_SIDE_KEEP_GIB", "4")) * 2 ** 30)
        self._in_side = False
        self.lane = None         # set while a training step is captured with a side lane (side_run defers into it)
        self._ws2 = torch.empty(1 << 20, dtype=torch.uint8, device=self.device)
        self._ws2_peak = 0
        self._ws2_pinned = False

    # ------------------------------------------------------------------------------------------ plumbing
    @property
    def stream(self):
        # torch's CURRENT stream of this device (a hipGraph capture swaps it), by the raw-handle call: one C call instead of
        # building a torch.cuda.Stream object per launch (~1800 launches per training step)
        return C.c_void_p(_raw_stream(self._dev_index))

    @contextlib.contextmanager
    def side(self, tag, *tensors, kind=2):
        """Launches inside the block go to the side stream, ordered behind everything queued on the current stream so far.
        `tensors` are the operands that live in the current stream's memory pool: the allocator must not hand their blocks out
        again before the side stream is done with them (they are held until join_side).  Inline (no second stream) while the launches are being
        bracketed for a profile, inside a hipGraph capture, or with SG_SIDE_WGRAD=0."""
        if (not self._side_on or self._in_side or (self._ws_pinned and not self._ws2_pinned)
                or (self._side_mode in (2, 3) and self._side_mode != kind)
                or (self._prof is not None and (tag or self._prof_all))):
            yield
            return
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        sd = self._side_stream
        sd.wait_stream(main)
        # the operands live in the main stream's pool: they are kept alive until the main stream has joined the side stream
        # (join_side), after which any reuse of their blocks is ordered behind the side stream's reads.  (record_stream would
        # do, but every recorded block costs the allocator an event it polls on later allocations.)
        for t in tensors:
            if t is not None:
                self._side_keep.append(t)
                self._side_kept += t.numel() * t.element_size()
        self._in_side = True
        self.side_launches += 1
        try:
            with torch.cuda.stream(sd):
                yield
        finally:
            self._in_side = False
            self._side_dirty = True
            self._side_group_blocks += 1
            if self._side_group_blocks >= self._SIDE_GROUP:
                self._close_side_group(sd)
            while self._side_groups and self._side_groups[0][0].query():   # the side stream is past this group's launches
                ev, ts = self._side_groups.popleft()
                self._side_kept -= sum(t.numel() * t.element_size() for t in ts)
                self._side_free_events.append(ev)
            # The host queues a step well ahead of the device, so completed events alone release late.  Past SG_SIDE_KEEP_GIB
            # (default 4) of held operands the main stream joins the side stream - a device-side wait, no host sync - and
            # everything held is dropped: the peak of a DeepLabv3+ 512 x 512 bs 16 step falls from 35.5 to ~24 GiB (25.2
            # without the second stream) for ~8 joins per backward pass.  Not more often than every 16 side blocks: a
            # join makes the chain wait for the filter gradients queued so far, and the U-Nets' full-resolution layers hold
            # 2 GiB per block (a join every other layer cost Res34-UNet 3 % of its step).
            self._side_blocks_since_join += 1
            if self._side_kept > self._side_keep_bound and self._side_blocks_since_join >= self._SIDE_JOIN_MIN_BLOCKS:
                self.join_side()

    _SIDE_GROUP = 4   # side blocks per release event (an event costs the host a few microseconds)
    _SIDE_JOIN_MIN_BLOCKS = 16

    def side_run(self, tag, tensors, fn, kind=2):
        """`fn()` - the launches of one filter gradient - beside the input-gradient chain.  Eager step: inside side() (second
        stream behind an event).  While a training step is being captured with lanes (runtime.GraphedTrainStep sets
        `self.lane`): NOT launched now but deferred - the capture collects the segment's filter-gradient calls and records
        them into a side graph of their own after the segment's main graph, which the replay launches on the second stream
        beside the NEXT segment.  `tensors` are the operands the call reads (kept alive for it either way)."""
        lane = self.lane
        if (lane is not None and self._side_on and not self._in_side
                and not (self._side_mode in (2, 3) and self._side_mode != kind)):
            lane.defer(fn, [t for t in tensors if t is not None])
            return
        with self.side(tag, *tensors, kind=kind):
            fn()

    def _close_side_group(self, sd):
        if self._side_keep:
            ev = self._side_free_events.pop() if self._side_free_events else torch.cuda.Event()
            ev.record(sd)
            self._side_groups.append((ev, self._side_keep))
            self._side_keep = []
        self._side_group_blocks = 0

    def side_kept_bytes(self) -> int:
        """Bytes of operands currently held for the side stream (tests)."""
        seen, n = set(), 0
        for t in [t for _, ts in self._side_groups for t in ts] + list(self._side_keep):
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                n += t.numel() * t.element_size()
        return n

    def join_side(self):
        """The current stream waits for everything queued on the side stream (before the gradients are read)."""
        if self._side_dirty:
            torch.cuda.current_stream(self.device).wait_stream(self._side_stream)
            self._side_dirty = False
            self._side_keep = []
            while self._side_groups:
                self._side_free_events.append(self._side_groups.popleft()[0])
            self._side_group_blocks = 0
            self._side_kept = 0
            self._side_blocks_since_join = 0

    def ws(self, nbytes: int):
        nbytes = int(nbytes)
        if self._in_side:   # the side stream's launches have a scratch buffer of their own
            if nbytes > self._ws2_peak:
                self._ws2_peak = nbytes
            if nbytes > self._ws2.numel():
                if self._ws2_pinned:
                    raise _lib.SgError(f"side-stream workspace request of {nbytes} B exceeds the private {self._ws2.numel()} B "
                                       "buffer of the hipGraph being captured (the sizing pass saw a smaller request)")
                torch.cuda.synchronize(self.device)   # (first steps only) nothing may still be using the old buffer
                self._ws2 = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
            return C.c_void_p(self._ws2.data_ptr()), C.c_size_t(self._ws2.numel())
        if nbytes > self._ws_peak:
            self._ws_peak = nbytes
        if nbytes > self._ws.numel():
            if self._ws_pinned:
                raise _lib.SgError(f"workspace request of {nbytes} B exceeds the private {self._ws.numel()} B buffer of "
                                   "the hipGraph being captured (the sizing pass saw a smaller request)")
            self._ws = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return C.c_void_p(self._ws.data_ptr()), C.c_size_t(self._ws.numel())

    @contextlib.contextmanager
    def private_ws(self, buf: torch.Tensor, side_buf: Optional[torch.Tensor] = None):
        """Launches inside the block use `buf` as their scratch.  A hipGraph bakes the scratch pointer into its kernel
        nodes, so every captured graph owns its buffer: the engine's shared one may be re-grown (= freed) by any
        later eager call, and a replay would then write into whatever tensor owns that block by then.  `side_buf`: the
        same for the side stream's launches (without it the filter gradients stay on the capturing stream)."""
        with self.lock:
            old, old_pin, old2, old_pin2 = self._ws, self._ws_pinned, self._ws2, self._ws2_pinned
            self._ws, self._ws_pinned = buf, True
            if side_buf is not None:
                self._ws2, self._ws2_pinned = side_buf, True
            try:
                yield
            finally:
                self._ws, self._ws_pinned = old, old_pin
                self._ws2, self._ws2_pinned = old2, old_pin2

    # -- in-run timing of tagged launches with HIP events recorded on the launch stream (bench.py roofline) --
    def profile_begin(self, all_convs=False):
        """all_convs: every GEMM-convolution launch (conv2d_fwd / dgrad / wgrad: Conv2D, SeparableConv pointwise,
        Conv2DTranspose) is bracketed under the tag "gemm_conv" as well (bench.py's whole-family figure)."""
        self._prof = {}
        self._prof_all = bool(all_convs)

    def _gemm_tag(self):
        return "gemm_conv" if self._prof_all and self._prof is not None else None

    def profile_end(self):
        prof, self._prof = getattr(self, "_prof", None) or {}, None
        self._prof_all = False
        torch.cuda.synchronize(self.device)
        out = {}
        for tag, evs in prof.items():
            out[tag] = sum(a.elapsed_time(b) for a, b in evs)  # milliseconds
            out[tag + "_launches"] = len(evs)
        return out

    def timed(self, tag):
        if self._prof is None or not tag:
            return _NO_TIMER
        return _Timed(self, tag)

    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def zeros(self, *shape, dtype=torch.float32):
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def cast(self, x, dtype, out=None):
        """fp32 <-> bf16 conversion of a tensor (sg_cast: round to nearest even)."""
        _chk(x, "x")
        if x.dtype == dtype and out is None:
            return x
        y = out if out is not None else torch.empty(x.shape, dtype=dtype, device=self.device)
        check(self.lib.sg_cast(self.h, self.stream, _dt(x), _dt(y), x.numel(), _ptr(x), _ptr(y)), "sg_cast")
        return y

    # ---------------------------------------------------------------------------------------------- conv
    def set_conv_x6(self, on: bool) -> bool:
        """Process-wide: run qualifying convolutions as six bf16 MFMA passes (True, default) or on the native fp32 MFMA."""
        return bool(self.lib.sg_set_conv_x6(int(bool(on))))

    @staticmethod
    def conv_desc(xshape, cout, kh, kw, stride=1, dilation=1, padding="same", x_ld=0, y_ld=0) -> ConvDesc:
        n, h, w, cin = xshape
        ho, wo, pt, pl = conv_out_geometry(h, w, kh, kw, stride, dilation, padding)
        return ConvDesc(n, h, w, cin, cout, kh, kw, stride, dilation, pt, pl, ho, wo, x_ld, y_ld)

    def conv2d_bn_in_ok(self, d: ConvDesc, dtype=torch.float32) -> bool:
        """Can the launches of convolution `d` apply a BatchNormalization(+ReLU) to their input themselves (sg_conv2d_fwd_stats_bn /
        sg_conv2d_wgrad_bn: the thin 1x1 and the patch kernels, fp32 storage)?"""
        return dtype == torch.float32 and bool(self.lib.sg_conv2d_bn_in_supported(self.h, SG_F32, C.byref(d)))

    @staticmethod
    def _bn_in(bn):
        """bn = (gamma, beta, mean, invstd-or-moving-variance, relu, infer, eps) -> the sg_bn_in struct (tensors kept by the caller)"""
        gamma, beta, mean, inv, relu, infer, eps = bn
        for t in (gamma, beta, mean, inv):
            _chk32(t, "bn_in")
        return _lib.BnIn(mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), int(bool(relu)), int(bool(infer)), float(eps))

    def conv2d_up2_ok(self, d: ConvDesc, dtype=torch.float32) -> bool:
        """Does the convolution `d` (on the up-sampled grid) take the fused UpSampling2D(2) -> Conv2D 3x3 kernels
        (SG_PRO_UP2 / SG_EPI_DOWN2 / SG_X_UP2, csrc/conv_x6p.h)?"""
        return dtype == torch.float32 and bool(self.lib.sg_conv2d_up2_supported(SG_F32, C.byref(d)))

    def conv2d_fwd(self, x, w, b=None, stride=1, dilation=1, padding="same", relu=False, out=None, desc=None,
                   want_stats=False, head_f32=False, planes=None, up2=False, x_planes=None, bn_in=None):
        """want_stats: also return the BatchNormalization statistics of y as (stats tensor [tiles,2,Cout], tiles), or
        None when this launch could not produce them (then BN computes its own).
        head_f32 (bf16 storage only): the output is fp32 - the softmax head, a thin 1x1 convolution (SG_HEAD_F32).
        up2: x is the SOURCE [N, H/2, W/2, Cin] of a nearest 2x up-sampling and `desc` (required) names the convolution on
        the up-sampled grid (SG_PRO_UP2: the sub-pixel kernel; the up-sampled tensor is never built).
        x_planes: split_planes(x), when the caller has them (sg_conv2d_fwd_stats_ap: the planes-in kernel then skips its own
        split; any other kernel ignores them).
        bn_in: (gamma, beta, mean, invstd | moving variance, relu, infer, eps) - x is the RAW input of that BatchNormalization(+ReLU)
        and the kernel applies it while loading (sg_conv2d_fwd_stats_bn; conv2d_bn_in_ok tells which launches can)."""
        _chk(x, "x"); _chk32(w, "w")
        kh, kw, cin, cout = w.shape
        d = desc or self.conv_desc(x.shape, cout, kh, kw, stride, dilation, padding)
        assert d.Cin == cin, (d.Cin, cin)
        if up2:
            assert desc is not None and tuple(x.shape) == (d.N, d.H // 2, d.W // 2, d.Cin), (tuple(x.shape), d.H, d.W)
            planes = None   # the summed-tap planes of the sub-pixel form are made per launch in the plain workspace
        head_f32 = bool(head_f32) and x.dtype == torch.bfloat16
        y = out if out is not None else self.empty(d.N, d.Ho, d.Wo, cout, dtype=torch.float32 if head_f32 else x.dtype)
        dt = _dt(x) | (SG_HEAD_F32 if head_f32 else 0)
        flags = (_lib.SG_EPI_BIAS if b is not None else 0) | (_lib.SG_EPI_RELU if relu else 0) | (_lib.SG_PRO_UP2 if up2 else 0)
        if planes is not None:  # this layer's weight planes, prepared once per step (runtime._Runtime.ensure_planes)
            wsp, wsn = C.c_void_p(planes), C.c_size_t(_lib.SG_WS_PREPARED)
        else:
            wsp, wsn = self.ws(self.lib.sg_conv2d_fwd_ws_bytes(C.byref(d)))
        if bn_in is not None:
            assert not up2 and x_planes is None
            bq = self._bn_in(bn_in)
            st = self.empty(self.lib.sg_conv2d_fwd_stats_bytes(C.byref(d)) // 4) if want_stats else None
            tiles = C.c_int(0)
            with self.timed(self._gemm_tag()):
                check(self.lib.sg_conv2d_fwd_stats_bn(self.h, self.stream, dt, C.byref(d), _ptr(x), _ptr(w), _ptr(b), _ptr(y), flags,
                                                      wsp, wsn, _ptr(st), C.byref(tiles) if want_stats else None, C.byref(bq)),
                      "sg_conv2d_fwd_stats_bn")
            if want_stats:
                return y, ((st, tiles.value) if tiles.value > 0 else None)
            return y
        if want_stats:
            st = self.empty(self.lib.sg_conv2d_fwd_stats_bytes(C.byref(d)) // 4)
            tiles = C.c_int(0)
            with self.timed(self._gemm_tag()):
                check(self.lib.sg_conv2d_fwd_stats_ap(self.h, self.stream, dt, C.byref(d), _ptr(x), _ptr(w), _ptr(b), _ptr(y),
                                                      flags, wsp, wsn, _ptr(st), C.byref(tiles), _ptr(x_planes)), "sg_conv2d_fwd_stats")
            return y, ((st, tiles.value) if tiles.value > 0 else None)
        with self.timed(self._gemm_tag()):
            if x_planes is not None:
                check(self.lib.sg_conv2d_fwd_stats_ap(self.h, self.stream, dt, C.byref(d), _ptr(x), _ptr(w), _ptr(b), _ptr(y), flags,
                                                      wsp, wsn, None, None, _ptr(x_planes)), "sg_conv2d_fwd_stats_ap")
            else:
                check(self.lib.sg_conv2d_fwd_ws(self.h, self.stream, dt, C.byref(d), _ptr(x), _ptr(w), _ptr(b), _ptr(y), flags,
                                                wsp, wsn), "sg_conv2d_fwd_ws")
        return y

    def conv2d_planes_in(self, d: ConvDesc, dgrad: bool) -> bool:
        """Does the fp32 forward (dgrad) launch of `d` read its activation as bf16 planes (csrc/conv_x6w.h)?  Then planes the
        caller already has (split_planes) save the launch its own split."""
        return bool(self.lib.sg_conv2d_planes_in(C.byref(d), 1 if dgrad else 0))

    def conv2d_dgrad(self, dy, w, d: ConvDesc, bias=None, relu=False, out=None, out_dtype=None, planes=None, res=None,
                     down2=False, dy_planes=None):
        """dx of the forward conv described by `d`; also Conv2DTranspose forward (then bias/relu apply).
        out_dtype = torch.bfloat16 with an fp32 dy: the backward of the fp32 softmax head of a bf16 model (SG_HEAD_F32).
        res: a gradient already collected for the same tensor, added in the kernel's epilogue (sg_conv2d_dgrad_acc: only for
        launches whose prepared planes are of kind 1, the slab kernels).
        down2: the conv's input was a nearest 2x up-sampling; dx is the gradient of its SOURCE, [N, H/2, W/2, Cin] (the 2 x 2
        cells added in the epilogue in up-sampling's backward order: SG_EPI_DOWN2)."""
        _chk(dy, "dy"); _chk32(w, "w")
        odt = out.dtype if out is not None else (out_dtype or dy.dtype)
        if down2:
            assert res is None and bias is None and not relu
            dx = out if out is not None else self.empty(d.N, d.H // 2, d.W // 2, d.Cin, dtype=odt)
        else:
            dx = out if out is not None else self.empty(d.N, d.H, d.W, d.Cin, dtype=odt)
        dt = (SG_BF16 | SG_HEAD_F32) if (odt == torch.bfloat16 and dy.dtype == torch.float32) else _dt(dy)
        if planes is not None:
            wsp, wsn = C.c_void_p(planes), C.c_size_t(_lib.SG_WS_PREPARED)
        else:
            wsp, wsn = self.ws(self.lib.sg_conv2d_dgrad_ws_bytes(C.byref(d)))
        flags = (_lib.SG_EPI_BIAS if bias is not None else 0) | (_lib.SG_EPI_RELU if relu else 0) | (_lib.SG_EPI_DOWN2 if down2 else 0)
        with self.timed(self._gemm_tag()):
            if res is not None:
                check(self.lib.sg_conv2d_dgrad_acc(self.h, self.stream, dt, C.byref(d), _ptr(dy), _ptr(w), _ptr(bias), _ptr(dx),
                                                   flags, wsp, wsn, _ptr(res)), "sg_conv2d_dgrad_acc")
            elif dy_planes is not None:
                check(self.lib.sg_conv2d_dgrad_ap(self.h, self.stream, dt, C.byref(d), _ptr(dy), _ptr(w), _ptr(bias), _ptr(dx),
                                                  flags, wsp, wsn, _ptr(dy_planes)), "sg_conv2d_dgrad_ap")
            else:
                check(self.lib.sg_conv2d_dgrad(self.h, self.stream, dt, C.byref(d), _ptr(dy), _ptr(w), _ptr(bias), _ptr(dx),
                                               flags, wsp, wsn), "sg_conv2d_dgrad")
        return dx

    def conv2d_dgrad_bnb_ok(self, d: ConvDesc, dtype=torch.float32) -> bool:
        """Does the input gradient of pointwise convolution `d` take the wide kernel's BatchNormalization-backward form
        (sg_conv2d_dgrad_bnb)?  Depends on the batch (the wide kernel wants >= 6144 rows)."""
        return dtype == torch.float32 and bool(self.lib.sg_conv2d_dgrad_bnb_supported(self.h, SG_F32, C.byref(d)))

    def conv2d_dgrad_bnb(self, dy_bn, x_bn, w, d: ConvDesc, gamma, beta, mean, invstd, dgamma, dbeta, relu, planes=None):
        """dx of pointwise convolution `d` AND the BatchNormalization's applied gradient dz, from the gradient dy_bn of that
        layer's output and its raw input x_bn (= the convolution's forward output): sg_conv2d_dgrad_bnb.  -> (dx, dz)"""
        _chk32(dy_bn, "dy"); _chk32(x_bn, "x"); _chk32(w, "w")
        assert dy_bn.shape == x_bn.shape and dy_bn.is_contiguous() and x_bn.is_contiguous()
        dx = self.empty(d.N, d.H, d.W, d.Cin)
        dz = torch.empty_like(dy_bn)
        rows = dy_bn.numel() // dy_bn.shape[-1]
        q = _lib.BnBwdIn(x_bn.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr() if (relu and beta is not None) else None,
                         dgamma.data_ptr(), dbeta.data_ptr(), dz.data_ptr(), int(bool(relu)), rows)
        if planes is not None:
            wsp, wsn = C.c_void_p(planes), C.c_size_t(_lib.SG_WS_PREPARED)
        else:
            wsp, wsn = self.ws(self.lib.sg_conv2d_dgrad_ws_bytes(C.byref(d)))
        with self.timed(self._gemm_tag()):
            check(self.lib.sg_conv2d_dgrad_bnb(self.h, self.stream, SG_F32, C.byref(d), _ptr(dy_bn), _ptr(w), _ptr(dx), wsp, wsn, C.byref(q)),
                  "sg_conv2d_dgrad_bnb")
        return dx, dz

    def conv2d_wgrad(self, x, dy, d: ConvDesc, want_bias=True, dw=None, db=None, x_up2=False, bn_in=None):
        """x_up2: x is the SOURCE [N, H/2, W/2, Cin] of the nearest 2x up-sampling the conv `d` read (SG_X_UP2: the patch
        kernel gathers x[n, h >> 1, w >> 1]; the bits of the filter gradient on the materialised tensor)."""
        _chk(x, "x"); _chk(dy, "dy")
        dt = (SG_BF16 | SG_HEAD_F32) if (x.dtype == torch.bfloat16 and dy.dtype == torch.float32) else _dt(x)
        if x_up2:
            assert tuple(x.shape) == (d.N, d.H // 2, d.W // 2, d.Cin), (tuple(x.shape), d.H, d.W)
            dt |= _lib.SG_X_UP2
        if dw is None:
            dw = self.empty(d.KH, d.KW, d.Cin, d.Cout)
        if want_bias and db is None:
            db = self.empty(d.Cout)
        need = self.lib.sg_conv2d_wgrad_ws_bytes(self.h, C.byref(d))
        wsp, wsn = self.ws(need)
        with self.timed(self._gemm_tag()):
            if bn_in is not None:   # x is the raw input of the BatchNormalization(+ReLU) in front of the layer (see conv2d_fwd)
                bq = self._bn_in(bn_in)
                check(self.lib.sg_conv2d_wgrad_bn(self.h, self.stream, dt, C.byref(d), _ptr(x), _ptr(dy), _ptr(dw),
                                                  _ptr(db) if want_bias else None, wsp, wsn, C.byref(bq)), "sg_conv2d_wgrad_bn")
            else:
                check(self.lib.sg_conv2d_wgrad(self.h, self.stream, dt, C.byref(d), _ptr(x), _ptr(dy), _ptr(dw),
                                               _ptr(db) if want_bias else None, wsp, wsn), "sg_conv2d_wgrad")
        return dw, (db if want_bias else None)

    def split_planes(self, x, out=None):
        """x [..., C] fp32 -> int16 tensor [3, rows, C]: the three bf16 planes of the exact split a1 + a2 + a3 = x (sg_split_planes)."""
        _chk32(x, "x")
        c = x.shape[-1]
        rows = x.numel() // c
        pl = out if out is not None else torch.empty((3, rows, c), dtype=torch.int16, device=x.device)
        check(self.lib.sg_split_planes(self.h, self.stream, _ptr(x), rows, c, c, _ptr(pl)), "sg_split_planes")
        return pl

    def conv2d_wgrad_planes_ok(self, d: ConvDesc) -> bool:
        return bool(self.lib.sg_conv2d_wgrad_planes_supported(self.h, C.byref(d)))

    def conv2d_wgrad_planes(self, x_planes, dy_planes, d: ConvDesc, dw=None):
        """The fp32 filter gradient from operands that are already split into bf16 planes (sg_conv2d_wgrad_planes): the bits of
        conv2d_wgrad on the fp32 tensors, without the per-launch VALU split."""
        if dw is None:
            dw = self.empty(d.KH, d.KW, d.Cin, d.Cout)
        wsp, wsn = self.ws(self.lib.sg_conv2d_wgrad_planes_ws_bytes(self.h, C.byref(d)))
        with self.timed(self._gemm_tag()):
            check(self.lib.sg_conv2d_wgrad_planes(self.h, self.stream, C.byref(d), _ptr(x_planes), _ptr(dy_planes), _ptr(dw), wsp, wsn),
                  "sg_conv2d_wgrad_planes")
        return dw

    def bias_grad(self, dy, db):
        """db[C] = column sums of dy[..., C] (bias gradient of Conv2DTranspose / stand-alone use)."""
        c = dy.shape[-1]
        rows = dy.numel() // c
        wsp, wsn = self.ws(self.lib.sg_bias_grad_ws_bytes(self.h, rows, c))
        check(self.lib.sg_bias_grad(self.h, self.stream, _dt(dy), rows, c, c, _ptr(dy), _ptr(db), wsp, wsn), "sg_bias_grad")
        return db

    # --------------------------------------------------------------------------------------- depthwise
    def dwconv_fwd(self, x, w, stride=1, pre_relu=False, out=None, desc=None, bn=None):
        """bn = (gamma, beta, mean, invstd, relu): x is the RAW input of a training-mode BatchNormalization(+ReLU) whose
        output this depthwise convolution consumes; the normalisation is applied in the gather (sg_dwconv2d_fwd_bn)."""
        _chk(x, "x"); _chk(w, "w")
        kh, kw, c = w.shape[0], w.shape[1], w.shape[2]
        d = desc or self.conv_desc(x.shape, c, kh, kw, stride, 1, "same")
        y = out if out is not None else self.empty(d.N, d.Ho, d.Wo, c, dtype=x.dtype)
        if bn is not None:
            gamma, beta, mean, invstd, relu = bn
            check(self.lib.sg_dwconv2d_fwd_bn(self.h, self.stream, _dt(x), C.byref(d), _ptr(x), _ptr(w), _ptr(y), _ptr(gamma),
                                              _ptr(beta), _ptr(mean), _ptr(invstd), int(relu)), "sg_dwconv2d_fwd_bn")
            return y
        check(self.lib.sg_dwconv2d_fwd(self.h, self.stream, _dt(x), C.byref(d), _ptr(x), _ptr(w), _ptr(y), int(pre_relu)),
              "sg_dwconv2d_fwd")
        return y

    def dwconv_dgrad(self, dy, w, d: ConvDesc, x=None, pre_relu=False, out=None, res=None):
        """res: a gradient already collected for the same input tensor, added to the result inside the kernel
        (sg_dwconv2d_dgrad_acc: stride-1 3x3, W % 4 == 0, C % 4 == 0 only - dwconv_dgrad_acc_ok)."""
        dx = out if out is not None else self.empty(d.N, d.H, d.W, d.Cin, dtype=dy.dtype)
        if res is not None:
            check(self.lib.sg_dwconv2d_dgrad_acc(self.h, self.stream, _dt(dy), C.byref(d), _ptr(dy), _ptr(w), _ptr(x), _ptr(dx),
                                                 int(pre_relu), _ptr(res)), "sg_dwconv2d_dgrad_acc")
            return dx
        check(self.lib.sg_dwconv2d_dgrad(self.h, self.stream, _dt(dy), C.byref(d), _ptr(dy), _ptr(w), _ptr(x), _ptr(dx),
                                         int(pre_relu)), "sg_dwconv2d_dgrad")
        return dx

    def dwconv_dgrad_bnsums(self, dy, w, d: ConvDesc, bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta, bn_relu, dgamma, dbeta,
                            x=None, pre_relu=False, res=None, out=None):
        """dwconv_dgrad whose result is the output gradient of a training-mode BatchNormalization (raw input bn_x): also writes
        that layer's dgamma / dbeta (sg_dwconv2d_dgrad_bnsums); follow with bn_train_bwd_apply.  Geometry: dwconv_dgrad_acc_ok."""
        dx = out if out is not None else self.empty(d.N, d.H, d.W, d.Cin, dtype=dy.dtype)
        wsp, wsn = self.ws(self.lib.sg_dwconv2d_dgrad_bnsums_ws_bytes(self.h, C.byref(d)))
        check(self.lib.sg_dwconv2d_dgrad_bnsums(self.h, self.stream, _dt(dy), C.byref(d), _ptr(dy), _ptr(w), _ptr(x), _ptr(dx),
                                                int(pre_relu), _ptr(res), _ptr(bn_x), _ptr(bn_mean), _ptr(bn_invstd), _ptr(bn_gamma),
                                                _ptr(bn_beta), int(bn_relu), _ptr(dgamma), _ptr(dbeta), wsp, wsn),
              "sg_dwconv2d_dgrad_bnsums")
        return dx

    @staticmethod
    def dwconv_dgrad_acc_ok(d: ConvDesc) -> bool:
        return (d.KH == 3 and d.KW == 3 and d.stride == 1 and d.dilation == 1 and d.pad_t == 1 and d.pad_l == 1 and d.Ho == d.H
                and d.Wo == d.W and d.W % 4 == 0 and d.Cin % 4 == 0 and d.x_ld == 0 and d.y_ld == 0)

    def dwconv_wgrad(self, x, dy, d: ConvDesc, pre_relu=False, dw=None, bn=None):
        if dw is None:
            dw = self.empty(d.KH, d.KW, d.Cin, 1)
        need = self.lib.sg_dwconv2d_wgrad_ws_bytes(self.h, C.byref(d))
        wsp, wsn = self.ws(need)
        if bn is not None:  # see dwconv_fwd
            gamma, beta, mean, invstd, relu = bn
            check(self.lib.sg_dwconv2d_wgrad_bn(self.h, self.stream, _dt(x), C.byref(d), _ptr(x), _ptr(dy), _ptr(dw), _ptr(gamma),
                                                _ptr(beta), _ptr(mean), _ptr(invstd), int(relu), wsp, wsn), "sg_dwconv2d_wgrad_bn")
            return dw
        check(self.lib.sg_dwconv2d_wgrad(self.h, self.stream, _dt(x), C.byref(d), _ptr(x), _ptr(dy), _ptr(dw),
                                         int(pre_relu), wsp, wsn), "sg_dwconv2d_wgrad")
        return dw

    # ---------------------------------------------------------------------------------------------- BN
    def bn_train_fwd(self, x, gamma, beta, mm, mv, relu=False, momentum=0.99, eps=1e-3, out=None):
        _chk(x, "x")
        c = x.shape[-1]
        rows = x.numel() // c
        y = out if out is not None else torch.empty_like(x)
        mean, invstd = self.empty(c), self.empty(c)
        wsp, wsn = self.ws(self.lib.sg_bn_ws_bytes(self.h, rows, c))
        check(self.lib.sg_bn_train_fwd(self.h, self.stream, _dt(x), rows, c, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mm),
                                       _ptr(mv), _ptr(y), _ptr(mean), _ptr(invstd), momentum, eps, int(relu),
                                       int(x.dim() == 4), wsp, wsn), "sg_bn_train_fwd")
        return y, mean, invstd

    def bn_train_fwd_from_tiles(self, x, stats, tiles, gamma, beta, mm, mv, relu=False, momentum=0.99, eps=1e-3, out=None,
                                apply=True):
        """Training forward with the statistics the producing conv left in `stats` (conv2d_fwd(want_stats=True)).
        apply=False: statistics (and the moving averages) only - the consumer applies the normalisation itself
        (dwconv_fwd(bn=...)); returns (None, mean, invstd)."""
        c = x.shape[-1]
        rows = x.numel() // c
        y = (out if out is not None else torch.empty_like(x)) if apply else None
        mean, invstd = self.empty(c), self.empty(c)
        wsp, wsn = self.ws(self.lib.sg_bn_tiles_ws_bytes(self.h, int(tiles), c))
        check(self.lib.sg_bn_train_fwd_tiles(self.h, self.stream, _dt(x), rows, c, _ptr(stats), int(tiles), _ptr(mm), _ptr(mv),
                                             _ptr(mean), _ptr(invstd), momentum, eps, int(x.dim() == 4), wsp, wsn),
              "sg_bn_train_fwd_tiles")
        if apply:
            check(self.lib.sg_bn_apply(self.h, self.stream, _dt(x), rows, c, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean),
                                       _ptr(invstd), _ptr(y), int(relu)), "sg_bn_apply")
        return y, mean, invstd

    def bn_train_bwd_apply(self, x, dy, gamma, beta, mean, invstd, dgamma, dbeta, relu=False, out=None):
        """dx of a training-mode BatchNormalization whose column sums dgamma / dbeta are already there (dwconv_dgrad_bnsums)."""
        c = x.shape[-1]
        dx = out if out is not None else torch.empty_like(x)
        check(self.lib.sg_bn_train_bwd_apply(self.h, self.stream, _dt(x), x.numel() // c, c, _ptr(x), _ptr(dy), _ptr(gamma),
                                             _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(dgamma), _ptr(dbeta), _ptr(dx), int(relu)),
              "sg_bn_train_bwd_apply")
        return dx

    def bn_train_bwd(self, x, y, dy, gamma, mean, invstd, relu=False, out=None, dgamma=None, dbeta=None, beta=None):
        c = x.shape[-1]
        rows = x.numel() // c
        dx = out if out is not None else torch.empty_like(x)
        dgamma = dgamma if dgamma is not None else self.empty(c)
        dbeta = dbeta if dbeta is not None else self.empty(c)
        wsp, wsn = self.ws(self.lib.sg_bn_ws_bytes(self.h, rows, c))
        check(self.lib.sg_bn_train_bwd(self.h, self.stream, _dt(x), rows, c, _ptr(x), _ptr(y), _ptr(dy), _ptr(gamma),
                                       _ptr(beta), _ptr(mean), _ptr(invstd), _ptr(dx), _ptr(dgamma), _ptr(dbeta), int(relu),
                                       wsp, wsn),
              "sg_bn_train_bwd")
        return dx, dgamma, dbeta

    def bn_infer(self, x, gamma, beta, mm, mv, relu=False, eps=1e-3, out=None):
        _chk(x, "x")
        c = x.shape[-1]
        y = out if out is not None else torch.empty_like(x)
        check(self.lib.sg_bn_infer(self.h, self.stream, _dt(x), x.numel() // c, c, _ptr(x), _ptr(gamma), _ptr(beta),
                                   _ptr(mm), _ptr(mv), _ptr(y), eps, int(relu)), "sg_bn_infer")
        return y

    # ------------------------------------------------------------------------------------- element-wise
    def act_fwd(self, x, act, out=None):
        _chk(x, "x")
        y = out if out is not None else torch.empty_like(x)
        check(self.lib.sg_act_fwd(self.h, self.stream, _dt(x), act, x.numel(), _ptr(x), _ptr(y)), "sg_act_fwd")
        return y

    def act_bwd(self, y, dy, act, out=None, accumulate=False):
        dx = out if out is not None else torch.empty_like(y)
        check(self.lib.sg_act_bwd(self.h, self.stream, _dt(y), act, y.numel(), _ptr(y), _ptr(dy), _ptr(dx), int(accumulate)),
              "sg_act_bwd")
        return dx

    def add_n(self, xs: Sequence[torch.Tensor], relu=False, out=None):
        assert 1 <= len(xs) <= 8
        for t in xs:
            _chk(t, "add_n operand")
        arr = (C.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
        y = out if out is not None else torch.empty_like(xs[0])
        check(self.lib.sg_add_n(self.h, self.stream, _dt(xs[0]), len(xs), arr, xs[0].numel(), _ptr(y), int(relu)), "sg_add_n")
        return y

    def add2_bn(self, a, b, bn_a=None, bn_b=None, relu=False, infer=False, eps=1e-3, out=None, relu_a=False, relu_b=False):
        """relu?(f_a(a) + f_b(b)): bn_x = (mean, invstd | moving variance, gamma, beta) applies that BatchNormalization to the
        operand on the way (sg_add2_bn), None leaves it as it is; relu_x: followed by that layer's fused ReLU."""
        _chk(a, "a"); _chk(b, "b")
        c = a.shape[-1]
        y = out if out is not None else torch.empty_like(a)
        pa = [_ptr(t) for t in bn_a] if bn_a is not None else [None] * 4
        pb = [_ptr(t) for t in bn_b] if bn_b is not None else [None] * 4
        check(self.lib.sg_add2_bn(self.h, self.stream, _dt(a), a.numel() // c, c, _ptr(a), _ptr(b), *pa, *pb, _ptr(y), int(relu),
                                  int(infer), float(eps), int(relu_a), int(relu_b)), "sg_add2_bn")
        return y

    def copy_channels(self, src, src_off, dst, dst_off, c, accumulate=False):
        rows = src.numel() // src.shape[-1]
        assert rows == dst.numel() // dst.shape[-1]
        assert src.dtype == dst.dtype, (src.dtype, dst.dtype)
        check(self.lib.sg_copy_channels(self.h, self.stream, _dt(src), rows, c, _ptr(src), src.shape[-1], src_off, _ptr(dst),
                                        dst.shape[-1], dst_off, int(accumulate)), "sg_copy_channels")
        return dst

    def concat(self, xs: Sequence[torch.Tensor], out=None):
        ctot = sum(t.shape[-1] for t in xs)
        y = out if out is not None else self.empty(*xs[0].shape[:-1], ctot, dtype=xs[0].dtype)
        off = 0
        for t in xs:
            self.copy_channels(t, 0, y, off, t.shape[-1])
            off += t.shape[-1]
        return y

    def softmax2_fwd(self, z, out=None):
        _chk32(z, "z")
        assert z.shape[-1] == 2
        p = out if out is not None else torch.empty_like(z)
        check(self.lib.sg_softmax2_fwd(self.h, self.stream, SG_F32, z.numel() // 2, _ptr(z), _ptr(p)), "sg_softmax2_fwd")
        return p

    def softmax2_bwd(self, p, dp, out=None):
        dz = out if out is not None else torch.empty_like(p)
        check(self.lib.sg_softmax2_bwd(self.h, self.stream, SG_F32, p.numel() // 2, _ptr(p), _ptr(dp), _ptr(dz)),
              "sg_softmax2_bwd")
        return dz

    def softmax_branch_fwd(self, z):
        n, b, c = z.shape
        p = torch.empty_like(z)
        check(self.lib.sg_softmax_branch_fwd(self.h, self.stream, _dt(z), n, b, c, _ptr(z), _ptr(p)), "sg_softmax_branch_fwd")
        return p

    def softmax_branch_bwd(self, p, dp):
        n, b, c = p.shape
        dz = torch.empty_like(p)
        check(self.lib.sg_softmax_branch_bwd(self.h, self.stream, _dt(p), n, b, c, _ptr(p), _ptr(dp), _ptr(dz)),
              "sg_softmax_branch_bwd")
        return dz

    # -------------------------------------------------------------------------------------------- gates
    def bcast_mul_fwd(self, x, g, mode, out=None, accumulate=False):
        n, h, w, c = x.shape
        y = out if out is not None else torch.empty_like(x)
        check(self.lib.sg_bcast_mul_fwd(self.h, self.stream, _dt(x), n, h * w, c, mode, _ptr(x), _ptr(g), _ptr(y),
                                        int(accumulate)), "sg_bcast_mul_fwd")
        return y

    def bcast_mul_bwd(self, x, g, dy, mode, dx=None, accumulate_dx=False):
        n, h, w, c = x.shape
        if dx is None:
            dx = torch.empty_like(x)
            accumulate_dx = False
        dg = torch.empty_like(g)
        wsp, wsn = self.ws(self.lib.sg_bcast_mul_bwd_ws_bytes(self.h, n, h * w, c, mode))
        check(self.lib.sg_bcast_mul_bwd(self.h, self.stream, _dt(x), n, h * w, c, mode, _ptr(x), _ptr(g), _ptr(dy), _ptr(dx),
                                        _ptr(dg), int(accumulate_dx), wsp, wsn), "sg_bcast_mul_bwd")
        return dx, dg

    def scse_fwd(self, x, s, cl, out=None):
        n, h, w, c = x.shape
        y = out if out is not None else torch.empty_like(x)
        check(self.lib.sg_scse_fwd(self.h, self.stream, _dt(x), n, h * w, c, _ptr(x), _ptr(s), _ptr(cl), _ptr(y)), "sg_scse_fwd")
        return y

    def scse_bwd(self, x, s, cl, dy):
        n, h, w, c = x.shape
        dx, ds, dc = torch.empty_like(x), torch.empty_like(s), torch.empty_like(cl)
        wsp, wsn = self.ws(self.lib.sg_scse_bwd_ws_bytes(self.h, n, h * w, c))
        check(self.lib.sg_scse_bwd(self.h, self.stream, _dt(x), n, h * w, c, _ptr(x), _ptr(s), _ptr(cl), _ptr(dy), _ptr(dx),
                                   _ptr(ds), _ptr(dc), wsp, wsn), "sg_scse_bwd")
        return dx, ds, dc

    def bam_fwd(self, x, mc, ms, out=None):
        n, h, w, c = x.shape
        y = out if out is not None else torch.empty_like(x)
        check(self.lib.sg_bam_fwd(self.h, self.stream, _dt(x), n, h * w, c, _ptr(x), _ptr(mc), _ptr(ms), _ptr(y)), "sg_bam_fwd")
        return y

    def bam_bwd(self, x, mc, ms, dy):
        n, h, w, c = x.shape
        dx, dmc, dms = torch.empty_like(x), torch.empty_like(mc), torch.empty_like(ms)
        wsp, wsn = self.ws(self.lib.sg_bam_bwd_ws_bytes(self.h, n, h * w, c))
        check(self.lib.sg_bam_bwd(self.h, self.stream, _dt(x), n, h * w, c, _ptr(x), _ptr(mc), _ptr(ms), _ptr(dy), _ptr(dx),
                                  _ptr(dmc), _ptr(dms), wsp, wsn), "sg_bam_bwd")
        return dx, dmc, dms

    # ------------------------------------------------------------------------------------------ pooling
    def maxpool_fwd(self, x, k, stride, padding="valid", out=None, want_idx=False):
        """want_idx: the training form - also returns the uint8 tensor of winning window cells that maxpool_bwd_idx routes
        the gradient by (no x / y needed in the backward)."""
        n, h, w, c = x.shape
        if padding == "same":
            ho, pt, _ = same_pad(h, k, stride)
            wo, pl, _ = same_pad(w, k, stride)
        else:
            ho, wo, pt, pl = (h - k) // stride + 1, (w - k) // stride + 1, 0, 0
        y = out if out is not None else self.empty(n, ho, wo, c, dtype=x.dtype)
        if want_idx:
            idx = torch.empty(n, ho, wo, c, dtype=torch.uint8, device=self.device)
            check(self.lib.sg_maxpool_fwd_idx(self.h, self.stream, _dt(x), n, h, w, c, k, stride, pt, pl, ho, wo, _ptr(x), _ptr(y),
                                              _ptr(idx)), "sg_maxpool_fwd_idx")
            return y, (k, stride, pt, pl, ho, wo), idx
        check(self.lib.sg_maxpool_fwd(self.h, self.stream, _dt(x), n, h, w, c, k, stride, pt, pl, ho, wo, _ptr(x), _ptr(y)),
              "sg_maxpool_fwd")
        return y, (k, stride, pt, pl, ho, wo)

    def maxpool_bwd_idx(self, dy, idx, xshape, geom, out=None):
        n, h, w, c = xshape
        k, stride, pt, pl, ho, wo = geom
        dx = out if out is not None else self.empty(n, h, w, c, dtype=dy.dtype)
        check(self.lib.sg_maxpool_bwd_idx(self.h, self.stream, _dt(dy), n, h, w, c, k, stride, pt, pl, ho, wo, _ptr(dy), _ptr(idx),
                                          _ptr(dx)), "sg_maxpool_bwd_idx")
        return dx

    def maxpool_bwd(self, x, y, dy, geom, out=None):
        n, h, w, c = x.shape
        k, stride, pt, pl, ho, wo = geom
        dx = out if out is not None else torch.empty_like(x)
        check(self.lib.sg_maxpool_bwd(self.h, self.stream, _dt(x), n, h, w, c, k, stride, pt, pl, ho, wo, _ptr(x), _ptr(y),
                                      _ptr(dy), _ptr(dx)), "sg_maxpool_bwd")
        return dx

    def avgpool_fwd(self, x, kh, kw, out=None):
        n, h, w, c = x.shape
        y = out if out is not None else self.empty(n, h // kh, w // kw, c, dtype=x.dtype)
        wsp, wsn = self.ws(self.lib.sg_avgpool_ws_bytes(self.h, n, h, w, c, kh, kw))
        check(self.lib.sg_avgpool_fwd(self.h, self.stream, _dt(x), n, h, w, c, kh, kw, _ptr(x), _ptr(y), wsp, wsn), "sg_avgpool_fwd")
        return y

    def avgpool_bwd(self, dy, xshape, kh, kw, out=None, accumulate=False):
        n, h, w, c = xshape
        dx = out if out is not None else self.empty(n, h, w, c, dtype=dy.dtype)
        check(self.lib.sg_avgpool_bwd(self.h, self.stream, _dt(dy), n, h, w, c, kh, kw, _ptr(dy), _ptr(dx), int(accumulate)),
              "sg_avgpool_bwd")
        return dx

    def upsample_fwd(self, x, sh, sw=None, out=None, out_ld=0):
        sw = sh if sw is None else sw
        n, h, w, c = x.shape
        y = out if out is not None else self.empty(n, h * sh, w * sw, c, dtype=x.dtype)
        check(self.lib.sg_upsample_nearest_fwd(self.h, self.stream, _dt(x), n, h, w, c, sh, sw, _ptr(x), _ptr(y), out_ld),
              "sg_upsample_nearest_fwd")
        return y

    def upsample_bwd(self, dy, xshape, sh, sw=None, out=None, accumulate=False, dy_ld=0):
        sw = sh if sw is None else sw
        n, h, w, c = xshape
        dx = out if out is not None else self.empty(n, h, w, c, dtype=dy.dtype)
        check(self.lib.sg_upsample_nearest_bwd(self.h, self.stream, _dt(dy), n, h, w, c, sh, sw, _ptr(dy), dy_ld, _ptr(dx),
                                               int(accumulate)), "sg_upsample_nearest_bwd")
        return dx

    # ------------------------------------------------------------------------------ loss / metrics / Adam
    def loss_fwd(self, kind, p, y_true):
        _chk32(p, "p"); _chk32(y_true, "y_true")
        rows = p.numel() // 2
        out = self.empty(1)
        wsp, wsn = self.ws(self.lib.sg_loss_ws_bytes(self.h, rows))
        check(self.lib.sg_loss_fwd(self.h, self.stream, kind, rows, y_true.shape[-1], _ptr(p), _ptr(y_true), _ptr(out), wsp, wsn),
              "sg_loss_fwd")
        return out

    def loss_bwd(self, kind, p, y_true, scale=1.0, out=None):
        dp = out if out is not None else torch.empty_like(p)
        check(self.lib.sg_loss_bwd(self.h, self.stream, kind, p.numel() // 2, y_true.shape[-1], _ptr(p), _ptr(y_true), _ptr(dp),
                                   float(scale)), "sg_loss_bwd")
        return dp

    def confusion_counts(self, p, y_true, out=None):
        if out is None:
            out = torch.zeros(4, dtype=torch.int64, device=self.device)
        check(self.lib.sg_confusion_counts(self.h, self.stream, p.numel() // 2, y_true.shape[-1], _ptr(p), _ptr(y_true),
                                           C.c_void_p(out.data_ptr())), "sg_confusion_counts")
        return out

    def adam_step(self, w, m, v, g, lr_t, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, lr_dev=None):
        """lr_dev: a one-float device tensor holding lr_t (a captured training step; lr_t is then ignored)."""
        if lr_dev is not None:
            check(self.lib.sg_adam_step_lr(self.h, self.stream, w.numel(), _ptr(w), _ptr(m), _ptr(v), _ptr(g), _ptr(lr_dev),
                                           beta1, beta2, eps, float(grad_scale)), "sg_adam_step_lr")
            return
        check(self.lib.sg_adam_step(self.h, self.stream, w.numel(), _ptr(w), _ptr(m), _ptr(v), _ptr(g), float(lr_t), beta1,
                                    beta2, eps, float(grad_scale)), "sg_adam_step")

    def edge_labels(self, label, iterations=5):
        """label [N,H,W] float (gray/255) -> y_true [N,H,W,4] as train_data_gen builds it (DeepLabv3plus.py:70-100)."""
        _chk(label, "label")
        n, h, w = label.shape
        y = self.empty(n, h, w, 4)
        check(self.lib.sg_edge_labels(self.h, self.stream, n, h, w, int(iterations), _ptr(label), _ptr(y)), "sg_edge_labels")
        return y

    # -------------------------------------------------------------------------------------- inference tail
    def argmax_accumulate(self, p, canvas, y0, x0):
        th, tw = p.shape[-3], p.shape[-2]
        ch, cw = canvas.shape
        assert canvas.dtype == torch.int8
        check(self.lib.sg_argmax_accumulate_i8(self.h, self.stream, _ptr(p), th, tw, C.c_void_p(canvas.data_ptr()), ch, cw,
                                               y0, x0), "sg_argmax_accumulate_i8")

    def vote_ge(self, masks: Sequence[torch.Tensor], k: int):
        arr = (C.c_void_p * len(masks))(*[m.data_ptr() for m in masks])
        out = torch.empty_like(masks[0])
        check(self.lib.sg_vote_ge(self.h, self.stream, len(masks), arr, masks[0].numel(), k, C.c_void_p(out.data_ptr())),
              "sg_vote_ge")
        return out

    def fill(self, t, value=0.0):
        check(self.lib.sg_fill_f32(self.h, self.stream, _ptr(t), t.numel(), float(value)), "sg_fill_f32")
        return t


    def u8_to_f32(self, src, div, sub=0.0):
        """float32(src) / div - sub of a uint8 device tensor (decode_img / decode_lbel's normalisation)."""
        assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
        dst = torch.empty(src.shape, dtype=torch.float32, device=self.device)
        check(self.lib.sg_u8_to_f32(self.h, self.stream, src.numel(), _ptr(src), _ptr(dst), float(div), float(sub)), "sg_u8_to_f32")
        return dst

    def resize_linear_u8(self, src, oh, ow):
        """cv.resize(img, (ow, oh)) (default INTER_LINEAR, OpenCV's fixed-point arithmetic) of uint8 [N,H,W,C] or [N,H,W]."""
        assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous() and src.dim() in (3, 4)
        n, h, w = src.shape[:3]
        c = src.shape[3] if src.dim() == 4 else 1
        dst = torch.empty((n, oh, ow) + tuple(src.shape[3:]), dtype=torch.uint8, device=self.device)
        check(self.lib.sg_resize_linear_u8(self.h, self.stream, n, h, w, c, _ptr(src), int(oh), int(ow), _ptr(dst)),
              "sg_resize_linear_u8")
        return dst

    def scale(self, t, a):
        check(self.lib.sg_scale_f32(self.h, self.stream, _ptr(t), t.numel(), float(a)), "sg_scale_f32")
        return t


_engines = {}


def get_engine(device: int = 0) -> Engine:
    if device not in _engines:
        _engines[device] = Engine(device)
    return _engines[device]
