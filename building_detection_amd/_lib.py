"""ctypes binding of libsegengine.so (the C ABI declared in include/segengine.h).

This is the only place the Python host touches native code.  There is NO CPU fallback: if the shared library
is missing, or no gfx950 device is visible when a context is requested, the import / call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsegengine.so")

SG_F32, SG_BF16, SG_I64 = 0, 1, 2
SG_HEAD_F32 = 0x100  # OR-ed into the dtype of a thin 1x1 conv on bf16 storage: its few-channel side is fp32 (softmax head)
SG_COMM_ID_BYTES = 128
SG_EPI_BIAS, SG_EPI_RELU = 1, 2
SG_PRO_UP2, SG_EPI_DOWN2 = 16, 32   # UpSampling2D(2) -> Conv2D 3x3 fused: forward prologue / dgrad epilogue (segengine.h)
SG_X_UP2 = 0x200                   # ... and the dtype flag of its filter gradient
SG_ACT_RELU, SG_ACT_SIGMOID = 0, 1
SG_LOSS_CE2, SG_LOSS_FOCAL, SG_LOSS_EDGE_FOCAL = 0, 1, 2


class SgError(RuntimeError):
    pass


class PlanesJob(C.Structure):
    """Mirror of `sg_planes_job` (include/segengine.h): one weight tensor -> its prepared bf16 operand planes."""

    _fields_ = [("w_off", C.c_int64), ("out_off", C.c_int64)] + [(n, C.c_int32) for n in (
        "kind", "K", "N", "Kpad", "Npad", "Ck", "Ckp", "s_tap", "s_k", "s_n", "npl", "block0", "nblocks", "kd")]


SG_WS_PREPARED = C.c_size_t(-1).value


class ConvDesc(C.Structure):
    """Mirror of `sg_conv_desc` (include/segengine.h)."""

    _fields_ = [(n, C.c_int32) for n in (
        "N", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "dilation", "pad_t", "pad_l", "Ho", "Wo",
        "x_ld", "y_ld")]


class BnIn(C.Structure):
    """Mirror of `sg_bn_in` (include/segengine.h): a BatchNormalization (+ReLU) applied to a convolution's input in its loader."""

    _fields_ = [("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("relu", C.c_int32), ("infer", C.c_int32), ("eps", C.c_float)]


class BnBwdIn(C.Structure):
    """Mirror of `sg_bn_bwd_in` (include/segengine.h): a BatchNormalization's backward apply evaluated inside a pointwise dgrad."""

    _fields_ = [("x", C.c_void_p), ("mean", C.c_void_p), ("invstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dz", C.c_void_p), ("relu", C.c_int32), ("rows", C.c_int64)]


_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_dp = C.POINTER(ConvDesc)
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes).  Kept in the order of include/segengine.h.
_SIGNATURES = {
    "sg_abi_version": (_i, []),
    "sg_last_error": (C.c_char_p, []),
    "sg_set_conv_x6": (_i, [_i]),
    "sg_create": (_i, [_i, _pp]),
    "sg_destroy": (_i, [_vp]),
    "sg_num_cus": (_i, [_vp]),
    "sg_conv2d_fwd": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i]),
    "sg_conv2d_fwd_ws_bytes": (_sz, [_dp]),
    "sg_conv2d_fwd_ws": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "sg_conv2d_up2_supported": (_i, [_i, _dp]),
    "sg_conv2d_fwd_stats_bytes": (_sz, [_dp]),
    "sg_conv2d_fwd_stats": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, C.POINTER(C.c_int)]),
    "sg_get_conv_x6": (_i, []),
    "sg_conv2d_planes_job": (_i, [_vp, _i, _dp, _i, C.POINTER(PlanesJob), C.POINTER(C.c_size_t)]),
    "sg_prepare_planes": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i]),
    "sg_bn_tiles_ws_bytes": (_sz, [_vp, _i, _i]),
    "sg_bn_train_fwd_tiles": (_i, [_vp, _vp, _i, _i64, _i, _vp, _i, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _i, _vp, _sz]),
    "sg_bn_apply": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "sg_conv2d_dgrad_ws_bytes": (_sz, [_dp]),
    "sg_conv2d_dgrad": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "sg_conv2d_dgrad_acc": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "sg_conv2d_wgrad_ws_bytes": (_sz, [_vp, _dp]),
    "sg_conv2d_wgrad": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "sg_conv2d_bn_in_supported": (_i, [_vp, _i, _dp]),
    "sg_conv2d_fwd_stats_bn": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, C.POINTER(C.c_int), C.POINTER(BnIn)]),
    "sg_conv2d_wgrad_bn": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(BnIn)]),
    "sg_conv2d_dgrad_bnb_supported": (_i, [_vp, _i, _dp]),
    "sg_conv2d_dgrad_bnb": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _sz, C.POINTER(BnBwdIn)]),
    "sg_split_planes": (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp]),
    "sg_conv2d_planes_in": (_i, [_dp, _i]),
    "sg_conv2d_fwd_stats_ap": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp, C.POINTER(C.c_int), _vp]),
    "sg_conv2d_dgrad_ap": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "sg_conv2d_wgrad_planes_supported": (_i, [_vp, _dp]),
    "sg_conv2d_wgrad_planes_ws_bytes": (_sz, [_vp, _dp]),
    "sg_conv2d_wgrad_planes": (_i, [_vp, _vp, _dp, _vp, _vp, _vp, _vp, _sz]),
    "sg_bias_grad_ws_bytes": (_sz, [_vp, _i64, _i]),
    "sg_bias_grad": (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _sz]),
    "sg_dwconv2d_fwd": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _i]),
    "sg_dwconv2d_dgrad": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i]),
    "sg_dwconv2d_dgrad_acc": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sg_dwconv2d_dgrad_bnsums_ws_bytes": (_sz, [_vp, _dp]),
    "sg_dwconv2d_dgrad_bnsums": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _i, _vp] + [_vp] * 5 + [_i, _vp, _vp, _vp, _sz]),
    "sg_dwconv2d_fwd_bn": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i]),
    "sg_dwconv2d_wgrad_bn": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "sg_dwconv2d_wgrad_ws_bytes": (_sz, [_vp, _dp]),
    "sg_dwconv2d_wgrad": (_i, [_vp, _vp, _i, _dp, _vp, _vp, _vp, _i, _vp, _sz]),
    "sg_dense_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i]),
    "sg_bn_ws_bytes": (_sz, [_vp, _i64, _i]),
    "sg_bn_train_fwd": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _vp, _sz]),
    "sg_bn_train_bwd": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "sg_bn_train_bwd_apply": (_i, [_vp, _vp, _i, _i64, _i] + [_vp] * 9 + [_i]),
    "sg_bn_infer": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i]),
    "sg_add2_bn": (_i, [_vp, _vp, _i, _i64, _i] + [_vp] * 11 + [_i, _i, _f, _i, _i]),
    "sg_act_fwd": (_i, [_vp, _vp, _i, _i, _i64, _vp, _vp]),
    "sg_act_bwd": (_i, [_vp, _vp, _i, _i, _i64, _vp, _vp, _vp, _i]),
    "sg_add_n": (_i, [_vp, _vp, _i, _i, _pp, _i64, _vp, _i]),
    "sg_copy_channels": (_i, [_vp, _vp, _i, _i64, _i, _vp, _i, _i, _vp, _i, _i, _i]),
    "sg_softmax2_fwd": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "sg_softmax2_bwd": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "sg_softmax_branch_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sg_softmax_branch_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sg_bcast_mul_fwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp, _i]),
    "sg_bcast_mul_bwd_ws_bytes": (_sz, [_vp, _i, _i64, _i, _i]),
    "sg_bcast_mul_bwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _sz]),
    "sg_scse_fwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    "sg_scse_bwd_ws_bytes": (_sz, [_vp, _i, _i64, _i]),
    "sg_scse_bwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "sg_bam_fwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp]),
    "sg_bam_bwd_ws_bytes": (_sz, [_vp, _i, _i64, _i]),
    "sg_bam_bwd": (_i, [_vp, _vp, _i, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz]),
    "sg_maxpool_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "sg_maxpool_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "sg_maxpool_fwd_idx": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sg_maxpool_bwd_idx": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sg_avgpool_ws_bytes": (_sz, [_vp, _i, _i, _i, _i, _i, _i]),
    "sg_avgpool_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz]),
    "sg_avgpool_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i]),
    "sg_upsample_nearest_fwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i]),
    "sg_upsample_nearest_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i]),
    "sg_loss_ws_bytes": (_sz, [_vp, _i64]),
    "sg_loss_fwd": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _sz]),
    "sg_loss_bwd": (_i, [_vp, _vp, _i, _i64, _i, _vp, _vp, _vp, _f]),
    "sg_confusion_counts": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp]),
    "sg_adam_step": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f]),
    "sg_adam_step_lr": (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f]),
    "sg_edge_labels": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sg_resize_linear_u8": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "sg_argmax_accumulate_i8": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i]),
    "sg_vote_ge": (_i, [_vp, _vp, _i, _pp, _i64, _i, _vp]),
    "sg_mask_objects_ws_bytes": (_sz, [_i, _i]),
    "sg_mask_objects": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _sz, _vp, _vp, _i, _vp, _vp]),
    "sg_mask_split_words": (_i64, [_i, _i, _i, _i, _i, _i]),
    "sg_mask_split": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "sg_u8_to_f32": (_i, [_vp, _vp, _i64, _vp, _vp, _f, _f]),
    "sg_cast": (_i, [_vp, _vp, _i, _i, _i64, _vp, _vp]),
    "sg_fill_f32": (_i, [_vp, _vp, _vp, _i64, _f]),
    "sg_trace_mark": (_i, [_vp, _vp, _i, _i]),
    "sg_scale_f32": (_i, [_vp, _vp, _vp, _i64, _f]),
    "sg_comm_probe": (_i, [_vp]),
    "sg_comm_unique_id": (_i, [_vp]),
    "sg_comm_init": (_i, [_vp, _i, _i, _i, _pp]),
    "sg_comm_allreduce_sum": (_i, [_vp, _vp, _i, _vp, _i64]),
    "sg_comm_rank": (_i, [_vp]),
    "sg_comm_nranks": (_i, [_vp]),
    "sg_comm_destroy": (_i, [_vp]),
}

_lib = None
_lock = threading.Lock()


def load():
    """dlopen libsegengine.so (once) and attach the signatures.  Raises SgError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch must be imported first: it brings its own HIP runtime (libamdhip64) and a second copy loaded
        # ahead of it by this library would not see the device torch allocates on.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise SgError(
                f"{LIB_PATH} is missing - build it with `make -C building_detection_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI and the binding diverge
            fn.restype = res
            fn.argtypes = args
        if lib.sg_abi_version() != 1:
            raise SgError(f"libsegengine ABI {lib.sg_abi_version()} != binding ABI 1")
        _lib = lib
    return _lib


def exported_symbols():
    return list(_SIGNATURES)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().sg_last_error().decode("utf-8", "replace")
        raise SgError(f"{what or 'libsegengine'} failed (rc={rc}): {msg}")


class Context:
    """One `sg_ctx` per device (creation fails without a gfx950 GPU)."""

    def __init__(self, device: int = 0):
        lib = load()
        h = C.c_void_p()
        check(lib.sg_create(device, C.byref(h)), "sg_create")
        self.handle = h
        self.device = device
        self.num_cus = lib.sg_num_cus(h)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                load().sg_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
