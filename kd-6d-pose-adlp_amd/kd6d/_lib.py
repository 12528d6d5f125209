"""ctypes binding of libkd6d.so (the C ABI declared in include/kd6d.h).

The library is the product; there is no CPU fallback.  Importing this module without a
built libkd6d.so raises immediately (build with `python kd-6d-pose-adlp_amd/build.py`
or `__graft_entry__.build()`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libkd6d.so")

KD6D_BF16 = 0
KD6D_F32 = 1
ACT_NONE, ACT_LEAKY, ACT_RELU = 0, 1, 2
GN_STATS_READY, GN_WS_ZEROED = 1, 2
MAX_SEG = 5
ABI_VERSION = 8
ACC_ACT, ACC_GRAD = 32, 52           # KD6D_ACC_ACT / KD6D_ACC_GRAD: fixed-point classes of kd6d_acc
NORM_GROUP, NORM_BATCH = 1, 2
BN_FUSED_REPLICAS = 8
NORM_MAX_CTILES = 8


class Seg(ctypes.Structure):
    _fields_ = [("in_h", ctypes.c_int32), ("in_w", ctypes.c_int32),
                ("out_h", ctypes.c_int32), ("out_w", ctypes.c_int32),
                ("in_row0", ctypes.c_int32), ("out_row0", ctypes.c_int32)]


class ConvGeom(ctypes.Structure):
    _fields_ = [("nseg", ctypes.c_int32), ("batch", ctypes.c_int32),
                ("cin", ctypes.c_int32), ("cout", ctypes.c_int32),
                ("ksize", ctypes.c_int32), ("stride", ctypes.c_int32),
                ("pad", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("seg", Seg * MAX_SEG)]


class Levels(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("batch", ctypes.c_int32),
                ("h", ctypes.c_int32 * MAX_SEG), ("w", ctypes.c_int32 * MAX_SEG),
                ("anchor_stride", ctypes.c_float * MAX_SEG), ("anchor_size", ctypes.c_float * MAX_SEG)]


class WgradItem(ctypes.Structure):
    _fields_ = [("geom", ConvGeom), ("x", ctypes.c_void_p), ("dy", ctypes.c_void_p), ("dw", ctypes.c_void_p),
                ("dbias", ctypes.c_void_p)]


class ConvNorm(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("groups", ctypes.c_int32), ("act", ctypes.c_int32), ("eps", ctypes.c_float),
                ("momentum", ctypes.c_float)] + [(n, ctypes.c_void_p) for n in (
                    "gamma", "beta", "y", "stats", "counters", "running_mean", "running_var", "save_mean", "save_invstd")]


class BnIn(ctypes.Structure):
    _fields_ = [("sums", ctypes.c_void_p), ("replicas", ctypes.c_int32), ("act", ctypes.c_int32), ("eps", ctypes.c_float),
                ("momentum", ctypes.c_float)] + [(n, ctypes.c_void_p) for n in (
                    "gamma", "beta", "running_mean", "running_var", "save_mean", "save_invstd")]


MAX_GT = 4
MAX_ZERO = 8


class GnItem(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("x", "dz", "dx", "gamma", "beta", "stats", "gsum_ws", "dgamma", "dbeta")]


class ZeroList(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("pad_", ctypes.c_int32), ("ptr", ctypes.c_void_p * MAX_ZERO),
                ("bytes", ctypes.c_int64 * MAX_ZERO)]


_P = ctypes.c_void_p
_I = ctypes.c_int
_I64 = ctypes.c_int64
_F = ctypes.c_float
_G = ctypes.POINTER(ConvGeom)
_L = ctypes.POINTER(Levels)
_D = ctypes.c_double

# name -> argtypes (every function returns int unless listed in _RESTYPE)
SIGNATURES = {
    "kd6d_abi_version": [],
    "kd6d_device_cu_count": [],
    "kd6d_mark": [_P, _P],
    "kd6d_set_option": [ctypes.c_char_p, ctypes.c_longlong],
    "kd6d_get_option": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)],
    "kd6d_reset_options": [],
    "kd6d_ctx_create": [ctypes.POINTER(_P)],
    "kd6d_ctx_destroy": [_P],
    "kd6d_ctx_current": [],
    "kd6d_ctx_make_current": [_P],
    "kd6d_ctx_set_option": [_P, ctypes.c_char_p, ctypes.c_longlong],
    "kd6d_ctx_get_option": [_P, ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)],
    "kd6d_ctx_reset_options": [_P],
    "kd6d_ctx_barrier_timeouts": [_P],
    "kd6d_ctx_conv2d_pair_begin": [_P],
    "kd6d_ctx_conv2d_pair_end": [_P],
    "kd6d_ctx_conv2d_pair_pending": [_P],
    "kd6d_zero_regions": [ctypes.POINTER(ZeroList), _P, _I, _P],
    "kd6d_uniform_keys": [_P, _I64, _P, ctypes.c_uint64, _P],
    "kd6d_conv2d_fwd": [_G, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _I, _P, _I64, _P],
    "kd6d_conv2d_fwd_block": [_G, _I, _P, ctypes.POINTER(BnIn), _P, _P, _P, _P, _I, _P],
    "kd6d_conv2d_fwd_norm_fusable": [_G, _I, _I, _I],
    "kd6d_conv2d_fwd_norm": [_G, _I, _P, _P, _P, _P, ctypes.POINTER(ConvNorm), _P],
    "kd6d_conv2d_dgrad": [_G, _I, _P, _P, _P, _I, _P],
    "kd6d_conv2d_wgrad": [_G, _I, _P, _P, _P, _I64, _P, _I64, _I, _P],
    "kd6d_conv2d_wgrad_parts": [_G, _I, _I, _I],
    "kd6d_acc_read": [_P, _I64, _I, _P, _I, _I, _P],
    "kd6d_grad_acc_resolve": [_P, _I, _I, _P, _I64, _P, _P],
    "kd6d_grad_acc_resolve_part_groups": [_I],
    "kd6d_wgrad_group_supported": [_G, _I],
    "kd6d_wgrad_group_plan": [ctypes.POINTER(WgradItem), _I, _I, _I, _P, _I64, ctypes.POINTER(ctypes.c_int32)],
    "kd6d_wgrad_group_launch": [_P, _I, _I, _P, _P],
    "kd6d_pack_dgrad_weights": [_I, _P, _P, _P, _I, _I, _P],
    "kd6d_colstats": [_I, _P, _I64, _I, _P, _P, _P],
    "kd6d_bn_train_fwd": [_I, _I, _P, _P, _I64, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P],
    "kd6d_bn_train_bwd_reduce": [_I, _I, _P, _P, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P],
    "kd6d_bn_train_bwd_apply": [_I, _I, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P],
    "kd6d_barrier_timeouts": [],
    "kd6d_conv2d_pair_begin": [],
    "kd6d_conv2d_pair_end": [],
    "kd6d_conv2d_pair_pending": [],
    "kd6d_bn_pool_train_fwd": [_I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P],
    "kd6d_bn_pool_train_bwd": [_I, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P],
    "kd6d_bn_train_bwd": [_I, _I, _P, _P, _P, _I64, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _P],
    "kd6d_gn_relu_fwd": [_I, _I, _P, _P, ctypes.POINTER(ctypes.c_int32), _I, _I, _I, _I, _P, _P, _F, _P, _I, _P],
    "kd6d_gn_relu_bwd": [_I, _I, _P, _P, _P, ctypes.POINTER(ctypes.c_int32), _I, _I, _I, _I, _P, _P, _F, _P,
                         _P, _P, _P, _I64, _I, _P],
    "kd6d_gn_relu_bwd_pair": [_I, _I, ctypes.POINTER(GnItem), ctypes.POINTER(GnItem), ctypes.POINTER(ctypes.c_int32),
                              _I, _I, _I, _I, _F, _I64, _I, _P],
    "kd6d_maxpool2_fwd": [_I, _P, _P, _I, _I, _I, _I, _P],
    "kd6d_maxpool2_bwd": [_I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "kd6d_upsample2_add": [_I, _P, _P, _P, _I, _I, _I, _I, _P],
    "kd6d_sumpool2": [_I, _P, _P, _I, _I, _I, _I, _I, _P],
    "kd6d_eltwise": [_I, _I, _P, _P, _P, _I64, _P],
    "kd6d_image_to_nhwc": [_I, _P, _P, _I, _I, _I, _I, _I, _P],
    "kd6d_sinkhorn_div_fwd_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P],
    "kd6d_sinkhorn_max_points": [],
    "kd6d_sinkhorn_dense_workspace_floats": [_I, _I, _I],
    "kd6d_sinkhorn_dense_diameter": [_P, _P, _I, _I, _I, _P, _P, _P],
    "kd6d_sinkhorn_dense_fwd_bwd": [_P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _F, _D, _P, _I64, _P, _P, _P, _P],
    "kd6d_teacher_select": [_L, _P, _P, _P, _F, _F, _F, _I, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "kd6d_pose_candidates": [_L, _P, _P, _P, _P, _P, _F, _F, _F, _I, _P, _P, _P, _P],
    "kd6d_ssc_assign": [_L, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P],
    "kd6d_focal_fwd": [_P, _P, _I, _F, _F, _P, _P, _P],
    "kd6d_focal_bwd": [_I, _P, _P, _I, _F, _F, _P, _P, _P],
    "kd6d_student_points": [_L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.POINTER(ctypes.c_float),
                            _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    "kd6d_kd_mean": [_P, _P, _I, _P, _P, _P],
    "kd6d_loss_backward": [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _F, _F, _I, _I,
                           _P, _P, _P],
    "kd6d_dzi_crop": [_P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P],
    "kd6d_sumsq": [_P, _I64, _P, _P],
    "kd6d_clip_adamw": [_P, _P, _P, _P, _I64, _P, _P, _D, _D, _D, _D, _D, _D, _I64, _P, _P, _P],
    "kd6d_set_hyper": [_P, _D, _D, _D, _I64, _P],
    "kd6d_cast_f32_to_bf16": [_P, _P, _I64, _P],
    "kd6d_comm_unique_id": [_P],
    "kd6d_comm_init": [ctypes.POINTER(_P), _I, _I, _P],
    "kd6d_comm_rank": [_P],
    "kd6d_comm_world": [_P],
    "kd6d_comm_version": [],
    "kd6d_comm_allreduce": [_P, _P, _I64, _I, _P],
    "kd6d_comm_broadcast": [_P, _P, _I64, _I, _P],
    "kd6d_comm_destroy": [_P],
}
_RESTYPE = {"kd6d_last_error": ctypes.c_char_p, "kd6d_ctx_current": ctypes.c_void_p}


class Kd6dError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libkd6d.so not found at %s -- the HIP extension is the product path and has no "
            "fallback; build it with `python kd-6d-pose-adlp_amd/build.py`" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.kd6d_last_error.restype = ctypes.c_char_p
    lib.kd6d_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, ctypes.c_int64 if name.endswith(("_workspace_floats", "_group_plan")) else ctypes.c_int)
    if lib.kd6d_abi_version() != ABI_VERSION:
        raise ImportError("libkd6d.so ABI version %d != expected %d" % (lib.kd6d_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def check(rc, what=""):
    if rc != 0:
        msg = lib.kd6d_last_error()
        raise Kd6dError("%s failed (rc=%d): %s" % (what or "kd6d call", rc, msg.decode() if msg else "?"))
