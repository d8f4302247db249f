"""ctypes binding of include/fav.h (the C-ABI drop-in boundary).

Loads ``failure_aware_vision_amd/lib/libfav_hip.so`` (built in-tree by
``__graft_entry__.build()`` / ``csrc/Makefile``).  There is NO fallback: if the
library is missing the import of this module raises, and if no gfx950 device is
present ``fav_create`` returns FAV_ERR_NO_DEVICE, which ``Backend`` turns into a
RuntimeError.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FAV_LIB_PATH") or os.path.join(_HERE, "lib", "libfav_hip.so")   # FAV_LIB_PATH: A/B of two builds

FAV_OK = 0
STATUS_NAMES = {0: "FAV_OK", 1: "FAV_ERR_INVALID_ARG", 2: "FAV_ERR_BAD_BLOB", 3: "FAV_ERR_NO_WEIGHTS",
                4: "FAV_ERR_HIP", 5: "FAV_ERR_NO_DEVICE", 6: "FAV_ERR_UNSUPPORTED"}
LAYOUT_NHWC_U8, LAYOUT_NHWC_F32 = 0, 1
ARCH_RESNET18_CIFAR, ARCH_RESNET50, ARCH_VIT_B16, ARCH_VIT_TINY = 0, 1, 2, 3
CONF_MAX_SOFTMAX, CONF_ENTROPY = 0, 1
MATH_BF16, MATH_F32_EXACT = 0, 1
K_STEM, K_CONV, K_MAXPOOL, K_AVGPOOL, K_DROPOUT, K_HEAD, K_COUNT = 0, 1, 2, 3, 4, 5, 6
KERNEL_CLASS_NAMES = ("stem_im2col", "conv_igemm", "maxpool", "avgpool", "entry_dropout", "head")


class FavConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("device", C.c_int32), ("arch", C.c_int32), ("num_classes", C.c_int32),
        ("in_h", C.c_int32), ("in_w", C.c_int32), ("max_batch", C.c_int32),
        ("mean", C.c_float * 3), ("stdev", C.c_float * 3),
        ("n_samples", C.c_int32), ("site_mask", C.c_uint32), ("dropout_p", C.c_float), ("seed", C.c_uint64),
        ("temperature", C.c_float), ("conf_kind", C.c_int32), ("tau", C.c_float), ("math_mode", C.c_int32),
        ("chunk_a", C.c_int32), ("chunk_b", C.c_int32), ("regroup_block", C.c_int32), ("n_members", C.c_int32),
        ("tail_min_rows", C.c_int32), ("ens_grouped_max", C.c_int32), ("vit_streams", C.c_int32), ("stem_fused", C.c_int32),
    ]


class FavDropoutDesc(C.Structure):
    _fields_ = [("site", C.c_int32), ("threshold", C.c_uint32), ("scale", C.c_float), ("seed", C.c_uint64),
                ("v0", C.c_int64), ("n_img", C.c_int32), ("first_image_index", C.c_int64)]


class FavConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("y", C.c_void_p),
                ("n_frames", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("relu", C.c_int32), ("out_f32", C.c_int32), ("math_mode", C.c_int32), ("drop", FavDropoutDesc)]


class FavLinearDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("y", C.c_void_p),
                ("rows", C.c_int64), ("K", C.c_int32), ("N", C.c_int32), ("act", C.c_int32)]


class FavTailDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("wb", C.c_void_p), ("bias_b", C.c_void_p), ("wc", C.c_void_p), ("bias_c", C.c_void_p),
                ("res", C.c_void_p), ("y", C.c_void_p), ("wa", C.c_void_p), ("bias_a", C.c_void_p), ("t1n", C.c_void_p),
                ("n_frames", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cmid", C.c_int32), ("Nred", C.c_int32),
                ("drop", FavDropoutDesc), ("res_entry", C.c_int32), ("entry_site", C.c_int32)]


class FavProfile(C.Structure):
    _fields_ = [("ms", C.c_double * K_COUNT), ("flops", C.c_double * K_COUNT), ("bytes", C.c_double * K_COUNT),
                ("launches", C.c_int64 * K_COUNT)]


class FavOpProfile(C.Structure):
    _fields_ = [("op_index", C.c_int32), ("kind", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("Cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("stride", C.c_int32), ("reserved", C.c_int32), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double), ("launches", C.c_int64)]


_SIGNATURES = {
    "fav_abi_version": (C.c_int32, []),
    "fav_default_config": (None, [C.POINTER(FavConfig), C.c_int32]),
    "fav_create": (C.c_int, [C.POINTER(FavConfig), C.POINTER(C.c_void_p)]),
    "fav_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "fav_load_member_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t]),
    "fav_destroy": (None, [C.c_void_p]),
    "fav_check_blob": (C.c_int, [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "fav_plan_schedule": (C.c_int, [C.POINTER(FavConfig), C.c_int32, C.c_char_p, C.c_size_t]),
    "fav_last_error": (C.c_char_p, [C.c_void_p]),
    "fav_classify": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fav_classify_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "fav_classify_records": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "fav_classify_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "fav_get_logits": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]),
    "fav_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "fav_get_profile": (C.c_int, [C.c_void_p, C.POINTER(FavProfile), C.c_int32]),
    "fav_get_op_profile": (C.c_int, [C.c_void_p, C.POINTER(FavOpProfile), C.c_int32, C.POINTER(C.c_int32)]),
    "fav_op_conv2d": (C.c_int, [C.POINTER(FavConvDesc), C.c_void_p]),
    "fav_op_bottleneck_tail": (C.c_int, [C.POINTER(FavTailDesc), C.c_void_p]),
    "fav_op_linear_streamk": (C.c_int, [C.POINTER(FavLinearDesc), C.c_void_p]),
    "fav_op_stem_im2col": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.c_void_p, C.c_void_p]),
    "fav_op_stem_pool": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fav_op_maxpool3x3s2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fav_op_avgpool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(FavDropoutDesc),
                                 C.c_void_p]),
    "fav_op_entry_dropout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(FavDropoutDesc),
                                       C.c_void_p]),
    "fav_op_entry_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.POINTER(FavDropoutDesc), C.c_void_p]),
    "fav_op_layernorm": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                   C.c_void_p]),
    "fav_op_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fav_op_vit_assemble": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fav_op_head": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    """Load the shared library once.  torch is imported first so that the HIP
    runtime both sides use is the single libamdhip64.so.7 torch already mapped
    (device pointers and streams are then interchangeable)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C failure_aware_vision_amd/csrc`).  There is no CPU fallback for this path.")
    import torch  # noqa: F401  (maps torch's libamdhip64 first)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class FavError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def check(status: int, handle=None):
    if status != FAV_OK:
        msg = load().fav_last_error(handle)
        raise FavError(status, (msg or b"").decode("utf-8", "replace"))
