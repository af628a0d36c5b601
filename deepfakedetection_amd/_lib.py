"""ctypes binding of libdfd_hip.so (declared in include/dfd_hip.h).

The product path has no CPU or ATen fallback: if the shared object is missing or a
symbol cannot be resolved, importing/using the kernels raises immediately.
"""

from __future__ import annotations

import ctypes
from ctypes import POINTER, Structure, c_double, c_float, c_int, c_int64, c_long, c_size_t, c_uint32, c_void_p
from pathlib import Path

import os as _os

# DFD_LIB_PATH: load another build of the same library (kernel timing experiments)
LIB_PATH = Path(_os.environ.get("DFD_LIB_PATH") or Path(__file__).resolve().parent / "libdfd_hip.so")

DFD_OK = 0
ERRORS = {-1: "DFD_EINVAL", -2: "DFD_EUNSUPPORTED", -3: "DFD_ELAUNCH", -4: "DFD_EWORKSPACE"}
F32, BF16 = 0, 1
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GELU = 0, 1, 2, 3
PRO_NONE, PRO_BN_ACT, PRO_BN_ACT_GATE, PRO_AFFINE2 = 0, 1, 2, 3
MAX_PARTIALS = 1024
ADAMW_TABLE_COLS = 5
ADAMW_HP_LEN = 8


class DwShape(Structure):
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "C", "Ho", "Wo", "k", "stride", "pad_top", "pad_left")]


class StemShape(Structure):
    _fields_ = [(n, c_int) for n in ("N", "H", "W", "Cout", "Ho", "Wo", "k", "stride", "pad_top", "pad_left")]


class Mat(Structure):
    """struct dfd_mat: element (b, h, r, c) at base + b*sb + h*sh + r*sr + c*sc (element units)."""

    _fields_ = [(n, c_long) for n in ("sb", "sh", "sr", "sc")]


class Prologue(Structure):
    _fields_ = [
        ("mode", c_int), ("act", c_int), ("HW", c_int), ("_pad", c_int),
        ("a2", c_void_p), ("coef", c_void_p), ("gate", c_void_p),
    ]


P = c_void_p
_PI = POINTER(c_int)

# name -> (restype, argtypes); every symbol include/dfd_hip.h declares
SIGNATURES: dict[str, tuple] = {
    "dfd_version": (c_int, []),
    "dfd_tune": (c_int, [c_int, c_int]),
    "dfd_dw_mm_plan": (c_int, [POINTER(DwShape), c_int, _PI]),
    "dfd_bn_finalize": (c_int, [P, c_int, c_int, c_double, P, P, P, P, c_float, c_float, P, P]),
    "dfd_bn_eval_coeffs": (c_int, [P, P, P, P, c_float, c_int, P, P]),
    "dfd_bn_bwd_finalize": (c_int, [P, c_int, c_int, c_double, P, P, c_int, P, P, c_int, P, P]),
    "dfd_bn_act_apply": (c_int, [c_int, P, P, c_int, P, P, P, c_int, c_int, c_int, P]),
    "dfd_bn_bwd_reduce": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, P, c_int, _PI, P]),
    "dfd_gemm_bias_act": (c_int, [c_int, P, P, c_int, c_int, c_int, P, c_int, P, P, c_int, P, P, P]),
    "dfd_gemm_plan": (c_int, [c_int, c_int, c_int]),
    "dfd_pw_ntd_plan": (c_int, [c_int, c_int, c_int]),
    "dfd_bias_grad_ws": (c_size_t, [c_int, c_int, c_int]),
    "dfd_bias_grad": (c_int, [c_int, P, P, c_int, c_int, c_int, P, c_int, P, c_size_t, P]),
    "dfd_act_bn_bwd": (c_int, [c_int, P, P, P, P, P, c_int, P, c_int, c_int, c_int, P, c_int, _PI, P]),
    "dfd_act_bn_bwd_se": (c_int, [c_int, P, P, P, P, P, c_int, P, c_int, c_int, c_int, P, c_int, _PI, P, P, c_int, P, P, P, P, c_int, P]),
    "dfd_prep_weights_multi": (c_int, [P, c_int, P]),
    "dfd_resize_crop_u8": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "dfd_augment_u8": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "dfd_image_prep": (c_int, [P, P, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), P, P, P]),
    "dfd_pool_ws": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dfd_pool_act": (c_int, [c_int, P, P, c_int, P, c_int, c_int, c_int, P, c_size_t, P]),
    "dfd_pool_bwd_reduce": (c_int, [c_int, P, P, P, c_int, P, c_int, c_int, c_int, P, c_size_t, P]),
    "dfd_scale_rows": (c_int, [c_int, P, P, P, c_int, c_int, c_int, P]),
    "dfd_se_fc_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P]),
    "dfd_se_fc_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, P, P]),
    "dfd_dwconv_fwd": (c_int, [c_int, P, P, c_int, P, P, POINTER(DwShape), P, c_int, _PI, P]),
    "dfd_dwconv_bwd_data": (c_int, [c_int, P, P, P, P, P, P, c_int, P, POINTER(DwShape), P, c_int, _PI, P]),
    "dfd_dwconv_bwd_weight": (c_int, [c_int, P, P, P, P, P, c_int, P, POINTER(DwShape), c_int, P, c_size_t, P]),
    "dfd_dwconv_bwd_weight_ws": (c_size_t, [POINTER(DwShape)]),
    "dfd_dwconv_bwd_fused": (c_int, [c_int, P, P, P, P, P, P, c_int, P, P, POINTER(DwShape), P, c_int, _PI, c_int, P, c_size_t, P]),
    "dfd_pwconv_fwd": (c_int, [c_int, P, POINTER(Prologue), P, P, P, c_int, c_int, c_int, P, c_int, _PI, P]),
    "dfd_pwconv_wgrad": (c_int, [c_int, P, POINTER(Prologue), c_int, P, POINTER(Prologue), c_int, c_int, P, c_int, P, c_size_t, P]),
    "dfd_pwconv_wgrad_ws": (c_size_t, [c_int, c_int, c_int]),
    "dfd_pwconv_bwd_fused": (c_int, [c_int, P, P, P, P, P, P, c_int, c_int, c_int, P, P, c_int, P, c_size_t, P]),
    "dfd_pw_prep_weights": (c_int, [c_int, P, P, P, c_int, c_int, P]),
    "dfd_stem_conv_fwd": (c_int, [c_int, P, P, P, POINTER(StemShape), P, c_int, _PI, P]),
    "dfd_stem_conv_wgrad": (c_int, [c_int, P, P, P, P, P, POINTER(StemShape), c_int, P, c_size_t, P]),
    "dfd_stem_conv_wgrad_ws": (c_size_t, [POINTER(StemShape)]),
    "dfd_dropout": (c_int, [P, P, c_float, P, c_int, P]),
    "dfd_linear_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "dfd_linear_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dfd_ce_loss": (c_int, [P, P, c_int, c_int, c_float, c_float, P, P, P, P]),
    "dfd_softmax_argmax": (c_int, [P, c_int, c_int, P, P, P]),
    "dfd_adamw_step": (c_int, [P, c_int, P, P]),
    # ---- ABI 111
    "dfd_bn_eval_coeffs_multi": (c_int, [P, c_int, P]),
    "dfd_sum_batch_begin": (c_int, []),
    "dfd_sum_batch_end": (c_int, []),
    "dfd_sum_batch_end_deferred": (c_int, []),
    "dfd_sum_passengers_flush": (c_int, [P]),
    "dfd_sum_passengers_discard": (c_int, []),
    "dfd_se_fwd": (c_int, [c_int, P, P, c_int, c_int, c_int, c_int, P, P, P, P, c_int, c_int, P, P, P, P, P, c_size_t, P]),
    "dfd_se_bwd": (c_int, [c_int, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_int, c_int, P, P, P, P, P, P, c_int,
                           P, c_size_t, P, P]),
    # ---- ABI 112 (eval form of the MBConv block)
    "dfd_pwconv_fwd_eval": (c_int, [c_int, P, P, P, c_int, P, c_int, c_int, c_int, P]),
    "dfd_dwconv_fwd_eval_tiles": (c_int, [c_int, P]),
    "dfd_dwconv_fwd_eval": (c_int, [c_int, P, P, P, c_int, P, P, P, P, P]),
    "dfd_se_fwd_parts": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, c_int, c_int, P, P, P, P, P]),
    # ---- ABI 110
    "dfd_bn_finalize_ex": (c_int, [P, c_int, c_int, c_double, P, P, P, P, P, P, c_float, c_float, P, P]),
    "dfd_bn_eval_coeffs_ex": (c_int, [P, P, P, P, P, P, c_float, c_int, P, P]),
    "dfd_bn_bwd_finalize_ex": (c_int, [P, c_int, c_int, c_double, P, P, P, P, c_int, P, P, P, P, c_int, P, P]),
    "dfd_affine2_apply": (c_int, [c_int, P, P, P, P, c_long, c_int, P]),
    "dfd_bn_add_act": (c_int, [c_int, P, P, P, c_int, P, c_long, c_int, P]),
    "dfd_bn_add_act_bwd": (c_int, [c_int, P, P, P, P, c_int, P, c_long, c_int, P, c_int, _PI, P]),
    "dfd_channel_stats": (c_int, [c_int, P, c_long, c_int, P, c_int, _PI, P]),
    "dfd_sum_rows": (c_int, [P, c_int, c_long, P, c_int, P]),
    "dfd_sum_rows_deferred": (c_int, [P, c_int, c_long, P, c_int, P]),
    "dfd_up2_act_fwd": (c_int, [c_int, P, c_int, P, c_int, c_int, c_int, c_int, P]),
    "dfd_up2_act_bwd": (c_int, [c_int, P, P, c_int, P, c_int, c_int, c_int, c_int, P, P]),
    "dfd_subsample_add": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "dfd_subsample_add_bwd": (c_int, [c_int, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "dfd_bgemm": (c_int, [c_int, P, POINTER(Mat), c_int, P, POINTER(Mat), c_int, P, POINTER(Mat), P, c_float, c_int, c_int,
                          c_int, c_int, c_int, c_int, c_int, P]),
    "dfd_attn_softmax_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dfd_attn_softmax_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dfd_bias_gather": (c_int, [P, P, P, c_int, c_int, c_long, P]),
    "dfd_bias_scatter": (c_int, [P, P, P, c_int, c_int, c_long, c_int, P]),
    "dfd_im2col": (c_int, [c_int, P, P, c_int, P, POINTER(DwShape), P]),
    "dfd_conv_fwd": (c_int, [c_int, P, POINTER(DwShape), P, c_int, P, c_int, P, P, c_int, POINTER(c_int), P]),
    "dfd_conv_wgrad_ws": (c_size_t, [POINTER(DwShape), c_int]),
    "dfd_conv_wgrad": (c_int, [c_int, P, P, c_int, P, POINTER(DwShape), P, c_int, P, c_int, P, c_size_t, P]),
    "dfd_col2im": (c_int, [c_int, P, P, POINTER(DwShape), P]),
    "dfd_conv_weight_perm": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "dfd_layernorm_fwd": (c_int, [c_int, P, P, P, c_float, P, P, c_long, c_int, P]),
    "dfd_layernorm_bwd": (c_int, [c_int, P, P, P, P, P, P, P, c_int, _PI, c_long, c_int, P]),
    "dfd_copy_rows": (c_int, [c_int, P, P, P, P, c_long, c_int, P]),
    "dfd_add_rowtable": (c_int, [c_int, P, P, P, c_long, c_int, c_int, P]),
    "dfd_rowtable_grad_ws": (c_size_t, [c_int, c_int]),
    "dfd_rowtable_grad": (c_int, [c_int, P, P, c_long, c_int, c_int, c_int, P, c_size_t, P]),
    "dfd_avgpool_fwd": (c_int, [c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "dfd_avgpool_bwd": (c_int, [c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "dfd_relpos_bias_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "dfd_relpos_bias_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "dfd_axpby": (c_int, [P, P, c_float, c_float, P, P, c_long, P]),
    "dfd_add": (c_int, [c_int, P, P, P, c_long, P]),
    "dfd_rand": (c_int, [P, c_uint32, c_float, P, c_long, P]),
    "dfd_step_tick": (c_int, [P, c_int, P, P]),
    "dfd_mx_quant_weights_multi": (c_int, [P, c_int, P]),
    "dfd_mx_quant_rows": (c_int, [c_int, P, POINTER(Prologue), P, P, c_long, c_int, P]),
    "dfd_mx_gemm": (c_int, [P, P, P, P, c_int, P, c_long, c_int, c_int, P]),
    "dfd_wattn_parts": (c_int, [c_int]),
    "dfd_wattn_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "dfd_wattn_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "dfd_attn_scores": (c_int, [P, P, P, P, c_float, c_int, c_int, c_int, c_int, c_int, P]),
    "dfd_attn_apply": (c_int, [P, c_int, P, P, c_float, c_int, c_int, c_int, c_int, c_int, P]),
    "dfd_coord_mlp_fwd_multi": (c_int, [P, c_int, P]),
    "dfd_coord_mlp_bwd_multi": (c_int, [P, c_int, P]),
    "dfd_relpos_bias_fwd_multi": (c_int, [P, c_int, P]),
    "dfd_relpos_bias_bwd_multi": (c_int, [P, c_int, P]),
}

class BnEvalJob(Structure):
    """dfd_bn_eval_job"""
    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("conv_bias", c_void_p), ("ls", c_void_p), ("running_mean", c_void_p),
                ("running_var", c_void_p), ("bnstate", c_void_p), ("eps", c_float), ("C", c_int)]


class MxJob(Structure):
    """struct dfd_mx_job (include/dfd_hip.h)."""

    _fields_ = [("src", c_void_p), ("q", c_void_p), ("scale", c_void_p), ("kn", c_void_p), ("N", c_int), ("K", c_int),
                ("kn_dtype", c_int), ("_pad", c_int)]


class CmlpJob(Structure):
    """struct dfd_cmlp_job (include/dfd_hip.h)."""

    _fields_ = [(n, c_void_p) for n in ("coords", "w0", "b0", "w2", "table", "dtable", "dw0", "db0", "dw2")] + \
               [(n, c_int) for n in ("T", "D", "Hd", "_pad")]


class RelposJob(Structure):
    """struct dfd_relpos_job (include/dfd_hip.h)."""

    _fields_ = [(n, c_void_p) for n in ("table", "idx", "full", "dfull", "dtable")] + [(n, c_int) for n in ("H", "T", "n_local", "n_global")]


class PrepJob(Structure):
    """struct dfd_prep_job (include/dfd_hip.h)."""

    _fields_ = [("src", c_void_p), ("nk", c_void_p), ("kn", c_void_p), ("N", c_int), ("K", c_int), ("dtype", c_int),
                ("_pad", c_int)]


_lib: ctypes.CDLL | None = None


def load() -> ctypes.CDLL:
    """Load libdfd_hip.so and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: its bundled HIP runtime must be the one this process uses.  If libdfd_hip.so is
    # dlopen'ed before torch, the loader resolves libamdhip64 to the system copy, torch then brings its
    # own, and every launch on a torch stream fails (two runtimes, foreign stream handles).
    import torch  # noqa: F401

    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m deepfakedetection_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise RuntimeError(f"libdfd_hip.so does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    # DFD_TUNE="key=value,key=value": planner knobs of include/dfd_hip.h (dfd_tune) from the command line — A/B runs of
    # bench.py / the layer scripts without a rebuild.  Applied once, before anything launches.
    for item in filter(None, _os.environ.get("DFD_TUNE", "").split(",")):
        key, _, value = item.partition("=")
        if lib.dfd_tune(int(key), int(value)) != DFD_OK:
            raise RuntimeError(f"DFD_TUNE: unknown key in {item!r}")
    _lib = lib
    return lib


def check(code: int, op: str, detail: str = "") -> None:
    if code != DFD_OK:
        raise RuntimeError(f"{op} failed with {ERRORS.get(code, code)} {detail}")


__all__ = [
    "ACT_GELU", "ACT_NONE", "ACT_RELU", "ACT_SILU", "ADAMW_HP_LEN", "ADAMW_TABLE_COLS", "BF16", "F32",
    "MAX_PARTIALS", "Mat", "PRO_AFFINE2", "PRO_BN_ACT", "PRO_BN_ACT_GATE", "PRO_NONE", "DwShape", "Prologue",
    "StemShape", "SIGNATURES", "check", "load", "c_int64",
]
