"""ctypes binding of libunet_hip.so (the C ABI declared in include/unet_hip.h).

The product path has NO fallback: if the HIP library is missing or does not
export a declared symbol, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

import torch  # noqa: F401  -- FIRST: torch ships its own libamdhip64; loading libunet_hip.so before it binds /opt/rocm's copy and the process ends
#                    up with two HIP runtimes (the second one reports "no ROCm-capable device": seen with build() + smoke() in one process)

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libunet_hip.so"
HEADER = _HERE.parent / "include" / "unet_hip.h"

UNET_OK = 0
CONV_RELU = 1
CONV_MASK = 4
CONV_FWD = 0
CONV_DGRAD = 1
F32 = 0
BF16 = 1

c_float_p = C.POINTER(C.c_float)
vp = C.c_void_p


class Tuning(C.Structure):
    """unet_tuning: the kernel-selection switches of one launch (include/unet_hip.h).  Start from Tuning.default()."""
    _fields_ = [(n, C.c_int) for n in ("conv_splitk", "mfma_shape", "f32_big_tile", "bf16_big_tile", "t256_tiles_per_wg", "t256_sliver",
                                       "conv1x1_gemm", "wgrad_mfma_shape", "wgrad_bf16_k4", "wgrad_1x1", "wgrad_narrow", "plan_batch", "wgrad_wgs", "conv_smallcin", "conv_head1x1")]

    @classmethod
    def default(cls, **over) -> "Tuning":
        t = cls()
        lib.unet_tuning_default(C.byref(t))
        for k, v in over.items():
            if k not in dict(cls._fields_):
                raise KeyError(f"unet_tuning has no field {k!r}")
            setattr(t, k, int(v))
        return t


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", vp), ("x_cs", C.c_int), ("x_co", C.c_int),
        ("wp", vp),
        ("bias", vp),
        ("res", vp), ("res_cs", C.c_int), ("res_co", C.c_int),
        ("mask", vp), ("mask_cs", C.c_int), ("mask_co", C.c_int),
        ("y", vp), ("y_cs", C.c_int), ("y_co", C.c_int),
        ("N", C.c_int), ("IH", C.c_int), ("IW", C.c_int), ("Cin", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int),
        ("ks", C.c_int), ("stride", C.c_int),
        ("kind", C.c_int), ("flags", C.c_int),
        ("colsum", vp),
        ("colsumsq", vp),
        ("cout_begin", C.c_int), ("cout_count", C.c_int),
        ("wp_img_stride", C.c_longlong),
        ("dtype", C.c_int), ("y_f32", C.c_int),
        ("splitk_ws", vp), ("splitk_ws_floats", C.c_size_t),
        ("tuning", C.POINTER(Tuning)),
        ("pixel_shuffle", C.c_int),
        ("ps_tail", vp), ("ps_tail_cs", C.c_int), ("ps_tail_co", C.c_int), ("ps_tail_c", C.c_int), ("ps_tail_at", C.c_int),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", vp), ("x_cs", C.c_int), ("x_co", C.c_int),
        ("dy", vp), ("dy_cs", C.c_int), ("dy_co", C.c_int),
        ("dw", vp),
        ("dbias", vp),
        ("N", C.c_int), ("IH", C.c_int), ("IW", C.c_int), ("Cin", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int), ("ks", C.c_int), ("stride", C.c_int),
        ("workspace", vp), ("workspace_floats", C.c_size_t),
        ("accumulate", C.c_int),
        ("dtype", C.c_int),
        ("tuning", C.POINTER(Tuning)),
    ]


class PackJob(C.Structure):
    _fields_ = [("w", vp), ("wp", vp), ("Cout", C.c_int), ("Cin", C.c_int), ("ks", C.c_int), ("mode", C.c_int), ("out_scale", vp)]


def declared_symbols() -> list[str]:
    """Every function name include/unet_hip.h declares (used by the CPU-side ABI test)."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(unet_[a-z0-9_]+)\s*\(", txt)))


def _load() -> C.CDLL:
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -m unet_amd.build` (hipcc --offload-arch=gfx950). "
            "The MI355X path has no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"libunet_hip.so does not export: {missing}")
    return lib


lib = _load()

i, ll, f, sz = C.c_int, C.c_longlong, C.c_float, C.c_size_t
_sig = {
    "unet_abi_version": (i, []),
    "unet_last_error": (C.c_char_p, []),
    "unet_conv2d_colsum_rows": (i, [C.POINTER(ConvDesc)]),
    "unet_conv2d": (i, [C.POINTER(ConvDesc), vp]),
    "unet_conv2d_variant": (i, [C.POINTER(ConvDesc)]),
    "unet_conv2d_splitk_workspace": (sz, [C.POINTER(ConvDesc)]),
    "unet_tuning_default": (None, [C.POINTER(Tuning)]),
    "unet_pack_batch_table_bytes": (sz, [i]),
    "unet_pack_batch_build": (i, [C.POINTER(PackJob), i, i, vp, C.POINTER(C.c_uint)]),
    "unet_pack_batch_run": (i, [vp, i, C.c_uint, i, vp]),
    "unet_pack_weights_size": (sz, [i, i, i, i]),
    "unet_pack_weights": (i, [vp, vp, i, i, i, i, vp]),
    "unet_pack_weights_strided": (i, [vp, ll, ll, vp, i, i, vp]),
    "unet_row_softmax": (i, [vp, i, i, vp, i, i, ll, i, vp]),
    "unet_row_softmax_bwd": (i, [vp, i, i, vp, i, i, vp, i, i, ll, i, vp]),
    "unet_conv2d_wgrad_workspace": (sz, [C.POINTER(WgradDesc)]),
    "unet_conv2d_wgrad": (i, [C.POINTER(WgradDesc), vp]),
    "unet_bn_stats_rows": (i, [ll]),
    "unet_bn_stats": (i, [vp, i, i, ll, i, vp, vp]),
    "unet_bn_finalize": (i, [vp, vp, i, ll, i, vp, vp, vp, vp, f, f, vp, vp, vp, vp, vp, vp]),
    "unet_bn_eval_coeffs": (i, [vp, vp, vp, vp, f, i, vp, vp, vp]),
    "unet_affine_act": (i, [vp, i, i, vp, vp, vp, i, i, vp, vp, vp, i, i, ll, i, i, vp]),
    "unet_bn_bwd_reduce": (i, [vp, i, i, vp, i, i, vp, i, i, vp, vp, ll, i, vp, vp]),
    "unet_bn_bwd_finalize": (i, [vp, i, ll, i, vp, vp, vp, vp, vp]),
    "unet_bn_bwd_apply": (i, [vp, i, i, vp, i, i, vp, i, i, vp, vp, vp, vp, vp, vp, i, i, vp, i, i, i, ll, i, vp]),
    "unet_maxpool3x3s2": (i, [vp, i, i, vp, i, i, vp, i, i, i, i, i, i, vp]),
    "unet_maxpool3x3s2_bwd": (i, [vp, i, i, vp, vp, i, i, i, i, i, i, i, i, i, vp]),
    "unet_avgpool2_ceil": (i, [vp, i, i, vp, i, i, i, i, i, i, i, i, vp]),
    "unet_avgpool2_ceil_bwd": (i, [vp, i, i, vp, i, i, i, i, i, i, i, i, i, vp]),
    "unet_shuffle_blur": (i, [vp, i, i, vp, i, i, i, i, i, i, i, vp]),
    "unet_shuffle_blur_bwd": (i, [vp, i, i, vp, i, i, vp, i, i, i, i, i, i, i, vp]),
    "unet_shuffle_bwd_xmask": (i, [vp, i, i, vp, i, i, vp, i, i, i, i, i, i, vp]),
    "unet_resize_nearest": (i, [vp, i, i, vp, i, i, i, i, i, i, i, i, vp]),
    "unet_resize_nearest_bwd": (i, [vp, i, i, vp, i, i, i, i, i, i, i, i, vp]),
    "unet_nchw_to_nhwc": (i, [vp, vp, i, i, i, i, i, i, vp]),
    "unet_nhwc_to_nchw": (i, [vp, i, i, vp, i, i, i, i, vp]),
    "unet_copy_slice": (i, [vp, i, i, vp, i, i, ll, i, i, vp]),
    "unet_relu_mask": (i, [vp, i, i, vp, i, i, vp, i, i, ll, i, vp]),
    "unet_colsum": (i, [vp, i, i, ll, i, vp, vp, vp]),
    "unet_colsum_workspace": (sz, [ll, i]),
    "unet_dot": (i, [vp, i, i, vp, i, i, ll, i, vp, vp, vp]),
    "unet_ce_workspace": (sz, [ll]),
    "unet_ce_fwd": (i, [vp, i, i, vp, vp, ll, i, vp, vp, vp, vp]),
    "unet_ce_fwd_parts": (i, [vp, i, i, vp, vp, ll, i, vp, vp, vp]),
    "unet_ce_bwd": (i, [vp, i, i, vp, vp, ll, i, vp, f, vp, i, i, vp]),
    "unet_focal_fwd": (i, [vp, i, i, vp, vp, ll, i, f, vp, vp, vp]),
    "unet_focal_bwd": (i, [vp, i, i, vp, vp, ll, i, f, f, vp, i, i, vp]),
    "unet_regloss_fwd": (i, [vp, i, i, vp, ll, i, f, vp, vp, vp]),
    "unet_regloss_bwd": (i, [vp, i, i, vp, ll, i, f, f, vp, i, i, vp]),
    "unet_softmax_argmax": (i, [vp, i, i, i, i, i, i, vp, vp, vp]),
    "unet_adam_step": (i, [vp, vp, vp, vp, vp, ll, c_float_p, f, f, f, f, i, f, vp]),
    "unet_adam_hyper_floats": (i, []),
    "unet_adam_fill_hyper": (i, [c_float_p, c_float_p, f, f, f, f, i, f]),
    "unet_adam_step_dev": (i, [vp, vp, vp, vp, vp, ll, vp, vp]),
    "unet_mosaic_accumulate": (i, [vp, i, i, i, vp, vp, i, i, i, i, vp]),
    "unet_mosaic_finalize": (i, [vp, vp, i, i, i, vp, vp]),
    "unet_raster_nodata_zero": (i, [vp, i, i, ll, C.c_double, vp]),
    "unet_window_nonzero": (i, [vp, i, i, ll, i, vp, i, i, i, vp, vp]),
    "unet_window_gather": (i, [vp, i, i, ll, ll, i, vp, i, i, i, i, vp, i, i, i, vp]),
    "unet_mosaic_accumulate_windows": (i, [vp, i, i, i, i, i, vp, i, i, i, i, vp, vp, i, i, i, i, vp]),
    "unet_tiff_lzw_decode": (ll, [vp, ll, vp, ll]),
    "unet_tiff_packbits_decode": (ll, [vp, ll, vp, ll]),
    "unet_mosaic_finalize_rows": (i, [vp, vp, i, i, i, i, i, vp, c_float_p, vp]),
    "unet_tiles_stage": (i, [vp, i, i, i, i, i, i, C.c_ulonglong, C.c_ulonglong, vp, vp]),
    "unet_mask_stage": (i, [vp, i, i, i, i, C.c_ulonglong, C.c_ulonglong, vp, i, vp]),
    "unet_dice_counts": (i, [vp, vp, ll, i, vp, vp]),
}
# bf16-storage twins: same argument lists (every tensor is a void pointer on this side)
for _n in ("bn_stats", "affine_act", "bn_bwd_reduce", "bn_bwd_apply", "maxpool3x3s2", "maxpool3x3s2_bwd", "avgpool2_ceil",
           "avgpool2_ceil_bwd", "shuffle_blur", "shuffle_blur_bwd", "shuffle_bwd_xmask", "resize_nearest", "resize_nearest_bwd", "nchw_to_nhwc", "copy_slice", "ce_bwd", "focal_bwd"):
    _sig[f"unet_{_n}_bf16"] = _sig[f"unet_{_n}"]
for _n in ("pack_weights_strided", "row_softmax", "row_softmax_bwd", "relu_mask", "dot"):
    _sig[f"unet_{_n}_bf16"] = _sig[f"unet_{_n}"]
_sig["unet_sa_fused_supported"] = (i, [i, i])
_sig["unet_sa_pack_elems"] = (sz, [i, i])
_sig["unet_sa_pack_bf16"] = (i, [vp, i, i, i, i, i, vp, vp])
_sig["unet_sa_fwd_bf16"] = (i, [vp, i, i, i, i, i, vp, vp, i, i, vp, vp])
_sig["unet_sa_rowdot_bf16"] = (i, [vp, i, i, vp, i, i, i, i, i, vp, vp])
_sig["unet_sa_bwd_bf16"] = (i, [vp, i, i, i, i, i, vp, i, i, vp, vp, vp, vp, vp, vp, vp])
_sig["unet_cast_slice_bf16"] = (i, [vp, i, i, vp, i, i, ll, i, vp])
_sig["unet_pack_weights_size_bf16"] = _sig["unet_pack_weights_size"]
_sig["unet_pack_weights_bf16"] = _sig["unet_pack_weights"]

for _name, (_res, _args) in _sig.items():
    _fn = getattr(lib, _name)
    _fn.restype = _res
    _fn.argtypes = _args

_undeclared = [s for s in declared_symbols() if s not in _sig]
if _undeclared:
    raise ImportError(f"ctypes signatures missing for: {_undeclared}")

if lib.unet_abi_version() != 8:
    raise ImportError("libunet_hip.so ABI version mismatch; rebuild with `python -m unet_amd.build --force`")


class UnetHipError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != UNET_OK:
        msg = lib.unet_last_error().decode(errors="replace")
        raise UnetHipError(f"{what}: rc={rc}: {msg}")

