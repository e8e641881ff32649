"""ctypes binding of libkivi_hip.so (C ABI: include/kivi_hip.h).

Fails loudly: if the shared library has not been built (python -m kivi_amd.build
or __graft_entry__.build()) importing any op raises; nothing here falls back to
PyTorch or to the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkivi_hip.so")
# tuning sessions (KIVI_TUNING=1; A/B of two builds inside one GPU session): KIVI_HIP_LIB=/path/to/other/libkivi_hip.so
from . import _tuning  # noqa: E402

LIB_PATH = _tuning.knob("KIVI_HIP_LIB", LIB_PATH)

_i64, _i32, _vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p

_GEMV_ARGS = [_vp, _i64, _i64,            # q / a + strides
              _vp, _i64, _i64, _i64,      # code + strides
              _vp, _vp, _i64, _i64, _i64,  # scale, mn + strides
              _vp, _i64, _i64]            # out + strides

class DecodeAttendArgs(ctypes.Structure):
    """kivi_decode_attend_args (include/kivi_hip.h), field for field."""
    _fields_ = [
        ("q", _vp), ("q_sb", _i64), ("q_sh", _i64),
        ("kres", _vp), ("kres_sb", _i64), ("kres_sh", _i64), ("kres_st", _i64),
        ("knew", _vp), ("knew_sb", _i64), ("knew_sh", _i64), ("k_res_len", _i32),
        ("scores", _vp), ("s_sb", _i64), ("s_sh", _i64),
        ("inv_scale", ctypes.c_float), ("mask", _vp), ("mask_sb", _i64),
        ("v_code", _vp), ("vc_sb", _i64), ("vc_sh", _i64), ("vc_sr", _i64),
        ("v_scale", _vp), ("v_mn", _vp), ("vs_sb", _i64), ("vs_sh", _i64), ("vs_sr", _i64),
        ("vres", _vp), ("vres_sb", _i64), ("vres_sh", _i64), ("vres_st", _i64), ("v_win_start", _i32), ("v_res_len", _i32),
        ("vnew", _vp), ("vnew_sb", _i64), ("vnew_sh", _i64), ("v_flush", _i32),
        ("out", _vp), ("out_sb", _i64), ("out_sh", _i64),
        ("B", _i32), ("nh", _i32), ("nh_kv", _i32), ("D", _i32), ("group_size", _i32), ("v_bits", _i32),
        ("Tq", _i64), ("Tv", _i64),
        ("workspace", _vp), ("workspace_bytes", _i64),
        ("k_code", _vp), ("kc_sb", _i64), ("kc_sh", _i64), ("kc_sp", _i64), ("kc_sr", _i64),
        ("k_scale", _vp), ("k_mn", _vp), ("ks_sb", _i64), ("ks_sh", _i64), ("ks_sp", _i64), ("ks_sr", _i64),
        ("k_page_tokens", _i64), ("k_bits", _i32),
    ]


class LayerDesc(ctypes.Structure):
    """kivi_layer_desc (include/kivi_hip.h), field for field."""
    _fields_ = [
        ("B", _i32), ("nh_kv", _i32), ("D", _i32), ("k_bits", _i32), ("v_bits", _i32), ("group_size", _i32),
        ("residual_length", _i32), ("inv_scale", ctypes.c_float),
        ("cap", _i64), ("page_tokens", _i64), ("v_window_rows", _i64), ("s_pitch", _i64),
        ("k_code", _vp), ("kc_sb", _i64), ("kc_sh", _i64), ("kc_sp", _i64), ("kc_sr", _i64),
        ("k_scale", _vp), ("k_mn", _vp), ("ks_sb", _i64), ("ks_sh", _i64), ("ks_sp", _i64), ("ks_sr", _i64),
        ("k_res", _vp), ("kr_sb", _i64), ("kr_sh", _i64), ("kr_st", _i64),
        ("v_code", _vp), ("vc_sb", _i64), ("vc_sh", _i64), ("vc_sr", _i64),
        ("v_scale", _vp), ("v_mn", _vp), ("vs_sb", _i64), ("vs_sh", _i64), ("vs_sr", _i64),
        ("v_res", _vp), ("vr_sb", _i64), ("vr_sh", _i64), ("vr_st", _i64),
        ("scores", _vp), ("s_sb", _i64), ("s_sh", _i64),
        ("workspace", _vp), ("workspace_bytes", _i64),
    ]


class GqaDecodeArgs(ctypes.Structure):
    """kivi_gqa_decode_args of include/kivi_hip.h (field order must match)."""
    _fields_ = [
        ("B", _i32), ("nh", _i32), ("nh_kv", _i32), ("D", _i32), ("group_size", _i32), ("bits", _i32),
        ("inv_scale", ctypes.c_float),
        ("q", _vp), ("q_sb", _i64), ("q_sh", _i64),
        ("mask", _vp), ("mask_sb", _i64),
        ("kt", _vp), ("kt_sb", _i64), ("kt_sh", _i64), ("kt_ss", _i64), ("Tq", _i64),
        ("kres", _vp), ("kres_sb", _i64), ("kres_sh", _i64), ("kres_st", _i64),
        ("knew", _vp), ("knew_sb", _i64), ("knew_sh", _i64), ("k_res_len", _i32),
        ("vt", _vp), ("vt_sb", _i64), ("vt_sh", _i64), ("vt_ss", _i64), ("Tv", _i64),
        ("vres", _vp), ("vres_sb", _i64), ("vres_sh", _i64), ("vres_st", _i64), ("v_win_start", _i32), ("v_res_len", _i32),
        ("vnew", _vp), ("vnew_sb", _i64), ("vnew_sh", _i64), ("v_flush", _i32),
        ("scores", _vp), ("s_sb", _i64), ("s_sh", _i64),
        ("stats", _vp), ("stats_bytes", _i64),
        ("workspace", _vp), ("workspace_bytes", _i64),
        ("out", _vp), ("out_sb", _i64), ("out_sh", _i64),
        ("residual_length", _i32), ("v_window_rows", _i64), ("kt_superblocks", _i64), ("vt_superblocks", _i64),
        ("flags", _i32),
        ("kt_range", _vp), ("vt_range", _vp),
        ("dyn_step", _vp),
    ]


class MfStep(ctypes.Structure):
    """kivi_mf_step (include/kivi_hip.h): the six lengths of a decode step, host copy of the device-resident struct."""
    _fields_ = [("Tq", _i64), ("Tv", _i64), ("k_res_len", _i32), ("v_res_len", _i32), ("v_win_start", _i32), ("v_flush", _i32)]


GQA_FORCE_SPLIT, GQA_FORCE_ROW, GQA_WINDOW_RING, GQA_DUMP_SCORES = 1, 2, 4, 8


def gqa_slices(n: int) -> int:
    """KIVI_GQA_SLICES(n): force the one-launch form with n slices per row (nh / nh_kv in {4, 8}; tests, tuning)."""
    return (n & 0xFF) << 8


class MfLayerDesc(ctypes.Structure):
    """kivi_mf_layer_desc (include/kivi_hip.h), field for field."""
    _fields_ = [
        ("B", _i32), ("nh_kv", _i32), ("D", _i32), ("bits", _i32), ("group_size", _i32), ("residual_length", _i32),
        ("inv_scale", ctypes.c_float),
        ("cap", _i64), ("v_window_rows", _i64), ("s_pitch", _i64),
        ("kt", _vp), ("kt_sb", _i64), ("kt_sh", _i64), ("kt_ss", _i64),
        ("vt", _vp), ("vt_sb", _i64), ("vt_sh", _i64), ("vt_ss", _i64),
        ("k_res", _vp), ("kr_sb", _i64), ("kr_sh", _i64), ("kr_st", _i64),
        ("v_res", _vp), ("vr_sb", _i64), ("vr_sh", _i64), ("vr_st", _i64),
        ("scores", _vp), ("s_sb", _i64), ("s_sh", _i64),
        ("stats", _vp), ("stats_bytes", _i64),
        ("workspace", _vp), ("workspace_bytes", _i64),
        ("flags", _i32),
        ("kt_range", _vp), ("vt_range", _vp),
    ]


# name -> (restype, argtypes); must list every symbol include/kivi_hip.h declares
SIGNATURES = {
    "kivi_abi_version": (_i32, []),
    "kivi_last_error": (ctypes.c_char_p, []),
    "kivi_device_error": (_i32, []),
    "kivi_quant_pack_lastdim": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "kivi_quant_pack_k_tmajor": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64,
                                        _i64, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "kivi_unpack_dequant_lastdim": (_i32, [_vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "kivi_pack_codes_lastdim": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp]),
    "kivi_unpack_codes_lastdim": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp]),
    "kivi_gemv_k": (_i32, _GEMV_ARGS + [_i32, _i32, _i32, _i32, _i64, _i32, _i32, _vp]),
    "kivi_gemv_v": (_i32, _GEMV_ARGS + [_i32, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "kivi_gemv_outer_dim": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "kivi_gemv_k_paged": (_i32, [_i32, _i64, _i64, _i64] + _GEMV_ARGS + [_i32, _i32, _i32, _i32, _i64, _i32, _i32, _vp]),
    "kivi_decode_scores": (_i32, [_i64, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64,
                                  _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i32, _vp, _i64, _i64,
                                  _i32, _i32, _i32, _i32, _i64, _i32, _i32, _vp]),
    "kivi_softmax_scaled": (_i32, [_vp, _vp, _i64, _i64, _i64, _i64, ctypes.c_float, _vp, _i64, _i32, _vp]),
    "kivi_decode_output": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64,
                                  _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _i64, _i32, _vp, _i64, _i64,
                                  _i32, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "kivi_decode_softmax_output": (_i32, [_vp, _i64, _i64, ctypes.c_float, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64,
                                          _i64, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _i64, _i64, _i32, _vp, _i64,
                                          _i64, _i32, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "kivi_decode_attend": (_i32, [ctypes.POINTER(DecodeAttendArgs), _vp]),
    "kivi_decode_layer": (_i32, [ctypes.POINTER(LayerDesc), ctypes.POINTER(_i64), _vp, _i64, _i64, _i32, _vp, _i64, _i64,
                                 _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp]),
    # (a KT / VT store = pointer + 3 word strides + the pointer to its range flags)
    "kivi_kt_pack": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "kivi_vt_pack": (_i32, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "kivi_kt_relayout": (_i32, [_i32, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                _i64, _i32, _i32, _i32, _vp]),
    "kivi_vt_relayout": (_i32, [_i32, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _i32,
                                _i64, _i32, _i32, _i32, _vp]),
    "kivi_gqa_scores": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i64, _i32,
                               _i32, _vp]),
    "kivi_gqa_output": (_i32, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i64, _i32,
                               _i32, _vp, _i64, _vp]),
    "kivi_gqa_decode": (_i32, [ctypes.POINTER(GqaDecodeArgs), _vp]),
    "kivi_mf_launch_plan": (_i32, [_i32, _i32, _i32, _i64, _i32, _i32, _i32, _i32, _i32]),
    "kivi_mf_decode_layer": (_i32, [ctypes.POINTER(MfLayerDesc), ctypes.POINTER(_i64), _vp, _i64, _i64, _i32, _vp, _i64, _i64,
                                    _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp]),
    "kivi_mf_decode_layer_dyn": (_i32, [ctypes.POINTER(MfLayerDesc), ctypes.POINTER(MfStep), _vp, _vp, _i64, _i64, _i32, _vp, _i64, _i64,
                                        _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _vp]),
    "kivi_mf_step_key": (_i64, [ctypes.POINTER(MfStep), _i32, _i32, _i32, _i32, _i32]),
    "kivi_mf_step_advance": (_i32, [ctypes.POINTER(MfStep), _i32, _i64]),
    "kivi_mf_step_upload": (_i32, [ctypes.POINTER(MfStep), _vp, _vp]),
    "kivi_gemv_awq": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i64, _vp]),
    "kivi_gemv_k_num_variants": (_i32, []),
    "kivi_gemv_k_variant_name": (ctypes.c_char_p, [_i32]),
    "kivi_gemv_k_variant": (_i32, [_i32] + _GEMV_ARGS + [_i32, _i32, _i32, _i32, _i64, _i32, _i32, _vp]),
    "kivi_gemv_v_num_variants": (_i32, []),
    "kivi_gemv_v_variant_name": (ctypes.c_char_p, [_i32]),
    "kivi_gemv_v_variant": (_i32, [_i32] + _GEMV_ARGS + [_i32, _i32, _i32, _i64, _i32, _i32, _i32, _vp]),
    "kivi_event_create": (_vp, []),
    "kivi_event_destroy": (None, [_vp]),
    "kivi_set_launch_events": (None, [_vp, _vp]),
    "kivi_event_elapsed_us": (ctypes.c_float, [_vp, _vp]),
    "kivi_last_timed_kernel": (ctypes.c_char_p, []),
    "kivi_debug_set_stamps": (None, [_vp]),
}

_lib = None


class KiviHipError(RuntimeError):
    rc = None


ABI_VERSION = 3      # include/kivi_hip.h: KIVI_ABI_VERSION (3: range words with an explicit byte 2, ticket ids always, device error word)


class KiviTimeout(KiviHipError):
    """KIVI_ETIMEOUT: a block of an EARLIER sliced launch gave up waiting for a partner (that step's output holds NaN for the unit);
    the error is cleared by being reported, the next call runs normally."""


class KiviUnsupported(KiviHipError):
    """KIVI_EUNSUPPORTED: valid request, no tuned kernel for this shape (callers may compose the unfused ops)."""


def load() -> ctypes.CDLL:
    """Load the library once; raise (never fall back) if it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KiviHipError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m kivi_amd.build). "
            "kivi_amd has no CPU / PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.kivi_abi_version() != ABI_VERSION:
        raise KiviHipError(f"ABI version mismatch: library reports {lib.kivi_abi_version()}, binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().kivi_last_error().decode(errors="replace")
        err = {-3: KiviUnsupported, -4: KiviTimeout}.get(rc, KiviHipError)(f"{what} failed (rc={rc}): {msg}")
        err.rc = rc
        raise err


def stream_ptr(t: torch.Tensor) -> ctypes.c_void_p:
    """hipStream_t of torch's current stream on the tensor's device."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise KiviHipError(f"{name} must live on the GPU (got device={t.device}); kivi_amd has no CPU path")


def ptr(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr())
