"""ctypes front-end of the CPU oracle (oracle/kivi_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under kivi_amd/ may import this module.

The functions mirror the reference's Python API (quant/new_pack.py,
quant/matmul.py) on CPU torch tensors so the parity tests read like the
reference's own scripts; all arithmetic happens in the C restatement.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkivi_oracle.so")

NAN_CUDA = 0  # NaN code -> 0   (reference CUDA path; what the HIP kernels do)
NAN_CPU = 1   # NaN code -> INT_MIN (reference functions executed on x86)


def build(force: bool = False) -> str:
    """Compile oracle/kivi_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "kivi_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "libkivi_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        i64, i32, vp = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p
        L.kivi_oracle_h2f.restype = ctypes.c_float
        L.kivi_oracle_h2f.argtypes = [ctypes.c_uint16]
        L.kivi_oracle_f2h.restype = ctypes.c_uint16
        L.kivi_oracle_f2h.argtypes = [ctypes.c_float]
        L.kivi_oracle_quant_pack_lastdim.argtypes = [vp, i64, i64, i32, i32, i32, vp, vp, vp]
        L.kivi_oracle_quant_pack_kcache.argtypes = [vp, i64, i64, i64, i32, i32, i32, vp, vp, vp]
        L.kivi_oracle_pack_tensor.argtypes = [vp, i64, i64, i64, i32, vp]
        L.kivi_oracle_unpack_tensor.argtypes = [vp, i64, i64, i64, i32, vp]
        L.kivi_oracle_unpack_dequant_lastdim.argtypes = [vp, vp, vp, i64, i64, i32, i32, vp]
        L.kivi_oracle_unpack_dequant_kcache.argtypes = [vp, vp, vp, i64, i64, i64, i32, i32, vp]
        L.kivi_oracle_gemv_outer_dim.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32, i32]
        L.kivi_oracle_bmm_fA_qB_outer.argtypes = [vp, i64, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32,
                                                  i32, i32]
        L.kivi_oracle_fakequant_bmm.argtypes = [vp, i64, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32]
        L.kivi_oracle_gemv_awq.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64, i32, i64, i32]
        _lib = L
    return _lib


def _c(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    assert t.device.type == "cpu", "the oracle runs on CPU tensors"
    assert t.dtype == dtype, f"expected {dtype}, got {t.dtype}"
    return t.contiguous()


def _p(t: torch.Tensor) -> ctypes.c_void_p:
    return ctypes.c_void_p(t.data_ptr())


def _chk(rc: int, what: str) -> None:
    if rc != 0:
        raise ValueError(f"{what}: oracle returned {rc}")


# ------------------------------------------------------------------ pack

def quantize_and_pack_along_last_dim(data: torch.Tensor, group_size: int, bit: int,
                                     nan_mode: int = NAN_CUDA):
    """new_pack.py:217-252. data (B, nh, D, T) fp16 -> code (B,nh,D,T/fpi) int32,
    scale, mn (B,nh,D,T/g) fp16."""
    assert data.dim() == 4
    B, nh, D, T = data.shape
    x = _c(data, torch.float16)
    fpi = 32 // bit
    code = torch.empty((B, nh, D, T // fpi), dtype=torch.int32)
    scale = torch.empty((B, nh, D, T // group_size), dtype=torch.float16)
    mn = torch.empty_like(scale)
    _chk(lib().kivi_oracle_quant_pack_lastdim(_p(x), B * nh * D, T, group_size, bit, nan_mode,
                                              _p(code), _p(scale), _p(mn)), "quant_pack_lastdim")
    return code, scale, mn


def quant_and_pack_vcache(v: torch.Tensor, group_size: int, bits: int, nan_mode: int = NAN_CUDA):
    """new_pack.py:30-48: same as above with scale/mn keepdim (…, ng, 1)."""
    code, scale, mn = quantize_and_pack_along_last_dim(v, group_size, bits, nan_mode)
    return code, scale.unsqueeze(-1), mn.unsqueeze(-1)


def quant_and_pack_kcache(k: torch.Tensor, group_size: int, bits: int, nan_mode: int = NAN_CUDA):
    """new_pack.py:8-27. k (B,nh,T,D) -> code (B,nh,T/fpi,D), scale/mn (B,nh,T/g,1,D)."""
    assert k.dim() == 4
    B, nh, T, D = k.shape
    x = _c(k, torch.float16)
    fpi = 32 // bits
    code = torch.empty((B, nh, T // fpi, D), dtype=torch.int32)
    scale = torch.empty((B, nh, T // group_size, 1, D), dtype=torch.float16)
    mn = torch.empty_like(scale)
    _chk(lib().kivi_oracle_quant_pack_kcache(_p(x), B * nh, T, D, group_size, bits, nan_mode,
                                             _p(code), _p(scale), _p(mn)), "quant_pack_kcache")
    return code, scale, mn


def pack_tensor(data: torch.Tensor, bits: int, pack_dim: int) -> torch.Tensor:
    """new_pack.py:86-107 for any pack_dim of an int32 tensor."""
    x = _c(data, torch.int32)
    shape = list(x.shape)
    outer = int(np.prod(shape[:pack_dim], dtype=np.int64))
    inner = int(np.prod(shape[pack_dim + 1:], dtype=np.int64))
    n = shape[pack_dim]
    fpi = 32 // bits
    out_shape = shape[:pack_dim] + [n // fpi] + shape[pack_dim + 1:]
    code = torch.empty(out_shape, dtype=torch.int32)
    _chk(lib().kivi_oracle_pack_tensor(_p(x), outer, n, inner, bits, _p(code)), "pack_tensor")
    return code


def unpack_tensor(code: torch.Tensor, bits: int, pack_dim: int) -> torch.Tensor:
    """new_pack.py:110-129 -> int16."""
    x = _c(code, torch.int32)
    shape = list(x.shape)
    outer = int(np.prod(shape[:pack_dim], dtype=np.int64))
    inner = int(np.prod(shape[pack_dim + 1:], dtype=np.int64))
    nw = shape[pack_dim]
    fpi = 32 // bits
    out = torch.empty(shape[:pack_dim] + [nw * fpi] + shape[pack_dim + 1:], dtype=torch.int16)
    _chk(lib().kivi_oracle_unpack_tensor(_p(x), outer, nw, inner, bits, _p(out)), "unpack_tensor")
    return out


def unpack_and_dequant_vcache(v_code, scale, mn, group_size: int, bits: int) -> torch.Tensor:
    """new_pack.py:69-83. scale/mn may carry the trailing keepdim axis."""
    code = _c(v_code, torch.int32)
    rows = int(np.prod(code.shape[:-1], dtype=np.int64))
    T = code.shape[-1] * (32 // bits)
    s = _c(scale.reshape(rows, T // group_size), torch.float16)
    m = _c(mn.reshape(rows, T // group_size), torch.float16)
    out = torch.empty(tuple(code.shape[:-1]) + (T,), dtype=torch.float16)
    _chk(lib().kivi_oracle_unpack_dequant_lastdim(_p(code), _p(s), _p(m), rows, T, group_size, bits, _p(out)),
         "unpack_dequant_lastdim")
    return out


def unpack_and_dequant_kcache(k_code, scale, mn, group_size: int, bits: int) -> torch.Tensor:
    """new_pack.py:51-66. code (B,nh,T/fpi,D), scale/mn (B,nh,T/g,1,D) -> (B,nh,T,D)."""
    code = _c(k_code, torch.int32)
    B, nh, nw, D = code.shape
    T = nw * (32 // bits)
    s = _c(scale.reshape(B * nh, T // group_size, D), torch.float16)
    m = _c(mn.reshape(B * nh, T // group_size, D), torch.float16)
    out = torch.empty((B, nh, T, D), dtype=torch.float16)
    _chk(lib().kivi_oracle_unpack_dequant_kcache(_p(code), _p(s), _p(m), B * nh, T, D, group_size, bits,
                                                 _p(out)), "unpack_dequant_kcache")
    return out


# ------------------------------------------------------------------ GEMV

def gemv_forward_outer_dim(in_feats, kernel, scaling_factors, zeros, bit: int, group_size: int,
                           nh: int, nh_kv: int, use_fma: bool = True) -> torch.Tensor:
    """gemv_cuda.cu:511-557 on the reference kernel-input layout."""
    x = _c(in_feats, torch.float16)
    BS, M, IC = x.shape
    assert M == 1, "the reference kernel is only correct for M == 1 (gemv_cuda.cu:354-360)"
    w = _c(kernel, torch.int32)
    s = _c(scaling_factors, torch.float16)
    z = _c(zeros, torch.float16)
    OC = z.shape[1] * group_size  # :524
    out = torch.empty((BS, M, OC), dtype=torch.float16)
    _chk(lib().kivi_oracle_gemv_outer_dim(_p(x), _p(w), _p(s), _p(z), _p(out), BS, IC, OC, bit, group_size,
                                          nh, nh_kv, int(use_fma)), "gemv_outer_dim")
    return out


def bmm_fA_qB_outer(group_size: int, fA, qB, scales, zeros, bits: int, use_fma: bool = True,
                    fakequant: bool = False) -> torch.Tensor:
    """cuda_bmm_fA_qB_outer (matmul.py:178-219) in hook-state coordinates.
    fA (B,nh,1,K) (last-dim-contiguous rows, any row stride), qB (B,nh_kv,K,N/fpi),
    scales/zeros (B,nh_kv,K,N/g) -> (B,nh,1,N) fp16."""
    assert fA.dim() == 4 and qB.dim() == 4
    B, nh, M, K = fA.shape
    assert M == 1
    nh_kv = qB.shape[1]
    fpi = 32 // bits
    N = qB.shape[-1] * fpi
    a = _c(fA.reshape(B * nh, K), torch.float16)
    w = _c(qB, torch.int32)
    s = _c(scales, torch.float16)
    z = _c(zeros, torch.float16)
    out = torch.empty((B, nh, 1, N), dtype=torch.float16)
    if fakequant:
        rc = lib().kivi_oracle_fakequant_bmm(_p(a), K, _p(w), _p(s), _p(z), _p(out), B * nh, K, N, bits,
                                             group_size, nh, nh_kv)
    else:
        rc = lib().kivi_oracle_bmm_fA_qB_outer(_p(a), K, _p(w), _p(s), _p(z), _p(out), B * nh, K, N, bits,
                                               group_size, nh, nh_kv, int(use_fma))
    _chk(rc, "bmm_fA_qB_outer")
    return out


def gemv_forward_awq(in_feats, kernel, scaling_factors, zeros, bit: int, group_size: int, use_fma: bool = True):
    """gemv_cuda.cu:201-246 (legacy inner-dim 4-bit GEMV)."""
    assert bit == 4
    x = _c(in_feats, torch.float16)
    B, IC = x.shape
    w = _c(kernel, torch.int32)
    OC = w.shape[0]
    s = _c(scaling_factors, torch.float16)
    z = _c(zeros, torch.float16)
    out = torch.empty((B, OC), dtype=torch.float16)
    _chk(lib().kivi_oracle_gemv_awq(_p(x), _p(w), _p(s), _p(z), _p(out), B, IC, OC, group_size, s.shape[1], int(use_fma)),
         "gemv_awq")
    return out
