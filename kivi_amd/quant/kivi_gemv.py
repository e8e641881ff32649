"""Stand-in for the reference's native module `kivi_gemv` (quant/csrc/pybind.cpp:5-8).

Same two callables, same tensor layouts; implemented over the C ABI.
"""
from __future__ import annotations

import torch

from .. import _lib


def gemv_forward_cuda_outer_dim(in_feats: torch.Tensor, kernel: torch.Tensor, scaling_factors: torch.Tensor,
                                zeros: torch.Tensor, bit: int, group_size: int, nh: int, nh_kv: int) -> torch.Tensor:
    """Reference gemv_cuda.cu:511-557 on its kernel-input layout:
    in_feats (BS, 1, IC) fp16, kernel (BS_kv, OC // fpi, IC) int32, scaling_factors / zeros (BS_kv, OC // g, IC) fp16
    -> (BS, 1, OC) fp16 with OC = zeros.size(1) * group_size (:524)."""
    for t, n in ((in_feats, "in_feats"), (kernel, "kernel"), (scaling_factors, "scaling_factors"), (zeros, "zeros")):
        _lib.require_gpu(t, n)
    if in_feats.dtype != torch.float16 or scaling_factors.dtype != torch.float16 or zeros.dtype != torch.float16:
        raise TypeError("in_feats, scaling_factors and zeros must be float16")
    if kernel.dtype != torch.int32:
        raise TypeError("kernel must be int32")
    nh, nh_kv = int(nh), int(nh_kv)  # the reference's stale tests pass bools (quant/gemv.py:117); nh_kv=0 is rejected below
    BS, M, IC = in_feats.shape
    if M != 1:
        raise NotImplementedError("the reference kernel is only correct for M == 1 (gemv_cuda.cu:354-360)")
    OC = zeros.shape[1] * group_size
    x, w = in_feats.contiguous(), kernel.contiguous()
    s, z = scaling_factors.contiguous(), zeros.contiguous()
    out = torch.empty((BS, M, OC), dtype=torch.float16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.kivi_gemv_outer_dim(_lib.ptr(x), _lib.ptr(w), _lib.ptr(s), _lib.ptr(z), _lib.ptr(out), BS, IC, OC,
                                       bit, group_size, nh, nh_kv, _lib.stream_ptr(x)), "kivi_gemv_outer_dim")
    return out


def gemv_forward_cuda(in_feats: torch.Tensor, kernel: torch.Tensor, scaling_factors: torch.Tensor, zeros: torch.Tensor,
                      bit: int, group_size: int) -> torch.Tensor:
    """Reference gemv_cuda.cu:201-246: legacy AWQ-style INNER-dim 4-bit GEMV (g64 / g128).
    in_feats (B, IC) fp16, kernel (OC, IC // 8) int32 packed along IC, scaling_factors / zeros (OC, >= IC // g) fp16
    ("zeros" is the fp16 group minimum, quant/gemv.py:188) -> (B, OC) fp16.  Only the reference's disabled test
    scripts call it; kept for surface parity."""
    for t, n in ((in_feats, "in_feats"), (kernel, "kernel"), (scaling_factors, "scaling_factors"), (zeros, "zeros")):
        _lib.require_gpu(t, n)
    if in_feats.dtype != torch.float16 or scaling_factors.dtype != torch.float16 or zeros.dtype != torch.float16:
        raise TypeError("in_feats, scaling_factors and zeros must be float16")
    if kernel.dtype != torch.int32:
        raise TypeError("kernel must be int32")
    B, IC = in_feats.shape
    OC = kernel.shape[0]
    x, w = in_feats.contiguous(), kernel.contiguous()
    s = scaling_factors if scaling_factors.stride(1) == 1 else scaling_factors.contiguous()
    z = zeros if (zeros.stride() == s.stride()) else zeros.contiguous()
    if z.stride() != s.stride():
        s = s.contiguous()
    out = torch.empty((B, OC), dtype=torch.float16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.kivi_gemv_awq(_lib.ptr(x), _lib.ptr(w), _lib.ptr(s), _lib.ptr(z), _lib.ptr(out), B, IC, OC, bit,
                                 group_size, s.stride(0), _lib.stream_ptr(x)), "kivi_gemv_awq")
    return out
