"""Group-wise asymmetric 2/4/8-bit quantise+pack and unpack+dequant of the KV cache.

Same names, argument order and returned shapes/dtypes as the reference's
quant/new_pack.py; every function runs as hand-written HIP on the GPU through
the C ABI (include/kivi_hip.h).  The "triton_" prefix is kept for call-site
compatibility only -- there is no Triton here.

Layout facts (reference quant/new_pack.py:86-129): fpi = 32 // bits codes per
int32 word, element i of a word sits at bit bits*i (LSB first).
"""
from __future__ import annotations

import torch

from .. import _lib

__all__ = [
    "triton_quantize_and_pack_along_last_dim", "quantize_and_pack_k_tmajor",
    "quant_and_pack_kcache", "quant_and_pack_vcache",
    "unpack_and_dequant_kcache", "unpack_and_dequant_vcache",
    "pack_tensor", "unpack_tensor",
]


def _check_fp16(t: torch.Tensor, name: str) -> None:
    _lib.require_gpu(t, name)
    if t.dtype != torch.float16:
        # the reference extension reads data_ptr<at::Half> (gemv_cuda.cu:526-529); bf16 is not supported there either
        raise TypeError(f"{name} must be float16, got {t.dtype}")


def triton_quantize_and_pack_along_last_dim(data: torch.Tensor, group_size: int, bit: int):
    """Fused quantise + pack along the last dim (reference new_pack.py:217-252).

    data (B, nh, D, T) fp16 -> code (B, nh, D, T // fpi) int32, scale, mn (B, nh, D, T // group_size) fp16.
    One kernel: min/max, scale, sub, div, clamp, round and pack in a single pass.

    If `data` is a transposed view of a (B, nh, T, D) tensor (what the hook builds
    with `key_states.transpose(2, 3)`), the per-channel kernel reads it in place and
    the reference's `.contiguous()` copy is not needed.
    """
    assert data.dim() == 4
    _check_fp16(data, "data")
    B, nh, D, T = data.shape
    assert T % group_size == 0  # new_pack.py:222
    if data.stride(2) == 1 and data.stride(3) != 1 and D > 1:
        return quantize_and_pack_k_tmajor(data.transpose(2, 3), group_size, bit)
    x = data.contiguous()
    fpi = 32 // bit
    code = torch.empty((B, nh, D, T // fpi), dtype=torch.int32, device=x.device)
    scale = torch.empty((B, nh, D, T // group_size), dtype=torch.float16, device=x.device)
    mn = torch.empty_like(scale)
    lib = _lib.load()
    _lib.check(lib.kivi_quant_pack_lastdim(_lib.ptr(x), _lib.ptr(code), _lib.ptr(scale), _lib.ptr(mn),
                                           B * nh * D, T, group_size, bit, _lib.stream_ptr(x)),
               "kivi_quant_pack_lastdim")
    return code, scale, mn


def quantize_and_pack_k_tmajor(k: torch.Tensor, group_size: int, bits: int, out=None, token_offset: int = 0):
    """Per-channel K quantise + pack straight from the un-transposed k (B, nh, T, D).

    Equivalent to triton_quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), g, bits)
    (the call at models/llama_kivi.py:345 / :436) without materialising the transpose.
    Returns code (B, nh, D, T // fpi), scale, mn (B, nh, D, T // g).  With `out=(code, scale, mn)`
    the result is written in place at `token_offset` of pre-allocated (capacity-strided) buffers.
    """
    assert k.dim() == 4
    _check_fp16(k, "k")
    B, nh, T, D = k.shape
    assert T % group_size == 0
    if k.stride(3) != 1:
        k = k.contiguous()
    fpi = 32 // bits
    if out is None:
        code = torch.empty((B, nh, D, T // fpi), dtype=torch.int32, device=k.device)
        scale = torch.empty((B, nh, D, T // group_size), dtype=torch.float16, device=k.device)
        mn = torch.empty_like(scale)
        token_offset = 0
    else:
        code, scale, mn = out
        assert token_offset % group_size == 0
        assert code.stride(3) == 1 and scale.stride(3) == 1 and scale.stride() == mn.stride()
        assert code.shape[3] * fpi >= token_offset + T and scale.shape[3] * group_size >= token_offset + T
    lib = _lib.load()
    _lib.check(lib.kivi_quant_pack_k_tmajor(
        _lib.ptr(k), k.stride(0), k.stride(1), k.stride(2),
        _lib.ptr(code), code.stride(0), code.stride(1), code.stride(2), token_offset // fpi,
        _lib.ptr(scale), _lib.ptr(mn), scale.stride(0), scale.stride(1), scale.stride(2), token_offset // group_size,
        B, nh, T, D, group_size, bits, _lib.stream_ptr(k)), "kivi_quant_pack_k_tmajor")
    return code, scale, mn


def quant_and_pack_kcache(k: torch.Tensor, group_size: int, bits: int):
    """Reference new_pack.py:8-27: k (B, nh, T, D) -> code (B, nh, T // fpi, D), scale, mn (B, nh, T // g, 1, D)."""
    assert len(k.shape) == 4
    code_T, scale_T, mn_T = quantize_and_pack_k_tmajor(k, group_size, bits)
    code = code_T.transpose(2, 3).contiguous()
    scale = scale_T.transpose(2, 3).contiguous().unsqueeze(3)
    mn = mn_T.transpose(2, 3).contiguous().unsqueeze(3)
    return code, scale, mn


def quant_and_pack_vcache(v: torch.Tensor, group_size: int, bits: int):
    """Reference new_pack.py:30-48: v (B, nh, T, D) -> code (B, nh, T, D // fpi), scale, mn (B, nh, T, D // g, 1)."""
    assert len(v.shape) == 4
    assert v.shape[-1] % group_size == 0
    code, scale, mn = triton_quantize_and_pack_along_last_dim(v.contiguous(), group_size, bits)
    return code, scale.unsqueeze(-1), mn.unsqueeze(-1)


def _dequant_lastdim(code: torch.Tensor, scale: torch.Tensor, mn: torch.Tensor, group_size: int, bits: int):
    _lib.require_gpu(code, "code")
    assert bits in [2, 4, 8]
    code = code.contiguous()
    rows = code.numel() // code.shape[-1] if code.shape[-1] else 0
    T = code.shape[-1] * (32 // bits)
    scale = scale.reshape(rows, T // group_size).contiguous()
    mn = mn.reshape(rows, T // group_size).contiguous()
    _check_fp16(scale, "scale")
    _check_fp16(mn, "mn")
    out = torch.empty(tuple(code.shape[:-1]) + (T,), dtype=torch.float16, device=code.device)
    lib = _lib.load()
    _lib.check(lib.kivi_unpack_dequant_lastdim(_lib.ptr(code), _lib.ptr(scale), _lib.ptr(mn), _lib.ptr(out), rows, T,
                                               group_size, bits, _lib.stream_ptr(code)), "kivi_unpack_dequant_lastdim")
    return out


def unpack_and_dequant_vcache(v_code: torch.Tensor, scale: torch.Tensor, mn: torch.Tensor, group_size: int, bits: int):
    """Reference new_pack.py:69-83: fp16(fp16(fp16(q) * scale) + mn), groups along the last dim."""
    assert len(v_code.shape) == 4
    return _dequant_lastdim(v_code, scale, mn, group_size, bits)


def unpack_and_dequant_kcache(k_code: torch.Tensor, scale: torch.Tensor, mn: torch.Tensor, group_size: int, bits: int):
    """Reference new_pack.py:51-66: code (B, nh, T // fpi, D), scale/mn (B, nh, T // g, 1, D) -> (B, nh, T, D)."""
    assert len(k_code.shape) == 4
    B, nh, nw, D = k_code.shape
    ng = nw * (32 // bits) // group_size
    code_T = k_code.transpose(2, 3)
    scale_T = scale.reshape(B, nh, ng, D).transpose(2, 3)
    mn_T = mn.reshape(B, nh, ng, D).transpose(2, 3)
    return _dequant_lastdim(code_T, scale_T, mn_T, group_size, bits).transpose(2, 3).contiguous()


def pack_tensor(data: torch.Tensor, bits: int, pack_dim: int) -> torch.Tensor:
    """Reference new_pack.py:86-107: OR 32 // bits int32 codes into one int32 along `pack_dim`."""
    _lib.require_gpu(data, "data")
    assert bits in [2, 4, 8], "Only 2, 4, 8 bits are supported"
    if data.dtype != torch.int32:
        raise TypeError(f"data must be int32, got {data.dtype}")
    fpi = 32 // bits
    assert data.shape[pack_dim] % fpi == 0, "Dimension length must be divisible by number of features per int"
    last = data.dim() - 1
    x = data.transpose(pack_dim, last).contiguous()
    rows = x.numel() // x.shape[-1]
    code = torch.empty(tuple(x.shape[:-1]) + (x.shape[-1] // fpi,), dtype=torch.int32, device=x.device)
    lib = _lib.load()
    _lib.check(lib.kivi_pack_codes_lastdim(_lib.ptr(x), _lib.ptr(code), rows, x.shape[-1], bits, _lib.stream_ptr(x)),
               "kivi_pack_codes_lastdim")
    return code.transpose(pack_dim, last).contiguous()


def unpack_tensor(v_code: torch.Tensor, bits: int, pack_dim: int) -> torch.Tensor:
    """Reference new_pack.py:110-129: int16 codes; like the reference only pack_dim 2 and 3 of a 4-D tensor."""
    _lib.require_gpu(v_code, "v_code")
    assert bits in [2, 4, 8]
    if pack_dim not in (2, 3) or v_code.dim() != 4:
        raise NotImplementedError
    fpi = 32 // bits
    x = v_code.transpose(pack_dim, 3).contiguous()
    rows = x.numel() // x.shape[-1] if x.shape[-1] else 0
    out = torch.empty(tuple(x.shape[:-1]) + (x.shape[-1] * fpi,), dtype=torch.int16, device=x.device)
    lib = _lib.load()
    _lib.check(lib.kivi_unpack_codes_lastdim(_lib.ptr(x), _lib.ptr(out), rows, x.shape[-1] * fpi, bits,
                                             _lib.stream_ptr(x)), "kivi_unpack_codes_lastdim")
    return out.transpose(pack_dim, 3).contiguous()
