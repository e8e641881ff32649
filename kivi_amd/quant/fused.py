"""The three fused launches of one KIVI decode step (include/kivi_hip.h, "fused decode step").

Not part of the reference's Python surface: the reference composes the same arithmetic from ~20 torch / Triton /
CUDA kernels per layer (models/llama_kivi.py:314-399).  kivi_amd.attention uses these when a tuned kernel covers
the shape and falls back to the reference-style composition (quant.matmul + torch ops) otherwise.
"""
from __future__ import annotations

import ctypes

import torch

from .. import _lib
from . import matmul as _matmul


def decode_scores(layer, query_states: torch.Tensor, key_states: torch.Tensor, scores: torch.Tensor) -> None:
    """scores[..., :kv_len] <- [fused qK^T over the packed K pages | q . (fp16 residual keys + new key)], and the new
    key is appended to layer.k_res (the caller bumps layer.k_res_len).  query (B, nh, 1, D), key (B, nh_kv, 1, D)."""
    cfg = layer.cfg
    B, nh, _, D = query_states.shape
    q = query_states if query_states.stride(3) == 1 else query_states.contiguous()
    k = key_states if key_states.stride(3) == 1 else key_states.contiguous()
    kc, ks, km, kr = layer.k_code, layer.k_scale, layer.k_mn, layer.k_res
    lib = _lib.load()
    hook = _matmul.launch_hook
    if hook is not None and layer.k_quant_len:
        hook("pre", "k", dict(B=B, nh=nh, nh_kv=layer.nh_kv, K=D, N=layer.k_quant_len, bits=cfg.k_bits,
                              group_size=cfg.group_size))
    _lib.check(lib.kivi_decode_scores(
        layer.page_tokens, kc.stride(2), ks.stride(2),
        _lib.ptr(q), q.stride(0), q.stride(1),
        _lib.ptr(kc), kc.stride(0), kc.stride(1), kc.stride(3),
        _lib.ptr(ks), _lib.ptr(km), ks.stride(0), ks.stride(1), ks.stride(3),
        _lib.ptr(kr), kr.stride(0), kr.stride(1), kr.stride(2),
        _lib.ptr(k), k.stride(0), k.stride(1), layer.k_res_len,
        _lib.ptr(scores), scores.stride(0), scores.stride(1),
        B, nh, layer.nh_kv, D, layer.k_quant_len, cfg.group_size, cfg.k_bits, _lib.stream_ptr(q)), "kivi_decode_scores")


def softmax_scaled(scores: torch.Tensor, probs: torch.Tensor, n: int, inv_scale: float, mask: torch.Tensor = None) -> None:
    """probs[..., :n] <- softmax_fp32(fp16(scores[..., :n] * inv_scale) (+ mask)) as fp16.  scores / probs are
    (B, nh, 1, pitch) buffers; mask is the reference's additive (B, 1, 1, n) fp16 mask or None."""
    B, nh = scores.shape[0], scores.shape[1]
    assert scores.is_contiguous() and probs.is_contiguous() and scores.dtype == probs.dtype == torch.float16
    lib = _lib.load()
    if mask is not None:
        assert mask.shape == (B, 1, 1, n) and mask.dtype == torch.float16 and mask.stride(3) == 1
    _lib.check(lib.kivi_softmax_scaled(_lib.ptr(scores), _lib.ptr(probs), B * nh, n, scores.stride(1), probs.stride(1),
                                       float(inv_scale), _lib.ptr(mask) if mask is not None else None,
                                       mask.stride(0) if mask is not None else 0, nh, _lib.stream_ptr(scores)),
               "kivi_softmax_scaled")


def decode_output(layer, probs: torch.Tensor, value_states: torch.Tensor, out: torch.Tensor,
                  softmax_inv_scale: float = None, mask: torch.Tensor = None) -> bool:
    """out (B, nh, 1, D) <- fused sV over the packed V + probs[..., Tv:] @ [fp16 V window | new value]; the new value is
    appended to the window and, when the window then exceeds R tokens, its oldest token is quantised into the cache.
    Returns True if that flush happened (the caller updates the lengths).
    With `softmax_inv_scale`, `probs` holds the PRE-softmax scores and scale + mask + softmax run inside the same launch."""
    cfg = layer.cfg
    B, nh = probs.shape[0], probs.shape[1]
    v = value_states if value_states.stride(3) == 1 else value_states.contiguous()
    vc, vs, vm, vr = layer.v_code, layer.v_scale, layer.v_mn, layer.v_res
    flush = layer.v_res_len + 1 > cfg.residual_length
    lib = _lib.load()
    if softmax_inv_scale is not None:
        if mask is not None:
            assert mask.dtype == torch.float16 and mask.stride(3) == 1
        _lib.check(lib.kivi_decode_softmax_output(
            _lib.ptr(probs), probs.stride(0), probs.stride(1), float(softmax_inv_scale),
            _lib.ptr(mask) if mask is not None else None, mask.stride(0) if mask is not None else 0,
            _lib.ptr(vc), vc.stride(0), vc.stride(1), vc.stride(2),
            _lib.ptr(vs), _lib.ptr(vm), vs.stride(0), vs.stride(1), vs.stride(2),
            _lib.ptr(vr), vr.stride(0), vr.stride(1), vr.stride(2), layer.v_res_start, layer.v_res_len,
            _lib.ptr(v), v.stride(0), v.stride(1), int(flush),
            _lib.ptr(out), out.stride(0), out.stride(1),
            B, nh, layer.nh_kv, layer.v_quant_len, layer.D, cfg.group_size, cfg.v_bits, _lib.stream_ptr(probs)),
            "kivi_decode_softmax_output")
        return flush
    _lib.check(lib.kivi_decode_output(
        _lib.ptr(probs), probs.stride(0), probs.stride(1),
        _lib.ptr(vc), vc.stride(0), vc.stride(1), vc.stride(2),
        _lib.ptr(vs), _lib.ptr(vm), vs.stride(0), vs.stride(1), vs.stride(2),
        _lib.ptr(vr), vr.stride(0), vr.stride(1), vr.stride(2), layer.v_res_start, layer.v_res_len,
        _lib.ptr(v), v.stride(0), v.stride(1), int(flush),
        _lib.ptr(out), out.stride(0), out.stride(1),
        B, nh, layer.nh_kv, layer.v_quant_len, layer.D, cfg.group_size, cfg.v_bits, _lib.stream_ptr(probs)),
        "kivi_decode_output")
    return flush


_WS = {}


def _workspace(device, rows: int, head_dim: int) -> torch.Tensor:
    """Zero-initialised scratch shared by every layer on a device (launches are stream-ordered), as
    kivi_decode_attend documents it: 64 KiB of arrival counters, chunk statistics of the row softmax (up to 64 chunks
    per row), fp32 partial outputs for up to 64 + 1 blocks per row (split-T)."""
    need = 65536 + 4096 + rows * 64 * 8 + 4 * rows * head_dim * 65
    ws = _WS.get(device)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(need, dtype=torch.uint8, device=device)
        _WS[device] = ws
    return ws


def decode_attend(layer, query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor,
                  scores: torch.Tensor, out: torch.Tensor, inv_scale: float, mask: torch.Tensor = None) -> bool:
    """Everything after the packed qK^T GEMV in one launch (kivi_decode_attend): residual scores + K append,
    scale + mask + softmax, packed sV + V window + V append (+ flush of the oldest window token).
    `scores[..., :Tq]` must already hold the packed part.  Returns True if the V flush happened."""
    cfg = layer.cfg
    B, nh, _, D = query_states.shape
    q = query_states if query_states.stride(3) == 1 else query_states.contiguous()
    k = key_states if key_states.stride(3) == 1 else key_states.contiguous()
    v = value_states if value_states.stride(3) == 1 else value_states.contiguous()
    vc, vs, vm, vr, kr = layer.v_code, layer.v_scale, layer.v_mn, layer.v_res, layer.k_res
    flush = layer.v_res_len + 1 > cfg.residual_length
    if mask is not None:
        assert mask.dtype == torch.float16 and mask.stride(3) == 1
    ws = _workspace(q.device, B * nh, D)
    a = _lib.DecodeAttendArgs(
        q=q.data_ptr(), q_sb=q.stride(0), q_sh=q.stride(1),
        kres=kr.data_ptr(), kres_sb=kr.stride(0), kres_sh=kr.stride(1), kres_st=kr.stride(2),
        knew=k.data_ptr(), knew_sb=k.stride(0), knew_sh=k.stride(1), k_res_len=layer.k_res_len,
        scores=scores.data_ptr(), s_sb=scores.stride(0), s_sh=scores.stride(1),
        inv_scale=float(inv_scale), mask=mask.data_ptr() if mask is not None else None,
        mask_sb=mask.stride(0) if mask is not None else 0,
        v_code=vc.data_ptr(), vc_sb=vc.stride(0), vc_sh=vc.stride(1), vc_sr=vc.stride(2),
        v_scale=vs.data_ptr(), v_mn=vm.data_ptr(), vs_sb=vs.stride(0), vs_sh=vs.stride(1), vs_sr=vs.stride(2),
        vres=vr.data_ptr(), vres_sb=vr.stride(0), vres_sh=vr.stride(1), vres_st=vr.stride(2),
        v_win_start=layer.v_res_start, v_res_len=layer.v_res_len,
        vnew=v.data_ptr(), vnew_sb=v.stride(0), vnew_sh=v.stride(1), v_flush=int(flush),
        out=out.data_ptr(), out_sb=out.stride(0), out_sh=out.stride(1),
        B=B, nh=nh, nh_kv=layer.nh_kv, D=D, group_size=cfg.group_size, v_bits=cfg.v_bits,
        Tq=layer.k_quant_len, Tv=layer.v_quant_len,
        workspace=ws.data_ptr(), workspace_bytes=ws.numel() * ws.element_size())
    lib = _lib.load()
    _lib.check(lib.kivi_decode_attend(ctypes.byref(a), _lib.stream_ptr(q)), "kivi_decode_attend")
    return flush
