"""The KIVI attention hook: decode / prefill cache logic of the reference's
LlamaFlashAttention_KIVI.forward (models/llama_kivi.py:265-466) on the HIP kernels.

Same arithmetic sequence as the reference decode branch (:314-399):
  scores  = cat([ fused qK^T over packed K , q @ K_residual^T ]) / sqrt(D)      (fp16)
  weights = softmax(scores, fp32) -> fp16
  out     = fused sV over packed V  +  weights[..., -L:] @ V_residual           (fp16)
with the cache policy of cache.py.  What changed is data movement only: the fused GEMVs read the
hook-state layout directly and write into a scores buffer (no torch.cat, no transposed copies), and
the cache is appended in place.
"""
from __future__ import annotations

import ctypes
import math
import warnings
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _tuning
from ._lib import KiviUnsupported
from .cache import KiviCacheTuple, KiviConfig, KiviLayerCache
from .cache_mf import KiviLayerCacheMF, make_layer_cache, supported as _mf_supported
from .quant import fused
from .quant.matmul import cuda_bmm_fA_qB_outer, gemv_k_paged

__all__ = ["kivi_attention_decode", "kivi_attention_prefill", "LlamaAttention_KIVI", "LlamaFlashAttention_KIVI",
           "MistralAttention_KIVI", "MistralFlashAttention_KIVI",
           "KiviConfig", "KiviLayerCache", "KiviLayerCacheMF", "make_layer_cache"]


def _row_buffer(layer: KiviLayerCache, name: str, nh: int) -> torch.Tensor:
    """(B, nh, 1, pitch) per-layer fp16 scratch row buffer, pitch = a multiple of 8 halves >= capacity + 1
    (16-byte stores of the fused GEMV)."""
    pitch = ((layer.cap + 1 + 7) // 8) * 8
    buf = getattr(layer, name, None)
    if buf is None or buf.shape[1] != nh:
        buf = torch.empty((layer.B, nh, 1, pitch), dtype=torch.float16, device=layer.k_code.device)
        setattr(layer, name, buf)
    return buf


def _scores_buffer(layer: KiviLayerCache, nh: int, kv_len: int) -> torch.Tensor:
    return _row_buffer(layer, "_scores", nh)[..., :kv_len]


def _composed_output(attn_weights: torch.Tensor, value_states: torch.Tensor, layer: KiviLayerCache, nh: int) -> torch.Tensor:
    """Output over [quantised V prefix | fp16 V window], one launch per reference op (llama_kivi.py:377-399)."""
    cfg = layer.cfg
    B, nh_kv, D = layer.B, layer.nh_kv, layer.D
    rep = nh // nh_kv
    layer.append_v(value_states)
    Tv, Lv = layer.v_quant_len, layer.v_res_len
    v_full = layer.v_res_view()
    w_full = attn_weights[..., Tv:].reshape(B, nh_kv, rep, Lv)
    if Tv == 0:
        attn_output = torch.matmul(w_full, v_full).view(B, nh, 1, D)     # :380
    else:
        vc, vs, vm = layer.v_quant_views()
        attn_output = cuda_bmm_fA_qB_outer(cfg.group_size, attn_weights[..., :Tv], vc, vs, vm, cfg.v_bits)   # :382
        attn_output += torch.matmul(w_full, v_full).view(B, nh, 1, D)    # :384
    layer.maybe_flush_v()                                                 # :386-399
    return attn_output


_NATIVE_STEP = _tuning.knob("KIVI_NATIVE_STEP", "1") != "0"   # tuning sessions: 0 = the Python bookkeeping path
_FUSION_ENV = _tuning.knob("KIVI_DECODE_FUSION")   # tuning sessions: "attend" (2 launches), "softmax" (3), "separate" (4)


def _fusion_level(layer, nh: int, kv_len: int) -> int:
    """How much of the decode step goes into the sV launch: 2 = everything after the packed qK^T (kivi_decode_attend),
    1 = softmax + output, 0 = output only (softmax as its own launch).  Level 2 covers every tuned shape: for short
    MHA rows the block that owns a row does the row's softmax before it starts streaming; for grouped queries / long
    rows the library splits rows over blocks and adds a row-statistics launch (kivi_gemv_v.hip, v_run)."""
    if _FUSION_ENV:
        return {"attend": 2, "softmax": 1, "separate": 0}[_FUSION_ENV]
    return 2


class KiviPerformanceWarning(UserWarning):
    """A decode step left the fused kernels for a slower composition (results unchanged)."""


def _drop_fusion(layer, attr: str, what: str, why: str = "") -> None:
    """Mark `layer` as unable to use one fusion level and say so ONCE per layer and level: a shape without a tuned kernel otherwise
    runs the slower composition forever without a word."""
    if getattr(layer, attr, False):
        return
    setattr(layer, attr, True)
    if _FUSION_ENV:          # a tuning session asked for the lower level
        return
    c = layer.cfg
    warnings.warn(f"kivi_amd: {what} for this cache (k_bits={c.k_bits} v_bits={c.v_bits} group={c.group_size} "
                  f"residual={c.residual_length} head_dim={layer.D} kv_heads={layer.nh_kv}){': ' + why if why else ''} -- the step "
                  f"runs as more, slower launches from now on; results are unchanged", KiviPerformanceWarning, stacklevel=4)


def _matmul_mod():
    from .quant import matmul
    return matmul


def _native_desc(layer: KiviLayerCache, nh: int):
    """The kivi_layer_desc of this layer (built once: buffers and strides never change), its state array."""
    cached = getattr(layer, "_native", None)
    if cached is not None and cached[2] == nh:
        return cached
    from . import _lib
    from .quant.fused import _workspace
    cfg = layer.cfg
    scores = _row_buffer(layer, "_scores", nh)
    ws = _workspace(layer.k_code.device, layer.B * nh, layer.D)
    kc, ks, kr, vc, vs, vr = layer.k_code, layer.k_scale, layer.k_res, layer.v_code, layer.v_scale, layer.v_res
    d = _lib.LayerDesc(
        B=layer.B, nh_kv=layer.nh_kv, D=layer.D, k_bits=cfg.k_bits, v_bits=cfg.v_bits, group_size=cfg.group_size,
        residual_length=cfg.residual_length, inv_scale=1.0 / math.sqrt(layer.D),
        cap=layer.cap, page_tokens=layer.page_tokens, v_window_rows=vr.shape[2], s_pitch=scores.shape[3],
        k_code=kc.data_ptr(), kc_sb=kc.stride(0), kc_sh=kc.stride(1), kc_sp=kc.stride(2), kc_sr=kc.stride(3),
        k_scale=ks.data_ptr(), k_mn=layer.k_mn.data_ptr(), ks_sb=ks.stride(0), ks_sh=ks.stride(1), ks_sp=ks.stride(2),
        ks_sr=ks.stride(3),
        k_res=kr.data_ptr(), kr_sb=kr.stride(0), kr_sh=kr.stride(1), kr_st=kr.stride(2),
        v_code=vc.data_ptr(), vc_sb=vc.stride(0), vc_sh=vc.stride(1), vc_sr=vc.stride(2),
        v_scale=vs.data_ptr(), v_mn=layer.v_mn.data_ptr(), vs_sb=vs.stride(0), vs_sh=vs.stride(1), vs_sr=vs.stride(2),
        v_res=vr.data_ptr(), vr_sb=vr.stride(0), vr_sh=vr.stride(1), vr_st=vr.stride(2),
        scores=scores.data_ptr(), s_sb=scores.stride(0), s_sh=scores.stride(1),
        workspace=ws.data_ptr(), workspace_bytes=ws.numel() * ws.element_size())
    state = (ctypes.c_int64 * 6)()
    layer._native = (d, state, nh, _lib.load().kivi_decode_layer, ws)   # ws: keeps the shared workspace alive
    return layer._native


def _decode_native(query_states, key_states, value_states, layer: KiviLayerCache, attention_mask, out=None) -> torch.Tensor:
    """The whole step (both launches + cache bookkeeping + K flush) through ONE library call (kivi_decode_layer):
    the host side of a layer step drops from ~40 us of Python to one ctypes call.  Same launches, same results as
    _decode_fused; raises KiviUnsupported (state untouched) when no tuned kernel covers the shape."""
    from . import _lib
    B, nh, _, D = query_states.shape
    d, state, _, fn, _ = _native_desc(layer, nh)
    q = query_states if query_states.stride(3) == 1 else query_states.contiguous()
    k = key_states if key_states.stride(3) == 1 else key_states.contiguous()
    v = value_states if value_states.stride(3) == 1 else value_states.contiguous()
    kv_seq_len = layer.kv_seq_len + 1
    mask_ptr, mask_sb = None, 0
    if attention_mask is not None:
        if attention_mask.size() != (B, 1, 1, kv_seq_len):
            raise ValueError(f"Attention mask should be of size {(B, 1, 1, kv_seq_len)}, but is {attention_mask.size()}")
        assert attention_mask.dtype == torch.float16 and attention_mask.stride(3) == 1
        mask_ptr, mask_sb = attention_mask.data_ptr(), attention_mask.stride(0)
    state[0], state[1], state[2] = layer.k_quant_len, layer.k_res_len, layer.v_quant_len
    state[3], state[4], state[5] = layer.v_res_start, layer.v_res_len, layer.kv_seq_len
    if out is None:
        out = torch.empty((B, nh, 1, D), dtype=torch.float16, device=q.device)
    else:
        assert out.shape == (B, nh, 1, D) and out.dtype == torch.float16 and out.stride(3) == 1
    hook = _matmul_mod().launch_hook
    if hook is not None and layer.k_quant_len:   # bench.py: bracket the qK^T dispatch (the first launch of the call)
        hook("pre", "k", dict(B=B, nh=nh, nh_kv=layer.nh_kv, K=D, N=layer.k_quant_len, bits=layer.cfg.k_bits,
                              group_size=layer.cfg.group_size, v_bits=layer.cfg.v_bits, Tv=layer.v_quant_len,
                              k_res=layer.k_res_len + 1, v_res=layer.v_res_len + 1))
    rc = fn(ctypes.byref(d), state, q.data_ptr(), q.stride(0), q.stride(1), nh, k.data_ptr(), k.stride(0), k.stride(1),
            v.data_ptr(), v.stride(0), v.stride(1), mask_ptr, mask_sb, out.data_ptr(), out.stride(0), out.stride(1),
            torch.cuda.current_stream(q.device).cuda_stream)
    # the library writes `state` after every phase it has enqueued (a refused step leaves it untouched apart from a
    # completed window compaction), so the lengths are read back whether or not the call succeeded
    layer.k_quant_len, layer.k_res_len, layer.v_quant_len = state[0], state[1], state[2]
    layer.v_res_start, layer.v_res_len, layer.kv_seq_len = state[3], state[4], state[5]
    if rc:
        _lib.check(rc, "kivi_decode_layer")
    return out


def _decode_fused(query_states, key_states, value_states, layer: KiviLayerCache, attention_mask) -> torch.Tensor:
    """The decode step in two launches (three when only the separate softmax fits; +1 when the K residual fills up): same arithmetic and roundings as the
    composed path below.  Raises KiviUnsupported when no tuned kernel covers the shape."""
    cfg = layer.cfg
    B, nh, _, D = query_states.shape
    kv_seq_len = layer.kv_seq_len + 1
    scores = _row_buffer(layer, "_scores", nh)
    probs = _row_buffer(layer, "_probs", nh)
    if attention_mask is not None and attention_mask.size() != (B, 1, 1, kv_seq_len):
        raise ValueError(f"Attention mask should be of size {(B, 1, 1, kv_seq_len)}, but is {attention_mask.size()}")
    assert layer.k_quant_len + layer.k_res_len + 1 <= layer.cap, "cache capacity exceeded"
    if layer.v_res_start + layer.v_res_len + 1 > layer.v_res.shape[2]:     # make room in the window buffer
        layer.compact_v_window()
    out = torch.empty((B, nh, 1, D), dtype=torch.float16, device=query_states.device)
    inv = 1.0 / math.sqrt(D)
    flushed = None
    level = _fusion_level(layer, nh, kv_seq_len)
    if level < 2:
        _drop_fusion(layer, "_attend_unfusable", "fusion level lowered")
    if level < 1:
        _drop_fusion(layer, "_softmax_unfusable", "fusion level lowered")
    if not getattr(layer, "_attend_unfusable", False):
        # two launches: packed qK^T GEMV (:324), then residual scores + K append + softmax + output + V append/flush
        # only the launches that may be refused sit inside the try (they write scratch rows until the attend launch
        # runs); the bookkeeping follows once they have been enqueued
        try:
            if layer.k_quant_len:
                gemv_k_paged(cfg.group_size, query_states, layer.k_code, layer.k_scale, layer.k_mn, layer.k_quant_len,
                             cfg.k_bits, out=scores[..., : layer.k_quant_len])
            flushed = fused.decode_attend(layer, query_states, key_states, value_states, scores, out, inv, attention_mask)
        except KiviUnsupported as e:             # e.g. rows too long for the LDS: use the three-launch form below
            _drop_fusion(layer, "_attend_unfusable", "the fused attend launch (residual scores + softmax + output in one launch) is not available", str(e))
        else:
            layer.k_res_len += 1
            layer.maybe_flush_k()                                          # :343-356
    if flushed is None:
        fused.decode_scores(layer, query_states, key_states, scores)      # :323-337 (+ the K append of :333-336)
        layer.k_res_len += 1                     # committed: the launch above appended the key
        layer.maybe_flush_k()                                              # :343-356
        if not getattr(layer, "_softmax_unfusable", False):
            try:   # scale + mask + softmax (:339, :364-375) inside the sV launch (:377-399)
                flushed = fused.decode_output(layer, scores, value_states, out, softmax_inv_scale=inv, mask=attention_mask)
            except KiviUnsupported as e:         # keep the softmax as its own launch
                _drop_fusion(layer, "_softmax_unfusable", "the softmax cannot be folded into the sV launch", str(e))
    if flushed is None:
        fused.softmax_scaled(scores, probs, kv_seq_len, inv, attention_mask)
        try:
            flushed = fused.decode_output(layer, probs, value_states, out)
        except KiviUnsupported:   # no tuned sV kernel for this head_dim / group: compose the output part
            out = _composed_output(probs[..., :kv_seq_len], value_states, layer, nh)
            layer.kv_seq_len = kv_seq_len
            return out
    layer.v_res_len += 1
    if flushed:
        layer.v_quant_len += 1
        layer.v_res_start += 1
        layer.v_res_len -= 1
    layer.kv_seq_len = kv_seq_len
    return out


def kivi_attention_decode(query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor,
                          layer: KiviLayerCache, attention_mask: Optional[torch.Tensor] = None,
                          fused_kernels: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One decode step for one layer.  query (B, nh, 1, D), key/value (B, nh_kv, 1, D), RoPE already applied.
    Mutates `layer` in place and returns attn_output (B, nh, 1, D) fp16 (before the o_proj transpose).
    `fused_kernels=False` forces the reference-style composition (one launch per reference op).
    `out`: optional preallocated (B, nh, 1, D) fp16 result buffer (static buffers of graph-captured callers)."""
    layer.ensure_room(1)     # the reference's tuple grows without bound; the in-place cache doubles when it is full
    if isinstance(layer, KiviLayerCacheMF):   # grouped queries on the matrix pipe: two launches for the whole step
        return layer.decode_step(query_states, key_states, value_states, attention_mask, out)
    res = _attention_decode(query_states, key_states, value_states, layer, attention_mask, fused_kernels, out)
    if out is not None and res is not out:
        out.copy_(res)
        return out
    return res


def _attention_decode(query_states, key_states, value_states, layer: KiviLayerCache, attention_mask, fused_kernels, out):
    cfg = layer.cfg
    B, nh, q_len, D = query_states.shape
    assert q_len == 1, "decode branch: one new token (the reference kernel is q_len == 1 only)"
    if fused_kernels and not getattr(layer, "_fused_unsupported", False):
        if (_NATIVE_STEP and _fusion_level(layer, nh, layer.kv_seq_len + 1) == 2
                and not getattr(layer, "_attend_unfusable", False)):
            try:
                return _decode_native(query_states, key_states, value_states, layer, attention_mask, out)
            except KiviUnsupported as e:         # the Python path below picks the next fusion level
                _drop_fusion(layer, "_attend_unfusable", "the one-call layer step (kivi_decode_layer) is not available", str(e))
        try:
            return _decode_fused(query_states, key_states, value_states, layer, attention_mask)
        except KiviUnsupported as e:          # shape without a tuned kernel: compose the unfused ops from now on
            _drop_fusion(layer, "_fused_unsupported", "no fused decode kernel covers the shape: composing the reference's op sequence "
                         "from the fused GEMVs and torch ops", str(e))
    nh_kv = layer.nh_kv
    rep = nh // nh_kv
    kv_seq_len = layer.kv_seq_len + 1                                    # llama_kivi.py:307-309
    g = cfg.group_size

    # ---- scores over [quantised K prefix | fp16 K residual]  (:323-341)
    Tq = layer.k_quant_len
    scores = _scores_buffer(layer, nh, kv_seq_len)
    if Tq:   # :324, reading the K pages in place and writing straight into the scores buffer (scratch: may still raise)
        gemv_k_paged(g, query_states, layer.k_code, layer.k_scale, layer.k_mn, Tq, cfg.k_bits, out=scores[..., :Tq])
    layer.append_k(key_states)                                           # :333-336
    k_full = layer.k_res_view()                                          # (B, nh_kv, L, D)
    att_qkfull = torch.matmul(query_states.reshape(B, nh_kv, rep, D), k_full.transpose(2, 3))  # :337 (repeat_kv folded)
    scores[..., Tq:].copy_(att_qkfull.view(B, nh, 1, -1))
    attn_weights = scores / math.sqrt(D)                                  # :339, fp16 division like the reference
    layer.maybe_flush_k()                                                 # :343-356

    if attn_weights.size() != (B, nh, 1, kv_seq_len):
        raise ValueError(f"Attention weights should be of size {(B, nh, 1, kv_seq_len)}, but is {attn_weights.size()}")
    if attention_mask is not None:                                        # :364-372
        if attention_mask.size() != (B, 1, 1, kv_seq_len):
            raise ValueError(f"Attention mask should be of size {(B, 1, 1, kv_seq_len)}, but is {attention_mask.size()}")
        attn_weights = attn_weights + attention_mask
        attn_weights = torch.max(attn_weights, torch.tensor(torch.finfo(attn_weights.dtype).min, device=attn_weights.device))
    attn_weights = F.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query_states.dtype)   # :375

    attn_output = _composed_output(attn_weights, value_states, layer, nh)    # :377-399
    layer.kv_seq_len = kv_seq_len
    return attn_output


def kivi_attention_prefill(query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor,
                           layer: KiviLayerCache, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Prompt pass (:401-452): attention over fp16 q/k/v (torch SDPA, outside the quantised hot path), then split K/V
    into the quantised prefix and the fp16 residual.

    `attention_mask` None: causal -- the flash class, whose mask argument is hard-wired to None (:420-423).  Given
    (additive, (bsz, 1, q_len, kv_len), causal structure included, as HF builds it): the eager class's
    `attn_weights + attention_mask` (:228-237), e.g. for left-padded batches; a wrong shape raises like the reference."""
    B, nh, T, D = query_states.shape
    rep = nh // layer.nh_kv
    k, v = key_states, value_states
    if rep > 1:
        k = k.repeat_interleave(rep, dim=1)
        v = v.repeat_interleave(rep, dim=1)
    if attention_mask is None:
        attn_output = F.scaled_dot_product_attention(query_states, k, v, is_causal=True)
    else:
        if tuple(attention_mask.shape) != (B, 1, T, key_states.shape[-2]):
            raise ValueError(f"Attention mask should be of size {(B, 1, T, key_states.shape[-2])}, but is "
                             f"{tuple(attention_mask.shape)}")
        attn_output = F.scaled_dot_product_attention(query_states, k, v, attn_mask=attention_mask.to(query_states.dtype))
    layer.prefill(key_states, value_states)
    return attn_output


# --------------------------------------------------------------------------- module hook

def _rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def _rope_inv_freq(config, head_dim: int, theta: float) -> torch.Tensor:
    """Rotary frequencies incl. config.json's rope_scaling (the reference delegates RoPE to HF's rotary_emb,
    llama_kivi.py:52, which honours it): default, "linear" and Llama-3.1's "llama3" are implemented; anything else
    raises instead of decoding with silently wrong positions."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    rs = getattr(config, "rope_scaling", None)
    if not rs:
        return inv_freq
    rs = dict(rs) if not isinstance(rs, dict) else rs
    kind = rs.get("rope_type", rs.get("type", "default"))
    if kind == "default":
        return inv_freq
    if kind == "linear":
        return inv_freq / float(rs["factor"])
    if kind == "llama3":
        factor, lo, hi = float(rs["factor"]), float(rs["low_freq_factor"]), float(rs["high_freq_factor"])
        old = float(rs["original_max_position_embeddings"])
        wavelen = 2 * math.pi / inv_freq
        scaled = torch.where(wavelen > old / lo, inv_freq / factor, inv_freq)
        smooth = (old / wavelen - lo) / (hi - lo)
        medium = (wavelen >= old / hi) & (wavelen <= old / lo)
        return torch.where(medium, (1 - smooth) * scaled / factor + smooth * scaled, scaled)
    raise NotImplementedError(f"rope_scaling type {kind!r} is not implemented (default / linear / llama3 are)")


_ROPE_CACHE = {}   # (device, dtype, head_dim, theta, rope_scaling, past_len, q_len) -> (cos, sin) of the current step


class LlamaAttention_KIVI(nn.Module):
    """Self-contained Llama attention block with the KIVI cache hook (reference: LlamaFlashAttention_KIVI,
    models/llama_kivi.py:264-466; constructor fields :22-61).

    `config` needs hidden_size, num_attention_heads, num_key_value_heads, max_position_embeddings, rope_theta and
    the four KIVI fields the reference monkey-patches onto the HF config (README.md:72-75): k_bits, v_bits,
    group_size, residual_length.  forward() keeps the reference signature and returns
    (attn_output, None, past_key_value) with past_key_value the 9-tuple of :454-455.
    """

    # the eager class adds `attention_mask` to the prompt pass's scores (:228-237); the flash subclass does not (:420-423)
    _prefill_uses_mask = True

    def __init__(self, config, layer_idx: Optional[int] = None):
        super().__init__()
        self.config = config
        self.layer_idx = layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = getattr(config, "head_dim", None) or self.hidden_size // self.num_heads
        self.num_key_value_heads = getattr(config, "num_key_value_heads", self.num_heads)
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.max_position_embeddings = getattr(config, "max_position_embeddings", 4096)
        self.rope_theta = getattr(config, "rope_theta", 10000.0)
        self.kivi = KiviConfig(config.k_bits, config.v_bits, config.group_size, config.residual_length)
        self.k_bits, self.v_bits = config.k_bits, config.v_bits
        self.group_size, self.residual_length = config.group_size, config.residual_length
        if self.head_dim * self.num_heads != self.hidden_size and getattr(config, "head_dim", None) is None:
            raise ValueError("hidden_size must be divisible by num_heads")
        bias = getattr(config, "attention_bias", False)
        self.q_proj = nn.Linear(self.hidden_size, self.num_heads * self.head_dim, bias=bias)
        self.k_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias)
        self.v_proj = nn.Linear(self.hidden_size, self.num_key_value_heads * self.head_dim, bias=bias)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=bias)
        self.register_buffer("inv_freq", _rope_inv_freq(config, self.head_dim, self.rope_theta), persistent=False)

    def _rope(self, q, k, position_ids):
        freqs = position_ids[:, :, None].float() * self.inv_freq[None, None, :].float()   # (B, T, D/2)
        emb = torch.cat((freqs, freqs), dim=-1)
        cos, sin = emb.cos()[:, None].to(q.dtype), emb.sin()[:, None].to(q.dtype)        # (B, 1, T, D)
        return q * cos + _rotate_half(q) * sin, k * cos + _rotate_half(k) * sin

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                position_ids: Optional[torch.LongTensor] = None, past_key_value=None, output_attentions: bool = False,
                use_cache: bool = False, **kwargs) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[tuple]]:
        bsz, q_len, _ = hidden_states.size()
        q = self.q_proj(hidden_states).view(bsz, q_len, self.num_heads, self.head_dim).transpose(1, 2)
        k = self.k_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        v = self.v_proj(hidden_states).view(bsz, q_len, self.num_key_value_heads, self.head_dim).transpose(1, 2)
        past_len = 0 if past_key_value is None else int(past_key_value[-1])
        if position_ids is None:   # consecutive positions: cos / sin are the same for every layer of this step
            key = (hidden_states.device, q.dtype, self.head_dim, self.rope_theta, repr(getattr(self.config, "rope_scaling", None)),
                   past_len, q_len)
            cs = _ROPE_CACHE.get(key)
            if cs is None:
                pos = torch.arange(past_len, past_len + q_len, device=hidden_states.device)[None]
                freqs = pos[:, :, None].float() * self.inv_freq[None, None, :].float()
                emb = torch.cat((freqs, freqs), dim=-1)
                cs = (emb.cos()[:, None].to(q.dtype), emb.sin()[:, None].to(q.dtype))
                _ROPE_CACHE.clear()
                _ROPE_CACHE[key] = cs
            cos, sin = cs
            q, k = q * cos + _rotate_half(q) * sin, k * cos + _rotate_half(k) * sin
        else:
            q, k = self._rope(q, k, position_ids)

        if past_key_value is not None:
            if isinstance(past_key_value, KiviCacheTuple):
                layer = past_key_value.layer
                if past_len != layer.kv_seq_len:
                    # the reference's tuple is an immutable snapshot; this one is a single-use handle on a cache that is
                    # appended in place -- replaying an older one would append twice and rotate at the wrong position
                    raise RuntimeError(
                        f"stale KIVI past_key_value: the tuple was issued at kv length {past_len}, its cache has since "
                        f"advanced to {layer.kv_seq_len}. In-place cache tuples are single-use; clone the cache "
                        f"(KiviLayerCache.clone()) to continue one prefix twice.")
            elif _mf_supported(self.kivi, self.head_dim, self.num_heads, self.num_key_value_heads):
                layer = KiviLayerCacheMF.from_tuple(self.kivi, past_key_value, self._capacity(past_len + 1), self.num_heads)
            else:  # a plain reference-style tuple: adopt it once
                layer = KiviLayerCache.from_tuple(self.kivi, past_key_value, self._capacity(past_len + 1))
            attn_output = kivi_attention_decode(q, k, v, layer, attention_mask)
        else:
            layer = make_layer_cache(self.kivi, bsz, self.num_key_value_heads, self.head_dim, self._capacity(q_len),
                                     hidden_states.device, q.dtype, num_heads=self.num_heads)
            attn_output = kivi_attention_prefill(q, k, v, layer, attention_mask if self._prefill_uses_mask else None)
        past = layer.as_tuple() if use_cache else None                                     # :454-455
        attn_output = attn_output.transpose(1, 2).reshape(bsz, q_len, self.num_heads * self.head_dim)
        return self.o_proj(attn_output), None, past

    def _capacity(self, needed: int) -> int:
        """Initial cache capacity: the prompt plus one residual window (the cache doubles when it runs out, like the
        reference's torch.cat-grown tuple, so memory follows the sequence; the matrix-pipe stores round it up to whole
        512-token super-blocks).  `config.kivi_max_cache_len` is an opt-in reservation for callers that know their final
        length; max_position_embeddings is never used (a 128k-context config would otherwise pre-allocate GBs per sequence
        for a 1k-token run)."""
        reserve = getattr(self.config, "kivi_max_cache_len", None) or 0
        return max(needed + self.residual_length, reserve)


class LlamaFlashAttention_KIVI(LlamaAttention_KIVI):
    """Reference: LlamaFlashAttention_KIVI (models/llama_kivi.py:264-466): same cache logic as the eager class; its prompt
    pass is causal flash attention with the mask argument hard-wired to None (:420-423), so `attention_mask` only
    reaches the decode steps."""
    _prefill_uses_mask = False


class MistralAttention_KIVI(LlamaAttention_KIVI):
    """Reference: MistralAttention_KIVI / MistralFlashAttention_KIVI (models/mistral_kivi.py:69-534).  Against the Llama
    hook the reference differs in three places, all reproduced or made unnecessary here:
      * projections never carry a bias (:96-99) -- `attention_bias` of the config is ignored;
      * grouped queries reach the fused GEMV through `repeat_kv_quant` copies of codes / scale / mn (:58-67, :381-385,
        :441-445, a 4x cache-sized copy per call at Mistral-7B's ratio); the kernels here map the nh / nh_kv query heads
        of a kv head themselves (gemv_cuda.cu:361-365 semantics), same results, no copy;
      * `config.sliding_window` exists but is never applied to the quantised history (:356-367 is commented out):
        the whole prefix stays attended.  The field is kept on the module for callers that inspect it."""

    def __init__(self, config, layer_idx: Optional[int] = None):
        if getattr(config, "attention_bias", False):
            from types import SimpleNamespace
            config = SimpleNamespace(**{**vars(config), "attention_bias": False})
        super().__init__(config, layer_idx)
        self.sliding_window = getattr(config, "sliding_window", None)


class MistralFlashAttention_KIVI(MistralAttention_KIVI):
    """Reference: MistralFlashAttention_KIVI (models/mistral_kivi.py:319-534): causal flash prompt pass, no mask."""
    _prefill_uses_mask = False
