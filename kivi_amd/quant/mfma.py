"""The MFMA-friendly cache layout (grouped-query decode since round 2, multi-head decode since round 3) (include/kivi_hip.h, "grouped queries on the matrix pipe").

Not part of the reference's Python surface: the reference keeps the hook-state tensors (models/llama_kivi.py:454-455)
and, for grouped queries, expands them nh / nh_kv times per call (models/mistral_kivi.py:58-67) or lets the CUDA kernel
map heads (quant/csrc/gemv_cuda.cu:361-365).  Here the same codes / scales / zero points live in super-blocks whose
words are matrix-core operands; `*_to_ref` reproduces the hook-state tensors bit for bit.
"""
from __future__ import annotations

import torch

from .. import _lib

SB_TOKENS = 512
SB_WORDS = 6144            # 2-bit codes: [codes 4096 | scale 1024 | mn 1024] words per super-block
SB_WORDS_4BIT = 10240      # 4-bit codes: [codes 8192 | scale 1024 | mn 1024] (kivi_mfma_layout.h, "KT4 / VT4")
BLOCK_TOKENS = 32


def sb_words(bits: int) -> int:
    return {2: SB_WORDS, 4: SB_WORDS_4BIT}[bits]


def supported(k_bits: int, v_bits: int, group_size: int, head_dim: int, residual_length: int, ratio: int) -> bool:
    """2-bit K and V: nh / nh_kv in {1, 4, 8}; 4-bit K and V: nh / nh_kv = 4 (round 4) or 1 (round 6)."""
    if not (group_size == 32 and head_dim == 128 and residual_length % 32 == 0 and k_bits == v_bits):
        return False
    return (k_bits == 2 and ratio in (1, 4, 8)) or (k_bits == 4 and ratio in (1, 4))


def _flag_words(B: int, nh_kv: int) -> int:
    return (B * nh_kv + 63) // 64 * 64          # the flags take whole 256-byte lines behind the super-blocks


def alloc_store(B: int, nh_kv: int, n_sb: int, device, bits: int = 2) -> torch.Tensor:
    """Zero-initialised storage of n_sb super-blocks per (batch row, kv head): logical shape (B, nh_kv, n_sb, sb_words(bits)) int32,
    in memory the super-block index sits outside the head index (the super-blocks in use form one dense region).
    The store's RANGE WORDS (include/kivi_hip.h: B * nh_kv int32; byte 0 marked by whatever writes a scale >= 256 into the unit,
    byte 1 by whatever writes a scale >= 2^-8, byte 2 by every library writer: `range_big` / `range_small`) live in the same allocation, right behind the
    super-blocks: `range_flags(store)` is the (B, nh_kv) view, and every wrapper below passes it along with the store."""
    W = sb_words(bits)
    main = B * n_sb * nh_kv * W
    flat = torch.zeros(main + _flag_words(B, nh_kv), dtype=torch.int32, device=device)
    return flat[:main].view(B, n_sb, nh_kv, W).permute(0, 2, 1, 3)


def range_flags(store: torch.Tensor) -> torch.Tensor:
    """(B, nh_kv) int32 view of a store's range flags (the store must come from alloc_store)."""
    B, nh_kv, n_sb = store.shape[0], store.shape[1], store.shape[2]
    main = B * n_sb * nh_kv * store.shape[3]
    stg = store.untyped_storage()
    if store.storage_offset() != 0 or stg.nbytes() != (main + _flag_words(B, nh_kv)) * 4:
        raise ValueError("not a store of kivi_amd.quant.mfma.alloc_store (its range flags live behind the super-blocks)")
    return torch.empty(0, dtype=torch.int32, device=store.device).set_(stg, main, (B, nh_kv), (nh_kv, 1))


def range_big(store: torch.Tensor) -> torch.Tensor:
    """(B, nh_kv) bool: a scale >= 256 was written into the unit (q'' / p'' are placed 2^10 lower: kivi_mfma_layout.h)."""
    return (range_flags(store) & 0xFF) != 0


def range_small(store: torch.Tensor) -> torch.Tensor:
    """(B, nh_kv) bool: the unit's writers keep the marks (byte 2) and every scale written into it so far is < 2^-8 (q'' / p'' are
    placed 2^8 higher).  A zero word -- nothing written, or written by something that does not mark -- is NOT small."""
    return (range_flags(store) & 0xFFFFFF) == 0x010000


def copy_store(dst: torch.Tensor, src: torch.Tensor) -> None:
    """The first src.shape[2] super-blocks of `dst` and its range flags <- `src` (cache growth, clone)."""
    dst[:, :, : src.shape[2]].copy_(src)
    range_flags(dst).copy_(range_flags(src))


def _st(store: torch.Tensor, bits: int = 2):
    assert store.dtype == torch.int32 and store.dim() == 4 and store.shape[3] == sb_words(bits) and store.stride(3) == 1
    return _lib.ptr(store), store.stride(0), store.stride(1), store.stride(2), _lib.ptr(range_flags(store))


def kt_pack(k: torch.Tensor, store: torch.Tensor, token_offset: int = 0, group_size: int = 32, bits: int = 2) -> None:
    """k (B, nh_kv, T, 128) fp16, T % 32 == 0 -> quantised per channel into `store` at token_offset."""
    _lib.require_gpu(k, "k")
    B, nh_kv, T, D = k.shape
    assert k.dtype == torch.float16 and k.stride(3) == 1 and (token_offset + T) <= store.shape[2] * SB_TOKENS
    lib = _lib.load()
    _lib.check(lib.kivi_kt_pack(_lib.ptr(k), k.stride(0), k.stride(1), k.stride(2), *_st(store, bits), token_offset, B, nh_kv, T, D,
                                group_size, bits, _lib.stream_ptr(k)), "kivi_kt_pack")


def vt_pack(v: torch.Tensor, store: torch.Tensor, group_size: int = 32, bits: int = 2) -> None:
    """v (B, nh_kv, T, 128) fp16, any T -> quantised per token (groups along the channel axis) into `store` from token 0:
    the prompt-pass V quantisation (llama_kivi.py:441-448) without the intermediate hook-state tensors."""
    _lib.require_gpu(v, "v")
    B, nh_kv, T, D = v.shape
    assert v.dtype == torch.float16 and v.stride(3) == 1 and T <= store.shape[2] * SB_TOKENS
    _lib.check(_lib.load().kivi_vt_pack(_lib.ptr(v), v.stride(0), v.stride(1), v.stride(2), *_st(store, bits), B, nh_kv, T, D,
                                        group_size, bits, _lib.stream_ptr(v)), "kivi_vt_pack")


def _relayout(fn, name, to_ref, store, code, scale, mn, T, D, group_size, bits):
    B, nh_kv = store.shape[0], store.shape[1]
    assert code.dtype == torch.int32 and scale.dtype == mn.dtype == torch.float16
    assert code.stride(3) == 1 and scale.stride(3) == 1 and scale.stride() == mn.stride()
    _lib.check(fn(int(to_ref), *_st(store, bits), _lib.ptr(code), code.stride(0), code.stride(1), code.stride(2), _lib.ptr(scale),
                  _lib.ptr(mn), scale.stride(0), scale.stride(1), scale.stride(2), B, nh_kv, T, D, group_size, bits,
                  _lib.stream_ptr(code)), name)


def kt_to_ref(store: torch.Tensor, T: int, D: int = 128, group_size: int = 32, bits: int = 2):
    """-> K_code_T (B, nh_kv, D, T / (32 / bits)) int32, K_scale_T, K_mn_T (B, nh_kv, D, T/32) fp16 of tokens [0, T)."""
    B, nh_kv = store.shape[0], store.shape[1]
    code = torch.empty((B, nh_kv, D, T // (32 // bits)), dtype=torch.int32, device=store.device)
    scale = torch.empty((B, nh_kv, D, T // group_size), dtype=torch.float16, device=store.device)
    mn = torch.empty_like(scale)
    _relayout(_lib.load().kivi_kt_relayout, "kivi_kt_relayout", True, store, code, scale, mn, T, D, group_size, bits)
    return code, scale, mn


def kt_from_ref(store: torch.Tensor, code: torch.Tensor, scale: torch.Tensor, mn: torch.Tensor, group_size: int = 32, bits: int = 2):
    T = code.shape[3] * (32 // bits)
    _relayout(_lib.load().kivi_kt_relayout, "kivi_kt_relayout", False, store, code, scale, mn, T, code.shape[2], group_size, bits)


def vt_to_ref(store: torch.Tensor, T: int, D: int = 128, group_size: int = 32, bits: int = 2):
    """-> V_code (B, nh_kv, T, D / (32 / bits)) int32, V_scale, V_mn (B, nh_kv, T, D/32) fp16 of tokens [0, T)."""
    B, nh_kv = store.shape[0], store.shape[1]
    code = torch.empty((B, nh_kv, T, D // (32 // bits)), dtype=torch.int32, device=store.device)
    scale = torch.empty((B, nh_kv, T, D // group_size), dtype=torch.float16, device=store.device)
    mn = torch.empty_like(scale)
    _relayout(_lib.load().kivi_vt_relayout, "kivi_vt_relayout", True, store, code, scale, mn, T, D, group_size, bits)
    return code, scale, mn


def vt_from_ref(store: torch.Tensor, code: torch.Tensor, scale: torch.Tensor, mn: torch.Tensor, group_size: int = 32, bits: int = 2):
    T = code.shape[2]
    _relayout(_lib.load().kivi_vt_relayout, "kivi_vt_relayout", False, store, code, scale, mn, T, code.shape[3] * (32 // bits), group_size, bits)


def gqa_scores(q: torch.Tensor, store: torch.Tensor, T: int, out: torch.Tensor, group_size: int = 32, bits: int = 2) -> None:
    """out[..., :T] <- packed qK^T.  q (B, nh, 1, 128) fp16, out (B, nh, 1, >= T) fp16 rows (16-byte aligned)."""
    B, nh, _, D = q.shape
    nh_kv = store.shape[1]
    assert q.dtype == out.dtype == torch.float16 and q.stride(3) == 1 and out.stride(3) == 1
    _lib.check(_lib.load().kivi_gqa_scores(_lib.ptr(q), q.stride(0), q.stride(1), *_st(store, bits), _lib.ptr(out), out.stride(0),
                                           out.stride(1), B, nh, nh_kv, D, T, group_size, bits, _lib.stream_ptr(q)),
               "kivi_gqa_scores")


_OUT_WS = {}


def gqa_output(probs: torch.Tensor, store: torch.Tensor, T: int, out: torch.Tensor = None, group_size: int = 32, bits: int = 2) -> torch.Tensor:
    """out[b, h, 0, :] = packed sV over tokens [0, T) of a VT store for given fp16 attention weights probs (B, nh, 1, >= T)
    (rows 16-byte aligned, pitch a multiple of 8): cuda_bmm_fA_qB_outer at llama_kivi.py:382 on the matrix-pipe layout,
    nh / nh_kv in {1, 4, 8} at 2 bits, 4 at 4 bits."""
    B, nh = probs.shape[0], probs.shape[1]
    nh_kv = store.shape[1]
    assert probs.dtype == torch.float16 and probs.stride(3) == 1
    if out is None:
        out = torch.empty((B, nh, 1, 128), dtype=torch.float16, device=probs.device)
    nsb = max(1, (T + SB_TOKENS - 1) // SB_TOKENS)
    need = 65536 + ((B * nh * 4 + 255) // 256) * 256 + B * nh_kv * nsb * 2 * (nh // nh_kv) * 128 * 4
    key = (str(probs.device), torch.cuda.current_stream(probs.device).cuda_stream)
    ws = _OUT_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(need, dtype=torch.uint8, device=probs.device)
        _OUT_WS[key] = ws
    _lib.check(_lib.load().kivi_gqa_output(_lib.ptr(probs), probs.stride(0), probs.stride(1), *_st(store, bits), _lib.ptr(out),
                                           out.stride(0), out.stride(1), B, nh, nh_kv, 128, T, group_size, bits, _lib.ptr(ws),
                                           ws.numel(), _lib.stream_ptr(probs)), "kivi_gqa_output")
    return out
