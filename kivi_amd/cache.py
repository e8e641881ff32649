"""Per-layer KIVI KV-cache state: the reference's 9-tuple, held in pre-allocated, in-place-appended buffers.

Reference contract (models/llama_kivi.py:454-455, read back at :315-322):
    (K_code_T, K_full, K_scale_T, K_mn_T, V_code, V_full, V_scale, V_mn, kv_seq_len)
      K_code_T (B, nh_kv, D, Tq/fpi) int32     K_scale_T, K_mn_T (B, nh_kv, D, Tq/g) fp16
      K_full   (B, nh_kv, 0..R-1, D) fp16 or None
      V_code   (B, nh_kv, Tv, D/fpi) int32     V_scale, V_mn (B, nh_kv, Tv, D/g) fp16
      V_full   (B, nh_kv, <=R, D) fp16
The reference grows every member with torch.cat -- packed V + scale + mn are re-copied EVERY step
(llama_kivi.py:393-395), packed K every R steps (:350-352).  Here a decode step moves only the new token(s):

  * V (per token) is stored exactly in the reference layout with spare rows at the end; appending a token
    writes one row and the 9-tuple members are plain views.
  * K (per channel) cannot be appended in the reference layout without striding every channel row by the
    capacity, which costs ~35 % of HBM efficiency on MI355X (profiles/).  It is stored in PAGES of
    `page_tokens` tokens, each page a contiguous (D, page_tokens/fpi) block = the reference layout of that token
    range; the fused GEMV reads pages directly (kivi_gemv_k_paged).  The reference-layout K members of the
    9-tuple are materialised (one copy) only if somebody actually indexes them.

Cache policy (llama_kivi.py:343-356, 386-399, 425-452):
  K: residual grows; when it holds exactly R tokens all R are quantised at once (per channel, groups of g tokens).
  V: sliding fp16 window of R tokens; when it holds R+1 the OLDEST token is quantised (per token, groups of g channels).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _tuning
from .quant import new_pack

PAGE_TOKENS = 2048   # = the tile of the default qK^T kernels (64 lanes x 2 words x 16 codes; 4-bit: 4 words x 8)


class KiviCacheTuple(tuple):
    """The reference's 9-tuple.  `t[-1]` / `t[8]` (running kv length, what HF's generate loop reads,
    llama_kivi.py:698, :916) is free; tensor members are built from the live cache on first access."""

    def __new__(cls, layer: "KiviLayerCache"):
        t = super().__new__(cls, (None,) * 8 + (layer.kv_seq_len,))
        t.layer = layer
        t._items = None
        return t

    def _materialise(self):
        if self._items is None:
            n = tuple.__getitem__(self, 8)
            if self.layer.kv_seq_len != n:
                raise RuntimeError(f"stale KIVI cache tuple: issued at kv length {n}, the in-place cache is now at "
                                   f"{self.layer.kv_seq_len}; index the members before the next decode step (or clone the cache)")
            self._items = self.layer._tuple_members() + (n,)
        return self._items

    def __getitem__(self, i):
        if isinstance(i, int) and i in (8, -1):
            return tuple.__getitem__(self, 8)
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())


@dataclass
class KiviConfig:
    k_bits: int = 2
    v_bits: int = 2
    group_size: int = 32
    residual_length: int = 32

    def __post_init__(self):
        assert self.k_bits in (2, 4) and self.v_bits in (2, 4), "the fused GEMV supports 2 and 4 bits (matmul.py:215)"
        assert self.residual_length % self.group_size == 0  # llama_kivi.py:344


class KiviLayerCache:
    """One layer's quantised KV cache with capacity `max_len` tokens, appended in place."""

    def __init__(self, cfg: KiviConfig, batch: int, num_kv_heads: int, head_dim: int, max_len: int,
                 device, dtype=torch.float16, page_tokens: int = None):
        assert dtype == torch.float16, "the reference extension is fp16 only (gemv_cuda.cu:526-529)"
        self.cfg = cfg
        R, g = cfg.residual_length, cfg.group_size
        self.B, self.nh_kv, self.D = batch, num_kv_heads, head_dim
        assert head_dim % g == 0 and head_dim % (32 // cfg.v_bits) == 0
        if page_tokens is None:
            # a K flush of R tokens must not straddle pages AND a page is whole tiles of the default qK^T kernels: the smallest common
            # multiple (R = 32 / 64 / 128: 2048; R = 96 or 192: 6144 -- the reference accepts any multiple of the group size, llama_kivi.py:344)
            page_tokens = PAGE_TOKENS * R // math.gcd(PAGE_TOKENS, R)
        assert page_tokens % R == 0 and page_tokens % g == 0, "a K flush of R tokens must not straddle pages"
        self.page_tokens = page_tokens
        self.cap = ((max_len + R - 1) // R) * R
        self.n_pages = (self.cap + page_tokens - 1) // page_tokens
        kf, vf = 32 // cfg.k_bits, 32 // cfg.v_bits
        dev = device
        # logical shape (B, nh_kv, P, D, page/fpi); in MEMORY the page index is outside the head index, so the pages in
        # use form one dense region and the spare capacity sits behind it (with the head index outside, every head's
        # unused pages would punch 64 KiB holes into the streamed range)
        self.k_code = self._paged((batch, num_kv_heads, self.n_pages, head_dim, page_tokens // kf), torch.int32, dev)
        self.k_scale = self._paged((batch, num_kv_heads, self.n_pages, head_dim, page_tokens // g), dtype, dev)
        self.k_mn = self._paged((batch, num_kv_heads, self.n_pages, head_dim, page_tokens // g), dtype, dev)
        self.k_res = torch.empty((batch, num_kv_heads, R, head_dim), dtype=dtype, device=dev)
        self.v_code = torch.empty((batch, num_kv_heads, self.cap, head_dim // vf), dtype=torch.int32, device=dev)
        self.v_scale = torch.empty((batch, num_kv_heads, self.cap, head_dim // g), dtype=dtype, device=dev)
        self.v_mn = torch.empty_like(self.v_scale)
        # fp16 V window: R (+1 transient) live tokens inside a 2R+1 buffer, compacted every R steps
        self.v_res = torch.empty((batch, num_kv_heads, 2 * R + 1, head_dim), dtype=dtype, device=dev)
        self.k_quant_len = 0   # tokens in the packed K prefix (multiple of R)
        self.k_res_len = 0     # tokens in the fp16 K residual (< R between steps)
        self.v_quant_len = 0   # tokens in the packed V prefix
        self.v_res_start = 0
        self.v_res_len = 0     # tokens in the fp16 V window (<= R between steps)
        self.kv_seq_len = 0

    @staticmethod
    def _paged(shape, dtype, device) -> torch.Tensor:
        B, h, P, D, W = shape
        if _tuning.flag("KIVI_K_HEAD_MAJOR"):   # tuning sessions: the plain (B, nh_kv, P, ...) memory order
            return torch.empty(shape, dtype=dtype, device=device)
        return torch.empty((B, P, h, D, W), dtype=dtype, device=device).permute(0, 2, 1, 3, 4)

    # ------------------------------------------------------------------ capacity
    def reserve(self, max_len: int) -> None:
        """Grow the capacity to at least `max_len` tokens (the reference's torch.cat-grown tuple has no limit,
        llama_kivi.py:350-352, :393-395): new page / row buffers, the live contents copied once.  Scratch buffers and the
        cached native descriptor of the attention hook are dropped and rebuilt on the next step."""
        R = self.cfg.residual_length
        cap = ((max_len + R - 1) // R) * R
        if cap <= self.cap:
            return
        n_pages = (cap + self.page_tokens - 1) // self.page_tokens

        def grown(x, dim, n):
            shape = list(x.shape)
            shape[dim] = n
            y = torch.empty(shape, dtype=x.dtype, device=x.device)
            y.narrow(dim, 0, x.shape[dim]).copy_(x)
            return y
        if n_pages > self.n_pages:
            def grown_pages(x):
                y = self._paged((x.shape[0], x.shape[1], n_pages, x.shape[3], x.shape[4]), x.dtype, x.device)
                y[:, :, : x.shape[2]].copy_(x)
                return y
            self.k_code, self.k_scale, self.k_mn = (grown_pages(x) for x in (self.k_code, self.k_scale, self.k_mn))
            self.n_pages = n_pages
        self.v_code, self.v_scale, self.v_mn = (grown(x, 2, cap) for x in (self.v_code, self.v_scale, self.v_mn))
        self.cap = cap
        for name in ("_native", "_scores", "_probs"):
            if hasattr(self, name):
                delattr(self, name)

    def ensure_room(self, tokens: int = 1) -> None:
        """Make room for `tokens` more tokens, doubling the capacity when it runs out (amortised O(1) copies)."""
        need = self.kv_seq_len + tokens
        if need > self.cap:
            self.reserve(max(need, 2 * self.cap))

    def clone(self) -> "KiviLayerCache":
        """Independent copy of the cache (what holding on to an old reference tuple gives for free): use it to continue one
        prefix twice (beam / contrastive / assisted decoding)."""
        import copy
        other = copy.copy(self)
        for name in ("k_code", "k_scale", "k_mn", "k_res", "v_code", "v_scale", "v_mn", "v_res"):
            src = getattr(self, name)
            dst = torch.empty_strided(src.shape, src.stride(), dtype=src.dtype, device=src.device)
            dst.copy_(src)
            setattr(other, name, dst)
        for name in ("_native", "_scores", "_probs"):
            other.__dict__.pop(name, None)
        return other

    # ------------------------------------------------------------------ the 9-tuple
    def k_quant_reference_layout(self):
        """(K_code_T, K_scale_T, K_mn_T) in the reference layout (B, nh_kv, D, Tq/...): gathers the pages (a copy)."""
        if self.k_quant_len == 0:
            return None, None, None
        kf, g = 32 // self.cfg.k_bits, self.cfg.group_size
        npg = (self.k_quant_len + self.page_tokens - 1) // self.page_tokens
        B, h, D = self.B, self.nh_kv, self.D

        def gather(x, per_tok):
            y = x[:, :, :npg].permute(0, 1, 3, 2, 4).reshape(B, h, D, -1)
            return y[..., : self.k_quant_len // per_tok].contiguous()
        return gather(self.k_code, kf), gather(self.k_scale, g), gather(self.k_mn, g)

    def v_quant_views(self):
        if self.v_quant_len == 0:
            return None, None, None
        n = self.v_quant_len
        return self.v_code[:, :, :n], self.v_scale[:, :, :n], self.v_mn[:, :, :n]

    def k_res_view(self) -> Optional[torch.Tensor]:
        return self.k_res[:, :, : self.k_res_len] if self.k_res_len else None

    def v_res_view(self) -> torch.Tensor:
        return self.v_res[:, :, self.v_res_start: self.v_res_start + self.v_res_len]

    def _tuple_members(self):
        kc, ks, km = self.k_quant_reference_layout()
        vc, vs, vm = self.v_quant_views()
        return (kc, self.k_res_view(), ks, km, vc, self.v_res_view(), vs, vm)

    def as_tuple(self) -> KiviCacheTuple:
        return KiviCacheTuple(self)

    def nbytes(self) -> int:
        """Resident cache bytes = what the reference's 9-tuple tensors would hold for the same state."""
        c = self.cfg
        per_k = self.D * self.k_quant_len * c.k_bits // 8 + 2 * self.D * (self.k_quant_len // c.group_size) * 2
        per_v = self.v_quant_len * self.D * c.v_bits // 8 + 2 * self.v_quant_len * (self.D // c.group_size) * 2
        res = (self.k_res_len + self.v_res_len) * self.D * 2
        return self.B * self.nh_kv * (per_k + per_v + res)

    def allocated_bytes(self) -> int:
        return sum(x.numel() * x.element_size() for x in (self.k_code, self.k_scale, self.k_mn, self.k_res, self.v_code,
                                                          self.v_scale, self.v_mn, self.v_res))

    # ------------------------------------------------------------------ K pages
    def _k_page(self, p: int):
        return self.k_code[:, :, p], self.k_scale[:, :, p], self.k_mn[:, :, p]

    def _quantise_k(self, key_states: torch.Tensor, t0: int) -> None:
        """Quantise tokens key_states (B, nh_kv, n, D), n % g == 0, into the packed prefix starting at token t0."""
        P, g = self.page_tokens, self.cfg.group_size
        n = key_states.shape[2]
        done = 0
        while done < n:
            p, off = divmod(t0 + done, P)
            take = min(n - done, P - off)
            new_pack.quantize_and_pack_k_tmajor(key_states[:, :, done: done + take], g, self.cfg.k_bits,
                                                out=self._k_page(p), token_offset=off)
            done += take

    # ------------------------------------------------------------------ prefill (llama_kivi.py:425-452)
    def prefill(self, key_states: torch.Tensor, value_states: torch.Tensor) -> None:
        """key/value_states (B, nh_kv, T, D) fp16 (any strides with a contiguous last dim)."""
        cfg = self.cfg
        R, g = cfg.residual_length, cfg.group_size
        T = key_states.shape[2]
        self.reserve(T)
        nq = (T // R) * R                      # quantised K prefix, fp16 remainder T % R
        if nq:
            self._quantise_k(key_states[:, :, :nq], 0)
        self.k_quant_len = nq
        self.k_res_len = T - nq
        if self.k_res_len:
            self.k_res[:, :, : self.k_res_len].copy_(key_states[:, :, nq:])
        nv = max(T - R, 0)                     # quantised V prefix, last min(T, R) tokens stay fp16
        if nv:
            code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(value_states[:, :, :nv].contiguous(), g,
                                                                                cfg.v_bits)
            self.v_code[:, :, :nv].copy_(code)
            self.v_scale[:, :, :nv].copy_(scale)
            self.v_mn[:, :, :nv].copy_(mn)
        self.v_quant_len = nv
        self.v_res_start = 0
        self.v_res_len = T - nv
        self.v_res[:, :, : self.v_res_len].copy_(value_states[:, :, nv:])
        self.kv_seq_len = T

    # ------------------------------------------------------------------ decode-step mutations
    def append_k(self, key_states: torch.Tensor) -> None:
        """llama_kivi.py:333-336: K residual += the new token (B, nh_kv, 1, D)."""
        assert key_states.shape[2] == 1
        assert self.k_quant_len + self.k_res_len + 1 <= self.cap, "cache capacity exceeded"
        self.k_res[:, :, self.k_res_len: self.k_res_len + 1].copy_(key_states)
        self.k_res_len += 1

    def maybe_flush_k(self) -> None:
        """llama_kivi.py:343-356: when the residual holds exactly R tokens, quantise all of them in place."""
        R = self.cfg.residual_length
        if self.k_res_len == R:
            self._quantise_k(self.k_res, self.k_quant_len)
            self.k_quant_len += R
            self.k_res_len = 0

    def compact_v_window(self) -> None:
        """Move the live window rows to the front of the 2R+1 buffer (once every ~R steps)."""
        live = self.v_res[:, :, self.v_res_start: self.v_res_start + self.v_res_len].clone()
        self.v_res[:, :, : self.v_res_len].copy_(live)
        self.v_res_start = 0

    def append_v(self, value_states: torch.Tensor) -> None:
        """llama_kivi.py:377: V window += the new token."""
        assert value_states.shape[2] == 1
        R = self.cfg.residual_length
        if self.v_res_start + self.v_res_len + 1 > self.v_res.shape[2]:
            self.compact_v_window()
        pos = self.v_res_start + self.v_res_len
        self.v_res[:, :, pos: pos + 1].copy_(value_states)
        self.v_res_len += 1
        assert self.v_res_len <= R + 1

    def maybe_flush_v(self) -> None:
        """llama_kivi.py:386-399: when the window holds R+1 tokens, quantise the oldest one in place."""
        R, g = self.cfg.residual_length, self.cfg.group_size
        if self.v_res_len > R:
            assert self.v_res_len == R + 1 and self.v_quant_len + 1 <= self.cap
            oldest = self.v_res[:, :, self.v_res_start: self.v_res_start + 1]
            code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(oldest.contiguous(), g, self.cfg.v_bits)
            n = self.v_quant_len
            self.v_code[:, :, n: n + 1].copy_(code)
            self.v_scale[:, :, n: n + 1].copy_(scale)
            self.v_mn[:, :, n: n + 1].copy_(mn)
            self.v_quant_len += 1
            self.v_res_start += 1
            self.v_res_len -= 1

    # ------------------------------------------------------------------ import of a plain reference tuple
    @classmethod
    def from_tuple(cls, cfg: KiviConfig, past, max_len: int) -> "KiviLayerCache":
        """Adopt a plain 9-tuple produced elsewhere (copies it into the in-place buffers once)."""
        kc, kfull, ks, km, vc, vfull, vs, vm, kv_len = past
        ref = vfull if vfull is not None else kfull
        B, nh_kv, _, D = ref.shape
        self = cls(cfg, B, nh_kv, D, max_len, ref.device, ref.dtype)
        kf, g, P = 32 // cfg.k_bits, cfg.group_size, self.page_tokens
        if kc is not None:
            self.k_quant_len = kc.shape[-1] * kf
            for p in range((self.k_quant_len + P - 1) // P):
                n = min(P, self.k_quant_len - p * P)
                self.k_code[:, :, p, :, : n // kf].copy_(kc[..., p * P // kf: (p * P + n) // kf])
                self.k_scale[:, :, p, :, : n // g].copy_(ks[..., p * P // g: (p * P + n) // g])
                self.k_mn[:, :, p, :, : n // g].copy_(km[..., p * P // g: (p * P + n) // g])
        if kfull is not None:
            self.k_res_len = kfull.shape[2]
            self.k_res[:, :, : self.k_res_len].copy_(kfull)
        if vc is not None:
            self.v_quant_len = vc.shape[2]
            self.v_code[:, :, : self.v_quant_len].copy_(vc)
            self.v_scale[:, :, : self.v_quant_len].copy_(vs)
            self.v_mn[:, :, : self.v_quant_len].copy_(vm)
        self.v_res_len = vfull.shape[2]
        self.v_res[:, :, : self.v_res_len].copy_(vfull)
        self.kv_seq_len = int(kv_len)
        return self
