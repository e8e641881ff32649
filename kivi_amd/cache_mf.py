"""Per-layer KIVI cache in the matrix-pipe layout (kivi_amd/csrc/kivi_mfma_layout.h): grouped-query models (round 2) and
multi-head models (round 3).

Same state machine and the same bits as KiviLayerCache (cache.py; reference contract models/llama_kivi.py:454-455, read
back at :315-322; Mistral: models/mistral_kivi.py:381-385, :441-445): the 9-tuple members are reproduced bit for bit by
the relayout kernels.  What differs is where the packed codes live: super-blocks of 512 tokens whose words are B operands
of v_mfma_f32_16x16x32_f16, so that the nh / nh_kv query heads of a kv head share every code on the matrix pipe instead of
costing one FMA each (the shared-unpack VALU kernels run at ~0.35 of the HBM roofline for nh / nh_kv = 4).

A decode step is ONE library call (kivi_mf_decode_layer: lengths, the launches, the K flush through kivi_kt_pack every R
steps).  Launches: rows whose scores fit the LDS (nh = nh_kv: 16 super-blocks + the residual, nh / nh_kv = 4: 18) and enough of them
-> one (mf_row_kernel / mf_row4_kernel; longer grouped-query rows: one launch of slices); otherwise two (packed qK^T + residual scores + K append + softmax statistics, then
softmax-on-the-fly + packed sV + fp16 window + V append / quantise).  Every store carries range flags (quant/mfma.py) that
keep the fp16 operands of the matrix pipe finite for any finite scale.

Round 4: also 4-bit K / V for nh / nh_kv = 4 (the reference's published Mistral-7B + KIVI-4 shape, docs/long_bench.md:35-53): the same
state machine and calls over 10240-word super-blocks (kivi_mfma_layout.h, "KT4 / VT4").  Round 6: 4-bit K / V of multi-head models
(Llama-2-7B / LongChat-7B-32K + KIVI-4, docs/long_bench.md:5-26) too.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _lib, _tuning
from .cache import KiviCacheTuple, KiviConfig
from .quant import mfma, new_pack

SB = mfma.SB_TOKENS
_SCRATCH = {}   # (device, stream) -> dict(scores, stats, ws): shared by the layers that decode on that stream (launches are
                # stream-ordered; two streams must not share score rows or arrival counters)
_WS_COUNTER_BYTES = 65536


def supported(cfg: KiviConfig, head_dim: int, num_heads: int, num_kv_heads: int) -> bool:
    if _tuning.flag("KIVI_NO_MFMA_LAYOUT"):     # tuning sessions: keep every model on the hook-state layout
        return False
    if num_heads == num_kv_heads and _tuning.flag("KIVI_NO_MFMA_MHA"):   # tuning sessions (A/B): multi-head models on the hook-state layout
        return False
    return num_heads % num_kv_heads == 0 and mfma.supported(cfg.k_bits, cfg.v_bits, cfg.group_size, head_dim,
                                                            cfg.residual_length, num_heads // num_kv_heads) \
        and cfg.residual_length <= 128


def _scratch(device, B: int, nh: int, nh_kv: int, pitch: int, nseg: int, slices: int):
    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    d = _SCRATCH.setdefault((str(device), stream), {})
    sc = d.get("scores")
    if sc is None or sc.shape[0] < B or sc.shape[1] < nh or sc.shape[3] < pitch:
        shape = (B, nh, 1, pitch) if sc is None else (max(B, sc.shape[0]), max(nh, sc.shape[1]), 1, max(pitch, sc.shape[3]))
        sc = torch.empty(shape, dtype=torch.float16, device=device)      # every dimension grows monotonically
        d["scores"] = sc
    st = d.get("stats")
    need = B * nh * nseg * 2
    if st is None or st.numel() < need:
        st = torch.empty(need, dtype=torch.float32, device=device)
        d["stats"] = st
    ws = d.get("ws")
    need = _WS_COUNTER_BYTES + B * nh_kv * 2 * (max(slices, 1) + 1) * (nh // nh_kv) * 128 * 4   # + 1: the window block's slot
    if ws is None or ws.numel() < need:
        ws = torch.zeros(need, dtype=torch.uint8, device=device)   # arrival counters start at zero; the kernel resets them
        d["ws"] = ws
    return sc, st, ws


class KiviLayerCacheMF:
    """One layer's quantised KV cache (capacity `max_len` tokens, appended in place): 2-bit with nh / nh_kv in {1, 4, 8}, 4-bit with
    nh / nh_kv in {1, 4} (multi-head 4-bit: round 6)."""

    layout = "mfma"

    def __init__(self, cfg: KiviConfig, batch: int, num_kv_heads: int, head_dim: int, max_len: int, device,
                 dtype=torch.float16, num_heads: int = None):
        assert dtype == torch.float16, "the reference extension is fp16 only (gemv_cuda.cu:526-529)"
        assert num_heads is not None and supported(cfg, head_dim, num_heads, num_kv_heads), \
            "matrix-pipe layout: group 32, head_dim 128, residual_length <= 128; 2-bit with nh / nh_kv in {1, 4, 8} or 4-bit with nh / nh_kv in {1, 4}"
        self.cfg = cfg
        R = cfg.residual_length
        self.B, self.nh_kv, self.D, self.nh = batch, num_kv_heads, head_dim, num_heads
        self.cap = ((max_len + R - 1) // R) * R
        self.n_sb = (self.cap + SB - 1) // SB
        self.kt = mfma.alloc_store(batch, num_kv_heads, self.n_sb, device, cfg.k_bits)
        self.vt = mfma.alloc_store(batch, num_kv_heads, self.n_sb, device, cfg.v_bits)
        self.k_res = torch.empty((batch, num_kv_heads, R, head_dim), dtype=dtype, device=device)
        # fp16 value window: a RING of R + 1 rows (row of window token t = (v_res_start + t) mod rows): nothing is ever compacted
        self.ring = True
        self.v_res = torch.empty((batch, num_kv_heads, R + 1, head_dim), dtype=dtype, device=device)
        self.k_quant_len = 0
        self.k_res_len = 0
        self.v_quant_len = 0
        self.v_res_start = 0
        self.v_res_len = 0
        self.kv_seq_len = 0
        self._native = None       # (descriptor, state array, scratch tensors) of kivi_mf_decode_layer
        self.flags = 0            # _lib.GQA_FORCE_SPLIT / GQA_FORCE_ROW (tests, tuning)

    # ------------------------------------------------------------------ capacity
    def reserve(self, max_len: int) -> None:
        R = self.cfg.residual_length
        cap = ((max_len + R - 1) // R) * R
        if cap <= self.cap:
            return
        n_sb = (cap + SB - 1) // SB
        if n_sb > self.n_sb:
            for name in ("kt", "vt"):
                old = getattr(self, name)
                new = mfma.alloc_store(self.B, self.nh_kv, n_sb, old.device, self.cfg.k_bits)
                mfma.copy_store(new, old)                      # super-blocks in use + the store's range flags
                setattr(self, name, new)
            self.n_sb = n_sb
        self.cap = cap
        self._native = None

    def ensure_room(self, tokens: int = 1) -> None:
        need = self.kv_seq_len + tokens
        if need > self.cap:
            self.reserve(max(need, 2 * self.cap))

    def clone(self) -> "KiviLayerCacheMF":
        import copy
        other = copy.copy(self)
        for name in ("kt", "vt"):
            src = getattr(self, name)
            dst = mfma.alloc_store(self.B, self.nh_kv, self.n_sb, src.device, self.cfg.k_bits)
            mfma.copy_store(dst, src)
            setattr(other, name, dst)
        for name in ("k_res", "v_res"):
            src = getattr(self, name)
            dst = torch.empty_strided(src.shape, src.stride(), dtype=src.dtype, device=src.device)
            dst.copy_(src)
            setattr(other, name, dst)
        other._native = None
        return other

    # ------------------------------------------------------------------ the 9-tuple
    def k_quant_reference_layout(self):
        if self.k_quant_len == 0:
            return None, None, None
        return mfma.kt_to_ref(self.kt, self.k_quant_len, self.D, self.cfg.group_size, self.cfg.k_bits)

    def v_quant_views(self):
        if self.v_quant_len == 0:
            return None, None, None
        return mfma.vt_to_ref(self.vt, self.v_quant_len, self.D, self.cfg.group_size, self.cfg.v_bits)

    def k_res_view(self):
        return self.k_res[:, :, : self.k_res_len] if self.k_res_len else None

    def v_res_view(self):
        s, n, rows = self.v_res_start, self.v_res_len, self.v_res.shape[2]
        if s + n <= rows:
            return self.v_res[:, :, s: s + n]
        return torch.cat([self.v_res[:, :, s:], self.v_res[:, :, : s + n - rows]], dim=2)      # the ring wraps (9-tuple reads only)

    def _tuple_members(self):
        kc, ks, km = self.k_quant_reference_layout()
        vc, vs, vm = self.v_quant_views()
        return (kc, self.k_res_view(), ks, km, vc, self.v_res_view(), vs, vm)

    def as_tuple(self) -> KiviCacheTuple:
        return KiviCacheTuple(self)

    def nbytes(self) -> int:
        c = self.cfg
        per_k = self.D * self.k_quant_len * c.k_bits // 8 + 2 * self.D * (self.k_quant_len // c.group_size) * 2
        per_v = self.v_quant_len * self.D * c.v_bits // 8 + 2 * self.v_quant_len * (self.D // c.group_size) * 2
        res = (self.k_res_len + self.v_res_len) * self.D * 2
        return self.B * self.nh_kv * (per_k + per_v + res)

    def allocated_bytes(self) -> int:
        return sum(x.numel() * x.element_size() for x in (self.kt, self.vt, self.k_res, self.v_res))

    # ------------------------------------------------------------------ prefill (llama_kivi.py:425-452)
    def prefill(self, key_states: torch.Tensor, value_states: torch.Tensor) -> None:
        cfg = self.cfg
        R, g = cfg.residual_length, cfg.group_size
        T = key_states.shape[2]
        self.reserve(T)
        if self.kv_seq_len:        # reuse of the object: the V slots are filled token by token later, start from clean storage
            for st in (self.kt, self.vt):
                st.zero_()
                mfma.range_flags(st).zero_()
        nq = (T // R) * R
        if nq:
            mfma.kt_pack(key_states[:, :, :nq], self.kt, 0, g, cfg.k_bits)
        self.k_quant_len = nq
        self.k_res_len = T - nq
        if self.k_res_len:
            self.k_res[:, :, : self.k_res_len].copy_(key_states[:, :, nq:])
        nv = max(T - R, 0)
        if nv:
            vq = value_states[:, :, :nv]
            if vq.stride(3) == 1 and vq.data_ptr() % 16 == 0 and all(st % 8 == 0 for st in vq.stride()[:3]):
                mfma.vt_pack(vq, self.vt, g, cfg.v_bits)           # one pass, straight into the layout
            else:                                                   # odd strides: through the hook-state tensors
                code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(vq.contiguous(), g, cfg.v_bits)
                mfma.vt_from_ref(self.vt, code, scale, mn, g, cfg.v_bits)
        self.v_quant_len = nv
        self.v_res_start = 0
        self.v_res_len = T - nv
        self.v_res[:, :, : self.v_res_len].copy_(value_states[:, :, nv:])
        self.kv_seq_len = T

    @classmethod
    def from_tuple(cls, cfg: KiviConfig, past, max_len: int, num_heads: int) -> "KiviLayerCacheMF":
        kc, kfull, ks, km, vc, vfull, vs, vm, kv_len = past
        ref = vfull if vfull is not None else kfull
        B, nh_kv, _, D = ref.shape
        self = cls(cfg, B, nh_kv, D, max_len, ref.device, ref.dtype, num_heads=num_heads)
        if kc is not None:
            self.k_quant_len = kc.shape[-1] * (32 // cfg.k_bits)
            mfma.kt_from_ref(self.kt, kc.contiguous(), ks.contiguous(), km.contiguous(), cfg.group_size, cfg.k_bits)
        if kfull is not None:
            self.k_res_len = kfull.shape[2]
            self.k_res[:, :, : self.k_res_len].copy_(kfull)
        if vc is not None:
            self.v_quant_len = vc.shape[2]
            mfma.vt_from_ref(self.vt, vc.contiguous(), vs.contiguous(), vm.contiguous(), cfg.group_size, cfg.v_bits)
        self.v_res_len = vfull.shape[2]
        self.v_res[:, :, : self.v_res_len].copy_(vfull)
        self.kv_seq_len = int(kv_len)
        return self

    # ------------------------------------------------------------------ decode step (llama_kivi.py:314-399)
    def _desc(self, nh: int, device):
        """kivi_mf_layer_desc of this cache (rebuilt when the stores were reallocated or the stream changed)."""
        stream = torch.cuda.current_stream(device).cuda_stream
        nat = self._native
        if nat is not None and nat[2] == (nh, stream):
            return nat
        pitch = ((self.n_sb * SB + self.cfg.residual_length + 1 + 7) // 8) * 8     # the longest row of any step the stores can hold
        scores, stats, ws = _scratch(device, self.B, nh, self.nh_kv, pitch, self.n_sb + 4, self.n_sb)
        kt, vt, kr, vr = self.kt, self.vt, self.k_res, self.v_res
        d = _lib.MfLayerDesc(
            B=self.B, nh_kv=self.nh_kv, D=self.D, bits=self.cfg.k_bits, group_size=self.cfg.group_size,
            residual_length=self.cfg.residual_length, inv_scale=1.0 / math.sqrt(self.D),
            cap=self.n_sb * SB, v_window_rows=vr.shape[2], s_pitch=scores.shape[3],
            kt=kt.data_ptr(), kt_sb=kt.stride(0), kt_sh=kt.stride(1), kt_ss=kt.stride(2),
            vt=vt.data_ptr(), vt_sb=vt.stride(0), vt_sh=vt.stride(1), vt_ss=vt.stride(2),
            k_res=kr.data_ptr(), kr_sb=kr.stride(0), kr_sh=kr.stride(1), kr_st=kr.stride(2),
            v_res=vr.data_ptr(), vr_sb=vr.stride(0), vr_sh=vr.stride(1), vr_st=vr.stride(2),
            scores=scores.data_ptr(), s_sb=scores.stride(0), s_sh=scores.stride(1),
            stats=stats.data_ptr(), stats_bytes=stats.numel() * 4,
            workspace=ws.data_ptr(), workspace_bytes=ws.numel(), flags=self._flags(),
            kt_range=mfma.range_flags(kt).data_ptr(), vt_range=mfma.range_flags(vt).data_ptr())
        state = (ctypes.c_int64 * 6)()
        self._native = (d, state, (nh, stream), _lib.load().kivi_mf_decode_layer, (scores, stats, ws))
        return self._native

    def _flags(self) -> int:
        return self.flags | (_lib.GQA_WINDOW_RING if self.ring else 0)

    def decode_step(self, query_states: torch.Tensor, key_states: torch.Tensor, value_states: torch.Tensor,
                    attention_mask: torch.Tensor = None, out: torch.Tensor = None) -> torch.Tensor:
        B, nh, _, D = query_states.shape
        assert nh == self.nh and B == self.B and D == self.D
        q, k, v = _rows16(query_states), _rows16(key_states), _rows16(value_states)
        kv_seq_len = self.kv_seq_len + 1
        mask_ptr, mask_sb = None, 0
        if attention_mask is not None:
            if attention_mask.size() != (B, 1, 1, kv_seq_len):
                raise ValueError(f"Attention mask should be of size {(B, 1, 1, kv_seq_len)}, but is {attention_mask.size()}")
            assert attention_mask.dtype == torch.float16 and attention_mask.stride(3) == 1
            mask_ptr, mask_sb = attention_mask.data_ptr(), attention_mask.stride(0)
        if out is None:
            out = torch.empty((B, nh, 1, D), dtype=torch.float16, device=q.device)
        else:
            assert out.shape == (B, nh, 1, D) and out.dtype == torch.float16 and out.stride(3) == 1
        d, state, key, fn, _ = self._desc(nh, q.device)     # key = (nh, stream handle the descriptor was built for = the current one)
        d.flags = self._flags()
        state[0], state[1], state[2] = self.k_quant_len, self.k_res_len, self.v_quant_len
        state[3], state[4], state[5] = self.v_res_start, self.v_res_len, self.kv_seq_len
        hook = _launch_hook()
        if hook is not None and self.k_quant_len:
            hook("pre", "k", dict(B=B, nh=nh, nh_kv=self.nh_kv, K=D, N=self.k_quant_len, bits=self.cfg.k_bits,
                                  group_size=self.cfg.group_size, v_bits=self.cfg.v_bits, Tv=self.v_quant_len,
                                  k_res=self.k_res_len + 1, v_res=self.v_res_len + 1))
        rc = fn(ctypes.byref(d), state, q.data_ptr(), q.stride(0), q.stride(1), nh, k.data_ptr(), k.stride(0), k.stride(1),
                v.data_ptr(), v.stride(0), v.stride(1), mask_ptr, mask_sb, out.data_ptr(), out.stride(0), out.stride(1), key[1])
        # the library writes `state` after every phase it has enqueued: read the lengths back whether or not the call succeeded
        self.k_quant_len, self.k_res_len, self.v_quant_len = state[0], state[1], state[2]
        self.v_res_start, self.v_res_len, self.kv_seq_len = state[3], state[4], state[5]
        if rc:
            _lib.check(rc, "kivi_mf_decode_layer")
        return out


    # ------------------------------------------------------------------ device-resident lengths (hipGraph capture; kivi_amd/graph.py)
    def host_step(self) -> "_lib.MfStep":
        """The six lengths of the NEXT decode step as a kivi_mf_step (include/kivi_hip.h)."""
        R = self.cfg.residual_length
        return _lib.MfStep(Tq=self.k_quant_len, Tv=self.v_quant_len, k_res_len=self.k_res_len, v_res_len=self.v_res_len,
                           v_win_start=self.v_res_start, v_flush=int(self.v_res_len + 1 > R))

    def apply_step(self, hs: "_lib.MfStep") -> None:
        """Take the lengths a driver advanced (kivi_mf_step_advance / the K flush) back into this cache."""
        self.k_quant_len, self.k_res_len, self.v_quant_len = int(hs.Tq), int(hs.k_res_len), int(hs.Tv)
        self.v_res_start, self.v_res_len = int(hs.v_win_start), int(hs.v_res_len)
        self.kv_seq_len = self.k_quant_len + self.k_res_len

    def decode_step_dyn(self, query_states, key_states, value_states, hs: "_lib.MfStep", dev_step: torch.Tensor, out: torch.Tensor,
                        attention_mask: torch.Tensor = None) -> torch.Tensor:
        """The attend phase of one step with the lengths read from `dev_step` on the device (kivi_mf_decode_layer_dyn): what a
        hipGraph captures.  No bookkeeping happens here -- the caller advances `hs` (kivi_mf_step_advance), flushes K when due
        (flush_k) and applies the lengths (apply_step); buffers must be static across replays."""
        B, nh, _, D = query_states.shape
        assert nh == self.nh and B == self.B and D == self.D
        for t in (query_states, key_states, value_states):
            assert t.stride(3) == 1 and t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0, "static 16-byte rows"
        mask_ptr, mask_sb = None, 0
        if attention_mask is not None:          # a static (B, 1, 1, pitch >= capacity) buffer the caller refills every step
            assert attention_mask.dtype == torch.float16 and attention_mask.stride(3) == 1 and attention_mask.shape[3] >= self.kv_seq_len + 1
            mask_ptr, mask_sb = attention_mask.data_ptr(), attention_mask.stride(0)
        assert out.shape == (B, nh, 1, D) and out.dtype == torch.float16 and out.stride(3) == 1
        d, _, key, _, _ = self._desc(nh, query_states.device)
        d.flags = self._flags()
        q, k, v = query_states, key_states, value_states
        rc = _lib.load().kivi_mf_decode_layer_dyn(ctypes.byref(d), ctypes.byref(hs), dev_step.data_ptr(), q.data_ptr(), q.stride(0), q.stride(1),
                                                  nh, k.data_ptr(), k.stride(0), k.stride(1), v.data_ptr(), v.stride(0), v.stride(1),
                                                  mask_ptr, mask_sb, out.data_ptr(), out.stride(0), out.stride(1), key[1])
        _lib.check(rc, "kivi_mf_decode_layer_dyn")
        return out

    def flush_k(self) -> None:
        """The K flush of llama_kivi.py:343-356 for a full residual: R tokens quantised per channel into the layout at token Tq."""
        R = self.cfg.residual_length
        assert self.k_res_len == R
        mfma.kt_pack(self.k_res, self.kt, self.k_quant_len, self.cfg.group_size, self.cfg.k_bits)
        self.k_quant_len += R
        self.k_res_len = 0


def _rows16(x):
    """16-byte loads of whole rows: unit inner stride, 16-byte aligned rows (a copy otherwise)."""
    ok = x.stride(3) == 1 and x.data_ptr() % 16 == 0 and x.stride(0) % 8 == 0 and x.stride(1) % 8 == 0
    return x if ok else x.contiguous()


_MATMUL = None


def _launch_hook():
    global _MATMUL
    if _MATMUL is None:           # (imported lazily: quant.matmul imports this package)
        from .quant import matmul
        _MATMUL = matmul
    return _MATMUL.launch_hook


def make_layer_cache(cfg: KiviConfig, batch: int, num_kv_heads: int, head_dim: int, max_len: int, device,
                     dtype=torch.float16, num_heads: int = None):
    """The cache class for a model shape: the matrix-pipe layout for grouped queries it covers, the hook-state layout
    (KiviLayerCache) otherwise."""
    from .cache import KiviLayerCache
    if num_heads is not None and supported(cfg, head_dim, num_heads, num_kv_heads):
        return KiviLayerCacheMF(cfg, batch, num_kv_heads, head_dim, max_len, device, dtype, num_heads=num_heads)
    return KiviLayerCache(cfg, batch, num_kv_heads, head_dim, max_len, device, dtype)
