"""Pure-PyTorch CPU port of the reference's fake-quant path (the CPU timing baseline of bench.py).

TEST / BASELINE INFRASTRUCTURE ONLY.  Restates quant_and_pack_{k,v}cache -> unpack_and_dequant_* ->
torch.matmul (quant/new_pack.py:8-83, procedure of quant/test.py:187-195) with vectorised torch ops so the
host's cores are actually used (the reference's pack_tensor is an O(T) Python loop, new_pack.py:100-106).
Checked bit-for-bit against the C oracle in tests/test_cpu_baseline.py.
"""
from __future__ import annotations

import torch


def _quantize_lastdim(x: torch.Tensor, group_size: int, bits: int):
    """new_pack.py:36-44 on groups of the last dim: returns int32 codes (same shape), scale, mn (..., ng, 1)."""
    shape = x.shape
    maxq = 2 ** bits - 1
    data = x.reshape(shape[:-1] + (shape[-1] // group_size, group_size))
    mn = data.amin(dim=-1, keepdim=True)
    mx = data.amax(dim=-1, keepdim=True)
    scale = (mx - mn) / maxq
    q = ((data - mn) / scale).clamp_(0, maxq).round_().nan_to_num_(0).to(torch.int32)
    return q.reshape(shape), scale, mn


def _pack_lastdim(codes: torch.Tensor, bits: int) -> torch.Tensor:
    fpi = 32 // bits
    c = codes.reshape(codes.shape[:-1] + (codes.shape[-1] // fpi, fpi)).to(torch.int64)
    shifts = (torch.arange(fpi, dtype=torch.int64) * bits)
    word = (c << shifts).sum(dim=-1)                       # fields do not overlap: sum == OR
    return ((word + 2 ** 31) % 2 ** 32 - 2 ** 31).to(torch.int32)


def _unpack_lastdim(code: torch.Tensor, bits: int) -> torch.Tensor:
    fpi = 32 // bits
    shifts = (torch.arange(fpi, dtype=torch.int32) * bits)
    return ((code.unsqueeze(-1) >> shifts) & (2 ** bits - 1)).reshape(code.shape[:-1] + (code.shape[-1] * fpi,))


def quant_pack_lastdim(x: torch.Tensor, group_size: int, bits: int):
    q, scale, mn = _quantize_lastdim(x, group_size, bits)
    return _pack_lastdim(q, bits), scale.squeeze(-1), mn.squeeze(-1)


def dequant_lastdim(code: torch.Tensor, scale: torch.Tensor, mn: torch.Tensor, group_size: int, bits: int, ws: dict = None, tag: str = ""):
    """new_pack.py:78-83: fp16(fp16(q) * scale) + mn with torch's per-op fp16 rounding (unpack -> fp16 mul -> fp16 add, un-fused).
    `ws` (bench.py's timing loop): a dict that keeps the intermediates of the previous call -- the same ops write into them (out=)
    instead of allocating ~0.5 GB of fresh pages per call, which on a shared host is what the timing then mostly measures."""
    fpi = 32 // bits
    if ws is None:
        q = _unpack_lastdim(code, bits).to(torch.float16)
        shape = q.shape
        q = q.reshape(shape[:-1] + (shape[-1] // group_size, group_size))
        return (q * scale.unsqueeze(-1) + mn.unsqueeze(-1)).reshape(shape)
    shape = code.shape[:-1] + (code.shape[-1] * fpi,)
    key = (tag, tuple(shape))
    if key not in ws:
        ws[key] = (torch.empty(code.shape + (fpi,), dtype=torch.int32), torch.empty(shape, dtype=torch.float16),
                   torch.empty(shape, dtype=torch.float16), torch.arange(fpi, dtype=torch.int32) * bits)
    qi, qh, out, shifts = ws[key]
    torch.bitwise_right_shift(code.unsqueeze(-1), shifts, out=qi)
    qi.bitwise_and_(2 ** bits - 1)
    qh.view(qi.shape).copy_(qi)                                                           # int32 -> fp16
    grp = shape[:-1] + (shape[-1] // group_size, group_size)
    torch.mul(qh.view(grp), scale.unsqueeze(-1), out=out.view(grp))                         # fp16 multiply, rounded
    out.view(grp).add_(mn.unsqueeze(-1))                                                  # fp16 add, rounded
    return out


def fakequant_decode_layer(q, a, k, v, group_size: int, bits: int, ws: dict = None):
    """One layer of the reference's CPU procedure for a decode step: pack K (per channel) and V (per token),
    unpack+dequantise both, then the two GEMVs.  q (B,nh,1,D), a (B,nh,1,T), k/v (B,nh,T,D).
    Returns (scores (B,nh,1,T), out (B,nh,1,D)) and the per-stage seconds.  `ws`: see dequant_lastdim; with it the packed cache
    of the first call is kept too (a decode step finds its cache packed) and pack_s is 0 from the second call on."""
    import time
    t0 = time.perf_counter()
    if ws is not None and "packed" in ws:
        kc, ks, km, vc, vs, vm = ws["packed"]
    else:
        kc, ks, km = quant_pack_lastdim(k.transpose(2, 3).contiguous(), group_size, bits)
        vc, vs, vm = quant_pack_lastdim(v, group_size, bits)
        if ws is not None:
            ws["packed"] = (kc, ks, km, vc, vs, vm)
    t1 = time.perf_counter()
    k_hat_T = dequant_lastdim(kc, ks, km, group_size, bits, ws, "k")  # (B,nh,D,T)
    v_hat = dequant_lastdim(vc, vs, vm, group_size, bits, ws, "v")    # (B,nh,T,D)
    t2 = time.perf_counter()
    if ws is None:
        scores = torch.matmul(q.float(), k_hat_T.float()).half()
        out = torch.matmul(a.float(), v_hat.float()).half()
    else:
        if "f32" not in ws:
            ws["f32"] = (torch.empty(k_hat_T.shape, dtype=torch.float32), torch.empty(v_hat.shape, dtype=torch.float32))
        kf, vf = ws["f32"]
        kf.copy_(k_hat_T)
        vf.copy_(v_hat)
        scores = torch.matmul(q.float(), kf).half()
        out = torch.matmul(a.float(), vf).half()
    t3 = time.perf_counter()
    return scores, out, dict(pack_s=t1 - t0, dequant_s=t2 - t1, gemv_s=t3 - t2)
