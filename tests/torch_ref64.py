"""A second, fully independent reference for the FULL-COVERAGE GPU parity tests (tests/test_fullcover_gpu.py): plain torch
on the GPU, fp64 arithmetic, no HIP kernel of this repo and no oracle code anywhere in it.

The CPU oracle (oracle/kivi_oracle.c) needs seconds per (batch row, kv head) unit, so the full-size tests hold 2-4 of up to 1024
units to it.  This module covers ALL units of the same launches:

  * quantise + pack: the reference's own op sequence (quant/new_pack.py:30-48 == :217-252: `mx - mn`, `/ max_int`, `x - mn`, `div_`,
    `clamp_`, `round_`, LSB-first OR) with every fp16 rounding applied explicitly to exact fp64 intermediates (a fp16 +, -, / rounded
    once from the exact value is what the fp16 op returns; 53 bits hold every intermediate exactly), so the result does not depend on
    how torch implements half arithmetic on this device.  Bit-exact bar.
  * unpack: `(code[idx // fpi] >> (idx % fpi) * bits) & mask` (quant/new_pack.py:110-129).
  * fused GEMV: `sum_k fA[k] * (scale * code + zero)` (quant/csrc/gemv_cuda.cu:401-426, head mapping :361-365) in fp64 -- the exact
    real-number value of what the CUDA kernel accumulates in fp32 -- rounded once to fp16.  GEMV bar 1e-3.
  * one decode step of the hook (models/llama_kivi.py:314-399) on top of those, vectorised over every unit, incl. the cache update
    (K flush :343-356, V flush :386-399): pre-softmax rows, output and the new 9-tuple.
"""
import math

import torch


def _h(x64: torch.Tensor) -> torch.Tensor:
    """exact fp64 value -> fp16, round to nearest even (through fp32: innocuous for sums / differences / quotients of fp16 numbers
    held exactly in fp64, see the module docstring), back to fp64 for the next exact step."""
    return x64.to(torch.float32).to(torch.float16)


def quant_pack_lastdim(x: torch.Tensor, g: int, bits: int):
    """new_pack.py:30-48 on a (..., T) fp16 tensor: codes int32 (..., T / fpi), scale, mn fp16 (..., T / g)."""
    assert x.dtype == torch.float16 and x.shape[-1] % g == 0
    fpi = 32 // bits
    maxq = float(2 ** bits - 1)
    lead, T = x.shape[:-1], x.shape[-1]
    xd = x.reshape(-1, T // g, g).double()
    mn = xd.amin(-1, keepdim=True)
    mx = xd.amax(-1, keepdim=True)
    scale = _h(_h(mx - mn).double() / maxq)                           # fp16((mx - mn)) / max_int, rounded to fp16
    d = _h(xd - mn).double()                                          # data - mn
    q = _h(d / scale.double())                                        # data.div_(scale)
    q = q.clamp(0, maxq).round().to(torch.int32).reshape(-1, T // fpi, fpi)    # clamp_ / round_ (half to even) / int32
    code = torch.zeros(q.shape[:-1], dtype=torch.int32, device=x.device)
    for i in range(fpi):                                               # LSB first; the top code lands in the sign bit
        code |= q[..., i] << (bits * i)
    return (code.reshape(*lead, T // fpi), scale.reshape(*lead, T // g), _h(mn).reshape(*lead, T // g))


def unpack_lastdim(code: torch.Tensor, bits: int) -> torch.Tensor:
    """new_pack.py:110-129 along the last dim: int32 words (..., W) -> codes (..., W * fpi) as int32."""
    fpi = 32 // bits
    sh = torch.arange(fpi, device=code.device, dtype=torch.int32) * bits
    return ((code.unsqueeze(-1) >> sh) & (2 ** bits - 1)).reshape(*code.shape[:-1], code.shape[-1] * fpi)


def dequant64(code, scale, mn, g: int, bits: int) -> torch.Tensor:
    """scale * code + zero, exact, fp64 (gemv_cuda.cu:407-413), groups of g along the last dim."""
    q = unpack_lastdim(code, bits).double()
    lead, T = q.shape[:-1], q.shape[-1]
    return (q.view(*lead, T // g, g) * scale.double().unsqueeze(-1) + mn.double().unsqueeze(-1)).view(*lead, T)


def _expand_heads(x, ratio):                                           # (b, nh_kv, ...) -> (b, nh, ...), hk = h // ratio (:361-365)
    return x if ratio == 1 else x.repeat_interleave(ratio, dim=1)


def scores64(q, kc, ks, km, g: int, bits: int, chunk: int = 4) -> torch.Tensor:
    """qK^T over the packed keys: q (B, nh, 1, D); K_code_T (B, nh_kv, D, T / fpi), K_scale_T / K_mn_T (B, nh_kv, D, T / g)
    -> (B, nh, 1, T) fp16."""
    B, nh = q.shape[:2]
    ratio = nh // kc.shape[1]
    out = []
    for b0 in range(0, B, chunk):
        sl = slice(b0, b0 + chunk)
        kd = _expand_heads(dequant64(kc[sl], ks[sl], km[sl], g, bits), ratio)              # (b, nh, D, T)
        out.append(_h(torch.matmul(q[sl].double(), kd)))
    return torch.cat(out, 0)


def output64(a, vc, vs, vm, g: int, bits: int, chunk: int = 4) -> torch.Tensor:
    """sV over the packed values: a (B, nh, 1, Tv) fp16 (any strides); V_code (B, nh_kv, Tv, D / fpi) -> (B, nh, 1, D) fp16."""
    B, nh = a.shape[:2]
    ratio = nh // vc.shape[1]
    out = []
    for b0 in range(0, B, chunk):
        sl = slice(b0, b0 + chunk)
        vd = _expand_heads(dequant64(vc[sl], vs[sl], vm[sl], g, bits), ratio)              # (b, nh, Tv, D)
        out.append(_h(torch.matmul(a[sl].double(), vd)))
    return torch.cat(out, 0)


def prefill_cache(k, v, k_bits, v_bits, g, R, chunk: int = 2):
    """llama_kivi.py:425-452 -> the 9-tuple, on the device of k / v."""
    T = k.shape[2]
    nq = (T // R) * R
    kc = ks = km = None
    if nq:
        parts = [quant_pack_lastdim(k[b0:b0 + chunk, :, :nq].transpose(2, 3).contiguous(), g, k_bits) for b0 in range(0, k.shape[0], chunk)]
        kc, ks, km = (torch.cat([p[i] for p in parts], 0) for i in range(3))
    k_full = k[:, :, nq:].contiguous() if T > nq else None
    vc = vs = vm = None
    if T > R:
        parts = [quant_pack_lastdim(v[b0:b0 + chunk, :, :T - R].contiguous(), g, v_bits) for b0 in range(0, v.shape[0], chunk)]
        vc, vs, vm = (torch.cat([p[i] for p in parts], 0) for i in range(3))
        v_full = v[:, :, T - R:].contiguous()
    else:
        v_full = v
    return (kc, k_full, ks, km, vc, v_full, vs, vm, T)


def decode_step(q, kn, vn, past, k_bits, v_bits, g, R, attention_mask=None, scores_override=None):
    """llama_kivi.py:314-399 for every unit at once.  Returns (attn_output fp16 (B, nh, 1, D), new 9-tuple, the fp16 rows fed to the
    softmax).  `scores_override`: rows to feed the softmax instead (stage B of the two-stage check, tests/test_mfma_gpu.py)."""
    B, nh, _, D = q.shape
    ratio = nh // kn.shape[1]
    kc, k_full, ks, km, vc, v_full, vs, vm, past_len = past
    att_q = scores64(q, kc, ks, km, g, k_bits) if kc is not None else None                   # :324
    k_full = torch.cat([k_full, kn], 2) if k_full is not None else kn                         # :333-336
    att_f = _h(torch.matmul(q.double(), _expand_heads(k_full, ratio).transpose(2, 3).double()))   # :337
    w = (torch.cat([att_q, att_f], -1) if att_q is not None else att_f) / math.sqrt(D)         # :339 (fp16 tensor / python float)
    if k_full.shape[2] == R:                                                                   # :343-356
        kc_n, ks_n, km_n = quant_pack_lastdim(k_full.transpose(2, 3).contiguous(), g, k_bits)
        k_full = None
        if kc is not None:
            kc, ks, km = torch.cat([kc, kc_n], 3), torch.cat([ks, ks_n], 3), torch.cat([km, km_n], 3)
        else:
            kc, ks, km = kc_n, ks_n, km_n
    if attention_mask is not None:                                                             # :364-372
        w = w + attention_mask
        w = torch.max(w, torch.tensor(torch.finfo(w.dtype).min, device=w.device, dtype=w.dtype))
    pre = w
    if scores_override is not None:
        w = scores_override
    p = torch.softmax(w, dim=-1, dtype=torch.float32).to(torch.float16)                        # :375
    v_full = torch.cat([v_full, vn], 2)                                                        # :377
    Lv = v_full.shape[2]
    win = _h(torch.matmul(p[..., -Lv:].double(), _expand_heads(v_full, ratio).double()))
    if vc is None:
        out = win                                                                              # :380
    else:
        out = output64(p[..., :-Lv], vc, vs, vm, g, v_bits) + win                              # :382-384 (fp16 add)
    if Lv > R:                                                                                 # :386-399
        vc_n, vs_n, vm_n = quant_pack_lastdim(v_full[:, :, :1].contiguous(), g, v_bits)
        v_full = v_full[:, :, 1:].contiguous()
        if vc is not None:
            vc, vs, vm = torch.cat([vc, vc_n], 2), torch.cat([vs, vs_n], 2), torch.cat([vm, vm_n], 2)
        else:
            vc, vs, vm = vc_n, vs_n, vm_n
    return out, (kc, k_full, ks, km, vc, v_full, vs, vm, past_len + 1), pre
