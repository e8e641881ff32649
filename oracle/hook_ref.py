"""CPU restatement of the reference attention hook's cache logic (models/llama_kivi.py:314-455).

TEST INFRASTRUCTURE ONLY.  Tuple-based and torch.cat-grown exactly like the reference; the quantise/pack and
the fused GEMV go through the C oracle (oracle/kivi_oracle.c), the fp16 residual matmuls / softmax through
torch on CPU.

PINNED: models/llama_kivi.py cannot be imported as a module here (transformers 4.43 API, flash-attn, CUDA extension),
but oracle/pin_hook.py executes the source of its two attention classes (LlamaAttention_KIVI :19-262,
LlamaFlashAttention_KIVI :264-466) on CPU with only the environment shimmed, and checks this restatement against them:
the 9-tuple bit for bit after the prompt pass and after EVERY decode step (MHA / GQA, 2 / 4 bit, with and without the
additive mask, prompts shorter and longer than the residual length), step outputs within the GEMV bar.  The reference's
outputs are committed as tests/golden/hook_*.npz (tests/test_hook_golden_cpu.py, tests/test_hook_gpu.py).
"""
from __future__ import annotations

import math

import torch

from . import kivi_oracle as O


def repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    if n_rep == 1:
        return x
    B, h, T, D = x.shape
    return x[:, :, None].expand(B, h, n_rep, T, D).reshape(B, h * n_rep, T, D)


def prefill_cache(key_states, value_states, k_bits, v_bits, group_size, residual_length):
    """llama_kivi.py:425-452 -> 9-tuple."""
    T = key_states.shape[-2]
    R = residual_length
    if T % R != 0:
        if T < R:
            k_quant, k_full = None, key_states
        else:
            k_quant = key_states[:, :, :-(T % R), :].contiguous()
            k_full = key_states[:, :, -(T % R):, :].contiguous()
    else:
        k_quant, k_full = key_states, None
    if k_quant is not None:
        kc, ks, km = O.quantize_and_pack_along_last_dim(k_quant.transpose(2, 3).contiguous(), group_size, k_bits)
    else:
        kc = ks = km = None
    if T <= R:
        vc = vs = vm = None
        v_full = value_states
    else:
        v_quant = value_states[:, :, :-R, :].contiguous()
        v_full = value_states[:, :, -R:, :].contiguous()
        vc, vs, vm = O.quantize_and_pack_along_last_dim(v_quant, group_size, v_bits)
    return (kc, k_full, ks, km, vc, v_full, vs, vm, T)


def prefill_attention_eager(query_states, key_states, value_states, attention_mask=None):
    """The eager class's prompt pass (llama_kivi.py:180-183, :222-240): fp16 q k^T / sqrt(D), + additive mask clamped at
    the fp16 minimum, fp32 softmax cast back to fp16, fp16 probs @ V.  `attention_mask` (bsz, 1, q_len, kv_len) carries
    the causal structure (HF builds it); None = no masking at all, as in the reference.  Inputs are fp16 CPU tensors;
    the matmuls run in fp32 and round once to fp16 (what a CUDA fp16 matmul with fp32 accumulate returns)."""
    nh, nh_kv = query_states.shape[1], key_states.shape[1]
    k = repeat_kv(key_states, nh // nh_kv).float()
    v = repeat_kv(value_states, nh // nh_kv).float()
    w = (query_states.float() @ k.transpose(2, 3) / math.sqrt(query_states.shape[-1])).half()
    if attention_mask is not None:
        if tuple(attention_mask.shape) != (query_states.shape[0], 1, query_states.shape[2], key_states.shape[2]):
            raise ValueError("Attention mask should be of size (bsz, 1, q_len, kv_seq_len)")
        w = (w.float() + attention_mask.float()).half()
        w = torch.max(w, torch.tensor(torch.finfo(torch.float16).min, dtype=torch.float16))
    probs = torch.softmax(w.float(), dim=-1).half()
    return (probs.float() @ v).half()


def decode_step(query_states, key_states, value_states, past, k_bits, v_bits, group_size, residual_length,
                attention_mask=None, scores_override=None, return_scores=False):
    """llama_kivi.py:314-399.  query (B,nh,1,D), key/value (B,nh_kv,1,D) -> (attn_output (B,nh,1,D), new 9-tuple).

    Two-stage checks of a fused implementation (tests/test_mfma_gpu.py): `return_scores` also returns the fp16 row the
    reference feeds its softmax (scores / sqrt(D) + mask, :339, :364-372); `scores_override` replaces that row (e.g. by
    the row the implementation under test produced) before the softmax, everything after it (:375-399) unchanged."""
    B, nh, q_len, D = query_states.shape
    nh_kv = key_states.shape[1]
    groups = nh // nh_kv
    kc, k_full, ks, km, vc, v_full, vs, vm, past_len = past
    kv_seq_len = past_len + 1
    if kc is not None:
        att_qkquant = O.bmm_fA_qB_outer(group_size, query_states, kc, ks, km, k_bits)                    # :324
    else:
        att_qkquant = None
    k_full = torch.cat([k_full, key_states], dim=2) if k_full is not None else key_states                # :333-336
    att_qkfull = torch.matmul(query_states.float(), repeat_kv(k_full, groups).transpose(2, 3).float()).half()  # :337
    if att_qkquant is not None:
        attn_weights = torch.cat([att_qkquant, att_qkfull], dim=-1) / math.sqrt(D)                        # :339
    else:
        attn_weights = att_qkfull / math.sqrt(D)
    if k_full.shape[-2] == residual_length:                                                               # :343-356
        assert residual_length % group_size == 0
        kc_n, ks_n, km_n = O.quantize_and_pack_along_last_dim(k_full.transpose(2, 3).contiguous(), group_size, k_bits)
        k_full = None
        if kc is not None:
            kc, ks, km = torch.cat([kc, kc_n], 3), torch.cat([ks, ks_n], 3), torch.cat([km, km_n], 3)
        else:
            kc, ks, km = kc_n, ks_n, km_n
    assert attn_weights.size() == (B, nh, q_len, kv_seq_len)
    if attention_mask is not None:
        attn_weights = attn_weights + attention_mask
        attn_weights = torch.max(attn_weights, torch.tensor(torch.finfo(attn_weights.dtype).min))
    pre_softmax = attn_weights
    if scores_override is not None:
        assert scores_override.shape == attn_weights.shape and scores_override.dtype == attn_weights.dtype
        attn_weights = scores_override
    attn_weights = torch.softmax(attn_weights, dim=-1, dtype=torch.float32).to(query_states.dtype)         # :375
    v_full = torch.cat([v_full, value_states], dim=2)                                                     # :377
    Lv = v_full.shape[-2]
    if vc is None:
        attn_output = torch.matmul(attn_weights.float(), repeat_kv(v_full, groups).float()).half()
    else:
        attn_output = O.bmm_fA_qB_outer(group_size, attn_weights[:, :, :, :-Lv], vc, vs, vm, v_bits)       # :382
        attn_output = attn_output + torch.matmul(attn_weights[:, :, :, -Lv:].float(),
                                                 repeat_kv(v_full, groups).float()).half()                # :384
    if Lv > residual_length:                                                                              # :386-399
        assert Lv == residual_length + 1
        vc_n, vs_n, vm_n = O.quantize_and_pack_along_last_dim(v_full[:, :, :1, :].contiguous(), group_size, v_bits)
        v_full = v_full[:, :, 1:, :].contiguous()
        if vc is not None:
            vc, vs, vm = torch.cat([vc, vc_n], 2), torch.cat([vs, vs_n], 2), torch.cat([vm, vm_n], 2)
        else:
            vc, vs, vm = vc_n, vs_n, vm_n
    new_past = (kc, k_full, ks, km, vc, v_full, vs, vm, kv_seq_len)
    if return_scores:
        return attn_output, new_past, pre_softmax
    return attn_output, new_past
