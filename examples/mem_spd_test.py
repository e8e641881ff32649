#!/usr/bin/env python3
"""End-to-end decode throughput / peak memory of a Llama-shaped model with the KIVI attention hook.

Counterpart of the reference's mem_spd_test.py (:8-12, :53-70: Llama-2-7b, k = v = 2 bit, g = 32, one prompt batch,
N new tokens, ms per generate + torch.cuda.max_memory_allocated) for a box without network: the weights are RANDOM
(Llama-2-7B architecture by default), so the generated tokens mean nothing -- time and memory do.  The decoder is
kivi_amd.llama.LlamaForCausalLM_KIVI (plain-torch embedding / RMSNorm / projections / SwiGLU MLP / lm_head around
kivi_amd.attention.LlamaAttention_KIVI, the drop-in for models/llama_kivi.py).  `--baseline` runs the same model with an fp16 KV cache and torch SDPA instead.

    python examples/mem_spd_test.py --batch 32 --prompt 2048 --gen 512          # BASELINE.json configs[2]
    python examples/mem_spd_test.py --batch 32 --prompt 2048 --gen 512 --baseline
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class Fp16Attention(nn.Module):
    """The un-quantised baseline: preallocated fp16 KV cache + torch SDPA."""

    def __init__(self, cfg, layer_idx):
        super().__init__()
        from kivi_amd.attention import LlamaAttention_KIVI
        inner = LlamaAttention_KIVI(cfg, layer_idx)   # reuse the projections / rotary code
        self.inner = inner
        self.cap = cfg.kivi_max_cache_len

    def forward(self, hidden_states, past_key_value=None, use_cache=True, **kw):
        m = self.inner
        bsz, q_len, _ = hidden_states.size()
        q = m.q_proj(hidden_states).view(bsz, q_len, m.num_heads, m.head_dim).transpose(1, 2)
        k = m.k_proj(hidden_states).view(bsz, q_len, m.num_key_value_heads, m.head_dim).transpose(1, 2)
        v = m.v_proj(hidden_states).view(bsz, q_len, m.num_key_value_heads, m.head_dim).transpose(1, 2)
        past_len = 0 if past_key_value is None else past_key_value[2]
        pos = torch.arange(past_len, past_len + q_len, device=q.device)[None].expand(bsz, -1)
        q, k = m._rope(q, k, pos)
        if past_key_value is None:
            kc = torch.empty((bsz, m.num_key_value_heads, self.cap, m.head_dim), device=q.device, dtype=q.dtype)
            vc = torch.empty_like(kc)
        else:
            kc, vc, _ = past_key_value
        kc[:, :, past_len:past_len + q_len] = k
        vc[:, :, past_len:past_len + q_len] = v
        n = past_len + q_len
        o = F.scaled_dot_product_attention(q, kc[:, :, :n], vc[:, :, :n], is_causal=(q_len > 1),
                                           enable_gqa=(m.num_heads != m.num_key_value_heads))
        o = o.transpose(1, 2).reshape(bsz, q_len, m.num_heads * m.head_dim)
        return m.o_proj(o), None, (kc, vc, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=2048)
    ap.add_argument("--gen", type=int, default=512)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=32)
    ap.add_argument("--intermediate", type=int, default=11008)
    ap.add_argument("--vocab", type=int, default=32000)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--group", type=int, default=32)
    ap.add_argument("--residual", type=int, default=32)
    ap.add_argument("--baseline", action="store_true", help="fp16 KV cache + torch SDPA instead of the KIVI hook")
    ap.add_argument("--graphs", action="store_true", help="replay the dense part of every decode step from hipGraphs")
    ap.add_argument("--repeats", type=int, default=1, help="whole generations (prompt pass + gen steps) timed back to back")
    ap.add_argument("--recipe", action="store_true",
                    help="the reference's mem_spd_test.py:8-12, :53-70: batch 96, prompt 160, 338 new tokens, residual 128, 3 repeats")
    args = ap.parse_args()
    if args.recipe:
        args.batch, args.prompt, args.gen, args.residual, args.repeats = 96, 160, 338, 128, 3
    dev = torch.device("cuda:0")
    cfg = SimpleNamespace(hidden_size=args.hidden, num_attention_heads=args.heads, num_key_value_heads=args.kv_heads,
                          num_hidden_layers=args.layers, intermediate_size=args.intermediate, vocab_size=args.vocab,
                          max_position_embeddings=args.prompt + args.gen + 1, rope_theta=10000.0, rms_norm_eps=1e-5, tie_word_embeddings=False,
                          k_bits=args.bits, v_bits=args.bits, group_size=args.group, residual_length=args.residual,
                          kivi_max_cache_len=args.prompt + args.gen + 1, attention_bias=False)
    torch.manual_seed(0)
    from kivi_amd.llama import LlamaForCausalLM_KIVI
    from kivi_amd.attention import LlamaAttention_KIVI
    with torch.device(dev):
        torch.set_default_dtype(torch.float16)
        model = LlamaForCausalLM_KIVI(cfg, Fp16Attention if args.baseline else LlamaAttention_KIVI)
        torch.set_default_dtype(torch.float32)
    for p in model.parameters():      # small weights keep the random activations finite through 32 layers
        if p.dim() > 1:
            p.data.normal_(0.0, 0.02)
    weights = sum(p.numel() * p.element_size() for p in model.parameters())
    ids = torch.randint(0, args.vocab, (args.batch, args.prompt), device=dev)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t_prefill = t_dec = 0.0
    t_all0 = time.time()
    for rep in range(args.repeats):
        t0 = time.time()
        logits, pasts = model(ids)
        torch.cuda.synchronize()
        t_prefill += time.time() - t0
        tok = logits.argmax(-1)
        if args.graphs:
            assert not args.baseline, "--graphs drives the KIVI hook (the fp16 baseline cache grows in shape every step)"
            torch.cuda.synchronize()    # (the graphs are built by the first graphed step: whole-step graphs for matrix-pipe caches)
        t1 = time.time()
        if args.graphs:
            model.decode_graphed(tok, pasts, args.prompt, args.gen)
        else:
            for _ in range(args.gen):
                logits, pasts = model(tok, pasts)
                tok = logits.argmax(-1)
        torch.cuda.synchronize()
        t_dec += time.time() - t1
    t_generate = (time.time() - t_all0) / args.repeats
    t_prefill /= args.repeats
    t_dec /= args.repeats
    if args.baseline:
        kv = kv_alloc = sum(p[0].numel() * 2 * 2 for p in pasts)
    else:
        kv = sum(p.layer.nbytes() for p in pasts)                 # what the reference's 9-tuples would hold
        kv_alloc = sum(p.layer.allocated_bytes() for p in pasts)  # incl. page / window slack of the in-place cache
    print(json.dumps({
        "mode": "fp16 KV + SDPA" if args.baseline else f"KIVI {args.bits}-bit g={args.group} R={args.residual}" + (" + hipGraph" if args.graphs else ""),
        "hipgraph_eager_captures_replays": getattr(model, "_last_graph_stats", None),
        "model": f"llama-shaped random weights: L={args.layers} h={args.hidden} nh={args.heads}/{args.kv_heads} ffn={args.intermediate}",
        "batch": args.batch, "prompt": args.prompt, "gen": args.gen,
        "repeats": args.repeats, "generate_ms": round(1e3 * t_generate, 1),     # what the reference prints as "used time"
        "prefill_s": round(t_prefill, 3), "decode_ms_per_step": round(1e3 * t_dec / args.gen, 3),
        "decode_tokens_per_s": round(args.batch * args.gen / t_dec, 1),
        "weights_bytes": weights, "kv_cache_bytes": kv, "kv_cache_allocated_bytes": kv_alloc,
        "max_memory_allocated": torch.cuda.max_memory_allocated()}))


if __name__ == "__main__":
    main()
