"""A Llama / Mistral-shaped decoder around the KIVI attention hook -- the callers' side of the hot path.

Counterpart of the reference's LlamaForCausalLM_KIVI / MistralForCausalLM_KIVI wrappers (models/llama_kivi.py:564-1000,
models/mistral_kivi.py:673-1100) reduced to what decoding needs: token embedding, pre-norm decoder blocks
(RMSNorm, LlamaAttention_KIVI, SwiGLU MLP), final norm, lm_head, greedy generate.  Parameter names follow the Hugging
Face checkpoints (model.embed_tokens, model.layers.N.self_attn.q_proj, ..., lm_head), so `load_state_dict` /
`from_pretrained` take an unmodified Llama-2 / Llama-3 / Mistral checkpoint directory; everything outside the attention
block is plain torch (rocBLAS / hipBLASLt GEMMs).  The reference patches k_bits / v_bits / group_size /
residual_length onto the HF config (README.md:72-75); the same four fields are read here.
"""
from __future__ import annotations

import glob
import json
import os
from types import SimpleNamespace
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .attention import LlamaAttention_KIVI


def make_config(d: dict, k_bits: int = 2, v_bits: int = 2, group_size: int = 32, residual_length: int = 32,
                max_cache_len: Optional[int] = None) -> SimpleNamespace:
    """HF config.json fields (+ the four KIVI fields) -> the namespace the modules read."""
    hidden, heads = d["hidden_size"], d["num_attention_heads"]
    return SimpleNamespace(
        hidden_size=hidden, num_attention_heads=heads, num_key_value_heads=d.get("num_key_value_heads", heads),
        num_hidden_layers=d["num_hidden_layers"], intermediate_size=d["intermediate_size"], vocab_size=d["vocab_size"],
        max_position_embeddings=d.get("max_position_embeddings", 4096), rope_theta=d.get("rope_theta", 10000.0),
        rms_norm_eps=d.get("rms_norm_eps", 1e-5), attention_bias=d.get("attention_bias", False),
        rope_scaling=d.get("rope_scaling"), head_dim=d.get("head_dim"), sliding_window=d.get("sliding_window"),
        tie_word_embeddings=d.get("tie_word_embeddings", False),
        k_bits=d.get("k_bits", k_bits), v_bits=d.get("v_bits", v_bits), group_size=d.get("group_size", group_size),
        residual_length=d.get("residual_length", residual_length),
        kivi_max_cache_len=max_cache_len)   # opt-in reservation; by default the cache starts prompt-sized and doubles


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.eps = eps

    def forward(self, x):
        return F.rms_norm(x, (x.shape[-1],), self.weight, self.eps)


class MLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.gate_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.up_proj = nn.Linear(cfg.hidden_size, cfg.intermediate_size, bias=False)
        self.down_proj = nn.Linear(cfg.intermediate_size, cfg.hidden_size, bias=False)

    def forward(self, x):
        return self.down_proj(F.silu(self.gate_proj(x)) * self.up_proj(x))


class DecoderLayer_KIVI(nn.Module):
    def __init__(self, cfg, layer_idx: int, attention_cls=LlamaAttention_KIVI):
        super().__init__()
        self.self_attn = attention_cls(cfg, layer_idx)
        self.mlp = MLP(cfg)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)

    def forward(self, x, past, attention_mask=None):
        a, _, past = self.self_attn(self.input_layernorm(x), attention_mask=attention_mask, past_key_value=past,
                                    use_cache=True)
        x = x + a
        return x + self.mlp(self.post_attention_layernorm(x)), past


class _Body(nn.Module):
    def __init__(self, cfg, attention_cls):
        super().__init__()
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([DecoderLayer_KIVI(cfg, i, attention_cls) for i in range(cfg.num_hidden_layers)])
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)


class LlamaForCausalLM_KIVI(nn.Module):
    """`past_key_values` is a list with one entry per layer: None before the prompt pass, afterwards the 9-tuple of
    models/llama_kivi.py:454-455 (here the lazy KiviCacheTuple over the in-place cache)."""

    def __init__(self, config, attention_cls=LlamaAttention_KIVI):
        super().__init__()
        self.config = config
        self.model = _Body(config, attention_cls)
        self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        if getattr(config, "tie_word_embeddings", False):
            self.lm_head.weight = self.model.embed_tokens.weight

    @torch.no_grad()
    def forward(self, input_ids: torch.LongTensor, past_key_values: Optional[List] = None, attention_mask=None,
                last_token_only: bool = True):
        """logits (B, 1 or T, vocab), new past_key_values.  `attention_mask`: the reference's additive (B, 1, 1, kv_len)
        fp16 mask for decode steps (llama_kivi.py:364-372); the prompt pass is causal over equal-length prompts."""
        pasts = past_key_values or [None] * len(self.model.layers)
        x = self.model.embed_tokens(input_ids)
        new = []
        for layer, past in zip(self.model.layers, pasts):
            x, p = layer(x, past, attention_mask if past is not None else None)
            new.append(p)
        if last_token_only:
            x = x[:, -1:]
        return self.lm_head(self.model.norm(x)), new

    @torch.no_grad()
    def generate(self, input_ids: torch.LongTensor, max_new_tokens: int) -> torch.LongTensor:
        """Greedy decoding of equal-length prompts (the recipe of the reference's mem_spd_test.py / example.py)."""
        logits, pasts = self.forward(input_ids)
        out = [input_ids]
        tok = logits.argmax(-1)
        for _ in range(max_new_tokens):
            out.append(tok)
            logits, pasts = self.forward(tok, pasts)
            tok = logits.argmax(-1)
        return torch.cat(out, dim=1)

    # ------------------------------------------------------------------ hipGraph decode
    # The dense part of a decode step is ~30 small launches per layer; in eager mode the host needs longer to enqueue
    # them than the GPU to run them.  Everything with static shapes is captured once per batch size into hipGraphs
    # (torch.cuda.CUDAGraph): per layer one graph from the block input to the rotated q / k / v, one from the attention
    # output to the block output, plus embedding and head; the KIVI step between them stays one eager
    # kivi_decode_layer call per layer (its lengths change every step).  Same kernels, same results as forward().
    # Round 4: when every layer's cache is in the matrix-pipe layout the attention launches read their lengths from device
    # memory (kivi_amd/graph.py), so the WHOLE step -- dense parts and attention of all layers -- is ONE graph (whole=True).
    def _build_graphs(self, B: int, device, whole: bool = False):
        cfg = self.config
        nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
        D, H = self.model.layers[0].self_attn.head_dim, cfg.hidden_size
        dt = self.lm_head.weight.dtype
        g = SimpleNamespace(B=B, tok=torch.zeros((B, 1), dtype=torch.long, device=device),
                            cos=torch.zeros((1, 1, 1, D), dtype=dt, device=device),
                            sin=torch.zeros((1, 1, 1, D), dtype=dt, device=device),
                            x=[torch.zeros((B, 1, H), dtype=dt, device=device) for _ in range(len(self.model.layers) + 1)],
                            attn=torch.zeros((B, nh, 1, D), dtype=dt, device=device), qkv=[], pre=[], post=[], whole=whole)

        def rot(t):
            return torch.cat((-t[..., D // 2:], t[..., : D // 2]), dim=-1)

        def pre(i):
            layer = self.model.layers[i]
            a = layer.self_attn
            if i == 0:
                g.x[0].copy_(self.model.embed_tokens(g.tok))
            h = layer.input_layernorm(g.x[i])
            q = a.q_proj(h).view(B, 1, nh, D).transpose(1, 2)
            k = a.k_proj(h).view(B, 1, nkv, D).transpose(1, 2)
            v = a.v_proj(h).view(B, 1, nkv, D).transpose(1, 2)
            g.qkv[i][0].copy_(q * g.cos + rot(q) * g.sin)
            g.qkv[i][1].copy_(k * g.cos + rot(k) * g.sin)
            g.qkv[i][2].copy_(v)

        def post(i):
            layer = self.model.layers[i]
            x = g.x[i] + layer.self_attn.o_proj(g.attn.transpose(1, 2).reshape(B, 1, nh * D))
            g.x[i + 1].copy_(x + layer.mlp(layer.post_attention_layernorm(x)))
            if i == len(self.model.layers) - 1:
                g.tok.copy_(self.lm_head(self.model.norm(g.x[i + 1])).argmax(-1))

        for _ in self.model.layers:
            g.qkv.append((torch.zeros((B, nh, 1, D), dtype=dt, device=device),
                          torch.zeros((B, nkv, 1, D), dtype=dt, device=device),
                          torch.zeros((B, nkv, 1, D), dtype=dt, device=device)))
        g.pre_fn, g.post_fn = pre, post
        if whole:               # captured by kivi_amd.graph.GraphedDecode together with the attention launches
            return g
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):          # warm-up outside capture (library workspaces, autotuning)
            for i in range(len(self.model.layers)):
                pre(i)
                post(i)
        torch.cuda.current_stream(device).wait_stream(side)
        pool = None
        for i in range(len(self.model.layers)):
            for fn, dst in ((pre, g.pre), (post, g.post)):
                cg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(cg, pool=pool):
                    fn(i)
                pool = pool or cg.pool()
                dst.append(cg)
        g.tok.zero_()
        return g

    def prepare_graphs(self, batch: int, device, whole: bool = False) -> None:
        """Capture the decode graphs for this batch size now (otherwise on the first graphed step)."""
        g = getattr(self, "_graphs", None)
        if g is None or g.B != batch or g.whole != whole:
            self._graphs = self._build_graphs(batch, device, whole)

    @torch.no_grad()
    def decode_graphed(self, tok: torch.LongTensor, past_key_values: List, position: int, steps: int) -> torch.LongTensor:
        """`steps` greedy decode steps from token `tok` (B, 1) at position `position` with the dense part replayed from
        hipGraphs; the caches in `past_key_values` are advanced in place.  Returns the (B, steps) tokens fed to the model
        (tok first); the token following them is left in the graph's token buffer (`self._graphs.tok`)."""
        from .attention import kivi_attention_decode
        from .cache_mf import KiviLayerCacheMF
        caches = [p.layer for p in past_key_values]
        whole = all(isinstance(c, KiviLayerCacheMF) for c in caches)
        self.prepare_graphs(tok.shape[0], tok.device, whole)
        g = self._graphs
        attn0 = self.model.layers[0].self_attn
        g.tok.copy_(tok)
        out = []
        if whole:
            from .graph import GraphedDecode, MfStepDriver
            # the driver and its captured graph live on the model, for the caches (held weakly: MfStepDriver.serves) and the static
            # buffers they were built for: a second decode_graphed call over the same caches replays the graph it already has instead
            # of paying an eager step and a 32-layer capture again (only a new geometry class or a reallocated buffer re-captures:
            # MfStepDriver.prepare).  Nothing here keeps a finished request's KV cache alive.
            st = getattr(self, "_graphed", None)
            if st is not None and st[0] is g and st[1].serves(caches):
                drv, gd = st[1], st[2]
                drv.resync()
            else:
                drv = MfStepDriver(caches)

                def body():
                    for i in range(len(self.model.layers)):
                        g.pre_fn(i)
                        drv.enqueue(i, *g.qkv[i], g.attn)
                        g.post_fn(i)

                gd = GraphedDecode(drv, body)
                self._graphed = (g, drv, gd)
            for _ in range(steps):
                out.append(g.tok.clone())
                freqs = position * attn0.inv_freq.float()
                emb = torch.cat((freqs, freqs), dim=-1)
                g.cos.copy_(emb.cos().view(1, 1, 1, -1))
                g.sin.copy_(emb.sin().view(1, 1, 1, -1))
                gd.step()
                position += 1
            self._last_graph_stats = (gd.eager, gd.captures, gd.replays)
            return torch.cat(out, dim=1)
        for _ in range(steps):
            out.append(g.tok.clone())
            freqs = position * attn0.inv_freq.float()
            emb = torch.cat((freqs, freqs), dim=-1)
            g.cos.copy_(emb.cos().view(1, 1, 1, -1))
            g.sin.copy_(emb.sin().view(1, 1, 1, -1))
            for i in range(len(self.model.layers)):
                g.pre[i].replay()
                q, k, v = g.qkv[i]
                kivi_attention_decode(q, k, v, caches[i], out=g.attn)
                g.post[i].replay()
            position += 1
        return torch.cat(out, dim=1)

    @torch.no_grad()
    def generate_graphed(self, input_ids: torch.LongTensor, max_new_tokens: int) -> torch.LongTensor:
        """generate() with the dense part of every decode step replayed from hipGraphs (see _build_graphs)."""
        logits, pasts = self.forward(input_ids)
        new = self.decode_graphed(logits.argmax(-1), pasts, input_ids.shape[1], max_new_tokens)
        self._graphed = None        # the caches of this request die with it: so does the graph captured over them
        return torch.cat([input_ids, new], dim=1)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda", dtype=torch.float16, **kivi):
        """`path`: a local HF checkpoint directory (config.json + *.safetensors).  `kivi`: k_bits, v_bits, group_size,
        residual_length, max_cache_len."""
        from safetensors.torch import load_file
        cfg = make_config(json.load(open(os.path.join(path, "config.json"))), **kivi)
        with torch.device(device):
            prev = torch.get_default_dtype()
            torch.set_default_dtype(dtype)
            try:
                model = cls(cfg)
            finally:
                torch.set_default_dtype(prev)
        state = {}
        for f in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
            state.update(load_file(f, device=str(device)))
        missing, unexpected = model.load_state_dict(state, strict=False)
        missing = [m for m in missing if "inv_freq" not in m and not (cfg.tie_word_embeddings and m == "lm_head.weight")]
        if missing:
            raise KeyError(f"checkpoint lacks {missing[:5]}{' ...' if len(missing) > 5 else ''}")
        return model


class MistralForCausalLM_KIVI(LlamaForCausalLM_KIVI):
    """Counterpart of models/mistral_kivi.py:921 (MistralForCausalLM_KIVI): the same decoder around MistralAttention_KIVI
    (bias-free projections, grouped queries mapped inside the kernels, `sliding_window` carried but -- like the reference,
    mistral_kivi.py:356-367 -- never applied to the quantised history)."""

    def __init__(self, config, attention_cls=None):
        from .attention import MistralAttention_KIVI
        super().__init__(config, attention_cls or MistralAttention_KIVI)
