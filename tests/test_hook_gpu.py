"""GPU parity of the attention hook (kivi_amd.attention) vs the CPU restatement of llama_kivi.py:314-455."""
import pytest
import torch

from helpers import gemv_close, make_kv, same_bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nh,nh_kv,T0,R", [(4, 4, 2100, 32), (8, 2, 4200, 128), (2, 1, 9000, 32)])
def test_long_context_small_batch_split_rows(oracle, nh, nh_kv, T0, R):
    """Few (b, kv head) rows and a long context: the sV launch splits every row over several blocks that meet in a
    workspace, and the row softmax runs as its own launches (several blocks per row, chunk statistics combined in a
    second launch).  Must match the reference logic like any other shape."""
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode
    from oracle import hook_ref as H
    B, D, g = 1, 128, 32
    cfg = KiviConfig(2, 2, g, R)
    k0, v0 = make_kv(1, B, nh_kv, T0, D), make_kv(2, B, nh_kv, T0, D)
    layer = KiviLayerCache(cfg, B, nh_kv, D, T0 + 16, "cuda")
    layer.prefill(k0.cuda(), v0.cuda())
    past = H.prefill_cache(k0, v0, 2, 2, g, R)
    for s in range(5):
        q = make_kv(100 + s, B, nh, 1, D)
        kn, vn = make_kv(200 + s, B, nh_kv, 1, D), make_kv(300 + s, B, nh_kv, 1, D)
        out = kivi_attention_decode(q.cuda(), kn.cuda(), vn.cuda(), layer)
        assert not getattr(layer, "_attend_unfusable", False) and not getattr(layer, "_fused_unsupported", False)
        ref, past = H.decode_step(q, kn, vn, past, 2, 2, g, R)
        ok, ratio = gemv_close(out, ref, rtol=3e-3, ulps=1)
        assert ok, (s, ratio)
    _cmp_cache(layer.as_tuple(), past)


def test_fused_step_partial_support():
    """head_dim 96: the qK^T kernel is tuned, the sV kernel is not -> the step mixes fused scores with a composed output
    and must still match the reference logic and keep the cache consistent."""
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode
    from oracle import hook_ref as H
    B, nh, nh_kv, D, T0, R = 1, 4, 2, 96, 40, 32
    cfg = KiviConfig(2, 2, 32, R)
    k0, v0 = make_kv(1, B, nh_kv, T0, D), make_kv(2, B, nh_kv, T0, D)
    layer = KiviLayerCache(cfg, B, nh_kv, D, 128, "cuda")
    layer.prefill(k0.cuda(), v0.cuda())
    past = H.prefill_cache(k0, v0, 2, 2, 32, R)
    for s in range(30):
        q = make_kv(100 + s, B, nh, 1, D)
        kn, vn = make_kv(200 + s, B, nh_kv, 1, D), make_kv(300 + s, B, nh_kv, 1, D)
        out = kivi_attention_decode(q.cuda(), kn.cuda(), vn.cuda(), layer)
        ref, past = H.decode_step(q, kn, vn, past, 2, 2, 32, R)
        ok, ratio = gemv_close(out, ref, rtol=3e-3, ulps=1)
        assert ok, (s, ratio)
        _cmp_cache(layer.as_tuple(), past)


def _golden_hook_cases():
    import glob, os
    return sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "hook_*.npz")))


@pytest.mark.parametrize("layout", ["hook", "auto", "row", "split"])
@pytest.mark.parametrize("path", _golden_hook_cases(), ids=lambda p: p.split("hook_")[-1][:-4])
def test_hook_matches_reference_class_fixtures(path, layout):
    """The HIP hook replays the inputs of fixtures recorded from the REFERENCE's own attention classes
    (oracle/pin_hook.py): step outputs within the hook bar, final 9-tuple bit-identical to the reference's.
    layout "hook": the hook-state layout (VALU kernels); "auto": what make_layer_cache gives a model of that shape -- the
    matrix-pipe layout for 2-bit / g = 32 / D = 128 with nh / nh_kv in {1, 4, 8}, i.e. what every real model and the bench
    run; "row" / "split": the matrix-pipe layout with the one-launch (mf_row_kernel / mf_row4_kernel) and the two-launch
    (mf_k_kernel + mf_v_kernel) form forced, a launch probe naming the kernel that ran."""
    from test_hook_golden_cpu import NAMES, load_case
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, KiviLayerCache, KiviLayerCacheMF, kivi_attention_decode, make_layer_cache
    c = load_case(path)
    cfg = KiviConfig(c["bits"], c["bits"], c["g"], c["R"])
    cap = c["T0"] + c["steps"] + 4
    expect = None
    if layout == "hook":
        layer = KiviLayerCache(cfg, c["B"], c["nh_kv"], c["D"], cap, "cuda")
    else:
        layer = make_layer_cache(cfg, c["B"], c["nh_kv"], c["D"], cap, "cuda", num_heads=c["nh"])
        ratio = c["nh"] // c["nh_kv"]
        if layout != "auto":
            if not isinstance(layer, KiviLayerCacheMF):
                pytest.skip("shape outside the matrix-pipe layout: covered by layout=auto")
            if layout == "row":
                if ratio not in (1, 4, 8):
                    pytest.skip("no one-launch kernel for this head ratio")
                layer.flags, expect = _lib.GQA_FORCE_ROW, (b"mf_row_kernel" if ratio == 1 else b"mf_row4_kernel")
            else:
                layer.flags, expect = _lib.GQA_FORCE_SPLIT, b"mf_k_kernel"
    layer.prefill(c["k0"].cuda(), c["v0"].cuda())
    lib = _lib.load()
    for s in range(c["steps"]):
        m = c["masks"][s].cuda() if c["masks"][s] is not None else None
        if expect is not None and s in (0, c["steps"] - 1):
            e0, e1 = lib.kivi_event_create(), lib.kivi_event_create()
            lib.kivi_set_launch_events(e0, e1)
        out = kivi_attention_decode(c["q"][s].cuda(), c["k"][s].cuda(), c["v"][s].cuda(), layer, attention_mask=m)
        if expect is not None and s in (0, c["steps"] - 1):
            torch.cuda.synchronize()
            assert expect in (lib.kivi_last_timed_kernel() or b""), lib.kivi_last_timed_kernel()
        # the fixture outputs come from the reference classes' CPU fp16 matmuls (their own rounding noise ~1e-3 on top of
        # the two fp16 partial sums): hook bar 3e-3 + that
        ok, ratio = gemv_close(out, c["out"][s], rtol=4e-3, ulps=1)
        assert ok, (s, ratio)
    t = layer.as_tuple()
    for n, a, b in zip(NAMES, t[:8], c["final"]):
        if b is None:
            assert a is None or a.numel() == 0, n
        else:
            assert a is not None and tuple(a.shape) == tuple(b.shape) and same_bits(a, b), n
    assert t[8] == c["final_len"]


def _cmp_cache(t_gpu, t_ref):
    names = ["K_code_T", "K_full", "K_scale_T", "K_mn_T", "V_code", "V_full", "V_scale", "V_mn"]
    for n, a, b in zip(names, t_gpu[:8], t_ref[:8]):
        if b is None:
            assert a is None or a.numel() == 0, n
            continue
        assert a is not None and tuple(a.shape) == tuple(b.shape), (n, None if a is None else a.shape, b.shape)
        assert same_bits(a, b), n
    assert t_gpu[8] == t_ref[8]


@pytest.mark.parametrize("fused_kernels", ["native", "python", False])
@pytest.mark.parametrize("nh,nh_kv,T0,R,g,bits", [(4, 4, 70, 32, 32, 2), (8, 2, 33, 32, 32, 2), (4, 2, 5, 32, 32, 2),
                                                   (4, 4, 130, 64, 32, 4), (4, 1, 128, 128, 64, 2), (6, 2, 40, 32, 32, 2),
                                                   (4, 4, 130, 64, 64, 2), (2, 2, 260, 128, 128, 2),
                                                   (2, 2, 150, 192, 64, 4)])   # R = 192: pages of lcm(2048, R) = 6144 tokens (R = 96, every step, every
                                                                              # unit, against the fp64 reference: test_residual_lengths_that_do_not_divide_a_page)
def test_decode_steps_match_reference_logic(oracle, nh, nh_kv, T0, R, g, bits, fused_kernels, monkeypatch):
    """"native": the one-call layer step (kivi_decode_layer); "python": the same launches with the bookkeeping in
    kivi_amd.attention; False: one launch per reference op."""
    import kivi_amd.attention as A
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode
    from oracle import hook_ref as H
    monkeypatch.setattr(A, "_NATIVE_STEP", fused_kernels == "native")
    fused_kernels = bool(fused_kernels)
    B, D = 2, 128
    steps = R + 9
    cfg = KiviConfig(bits, bits, g, R)
    k0, v0 = make_kv(1, B, nh_kv, T0, D), make_kv(2, B, nh_kv, T0, D)
    layer = KiviLayerCache(cfg, B, nh_kv, D, T0 + steps + 3, "cuda")
    layer.prefill(k0.cuda(), v0.cuda())
    past = H.prefill_cache(k0, v0, bits, bits, g, R)
    _cmp_cache(layer.as_tuple(), past)
    for s in range(steps):
        q = make_kv(100 + s, B, nh, 1, D)
        kn, vn = make_kv(200 + s, B, nh_kv, 1, D), make_kv(300 + s, B, nh_kv, 1, D)
        out = kivi_attention_decode(q.cuda(), kn.cuda(), vn.cuda(), layer, fused_kernels=fused_kernels)
        ref, past = H.decode_step(q, kn, vn, past, bits, bits, g, R)
        ok, ratio = gemv_close(out, ref, rtol=3e-3, ulps=1)   # fp32 softmax (GPU exp vs libm) + two fp16 partial sums on top of the GEMV bar
        assert ok, (s, ratio)
        _cmp_cache(layer.as_tuple(), past)             # cache contents are bit-identical at every step
    assert layer.nbytes() == sum(x.numel() * x.element_size() for x in past[:8] if x is not None)
    assert layer.as_tuple()[-1] == past[8] and len(layer.as_tuple()) == 9


def test_module_hook_prefill_then_decode():
    """The nn.Module hook with the reference forward() signature: tuple contract, shapes, adoption of plain tuples."""
    from types import SimpleNamespace

    from kivi_amd.attention import LlamaAttention_KIVI
    from kivi_amd.cache import KiviCacheTuple
    cfg = SimpleNamespace(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256,
                          rope_theta=10000.0, k_bits=2, v_bits=2, group_size=32, residual_length=32)
    torch.manual_seed(0)
    attn = LlamaAttention_KIVI(cfg).half().cuda()
    x = torch.randn(2, 45, 512, device="cuda", dtype=torch.float16)
    out, w, past = attn(x, use_cache=True)
    assert out.shape == (2, 45, 512) and w is None and isinstance(past, KiviCacheTuple) and len(past) == 9
    assert past[-1] == 45 and past[0].shape == (2, 2, 128, 2) and past[1].shape == (2, 2, 13, 128)
    assert past[4].shape == (2, 2, 13, 128 // 16) and past[5].shape == (2, 2, 32, 128)
    plain = tuple(None if t is None else (t.clone() if torch.is_tensor(t) else t) for t in past)
    outs = []
    for p in (past, plain):
        cur = p
        o = None
        for s in range(3):
            xs = torch.full((2, 1, 512), 0.01 * (s + 1), device="cuda", dtype=torch.float16)
            o, _, cur = attn(xs, past_key_value=cur, use_cache=True)
        outs.append(o)
        assert cur[-1] == 48 and torch.isfinite(o).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("nh_kv", [4, 8])
def test_fused_and_composed_paths_agree_with_mask(nh_kv):
    """Same inputs through the fused step (nh_kv = 8: MHA, the one-launch decode-row kernel; nh_kv = 4: grouped queries,
    qK^T + row softmax + shared-unpack sV) and the reference-style composition: identical cache contents, outputs
    within fp16 rounding of each other -- including the additive attention mask branch (llama_kivi.py:364-372)."""
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode
    B, nh, D, T0, R = 2, 8, 128, 100, 32
    cfg = KiviConfig(2, 2, 32, R)
    k0, v0 = make_kv(1, B, nh_kv, T0, D), make_kv(2, B, nh_kv, T0, D)
    la = KiviLayerCache(cfg, B, nh_kv, D, 256, "cuda")
    lb = KiviLayerCache(cfg, B, nh_kv, D, 256, "cuda")
    for lc in (la, lb):
        lc.prefill(k0.cuda(), v0.cuda())
    for s in range(40):
        q = make_kv(100 + s, B, nh, 1, D).cuda()
        kn, vn = make_kv(200 + s, B, nh_kv, 1, D).cuda(), make_kv(300 + s, B, nh_kv, 1, D).cuda()
        mask = torch.zeros((B, 1, 1, T0 + s + 1), dtype=torch.float16, device="cuda")
        mask[0, :, :, : 7 + s] = torch.finfo(torch.float16).min      # left padding of sequence 0
        oa = kivi_attention_decode(q, kn, vn, la, attention_mask=mask, fused_kernels=True)
        ob = kivi_attention_decode(q, kn, vn, lb, attention_mask=mask, fused_kernels=False)
        assert not getattr(la, "_fused_unsupported", False)
        ok, ratio = gemv_close(oa, ob.cpu(), rtol=2e-3, ulps=1)
        assert ok, (s, ratio)
        for x, y in zip(la.as_tuple()[:8], lb.as_tuple()[:8]):
            assert (x is None) == (y is None) and (x is None or same_bits(x, y))


@pytest.mark.parametrize("B,nh,nh_kv,T0,R", [(2, 8, 1, 4500, 128), (1, 16, 4, 2300, 32), (3, 4, 2, 6200, 64)])
def test_grouped_queries_long_rows_masked(B, nh, nh_kv, T0, R):
    """Grouped queries (ratio 8 / 4 / 2) with rows long enough for the multi-block row softmax, plus the additive
    mask: the fused step (qK^T launch, row-softmax launches, shared-unpack sV launch) against the reference-style
    composition -- outputs within fp16 rounding, cache contents bit-identical."""
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode
    D = 128
    cfg = KiviConfig(2, 2, 32, R)
    k0, v0 = make_kv(11, B, nh_kv, T0, D), make_kv(12, B, nh_kv, T0, D)
    la = KiviLayerCache(cfg, B, nh_kv, D, T0 + 8, "cuda")
    lb = KiviLayerCache(cfg, B, nh_kv, D, T0 + 8, "cuda")
    for lc in (la, lb):
        lc.prefill(k0.cuda(), v0.cuda())
    for s in range(4):
        q = make_kv(100 + s, B, nh, 1, D).cuda()
        kn, vn = make_kv(200 + s, B, nh_kv, 1, D).cuda(), make_kv(300 + s, B, nh_kv, 1, D).cuda()
        mask = torch.zeros((B, 1, 1, T0 + s + 1), dtype=torch.float16, device="cuda")
        mask[0, :, :, : 1000 + 7 * s] = torch.finfo(torch.float16).min      # left padding of sequence 0
        oa = kivi_attention_decode(q, kn, vn, la, attention_mask=mask, fused_kernels=True)
        ob = kivi_attention_decode(q, kn, vn, lb, attention_mask=mask, fused_kernels=False)
        assert not getattr(la, "_fused_unsupported", False) and not getattr(la, "_attend_unfusable", False)
        ok, ratio = gemv_close(oa, ob.cpu(), rtol=2e-3, ulps=1)
        assert ok, (s, ratio)
        for x, y in zip(la.as_tuple()[:8], lb.as_tuple()[:8]):
            assert (x is None) == (y is None) and (x is None or same_bits(x, y))


@pytest.mark.parametrize("n,pitch", [(33, 40), (1024, 1024), (4109, 4136), (5000, 5000), (16384, 16384), (20001, 20008)])
def test_softmax_scaled_kernel(n, pitch):
    from kivi_amd.quant import fused
    torch.manual_seed(n)
    B, nh = 2, 3
    scores = torch.zeros((B, nh, 1, pitch), dtype=torch.float16, device="cuda")
    scores[..., :n] = (torch.randn((B, nh, 1, n), device="cuda") * 20).half()
    probs = torch.full_like(scores, 7.0)
    inv = 1.0 / (128 ** 0.5)
    for mask in (None, torch.where(torch.rand((B, 1, 1, n), device="cuda") < 0.2, torch.finfo(torch.float16).min, 0.0).half()):
        fused.softmax_scaled(scores, probs, n, inv, mask)
        x = scores[..., :n] / (128 ** 0.5)                     # reference op sequence on torch (llama_kivi.py:339-375)
        if mask is not None:
            x = torch.max(x + mask, torch.tensor(torch.finfo(torch.float16).min, device="cuda"))
        ref = torch.softmax(x, dim=-1, dtype=torch.float32).half()
        got = probs[..., :n]
        assert torch.isfinite(got).all() and bool((probs[..., n:] == 7.0).all())
        err = (got.float() - ref.float()).abs()
        assert bool((err <= 1.5e-3 * ref.float() + 1e-7).all()), err.max().item()
        assert abs(got.float().sum(-1) - 1).max().item() < 2e-3


def test_decoder_wrapper_generate_fused_vs_composed():
    """A tiny random Llama-shaped model through kivi_amd.llama: greedy generation runs, and the fused decode step
    gives the same logits as the reference-style composition to fp16 rounding."""
    import kivi_amd.attention as A
    from kivi_amd.llama import LlamaForCausalLM_KIVI, make_config
    torch.manual_seed(0)
    cfg = make_config(dict(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=2,
                           intermediate_size=1024, vocab_size=320), residual_length=32, max_cache_len=128)
    model = LlamaForCausalLM_KIVI(cfg).half().cuda()
    for p in model.parameters():
        if p.dim() > 1:
            p.data.normal_(0.0, 0.05)
    ids = torch.randint(0, 320, (2, 40), device="cuda")
    out = model.generate(ids, 12)
    assert out.shape == (2, 52) and bool((out[:, :40] == ids).all())
    logits_f, pasts_f = model(ids)
    logits_c, pasts_c = model(ids)
    tok = logits_f.argmax(-1)
    orig = A.kivi_attention_decode
    for step in range(40):   # crosses the K flush (every 32 tokens) and flushes V every step
        lf, pasts_f = model(tok, pasts_f)
        A.kivi_attention_decode = lambda *a, **k: orig(*a, **{**k, "fused_kernels": False})
        try:
            lc, pasts_c = model(tok, pasts_c)
        finally:
            A.kivi_attention_decode = orig
        assert torch.isfinite(lf).all()
        assert (lf.float() - lc.float()).abs().max().item() <= 2e-2 * lc.float().abs().max().item() + 1e-3, step
        tok = lf.argmax(-1)
    assert pasts_f[0][-1] == 80


def test_decoder_wrapper_graphed_decode_equals_eager():
    """The hipGraph-replayed decode (dense part captured, KIVI step eager) generates the same tokens as the eager loop."""
    from kivi_amd.llama import LlamaForCausalLM_KIVI, make_config
    torch.manual_seed(1)
    cfg = make_config(dict(hidden_size=512, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=3,
                           intermediate_size=1024, vocab_size=500), residual_length=32, max_cache_len=160)
    model = LlamaForCausalLM_KIVI(cfg).half().cuda()
    for p in model.parameters():
        if p.dim() > 1:
            p.data.normal_(0.0, 0.05)
    ids = torch.randint(0, 500, (2, 37), device="cuda")
    a = model.generate(ids, 45)
    b = model.generate_graphed(ids, 45)
    assert a.shape == b.shape == (2, 82)
    assert bool((a == b).all()), (a != b).nonzero()[:4]


def test_full_size_row_kernel_vs_two_launch_path(monkeypatch):
    """BASELINE config C2 at full size (B=32, 32 heads, 4096-token cache, 2-bit, g=32, R=32): the one-launch decode-row
    kernel against the same step done with the stand-alone qK^T and attend launches (different summation order in the
    qK^T phase only): outputs within fp16 rounding, cache contents bit-identical, probabilities sum to one."""
    import kivi_amd.attention as A
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode
    B, nh, D, T0, R = 32, 32, 128, 4096, 32
    cfg = KiviConfig(2, 2, 32, R)
    gen = torch.Generator(device="cuda").manual_seed(5)
    k0 = torch.randn((B, nh, T0, D), device="cuda", dtype=torch.float16, generator=gen)
    v0 = torch.randn((B, nh, T0, D), device="cuda", dtype=torch.float16, generator=gen)
    la = KiviLayerCache(cfg, B, nh, D, T0 + 40, "cuda")
    lb = KiviLayerCache(cfg, B, nh, D, T0 + 40, "cuda")
    for lc in (la, lb):
        lc.prefill(k0, v0)
    del k0, v0
    for s in range(34):     # crosses a K flush (every 32 tokens) and a window compaction
        q = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        kn = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        vn = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        monkeypatch.setattr(A, "_NATIVE_STEP", True)
        oa = kivi_attention_decode(q, kn, vn, la)
        monkeypatch.setattr(A, "_NATIVE_STEP", False)     # Python bookkeeping: gemv_k_paged + kivi_decode_attend
        ob = kivi_attention_decode(q, kn, vn, lb)
        ok, ratio = gemv_close(oa, ob.cpu(), rtol=3e-3, ulps=1)   # hook bar; worst of 131k outputs per step (1-ulp score flips)
        assert ok, (s, ratio)
    for x, y in zip(la.as_tuple()[:8], lb.as_tuple()[:8]):
        assert (x is None) == (y is None) and (x is None or same_bits(x, y))
    assert la.as_tuple()[8] == lb.as_tuple()[8] == T0 + 34


def test_cache_grows_past_its_initial_capacity():
    """The reference's tuple grows without bound; the in-place cache doubles its capacity when it runs out.  A cache that
    starts too small must give bit-identical outputs and contents to one that was big from the start."""
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode
    B, nh, nh_kv, D, T0, R = 2, 4, 4, 128, 60, 32
    cfg = KiviConfig(2, 2, 32, R)
    k0, v0 = make_kv(1, B, nh_kv, T0, D), make_kv(2, B, nh_kv, T0, D)
    small = KiviLayerCache(cfg, B, nh_kv, D, T0 + 2, "cuda")       # capacity 64 tokens
    big = KiviLayerCache(cfg, B, nh_kv, D, 4096, "cuda")
    for lc in (small, big):
        lc.prefill(k0.cuda(), v0.cuda())
    cap0 = small.cap
    for s in range(150):
        q = make_kv(100 + s, B, nh, 1, D).cuda()
        kn, vn = make_kv(200 + s, B, nh_kv, 1, D).cuda(), make_kv(300 + s, B, nh_kv, 1, D).cuda()
        oa = kivi_attention_decode(q, kn, vn, small)
        ob = kivi_attention_decode(q, kn, vn, big)
        assert same_bits(oa, ob), s
    assert small.cap > cap0 and small.kv_seq_len == big.kv_seq_len == T0 + 150
    for x, y in zip(small.as_tuple()[:8], big.as_tuple()[:8]):
        assert (x is None) == (y is None) and (x is None or same_bits(x, y))


@pytest.mark.parametrize("nh,nh_kv", [(4, 4), (8, 2)])
def test_eager_prompt_pass_applies_the_mask(oracle, nh, nh_kv):
    """The eager class adds the additive mask to the prompt pass (llama_kivi.py:228-237; left-padded batches), the flash
    class runs causal attention without it (:420-423).  Outputs vs the CPU restatement of the eager arithmetic, cache
    tuples bit-identical either way (the cache never depends on the mask)."""
    import types
    from kivi_amd.attention import KiviConfig, kivi_attention_prefill, make_layer_cache
    from oracle import hook_ref as H
    B, T, D, g, R = 2, 75, 128, 32, 32
    q, k, v = make_kv(1, B, nh, T, D), make_kv(2, B, nh_kv, T, D), make_kv(3, B, nh_kv, T, D)
    neg = torch.finfo(torch.float16).min
    mask = torch.full((T, T), neg, dtype=torch.float16).triu(1)[None, None].repeat(B, 1, 1, 1)
    mask[1, :, :, :9] = neg                                   # batch row 1: nine left-padding positions
    mask[1, 0, torch.arange(9), torch.arange(9)] = 0.0        # (padded queries still see themselves: no empty rows)
    cfg = KiviConfig(2, 2, g, R)
    for m in (mask, None):
        layer = make_layer_cache(cfg, B, nh_kv, D, T + 64, "cuda", num_heads=nh)
        out = kivi_attention_prefill(q.cuda(), k.cuda(), v.cuda(), layer, None if m is None else m.cuda())
        causal = torch.full((T, T), neg, dtype=torch.float16).triu(1)[None, None].repeat(B, 1, 1, 1)
        ref = H.prefill_attention_eager(q, k, v, m if m is not None else causal)
        err = (out.cpu().float() - ref.float()).abs().max().item()
        assert err <= 4e-3 * max(1.0, ref.float().abs().max().item()), err
        _cmp_cache(layer.as_tuple(), H.prefill_cache(k, v, 2, 2, g, R))
    with pytest.raises(ValueError):
        kivi_attention_prefill(q.cuda(), k.cuda(), v.cuda(), make_layer_cache(cfg, B, nh_kv, D, T + 64, "cuda", num_heads=nh),
                               mask[:, :, :1].cuda())


@pytest.mark.parametrize("R,g,bits", [(96, 32, 2), (192, 64, 4)])
def test_residual_lengths_that_do_not_divide_a_page(R, g, bits):
    """llama_kivi.py:344 accepts any residual length that is a multiple of the group size.  The hook-state layout pages K in whole
    tiles of the qK^T kernels (2048 tokens); a flush of R tokens must not straddle a page, so R = 96 / 192 get pages of 6144 tokens.
    Steps through K flushes on both sides of the first page boundary, every unit against the fp64 torch reference."""
    import torch_ref64 as T64
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode
    B, nh, nh_kv, D = 3, 4, 2, 128
    T0 = 6144 - R - 20
    steps = R + 30
    cfg = KiviConfig(bits, bits, g, R)
    gen = torch.Generator(device="cuda").manual_seed(77)
    k0 = torch.randn((B, nh_kv, T0, D), device="cuda", dtype=torch.float16, generator=gen)
    v0 = torch.randn((B, nh_kv, T0, D), device="cuda", dtype=torch.float16, generator=gen)
    layer = KiviLayerCache(cfg, B, nh_kv, D, T0 + steps + 1, "cuda")
    assert layer.page_tokens == 6144 and layer.n_pages == 2
    layer.prefill(k0, v0)
    past = T64.prefill_cache(k0, v0, bits, bits, g, R)
    names = ["K_code_T", "K_full", "K_scale_T", "K_mn_T", "V_code", "V_full", "V_scale", "V_mn"]

    def same(a, b):
        if a is None or b is None:
            return (a is None or a.numel() == 0) and (b is None or b.numel() == 0)
        a, b = a.contiguous(), b.contiguous()
        return a.shape == b.shape and bool(torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a,
                                                       b.view(torch.int16) if b.dtype == torch.float16 else b))
    worst = 0.0
    for s in range(steps):
        q = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        kn = torch.randn((B, nh_kv, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        vn = torch.randn((B, nh_kv, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        out = kivi_attention_decode(q, kn, vn, layer)
        ref, past, _ = T64.decode_step(q, kn, vn, past, bits, bits, g, R)
        ok, ratio = gemv_close(out, ref, rtol=3e-3, ulps=1)
        assert ok, (s, ratio)
        worst = max(worst, ratio)
        if s % 16 == 0 or s == steps - 1:
            for n, a, r in zip(names, layer.as_tuple()[:8], past[:8]):
                assert same(a, r), (s, n)
    assert layer.as_tuple()[8] == past[8] == T0 + steps
    print(f"R = {R}: worst output ratio {worst:.3f} of 3e-3 (+1 ulp)")
