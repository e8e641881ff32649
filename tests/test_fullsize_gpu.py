"""GPU parity at the sizes bench.py and BASELINE.json's configs actually run, against the CPU oracle.

Heads are independent, so the full-size step runs on the GPU while the CPU restatement of the
hook (oracle/hook_ref.py, pinned against the reference's own attention classes by oracle/pin_hook.py) follows a few
sampled (batch row, kv head) slices of the SAME inputs: outputs within the hook bar at every step, the sampled
slices of the final 9-tuple bit-identical.  Reference call sites: models/llama_kivi.py:314-399 (decode branch),
models/mistral_kivi.py:381-445 (grouped queries), quant/csrc/gemv_cuda.cu:348-427 (the fused GEMV itself).
"""
import ctypes

import pytest
import torch

from helpers import gemv_close, same_bits

pytestmark = pytest.mark.gpu

NAMES = ["K_code_T", "K_full", "K_scale_T", "K_mn_T", "V_code", "V_full", "V_scale", "V_mn"]


class LaunchProbe:
    """Names the kernel of the first fused launch of the next step (the library stamps the event pair it is handed on
    that dispatch and remembers the kernel's source name: include/kivi_hip.h, kivi_set_launch_events)."""

    def __init__(self):
        from kivi_amd import _lib
        self.lib = _lib.load()
        self.e0, self.e1 = self.lib.kivi_event_create(), self.lib.kivi_event_create()

    def arm(self):
        self.lib.kivi_set_launch_events(self.e0, self.e1)

    def kernel(self):
        torch.cuda.synchronize()
        assert self.lib.kivi_event_elapsed_us(self.e0, self.e1) > 0
        return (self.lib.kivi_last_timed_kernel() or b"").decode()


_SPLIT = 1      # _lib.GQA_FORCE_SPLIT


def run_sampled(B, nh, nh_kv, T0, R, bits, g, steps, samples, seed, masked=False, expect_kernel=None, D=128, layout="hook",
                stage_ab=False, outlier=False, extra_flags=0):
    """Full-size decode steps on the GPU; hook_ref on the sampled (b, kv head) slices.  Returns the kernels seen.
    layout "hook": the hook-state layout (KiviLayerCache, VALU kernels); "auto": what make_layer_cache picks for the shape
    (the matrix-pipe layout for g=32 / D=128: 2-bit with nh / nh_kv in {1, 4, 8}, 4-bit with nh / nh_kv = 4).
    outlier: every 17th key channel x 12 (what per-channel K quantisation exists for).  With such keys the fp16 scores reach
    |s| ~ 100, where ONE ulp of a dominant score (0.0625 / sqrt(D)) moves its probability by 0.5 %, so the end-to-end bar is
    replaced by the stage check of tests/test_mfma_gpu.py (stage_ab, matrix-pipe layout only): A. the fp16 rows the softmax
    consumed (KIVI_GQA_DUMP_SCORES makes the one-launch kernels write them) within the GEMV bar of the reference's rows;
    B. the output within 2e-3 of the reference's attend half run on the rows the GPU produced."""
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, KiviLayerCache, kivi_attention_decode, make_layer_cache
    from oracle import hook_ref as H
    ratio = nh // nh_kv
    cfg = KiviConfig(bits, bits, g, R)
    gen = torch.Generator(device="cuda").manual_seed(seed)

    def keys(T):
        k = torch.randn((B, nh_kv, T, D), device="cuda", dtype=torch.float32, generator=gen)
        if outlier:
            k[..., ::17] *= 12.0
        return k.half()

    k0 = keys(T0)
    v0 = torch.randn((B, nh_kv, T0, D), device="cuda", dtype=torch.float16, generator=gen)
    if layout == "hook":
        layer = KiviLayerCache(cfg, B, nh_kv, D, T0 + steps + 1, "cuda")
    else:
        layer = make_layer_cache(cfg, B, nh_kv, D, T0 + steps + 1, "cuda", num_heads=nh)
    if stage_ab:
        assert layer.layout == "mfma"
        layer.flags |= _lib.GQA_DUMP_SCORES
    if extra_flags:
        layer.flags |= extra_flags
    layer.prefill(k0, v0)
    pasts = {}
    for (b, hk) in samples:
        pasts[(b, hk)] = H.prefill_cache(k0[b:b + 1, hk:hk + 1].cpu(), v0[b:b + 1, hk:hk + 1].cpu(), bits, bits, g, R)
    del k0, v0
    probe = LaunchProbe()
    seen = set()
    worst = 0.0
    for s in range(steps):
        q = torch.randn((B, nh, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        kn = keys(1)
        vn = torch.randn((B, nh_kv, 1, D), device="cuda", dtype=torch.float16, generator=gen)
        mask = None
        if masked:
            mask = torch.zeros((B, 1, 1, T0 + s + 1), dtype=torch.float16, device="cuda")
            for b in range(0, B, 2):                       # left padding on every other sequence, growing with the step
                mask[b, :, :, : 100 + 7 * b + s] = torch.finfo(torch.float16).min
        probe.arm()
        out = kivi_attention_decode(q, kn, vn, layer, attention_mask=mask)
        assert not getattr(layer, "_fused_unsupported", False) and not getattr(layer, "_attend_unfusable", False)
        seen.add(probe.kernel().split("<")[0].strip("( "))
        for (b, hk) in samples:
            hs = slice(hk * ratio, (hk + 1) * ratio)
            args = (q[b:b + 1, hs].cpu(), kn[b:b + 1, hk:hk + 1].cpu(), vn[b:b + 1, hk:hk + 1].cpu(), pasts[(b, hk)], bits, bits, g, R)
            m_cpu = None if mask is None else mask[b:b + 1].cpu()
            if stage_ab:
                n = T0 + s + 1
                x_gpu = layer._native[4][0][b:b + 1, hs, :, :n].cpu()
                ref, new_past, x_ref = H.decode_step(*args, attention_mask=m_cpu, return_scores=True)
                live = x_ref.float() > -60000
                ok, ra = gemv_close(torch.where(live, x_gpu.float(), 0.0), torch.where(live, x_ref.float(), 0.0), rtol=1e-3, ulps=1)
                assert ok, ("scores", s, b, hk, ra)
                assert torch.equal(x_gpu[~live], x_ref[~live])
                ref_b, _ = H.decode_step(*args, attention_mask=m_cpu, scores_override=x_gpu)
                ok, ratio_err = gemv_close(out[b:b + 1, hs], ref_b, rtol=2e-3, ulps=1)
                pasts[(b, hk)] = new_past
            else:
                ref, pasts[(b, hk)] = H.decode_step(*args, attention_mask=m_cpu)
                ok, ratio_err = gemv_close(out[b:b + 1, hs], ref, rtol=3e-3, ulps=1)      # the hook bar (tests/test_hook_gpu.py)
            worst = max(worst, ratio_err)
            assert ok, (s, b, hk, ratio_err)
    t = layer.as_tuple()
    for (b, hk) in samples:
        for n, a, r in zip(NAMES, t[:8], pasts[(b, hk)][:8]):
            if r is None:
                assert a is None or a.numel() == 0, n
                continue
            assert a is not None and same_bits(a[b:b + 1, hk:hk + 1], r), (n, b, hk)
        assert t[8] == pasts[(b, hk)][8] == T0 + steps
    if expect_kernel is not None:
        assert seen == {expect_kernel}, seen
    return seen, worst


@pytest.mark.parametrize("bits,masked", [(2, False), (4, True)])
def test_bench_shape_decode_row_kernel_vs_oracle(oracle, bits, masked):
    """BENCH / BASELINE config C2: B=32, 32 heads (1024 row units -> nothing splits), 4096-token prompt = two 2048-token
    K pages, g=32, R=32, 40 steps: crosses the K flush at 4096+32 (a third, partial page) and a V-window compaction, the
    V flush runs every step.  The launch probe proves it is the one-launch decode_row_kernel that was checked."""
    seen, worst = run_sampled(B=32, nh=32, nh_kv=32, T0=4096, R=32, bits=bits, g=32, steps=40,
                              samples=[(0, 0), (13, 7), (31, 31)], seed=11, masked=masked,
                              expect_kernel="decode_row_kernel")
    print("worst ratio vs the 3e-3 hook bar:", worst)


@pytest.mark.parametrize("T0,masked", [(4096, False), (4080, True)])
def test_bench_shape_mf_row_kernel_vs_oracle(oracle, T0, masked):
    """The shape bench.py runs since round 3: B=32, 32 heads, 2-bit g=32 R=32 on the matrix-pipe layout, one launch per layer
    (mf_row_kernel); 40 steps across the K flush (kivi_kt_pack at 4096 + 32 / 4080 + 16), the V flush of every step, a
    window compaction and the growth into a ninth super-block.  The launch probe proves which kernel was checked."""
    seen, worst = run_sampled(B=32, nh=32, nh_kv=32, T0=T0, R=32, bits=2, g=32, steps=40, samples=[(0, 0), (13, 7), (31, 31)],
                              seed=11, masked=masked, expect_kernel="mf_row_kernel", layout="auto")
    print("worst ratio vs the 3e-3 hook bar:", worst)


@pytest.mark.parametrize("T0,masked", [(4096, False), (4080, True)])
def test_bench_shape_mf_row_kernel_stages_on_outlier_keys(oracle, T0, masked):
    """The same shape and kernel with large-magnitude key channels, checked stage by stage on the sampled rows (the scores of
    mf_row_kernel never leave the LDS: its test instantiation writes the rows its softmax consumes), 40 steps through the K
    flush.  VERDICT r3 "what's weak" #2."""
    seen, worst = run_sampled(B=32, nh=32, nh_kv=32, T0=T0, R=32, bits=2, g=32, steps=40, samples=[(0, 0), (13, 7), (31, 31)],
                              seed=21, masked=masked, expect_kernel="mf_row_kernel", layout="auto", stage_ab=True, outlier=True)
    print("worst ratio vs the 2e-3 attend bar:", worst)


def test_config4_shape_mf_row4_kernel_stages_on_outlier_keys(oracle):
    """BASELINE config 4 (32 / 8 heads, 8k keys, R = 128, B = 64: one mf_row4_kernel launch per layer) likewise."""
    seen, worst = run_sampled(B=64, nh=32, nh_kv=8, T0=8192 - 4, R=128, bits=2, g=32, steps=8, samples=[(0, 0), (63, 7), (32, 3)],
                              seed=22, expect_kernel="mf_row4_kernel", layout="auto", stage_ab=True, outlier=True)
    print("worst ratio vs the 2e-3 attend bar:", worst)


@pytest.mark.parametrize("B,T0,kernel,split", [(64, 8192 - 4, "mf_row4_kernel", False), (2, 32768 + 125, "mf_row4_kernel", False),
                                               (2, 32768 + 125, "mf_k_kernel", True)])
def test_4bit_gqa_shapes_on_the_matrix_pipe_stages(oracle, B, T0, kernel, split):
    """4-bit K / V, 32 / 8 heads (the reference's Mistral-7B + KIVI-4 shape, docs/long_bench.md:35-53; round 4): BASELINE config 4's
    geometry (B = 64, 8k keys, R = 128: one mf_row4_kernel launch per layer, a block per unit, across a K flush at step 4) and the
    config-5 slice's row length (32k keys: since round 5 ONE launch with every row cut into 16 slices -- 16 units x 16 blocks that
    exchange their softmax statistics inside the launch; and, forced, the two-launch form mf_k_kernel + mf_v_kernel), keys with outlier
    channels, masks, stage by stage (rows the softmax consumed within 1e-3, attend half on those rows within 2e-3), 9-tuples of the
    sampled units bit-identical to the reference logic's."""
    seen, worst = run_sampled(B=B, nh=32, nh_kv=8, T0=T0, R=128, bits=4, g=32, steps=8, samples=[(0, 0), (B - 1, 7), (B // 2, 3)],
                              seed=23, expect_kernel=kernel, layout="auto", stage_ab=True, outlier=True, masked=(B == 2),
                              extra_flags=_SPLIT if split else 0)
    print("worst ratio vs the 2e-3 attend bar:", worst)


@pytest.mark.parametrize("B,T0,kernel,flags", [(1, 32768 + 13, "mf_k_kernel", 0), (1, 32768 + 13, "mf_row4_kernel", 20 << 8),
                                               (4, 4080, "mf_row_kernel", 0), (4, 6000, "mf_k_kernel", 0), (16, 4080, "mf_row_kernel", 0)])
def test_small_batch_mf_split_rows_vs_oracle(oracle, B, T0, kernel, flags):
    """Few (batch row, head) rows: rows longer than the LDS row (B = 1 x 32k keys) and fewer than 192 rows of more than 8
    super-blocks (B = 4 x 6000) run the two-launch form with the rows cut into slices (the ONE-launch sliced form of long multi-head
    rows, KIVI_GQA_SLICES(20): measured slower in round 6, kept and covered); rows of at most 8 super-blocks take the eight-wave row
    kernel whatever the batch (B = 4 / 16 x 4k)."""
    run_sampled(B=B, nh=32, nh_kv=32, T0=T0, R=32, bits=2, g=32, steps=6, samples=[(0, 0), (B - 1, 31)], seed=15, layout="auto",
                expect_kernel=kernel, extra_flags=flags)


@pytest.mark.parametrize("B,T0,R,bits,flags,kernel", [(8, 32768 + 100, 128, 2, 0, "mf_k_kernel"), (16, 16384 + 100, 128, 2, 0, "mf_k_kernel"),
                                                      (8, 32768 + 100, 128, 2, 5 << 8, "mf_row4_kernel"), (8, 32768 + 100, 128, 4, 0, "mf_k_kernel"),
                                                      (16, 16384 + 100, 128, 4, 3 << 8, "mf_row4_kernel")])
def test_longchat_shape_multi_head_long_rows_stages(oracle, B, T0, R, bits, flags, kernel):
    """The reference's LongChat-7B-32K configuration (docs/long_bench.md:5-26: 32 heads = 32 kv heads, KIVI-2 and KIVI-4, g = 32,
    R = 128) at 16k / 32k keys and B = 8 / 16: multi-head rows beyond 8192 keys on the matrix pipe -- the plan's two launches
    (mf_k_kernel -> mf_v_kernel) and, forced, ONE launch with the rows cut into 5 / 3 slices (mf_row4_kernel<R = 1>; measured slower,
    profiles/r06_long_rows.log) --, through a K flush (step 28), outlier key channels, masks, stage by stage against the oracle on
    sampled units, 9-tuples of the sampled units bit-identical.  (Every unit of the same shapes: tests/test_fullcover_gpu.py.)"""
    seen, worst = run_sampled(B=B, nh=32, nh_kv=32, T0=T0, R=R, bits=bits, g=32, steps=30, samples=[(0, 0), (B - 1, 31), (B // 2, 13)],
                              seed=24, expect_kernel=kernel, layout="auto", stage_ab=True, outlier=True, masked=True, extra_flags=flags)
    print("worst ratio vs the 2e-3 attend bar:", worst)


def test_bench_shape_prompt_4080_k_flush_mid_page(oracle):
    """bench.py's default prompt (4080 tokens: K residual starts at 16, the flush lands inside the timed region)."""
    run_sampled(B=32, nh=32, nh_kv=32, T0=4080, R=32, bits=2, g=32, steps=20, samples=[(5, 3), (30, 17)], seed=12,
                expect_kernel="decode_row_kernel")


@pytest.mark.parametrize("B,steps", [(64, 8), (2, 8)])
def test_config4_shape_gqa_8k_vs_oracle(oracle, B, steps):
    """BASELINE config 4 (Llama-3-8B attention shape): 32 query / 8 kv heads, 8k context, R=128.  B=64: 512 row units,
    rows are not split; B=2: 16 units, the sV rows are split over blocks and the row softmax runs multi-block.  T0 puts
    the K residual at 124 so the flush of 128 tokens (models/llama_kivi.py:343-356) happens at step 4."""
    samples = [(0, 0), (B - 1, 7), (B // 2, 3)]
    run_sampled(B=B, nh=32, nh_kv=8, T0=8192 + 124, R=128, bits=2, g=32, steps=steps, samples=samples, seed=13,
                masked=(B == 2))
    if B == 64:      # the layout the hook uses for this shape since round 2 (matrix pipe), round-3 kernels
        run_sampled(B=B, nh=32, nh_kv=8, T0=8192 + 124, R=128, bits=2, g=32, steps=steps, samples=samples, seed=13, layout="auto")


@pytest.mark.parametrize("B", [1, 2])
def test_config5_slice_gqa_32k_vs_oracle(oracle, B):
    """BASELINE config 5, per-GPU slice shape (Mistral-7B attention: 32 / 8 heads, 32k context, R=128) at B = 1-2:
    16-chunk row softmax, split-T sV with the maximal split.  Mistral call sites: models/mistral_kivi.py:381-385, 441-445."""
    run_sampled(B=B, nh=32, nh_kv=8, T0=32768 + 125, R=128, bits=2, g=32, steps=6, samples=[(0, 0), (B - 1, 5)], seed=14)


def test_full_size_sv_vs_oracle(oracle):
    """BASELINE C2 sV product alone (B=32, 32 heads, Tv=4064 packed tokens, the reference's non-contiguous probability
    slice, llama_kivi.py:382) on sampled heads vs the oracle, + linearity over the whole output."""
    from kivi_amd.quant import matmul, new_pack
    B, nh, Tv, D, g, bits, L = 32, 32, 4064, 128, 32, 2, 33
    gen = torch.Generator(device="cuda").manual_seed(3)
    v = torch.randn((B, nh, Tv, D), device="cuda", dtype=torch.float16, generator=gen)
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, g, bits)
    del v
    w = torch.softmax(torch.randn((B, nh, 1, Tv + L), device="cuda", generator=gen) * 3, -1).half()
    a = w[..., :-L]                                         # the reference's slice: row pitch Tv + L
    out = matmul.cuda_bmm_fA_qB_outer(g, a, code, scale, mn, bits)
    assert out.shape == (B, nh, 1, D) and torch.isfinite(out).all()
    for (b, h) in [(0, 0), (17, 5), (31, 31), (8, 30)]:
        ref = oracle.bmm_fA_qB_outer(g, a[b:b + 1, h:h + 1].cpu().contiguous(), code[b:b + 1, h:h + 1].cpu(),
                                     scale[b:b + 1, h:h + 1].cpu(), mn[b:b + 1, h:h + 1].cpu(), bits)
        ok, ratio = gemv_close(out[b:b + 1, h:h + 1], ref)
        assert ok, (b, h, ratio)
    a2 = torch.softmax(torch.randn((B, nh, 1, Tv), device="cuda", generator=gen), -1).half()
    o2 = matmul.cuda_bmm_fA_qB_outer(g, a2, code, scale, mn, bits)
    o12 = matmul.cuda_bmm_fA_qB_outer(g, (a.float() + a2.float()).half(), code, scale, mn, bits)
    lin = (out.float() + o2.float() - o12.float()).abs()
    assert (lin / o12.float().pow(2).mean(-1, keepdim=True).sqrt()).max().item() < 6e-3


def test_mistral_7b_attention_module_vs_oracle(oracle):
    """The Mistral hook as a module (models/mistral_kivi.py:69-534): Mistral-7B attention config (4096 hidden, 32 / 8
    heads, sliding_window 4096, R=128), prompt 300, 100 decode steps across the K flush at 384; the module's own
    post-RoPE q / k / v drive hook_ref.  Final 9-tuple bit-identical, step outputs within the hook bar."""
    from types import SimpleNamespace

    import models.mistral_kivi as M
    from kivi_amd.cache import KiviCacheTuple
    from oracle import hook_ref as H
    cfg = SimpleNamespace(hidden_size=4096, num_attention_heads=32, num_key_value_heads=8, max_position_embeddings=32768,
                          rope_theta=1000000.0, sliding_window=4096, k_bits=2, v_bits=2, group_size=32, residual_length=128)
    torch.manual_seed(0)
    attn = M.MistralFlashAttention_KIVI(cfg).half().cuda()
    assert attn.sliding_window == 4096 and attn.num_key_value_groups == 4
    with torch.no_grad():
        attn.o_proj.weight.copy_(torch.eye(4096))
    B, T0, steps = 2, 300, 100
    captured = {}
    import kivi_amd.attention as A
    orig_dec, orig_pre = A.kivi_attention_decode, A.kivi_attention_prefill

    def spy_dec(q, k, v, layer, attention_mask=None, **kw):
        captured["qkv"] = (q.cpu(), k.cpu(), v.cpu())
        return orig_dec(q, k, v, layer, attention_mask, **kw)

    def spy_pre(q, k, v, layer, attention_mask=None):
        captured["qkv"] = (q.cpu(), k.cpu(), v.cpu())
        return orig_pre(q, k, v, layer, attention_mask)

    A.kivi_attention_decode, A.kivi_attention_prefill = spy_dec, spy_pre
    try:
        with torch.no_grad():
            x = torch.randn(B, T0, 4096, device="cuda", dtype=torch.float16) * 0.3
            _, w, past = attn(x, use_cache=True)
            assert w is None and isinstance(past, KiviCacheTuple) and past[-1] == T0
            _, k0, v0 = captured["qkv"]
            ref_past = H.prefill_cache(k0, v0, 2, 2, 32, 128)
            for s in range(steps):
                xs = torch.randn(B, 1, 4096, device="cuda", dtype=torch.float16) * 0.3
                out, _, past = attn(xs, past_key_value=past, use_cache=True)
                q, k, v = captured["qkv"]
                ref, ref_past = H.decode_step(q, k, v, ref_past, 2, 2, 32, 128)
                got = out.view(B, 1, 32, 128).transpose(1, 2)              # o_proj is the identity
                ok, ratio = gemv_close(got, ref, rtol=3e-3, ulps=1)
                assert ok, (s, ratio)
    finally:
        A.kivi_attention_decode, A.kivi_attention_prefill = orig_dec, orig_pre
    assert past[-1] == T0 + steps == ref_past[8]
    for n, a, r in zip(NAMES, past[:8], ref_past[:8]):
        if r is None:
            assert a is None or a.numel() == 0, n
        else:
            assert same_bits(a, r), n
