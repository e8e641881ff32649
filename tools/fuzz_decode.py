"""Randomised parity sweep of the decode path: random model shapes, prompt lengths, bit widths, group sizes, residual lengths, masks
and input distributions through `kivi_attention_decode`, every unit of every step against the independent fp64 torch reference of
tests/torch_ref64.py (no oracle, no HIP code in the reference).  Bars as in tests/test_fullcover_gpu.py:

  * the 9-tuple after the prompt pass and after the last step: bit-identical;
  * matrix-pipe layout: the rows the softmax consumes at 1e-3 (+1 ulp: two fp16 roundings), the attend half on those rows at 2e-3
    (+1 ulp), masked scores identical;
  * the two GEMVs on the final cache (the reference's operator on the 9-tuple; kivi_gqa_scores / kivi_gqa_output on the matrix-pipe
    stores; peaked probability rows): the bare north_star bar, 1e-3;
  * end to end: reported against 3e-3, a case FAILS above 3x for scores below 4 (the reference softmax's own sensitivity to an ulp of a
    score), the allowance doubling with every binade of the largest score above that.

    python tools/fuzz_decode.py --seconds 600 --seed 1 > gpurun_out/fuzz.log

Every case is reproducible from its line (`--only SEED:INDEX`).  Exit code 1 if any case failed.
"""
import argparse
import math
import os
import random
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch_ref64 as T64          # noqa: E402
from helpers import gemv_close     # noqa: E402

NAMES = ["K_code_T", "K_full", "K_scale_T", "K_mn_T", "V_code", "V_full", "V_scale", "V_mn"]


def eq_bits(a, b):
    if a is None or b is None:
        return (a is None or a.numel() == 0) and (b is None or b.numel() == 0)
    if a.shape != b.shape:
        return False
    a, b = a.contiguous(), b.contiguous()
    if a.dtype == torch.float16:
        a, b = a.view(torch.int16), b.view(torch.int16)
    return bool(torch.equal(a, b))


def check_tuple(layer, past, what):
    t = layer.as_tuple()
    for n, a, r in zip(NAMES, t[:8], past[:8]):
        if not eq_bits(a, r):
            raise AssertionError(f"{what}: {n} differs")
    if t[8] != past[8]:
        raise AssertionError(f"{what}: length {t[8]} != {past[8]}")


def draw(rng: random.Random):
    ratio = rng.choice([1, 1, 2, 4, 4, 8])
    nh_kv = rng.choice([1, 2, 4, 8, 8, 32] if ratio == 1 else [1, 2, 4, 8])
    D = rng.choice([128, 128, 128, 64])
    g = rng.choice([32, 32, 32, 64])
    R = g * rng.choice([1, 2, 3, 4])
    if R > 128:
        R = 128
    k_bits = rng.choice([2, 2, 4])
    v_bits = k_bits if rng.random() < 0.85 else 6 - k_bits
    # prompt length: around the interesting boundaries (residual, super-block, launch-plan caps) or anywhere
    kind = rng.random()
    if kind < 0.15:
        T0 = rng.randint(1, 2 * R + 2)
    elif kind < 0.45:
        edge = rng.choice([512, 1024, 2048, 4096, 4608, 8192, 8320, 9216, 9344, 16384])
        T0 = max(1, edge + rng.randint(-R - 3, R + 3))
    else:
        T0 = rng.randint(1, rng.choice([600, 3000, 9000, 20000]))
    budget = 3_000_000                                              # B * nh_kv * T tokens: fp64 temporaries of ~3 GB
    Bmax = max(1, min(48, budget // (nh_kv * (T0 + 200))))
    B = rng.randint(1, Bmax)
    steps = rng.choice([1, 3, R + 2, rng.randint(1, 2 * R + 2)])
    if B * nh_kv * T0 > 1_000_000:
        steps = min(steps, 6)
    dist = rng.choice(["randn", "randn", "outlier", "small", "big", "mixed"])
    masked = rng.random() < 0.4
    flags = 0
    if rng.random() < 0.2:
        flags = rng.choice([1, 2, 2 << 8, 3 << 8, 4 << 8])          # KIVI_GQA_FORCE_SPLIT / _FORCE_ROW / KIVI_GQA_SLICES(n); an infeasible request falls back to the plan's two launches
    return dict(B=B, nh=nh_kv * ratio, nh_kv=nh_kv, D=D, g=g, R=R, k_bits=k_bits, v_bits=v_bits, T0=T0, steps=steps, dist=dist,
                masked=masked, flags=flags)


def make(shape, dist, gen):
    x = torch.randn(shape, device="cuda", dtype=torch.float32, generator=gen)
    if dist == "outlier":                                           # a few channels carry 8x the magnitude (what per-channel K quantisation is for)
        ch = torch.rand((shape[-1],), device="cuda", generator=gen) < 0.05
        x = x * torch.where(ch, 8.0, 1.0)
    elif dist == "small":
        x = x * 0.02
    elif dist == "big":
        x = x * 30.0
    elif dist == "mixed":                                           # per (batch row, head) scale over five decades: both range-word marks
        s = torch.exp(torch.randn(shape[:2] + (1, 1), device="cuda", generator=gen) * 3.0).clamp(1e-3, 300.0)
        x = x * s
    return x.half()


VERBOSE = False


def diagnose(layer, out, ref_b, x_gpu, past, c, s):
    """Stage-B failure: which units, and what their stores / probabilities look like."""
    from kivi_amd.quant import mfma
    g_, r_ = out.float(), ref_b.float()
    rms = r_.pow(2).mean(dim=-1, keepdim=True).sqrt()
    ratio = ((g_ - r_).abs() / (2e-3 * torch.maximum(r_.abs(), rms)).clamp_min(2.0 ** -24)).amax(dim=(-1, -2))     # (B, nh)
    B, nh = ratio.shape
    rat = nh // c["nh_kv"]
    kflag, vflag = mfma.range_flags(layer.kt), mfma.range_flags(layer.vt)
    p = torch.softmax(x_gpu.float(), -1).half()
    Lv = past[5].shape[2] + 1
    vs = past[6]
    order = ratio.flatten().argsort(descending=True)[:4]
    print(f"  step {s}: units over the bar: {(ratio > 1).sum().item()} of {B * nh}", flush=True)
    # (a) the packed sV product alone, on the REFERENCE's probabilities (no exponential of ours in it): kivi_gqa_output against fp64
    # (b) how sharp the reference's own rounding of p is: fp16(fp32 softmax) against fp16(fp64 softmax) of the same rows
    Tv = vs.shape[2] if vs is not None else 0
    part = None
    if Tv:
        pitch = (Tv + 7) // 8 * 8
        ap = torch.zeros((B, nh, 1, pitch), dtype=torch.float16, device="cuda")
        ap[..., :Tv] = p[..., :Tv]
        got = mfma.gqa_output(ap, layer.vt, Tv, None, c["g"], c["v_bits"]).float()
        want = T64.output64(p[..., :Tv], past[4], past[6], past[7], c["g"], c["v_bits"]).float()
        part = ((got - want).abs() / (1e-3 * torch.maximum(want.abs(), want.pow(2).mean(-1, keepdim=True).sqrt())).clamp_min(2.0 ** -24)).amax(dim=(-1, -2))
    rat_ = nh // c["nh_kv"]
    vwin = torch.cat([past[5], torch.zeros_like(past[5][:, :, :1])], 2)     # (the new token's values are not in `past`: magnitudes only)
    win_rms = torch.matmul(p[..., -Lv:].double(), T64._expand_heads(vwin, rat_).double()).pow(2).mean(-1).sqrt()[..., 0]
    pk_rms = want.double().pow(2).mean(-1).sqrt()[..., 0] if Tv else torch.zeros_like(win_rms)
    p64 = torch.softmax(x_gpu.double(), -1).to(torch.float32).half()
    flips = (p64 != p).sum(dim=(-1, -2))
    for o in order.tolist():
        b, h = o // nh, o % nh
        hk = h // rat
        pk = p[b, h, 0, :-Lv].float()
        pwin = p[b, h, 0, -Lv:].float()
        vsc = vs[b, hk].float() if vs is not None else torch.zeros(1)
        e = (g_[b, h, 0] - r_[b, h, 0]).abs().argmax().item()
        print(f"    (b {b}, h {h}) ratio {ratio[b, h].item():.2f}: vt word {vflag[b, hk].item():#x} kt word {kflag[b, hk].item():#x}; packed V scales "
              f"{vsc.min().item():.3g} .. {vsc.max().item():.3g}; window |v| max {past[5][b, hk].abs().max().item():.3g}; p: packed mass {pk.sum().item():.3g} "
              f"max {pk.max().item():.3g}, window mass {pwin.sum().item():.3g}; out[{e}] {g_[b, h, 0, e].item():.6g} ref {r_[b, h, 0, e].item():.6g} row rms {rms[b, h, 0, 0].item():.3g}; packed product on the reference's p: {(part[b, h].item() if part is not None else 0):.3f} of 1e-3; "
              f"p values that differ between an fp32 and an fp64 softmax: {flips[b, h].item()}; rms of the packed part {pk_rms[b, h].item():.3g}, of the window part (without the new token) {win_rms[b, h].item():.3g}", flush=True)


def run_case(c, seed):
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    B, nh, nh_kv, D, g, R, T0, steps = c["B"], c["nh"], c["nh_kv"], c["D"], c["g"], c["R"], c["T0"], c["steps"]
    kb, vb = c["k_bits"], c["v_bits"]
    cfg = KiviConfig(kb, vb, g, R)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    k0 = make((B, nh_kv, T0, D), c["dist"], gen)
    v0 = make((B, nh_kv, T0, D), c["dist"] if c["dist"] != "outlier" else "randn", gen)
    try:
        layer = make_layer_cache(cfg, B, nh_kv, D, T0 + steps + 1, "cuda", num_heads=nh)
    except AssertionError as e:                                     # a combination the cache classes refuse by contract
        return f"skipped (refused: {e})", {"A": 0.0, "B": 0.0, "E": 0.0, "G": 0.0}, None
    mf = getattr(layer, "layout", "") == "mfma"
    if mf:
        layer.flags |= _lib.GQA_DUMP_SCORES | c["flags"]
    layer.prefill(k0, v0)
    past = T64.prefill_cache(k0, v0, kb, vb, g, R)
    del k0, v0
    check_tuple(layer, past, "after the prompt pass")
    worst = {"A": 0.0, "B": 0.0, "E": 0.0, "G": 0.0}
    for s in range(steps):
        q = make((B, nh, 1, D), "randn" if c["dist"] in ("mixed", "outlier") else c["dist"], gen)
        kn = make((B, nh_kv, 1, D), c["dist"], gen)
        vn = make((B, nh_kv, 1, D), c["dist"] if c["dist"] != "outlier" else "randn", gen)
        n = T0 + s + 1
        mask = None
        if c["masked"] and s % 2 == 0:
            mask = torch.zeros((B, 1, 1, n), dtype=torch.float16, device="cuda")
            for b in range(0, B, 2):
                mask[b, :, :, : min(n - 1, (7 + 13 * b + s) % max(1, n))] = torch.finfo(torch.float16).min
        out = kivi_attention_decode(q, kn, vn, layer, attention_mask=mask)
        ref, new_past, pre = T64.decode_step(q, kn, vn, past, kb, vb, g, R, attention_mask=mask)
        if not torch.isfinite(ref.float()).all():
            return "skipped (the reference itself overflows fp16)", worst, mf
        if mf:
            x_gpu = layer._native[4][0][:B, :nh, :, :n]
            live = pre.float() > -60000
            ok, ra = gemv_close(torch.where(live, x_gpu.float(), 0.0), torch.where(live, pre.float(), 0.0), rtol=1e-3, ulps=1)
            worst["A"] = max(worst["A"], ra)
            if not ok:
                raise AssertionError(f"step {s}: scores ratio {ra:.3f} of 1e-3 (+1 ulp)")
            # masked positions: fp16(x + finfo.min) keeps x at the spacing of that binade (32), so two scores one ulp apart may land
            # one such step apart; identical whenever |x| < 16 (tests/test_fullcover_gpu.py asserts equality on randn inputs)
            xm, pm = x_gpu[~live].float(), pre[~live].float()
            if not bool(((xm == pm) | ((xm - pm).abs() <= 32.0)).all()):           # (equal covers -inf on both sides: inputs that overflow fp16 scores)
                raise AssertionError(f"step {s}: masked scores differ")
            ref_b, _, _ = T64.decode_step(q, kn, vn, past, kb, vb, g, R, attention_mask=mask, scores_override=x_gpu.contiguous())
            ok, rb = gemv_close(out, ref_b, rtol=2e-3, ulps=1)
            worst["B"] = max(worst["B"], rb)
            if not ok:
                if VERBOSE:
                    diagnose(layer, out, ref_b, x_gpu, past, c, s)
                raise AssertionError(f"step {s}: attend half ratio {rb:.3f} of 2e-3 (+1 ulp)")
        # end to end: the reference softmax turns one fp16 ulp u of a score into a relative change u of its probability (|x| in [2, 4):
        # u = 2e-3, the case the 2 x 3e-3 allowance was measured on); larger scores carry larger ulps, and the allowance follows them
        _, re_ = gemv_close(out, ref, rtol=3e-3)
        smax = pre.float().abs().masked_fill(pre.float() <= -60000, 0).max().item()
        allow = 3.0 * max(1.0, 2.0 ** (math.floor(math.log2(max(smax, 1e-9))) - 10) / 2.0 ** -9)
        worst["E"] = max(worst["E"], re_ / allow)
        if re_ > allow:
            raise AssertionError(f"step {s}: output ratio {re_:.3f} of 3e-3 (allowed {allow:.1f}: largest |score| {smax:.3g})")
        past = new_past
    check_tuple(layer, past, "after the last step")
    worst["G"] = check_ops(layer, past, c, gen, mf)
    return "ok", worst, mf


def check_ops(layer, past, c, gen, mf):
    """The two GEMVs on the final cache, bare north_star bar (1e-3): the reference's operator on the 9-tuple (hook-state kernels) and,
    on the matrix-pipe layout, kivi_gqa_scores / kivi_gqa_output on the layer's own stores.  Probabilities: peaked rows."""
    from kivi_amd.quant import matmul, mfma
    B, nh, D, g = c["B"], c["nh"], c["D"], c["g"]
    kc, _, ks, km, vc, _, vs, vm, _ = past
    worst = 0.0
    q = make((B, nh, 1, D), "randn", gen)
    if kc is not None:
        Tq = kc.shape[-1] * (32 // c["k_bits"])
        ref = T64.scores64(q, kc, ks, km, g, c["k_bits"])
        if torch.isfinite(ref.float()).all():
            ok, r = gemv_close(matmul.cuda_bmm_fA_qB_outer(g, q, kc, ks, km, c["k_bits"]), ref)
            worst = max(worst, r)
            if not ok:
                raise AssertionError(f"qK^T through the reference operator: ratio {r:.3f} of 1e-3")
            if mf:
                o2 = torch.empty((B, nh, 1, Tq), dtype=torch.float16, device="cuda")
                mfma.gqa_scores(q, layer.kt, Tq, o2, g, c["k_bits"])
                ok, r = gemv_close(o2, ref)
                worst = max(worst, r)
                if not ok:
                    raise AssertionError(f"kivi_gqa_scores: ratio {r:.3f} of 1e-3")
    if vc is not None:
        Tv = vc.shape[2]
        sharp = (0.5, 3.0, 12.0)[int(torch.randint(3, (1,), generator=torch.Generator().manual_seed(Tv + B)).item())]
        a = torch.softmax(torch.randn((B, nh, 1, Tv), device="cuda", generator=gen) * sharp, -1).half()
        ref = T64.output64(a, vc, vs, vm, g, c["v_bits"])
        if torch.isfinite(ref.float()).all():
            ok, r = gemv_close(matmul.cuda_bmm_fA_qB_outer(g, a, vc, vs, vm, c["v_bits"]), ref)
            worst = max(worst, r)
            if not ok:
                raise AssertionError(f"sV through the reference operator: ratio {r:.3f} of 1e-3")
            if mf:
                pitch = (Tv + 7) // 8 * 8
                ap = torch.zeros((B, nh, 1, pitch), dtype=torch.float16, device="cuda")
                ap[..., :Tv] = a
                ok, r = gemv_close(mfma.gqa_output(ap, layer.vt, Tv, None, g, c["v_bits"]), ref)
                worst = max(worst, r)
                if not ok:
                    raise AssertionError(f"kivi_gqa_output: ratio {r:.3f} of 1e-3")
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cases", type=int, default=100000)
    ap.add_argument("--only", default=None, help="SEED:INDEX of one case to re-run")
    ap.add_argument("--verbose", action="store_true", help="per-unit diagnostics of a failing attend half")
    a = ap.parse_args()
    global VERBOSE
    VERBOSE = a.verbose
    t0 = time.time()
    n_ok = n_fail = n_skip = n_mf = 0
    worst_all = {"A": 0.0, "B": 0.0, "E": 0.0, "G": 0.0}
    idx = 0
    if a.only:
        a.seed, idx = (int(x) for x in a.only.split(":"))
        a.max_cases = idx + 1
    print(f"# tools/fuzz_decode.py --seconds {a.seconds} --seed {a.seed}: case index, configuration, layout, status, worst ratios "
          f"(scores of 1e-3 +1 ulp / attend of 2e-3 +1 ulp / output of its allowance / the two GEMVs on the final cache of the bare 1e-3)", flush=True)
    while idx < a.max_cases and (a.only or time.time() - t0 < a.seconds):
        rng = random.Random(a.seed * 1_000_003 + idx)
        c = draw(rng)
        try:
            status, worst, mf = run_case(c, a.seed * 7919 + idx)
        except AssertionError as e:
            status, worst, mf = f"FAIL: {e}", {"A": -1, "B": -1, "E": -1, "G": -1}, None
        except Exception as e:                                      # an error code of the library, an unsupported combination
            status, worst, mf = f"ERROR: {type(e).__name__}: {e}", {"A": -1, "B": -1, "E": -1, "G": -1}, None
            traceback.print_exc(file=sys.stderr)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if status == "ok":
            n_ok += 1
            n_mf += bool(mf)
            for k in worst_all:
                worst_all[k] = max(worst_all[k], worst[k])
        elif status.startswith("skipped"):
            n_skip += 1
        else:
            n_fail += 1
        print(f"{a.seed}:{idx} {c} {'matrix-pipe' if mf else 'hook-state' if mf is not None else '?'} {status} "
              f"{worst['A']:.3f} {worst['B']:.3f} {worst['E']:.3f} {worst.get('G', 0.0):.3f}", flush=True)
        idx += 1
    print(f"# {n_ok} ok ({n_mf} on the matrix-pipe layout), {n_skip} skipped, {n_fail} FAILED in {time.time() - t0:.0f} s; worst ratios over the ok cases: "
          f"scores {worst_all['A']:.3f}, attend {worst_all['B']:.3f}, output {worst_all['E']:.3f} (of its allowance), GEMVs {worst_all['G']:.3f} of the bare 1e-3", flush=True)
    sys.exit(1 if n_fail else 0)


if __name__ == "__main__":
    main()
