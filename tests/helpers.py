"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def f16(arr: np.ndarray) -> torch.Tensor:
    """uint16 bit pattern array -> fp16 tensor."""
    return torch.from_numpy(arr.astype(np.uint16).view(np.int16).copy()).view(torch.float16)


def bits(t: torch.Tensor) -> torch.Tensor:
    return t.detach().cpu().contiguous().view(torch.int16)


def same_bits(a: torch.Tensor, b: torch.Tensor) -> bool:
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.shape != b.shape:
        return False
    if a.dtype == torch.float16:
        return bool((bits(a) == bits(b)).all())
    return bool((a == b).all())


def load_golden(prefix: str):
    out = {}
    for p in sorted(glob.glob(os.path.join(GOLD, prefix + "*.npz"))):
        out[os.path.basename(p)[:-4]] = dict(np.load(p))
    assert out, f"no golden fixtures match {prefix}"
    return out


# worst ratios seen by gemv_close since the last drain (tests/conftest.py writes them to the ratio log after every test)
RATIOS = []


def gemv_close(got: torch.Tensor, ref: torch.Tensor, rtol: float = 1e-3, ulps: float = 0.0):
    """north_star bar for the fp16 GEMV, bare: |a-b| <= rtol * max(|ref|, rms(ref_row)) -- no ulp term.
    A correctly rounded fp16 result 1 ulp from a NORMAL reference already fits under 1e-3 (ulp <= 9.77e-4 relative), so the only
    slack kept is one subnormal ulp (2^-24 absolute) where |ref| is below the fp16 normal range, where a relative bar has no
    meaning.  `ulps` (default 0; stage A of the decode-step checks passes 1): the compared tensor is NOT a GEMV output but the row the
    softmax consumes, fp16(fp16(score) / sqrt(D)) (llama_kivi.py:339) -- a second fp16 rounding after the GEMV's, so two correct
    GEMVs one ulp apart can land two ulps apart there; that many fp16 ulps of |ref| are added to the bound, and the bar is logged as
    such.  The HOOK-LEVEL bars (decode-step outputs: 3e-3 end to end, 2e-3 for the attend half, 1.5e-3 between two forms of the same
    step -- this repo's own bars, north_star fixes only the GEMV's) also pass ulps=1: a step's output is fp16(fp16(packed part) +
    fp16(window part)), three fp16 roundings where a GEMV has one (round 5 hid this ulp inside gemv_close for EVERY comparison; now the
    GEMV comparisons are bare and the ones that carry the ulp say so and are logged as "<rtol>+1ulp").  Returns (ok, worst ratio against that bar); every call is recorded (RATIOS) and lands in the ratio log."""
    g, r = got.detach().cpu().float(), ref.detach().cpu().float()
    if not r.numel():
        return True, 0.0
    rms = r.pow(2).mean(dim=-1, keepdim=True).sqrt()
    bound = rtol * torch.maximum(r.abs(), rms)
    bound = torch.where(r.abs() < 2.0 ** -14, bound + 2.0 ** -24, bound)
    if ulps:
        bound = bound + ulps * torch.finfo(torch.float16).eps * r.abs().clamp_min(2.0 ** -14)
    err = (g - r).abs()
    # inf / nan: equal non-finite values agree, anything else is a miss
    same = (g == r) | (torch.isnan(g) & torch.isnan(r))
    ratio_t = torch.where(same, torch.zeros_like(err), err / bound.clamp_min(2.0 ** -126))
    ratio_t = torch.where(torch.isnan(ratio_t), torch.full_like(ratio_t, float("inf")), ratio_t)
    ratio = ratio_t.max().item()
    RATIOS.append((f"{rtol:g}" + (f"+{ulps:g}ulp" if ulps else ""), ratio, int(r.numel())))
    return ratio <= 1.0, ratio


def make_kv(seed: int, B: int, nh_kv: int, T: int, D: int, kind: str = "randn"):
    g = torch.Generator().manual_seed(seed)
    if kind == "randn":
        return torch.randn((B, nh_kv, T, D), generator=g).half()
    if kind == "int":
        return torch.randint(10, (B, nh_kv, T, D), generator=g).half()
    if kind == "outlier":  # a few large-magnitude channels, like real K caches (KIVI paper fig. 2)
        x = torch.randn((B, nh_kv, T, D), generator=g)
        x[..., ::17] *= 12.0
        return x.half()
    if kind == "tiny":     # fp16 subnormal neighbourhood: groups whose range is 0, 1 or a few subnormal ulps -- scale rounds to
        # 0 for a one-ulp range and the reference's d / 0 = inf -> max code (kivi_quant.h)
        base = torch.randint(0, 64, (B, nh_kv, 1, D), generator=g)
        bits = (base + torch.randint(0, 2, (B, nh_kv, T, D), generator=g) * torch.randint(0, 4, (B, nh_kv, 1, D), generator=g))
        sign = torch.randint(0, 2, (B, nh_kv, 1, D), generator=g) * 0x8000
        return (bits + sign).to(torch.int32).to(torch.uint16).view(torch.float16)
    raise ValueError(kind)
