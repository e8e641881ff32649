#!/usr/bin/env python3
"""Pin the CPU oracle against the REAL reference and mint golden fixtures.

Runs only in the build container (needs /root/reference).  It imports the
reference's quant/new_pack.py unmodified, executes its pure-PyTorch functions
on seeded CPU inputs, checks that oracle/kivi_oracle.c reproduces every output
bit for bit (NaN->int in the reference's CPU flavour), and writes the inputs
and the REFERENCE's outputs to tests/golden/*.npz.  The committed fixtures are
what `pytest -m "not gpu"` (oracle vs reference) and `pytest -m gpu` (HIP vs
reference) check against on machines where /root/reference does not exist.

    python oracle/pin_reference.py            # verify + (re)write fixtures
    python oracle/pin_reference.py --check    # verify only
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import kivi_oracle as O  # noqa: E402

REF = os.environ.get("KIVI_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")


def load_reference():
    spec = importlib.util.spec_from_file_location("ref_new_pack", os.path.join(REF, "quant", "new_pack.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def u16(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def same_bits(a: torch.Tensor, b: torch.Tensor) -> bool:
    if a.dtype == torch.float16:
        return a.shape == b.shape and bool((a.contiguous().view(torch.int16) == b.contiguous().view(torch.int16)).all())
    return a.shape == b.shape and bool((a == b).all())


def gen_inputs(seed: int, shape, kind: str) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    if kind == "randn":
        return torch.randn(shape, generator=g).half()
    if kind == "randn_big":  # wide dynamic range, exercises fp16 rounding of x-mn and /scale
        return (torch.randn(shape, generator=g) * torch.exp(3 * torch.randn(shape, generator=g))).half()
    if kind == "int":        # quant/test.py:182 style integer-valued data
        return torch.randint(10, shape, generator=g).half()
    if kind == "const_groups":  # some groups constant -> scale 0 -> 0/0 = NaN path
        x = torch.randn(shape, generator=g).half()
        flat = x.view(-1, 32)
        flat[::3] = flat[::3, :1]
        flat[1] = 0
        return x
    raise ValueError(kind)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    warnings.filterwarnings("ignore")
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    R = load_reference()
    os.makedirs(GOLD, exist_ok=True)
    ok = True
    fixtures = {}

    # ---- 1. last-dim quantise+pack: quant_and_pack_vcache (new_pack.py:30-48) is the
    # pure-torch statement of triton_quantize_and_pack_along_last_dim (:217-252).
    lastdim_cases = [
        # name, shape (B,nh,rows,T'), group, bits, data kind
        ("cfg1_k_T", (1, 1, 128, 128), 32, 2, "randn"),     # BASELINE config 1, K transposed (D,T)
        ("cfg1_v", (1, 1, 128, 128), 32, 2, "randn"),
        ("v_b4", (2, 3, 5, 128), 32, 4, "randn"),
        ("v_b8", (1, 2, 4, 128), 64, 8, "randn"),
        ("v_g64", (2, 2, 6, 128), 64, 2, "randn_big"),
        ("v_g128_b4", (1, 2, 3, 256), 128, 4, "randn_big"),
        ("v_int", (1, 2, 8, 64), 32, 2, "int"),
        ("v_d64", (1, 1, 9, 64), 32, 2, "randn"),
        ("v_const", (1, 2, 6, 128), 32, 2, "const_groups"),
        ("v_const_b4", (1, 1, 6, 128), 32, 4, "const_groups"),
    ]
    for i, (name, shape, g, bits, kind) in enumerate(lastdim_cases):
        x = gen_inputs(100 + i, shape, kind)
        code, scale, mn = R.quant_and_pack_vcache(x.clone(), g, bits)
        o_code, o_scale, o_mn = O.quant_and_pack_vcache(x, g, bits, nan_mode=O.NAN_CPU)
        deq = R.unpack_and_dequant_vcache(code, scale, mn, g, bits)
        o_deq = O.unpack_and_dequant_vcache(code, scale, mn, g, bits)
        unp = R.unpack_tensor(code, bits, 3)
        o_unp = O.unpack_tensor(code, bits, 3)
        good = (same_bits(code, o_code) and same_bits(scale, o_scale) and same_bits(mn, o_mn)
                and same_bits(deq, o_deq) and same_bits(unp, o_unp))
        print(f"lastdim/{name:12s} shape={shape} g={g} bits={bits} {kind:12s} -> {'OK' if good else 'MISMATCH'}")
        ok &= good
        fixtures[f"lastdim_{name}"] = dict(x=u16(x), g=g, bits=bits, code=code.numpy(), scale=u16(scale.squeeze(-1)),
                                           mn=u16(mn.squeeze(-1)), deq=u16(deq), has_nan=int(kind == "const_groups"))

    # ---- 2. T-major K quantise+pack: quant_and_pack_kcache (new_pack.py:8-27)
    k_cases = [
        ("cfg1", (1, 1, 128, 128), 32, 2, "randn"),
        ("k_b4", (2, 2, 64, 128), 32, 4, "randn"),
        ("k_g64", (1, 3, 128, 64), 64, 2, "randn_big"),
        ("k_b8", (1, 1, 64, 32), 32, 8, "randn"),
        ("k_int", (1, 2, 64, 128), 32, 2, "int"),
    ]
    for i, (name, shape, g, bits, kind) in enumerate(k_cases):
        k = gen_inputs(200 + i, shape, kind)
        code, scale, mn = R.quant_and_pack_kcache(k.clone(), g, bits)
        o_code, o_scale, o_mn = O.quant_and_pack_kcache(k, g, bits, nan_mode=O.NAN_CPU)
        deq = R.unpack_and_dequant_kcache(code, scale, mn, g, bits)
        o_deq = O.unpack_and_dequant_kcache(code, scale, mn, g, bits)
        unp = R.unpack_tensor(code, bits, 2)
        o_unp = O.unpack_tensor(code, bits, 2)
        # cross-layout identity the hook relies on (llama_kivi.py:345): packing K^T along the
        # last dim gives the transpose of the T-major codes.
        c2, s2, m2 = R.quant_and_pack_vcache(k.transpose(2, 3).contiguous(), g, bits)
        ident = (same_bits(c2, code.transpose(2, 3).contiguous())
                 and same_bits(s2.squeeze(-1), scale.squeeze(-2).transpose(2, 3).contiguous())
                 and same_bits(m2.squeeze(-1), mn.squeeze(-2).transpose(2, 3).contiguous()))
        good = (same_bits(code, o_code) and same_bits(scale, o_scale) and same_bits(mn, o_mn)
                and same_bits(deq, o_deq) and same_bits(unp, o_unp) and ident)
        print(f"kcache/{name:12s} shape={shape} g={g} bits={bits} {kind:12s} -> {'OK' if good else 'MISMATCH'}")
        ok &= good
        fixtures[f"kcache_{name}"] = dict(k=u16(k), g=g, bits=bits, code=code.numpy(), scale=u16(scale),
                                          mn=u16(mn), deq=u16(deq))

    # ---- 3. pack_tensor / unpack_tensor round trip on raw integer codes (new_pack.py:86-129)
    gi = torch.Generator().manual_seed(7)
    for bits in (2, 4, 8):
        for pack_dim in (2, 3):
            data = torch.randint(0, 2 ** bits, (2, 2, 64, 32), generator=gi, dtype=torch.int32)
            code = R.pack_tensor(data, bits, pack_dim)
            good = same_bits(code, O.pack_tensor(data, bits, pack_dim))
            back = R.unpack_tensor(code, bits, pack_dim)
            good &= same_bits(back, O.unpack_tensor(code, bits, pack_dim)) and bool((back.int() == data).all())
            print(f"pack_tensor bits={bits} pack_dim={pack_dim} -> {'OK' if good else 'MISMATCH'}")
            ok &= good
            fixtures[f"packtensor_b{bits}_d{pack_dim}"] = dict(data=data.numpy(), bits=bits, pack_dim=pack_dim,
                                                               code=code.numpy())

    # ---- 4. fused-GEMV oracle: indexing / GQA mapping pinned through EXACT arithmetic.
    # Scales are powers of two, zero points and inputs small integers, so the reference's
    # fp16 dequant (unpack_and_dequant_vcache) and an fp64 matmul are exact and every
    # summation order gives the same fp32 value: the fused oracle must agree bit for bit.
    for gi_case, (name, (B, nh, nh_kv, K, N, g, bits)) in enumerate({
        "qk_mha": (2, 4, 4, 128, 256, 32, 2),   # qK^T: K=D, N=Tq
        "qk_gqa": (1, 8, 2, 128, 128, 32, 2),
        "sv_mha": (2, 2, 2, 96, 128, 32, 2),    # sV: K=Tv (not a multiple of 128), N=D
        "sv_gqa_b4": (1, 4, 1, 77, 128, 64, 4),
        "qk_b4": (1, 2, 2, 64, 192, 32, 4),
    }.items()):
        gg = torch.Generator().manual_seed(400 + gi_case)
        fpi = 32 // bits
        codes = torch.randint(0, 2 ** bits, (B, nh_kv, K, N), generator=gg, dtype=torch.int32)
        qB = R.pack_tensor(codes, bits, 3)
        scales = (2.0 ** torch.randint(-2, 2, (B, nh_kv, K, N // g), generator=gg)).half()
        zeros = (torch.randint(-8, 8, (B, nh_kv, K, N // g), generator=gg) / 4).half()
        fA = torch.randint(-4, 5, (B, nh, 1, K), generator=gg).half()
        deq = R.unpack_and_dequant_vcache(qB, scales.unsqueeze(-1), zeros.unsqueeze(-1), g, bits)  # (B,nh_kv,K,N)
        deq = deq.repeat_interleave(nh // nh_kv, dim=1)  # head h reads kv head h // ratio (gemv_cuda.cu:361-365)
        ref = torch.matmul(fA.double(), deq.double()).half()
        got = O.bmm_fA_qB_outer(g, fA, qB, scales, zeros, bits)
        got_nofma = O.bmm_fA_qB_outer(g, fA, qB, scales, zeros, bits, use_fma=False)
        got_fq = O.bmm_fA_qB_outer(g, fA, qB, scales, zeros, bits, fakequant=True)
        # and through the reference's transposed kernel-input layout (matmul.py:205,213-214)
        w_t = qB.reshape(-1, K, N // fpi).transpose(1, 2).contiguous()
        s_t = scales.reshape(-1, K, N // g).transpose(1, 2).contiguous()
        z_t = zeros.reshape(-1, K, N // g).transpose(1, 2).contiguous()
        got_t = O.gemv_forward_outer_dim(fA.reshape(B * nh, 1, K), w_t, s_t, z_t, bits, g, nh, nh_kv).view(B, nh, 1, N)
        good = same_bits(ref, got) and same_bits(ref, got_nofma) and same_bits(ref, got_fq) and same_bits(ref, got_t)
        print(f"gemv-exact/{name:10s} B={B} nh={nh}/{nh_kv} K={K} N={N} g={g} bits={bits} -> {'OK' if good else 'MISMATCH'}")
        ok &= good
        fixtures[f"gemvexact_{name}"] = dict(fA=u16(fA), qB=qB.numpy(), scales=u16(scales), zeros=u16(zeros), g=g,
                                             bits=bits, out=u16(ref))

    # ---- 5. reference fake-quant GEMV procedure on BASELINE config 1 (quant/test.py:187-195):
    # quant_and_pack_kcache -> unpack_and_dequant_kcache -> torch.matmul, randn inputs.
    g5 = torch.Generator().manual_seed(0)
    k = torch.randn((1, 1, 128, 128), generator=g5).half()
    q = torch.randn((1, 1, 1, 128), generator=g5).half()
    code, scale, mn = R.quant_and_pack_kcache(k.clone(), 32, 2)
    k_hat = R.unpack_and_dequant_kcache(code, scale, mn, 32, 2)
    ref_fq = torch.matmul(q.float(), k_hat.float().transpose(2, 3)).half()
    code_T = code.transpose(2, 3).contiguous()
    scale_T = scale.view(1, 1, -1, 128).transpose(2, 3).contiguous()
    mn_T = mn.view(1, 1, -1, 128).transpose(2, 3).contiguous()
    got_fq = O.bmm_fA_qB_outer(32, q, code_T, scale_T, mn_T, 2, fakequant=True)
    got_fused = O.bmm_fA_qB_outer(32, q, code_T, scale_T, mn_T, 2)
    err_fq = (got_fq.float() - ref_fq.float()).abs().max().item()
    rms = ref_fq.float().pow(2).mean().sqrt().item()
    err_fused = (got_fused.float() - ref_fq.float()).abs().max().item()
    good = err_fq <= 2e-3 * rms
    print(f"cfg1 fake-quant qK: max|oracle_fq-ref|={err_fq:.3e} (rms {rms:.2f}); fused-vs-fakequant {err_fused:.3e}"
          f" -> {'OK' if good else 'MISMATCH'}")
    ok &= good
    fixtures["cfg1_fakequant_qk"] = dict(k=u16(k), q=u16(q), code_T=code_T.numpy(), scale_T=u16(scale_T),
                                         mn_T=u16(mn_T), out_fakequant=u16(ref_fq))

    # ---- 6. fp16 division by the constant max_int: torch CUDA multiplies by 1/max_int in
    # fp32 (reciprocal), torch CPU divides.  Exhaustive proof that both round to the same half.
    allh = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).view(torch.float16)
    finite = torch.isfinite(allh)
    for maxq in (3, 15, 255):
        a = allh[finite].float()
        div = (a / maxq).half()
        rcp = (a * (torch.tensor(1.0, dtype=torch.float32) / maxq)).half()
        good = same_bits(div, rcp)
        print(f"x/{maxq} == x*(1/{maxq}) over all finite halves -> {'OK' if good else 'MISMATCH'}")
        ok &= good

    if not ok:
        print("ORACLE IS NOT PINNED: mismatch against the reference")
        return 1
    if not args.check:
        for name, d in fixtures.items():
            np.savez_compressed(os.path.join(GOLD, name + ".npz"), **d)
        total = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
        print(f"wrote {len(fixtures)} fixtures to {GOLD} ({total / 1024:.0f} KiB)")
    print("oracle pinned against", REF)
    return 0


if __name__ == "__main__":
    sys.exit(main())
