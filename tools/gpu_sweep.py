#!/usr/bin/env python3
"""One-shot GPU diagnostic: parity + timing of every compiled kernel variant at BASELINE config 2.

Writes gpurun_out/sweep.json and prints a table.  Timing: per-launch HIP events on the launch stream,
rotating over NBUF independent caches so the ~200 MB working set cannot live in the 256 MiB Infinity Cache.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kivi_amd._lib import KiviHipError  # noqa: E402
from kivi_amd.quant import matmul, new_pack  # noqa: E402


def time_launches(fn, nbuf, iters, warm=3):
    for i in range(warm):
        fn(i % nbuf)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for i in range(iters):
        evs[i][0].record()
        fn(i % nbuf)
        evs[i][1].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)  # us
    return ts[len(ts) // 2], ts[0], ts[-1]


def time_dispatches(fn, nbuf, iters, warm=3):
    """Per-dispatch kernel durations (hipExtLaunchKernelGGL start/stop events): what rocprofv3 reports."""
    from kivi_amd import _lib
    lib = _lib.load()
    for i in range(warm):
        fn(i % nbuf)
    torch.cuda.synchronize()
    evs = [(lib.kivi_event_create(), lib.kivi_event_create()) for _ in range(iters)]
    for i in range(iters):
        lib.kivi_set_launch_events(evs[i][0], evs[i][1])
        fn(i % nbuf)
    torch.cuda.synchronize()
    ts = sorted(lib.kivi_event_elapsed_us(a, b) for a, b in evs)
    for a, b in evs:
        lib.kivi_event_destroy(a)
        lib.kivi_event_destroy(b)
    return ts[len(ts) // 2], ts[0], ts[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--nh", type=int, default=32)
    ap.add_argument("--nh_kv", type=int, default=32)
    ap.add_argument("--T", type=int, default=4096)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--g", type=int, default=32)
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--nbuf", type=int, default=24)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--skip_pack", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--kcap", type=int, default=0, help="allocate K buffers with this token capacity (row stride) > T")
    args = ap.parse_args()
    B, nh, nh_kv, T, D, g, bits = args.B, args.nh, args.nh_kv, args.T, args.D, args.g, args.bits
    fpi = 32 // bits
    dev = torch.device("cuda:0")
    props = torch.cuda.get_device_properties(0)
    print(f"device: {props.name}  CUs={props.multi_processor_count}  mem={props.total_memory / 2**30:.0f} GiB")
    res = {"config": vars(args), "device": props.name, "k": [], "v": [], "pack": []}
    torch.manual_seed(0)

    # ---- buffers: buffer 0 = really quantised randn K / V, others = random bits (same bytes moved)
    t0 = time.time()
    k = torch.randn((B, nh_kv, T, D), device=dev, dtype=torch.float16)
    kc, ks, km = new_pack.quantize_and_pack_k_tmajor(k, g, bits)
    vc, vs, vm = new_pack.triton_quantize_and_pack_along_last_dim(k, g, bits)  # same tensor as V
    if args.kcap:
        cap = args.kcap
        kc2 = torch.zeros((B, nh_kv, D, cap // fpi), device=dev, dtype=torch.int32)
        ks2 = torch.zeros((B, nh_kv, D, cap // g), device=dev, dtype=torch.float16)
        km2 = torch.zeros_like(ks2)
        kc2[..., : T // fpi] = kc
        ks2[..., : T // g] = ks
        km2[..., : T // g] = km
        kc, ks, km = kc2[..., : T // fpi], ks2[..., : T // g], km2[..., : T // g]
        print("K row strides:", kc.stride(), ks.stride())
    Kbufs, Vbufs = [(kc, ks, km)], [(vc, vs, vm)]
    for i in range(1, args.nbuf):
        if args.kcap:
            c = torch.randint(-2**31, 2**31 - 1, (B, nh_kv, D, args.kcap // fpi), device=dev, dtype=torch.int32)[..., : T // fpi]
            s = (torch.rand((B, nh_kv, D, args.kcap // g), device=dev) + 0.5).half()[..., : T // g]
            m = torch.randn((B, nh_kv, D, args.kcap // g), device=dev).half()[..., : T // g]
        else:
            c = torch.randint(-2**31, 2**31 - 1, kc.shape, device=dev, dtype=torch.int32)
            s = (torch.rand(ks.shape, device=dev) + 0.5).half()
            m = torch.randn(km.shape, device=dev).half()
        Kbufs.append((c, s, m))
        c = torch.randint(-2**31, 2**31 - 1, vc.shape, device=dev, dtype=torch.int32)
        s = (torch.rand(vs.shape, device=dev) + 0.5).half()
        m = torch.randn(vm.shape, device=dev).half()
        Vbufs.append((c, s, m))
    q = torch.randn((B, nh, 1, D), device=dev, dtype=torch.float16)
    a = torch.softmax(torch.randn((B, nh, 1, T), device=dev), -1).half()
    torch.cuda.synchronize()
    print(f"buffers ready in {time.time() - t0:.1f}s, allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB")

    kbytes = B * nh_kv * (D * T * bits // 8 + 2 * D * (T // g) * 2) + B * nh * (D * 2 + T * 2)
    vbytes = B * nh_kv * (T * D * bits // 8 + 2 * T * (D // g) * 2) + B * nh * (T * 2 + D * 2)
    res["k_bytes"], res["v_bytes"] = kbytes, vbytes

    # ---- copy baseline (read + write 256 MiB each)
    src = torch.empty(64 * 2**20, device=dev, dtype=torch.int32)
    dsts = [torch.empty_like(src) for _ in range(3)]
    med, mn_, mx_ = time_launches(lambda i: dsts[i % 3].copy_(src), 3, 20)
    res["copy_256MiB_us"] = med
    print(f"torch copy 256 MiB: median {med:.1f} us -> {2 * src.numel() * 4 / med / 1e6:.2f} TB/s (read+write)")
    del src, dsts

    # ---- reference outputs from the generic-most variant for cross-checking
    names = matmul.bmm_variants(include_diagnostic=True)
    ref_k = None
    ref_v = None
    for kind, vid, name in names:
        if args.only and args.only not in name:
            continue
        bufs, x, nbytes, key = (Kbufs, q, kbytes, "k") if kind == "k" else (Vbufs, a, vbytes, "v")
        try:
            out0 = matmul.bmm_fA_qB_outer_variant(kind, vid, g, x, *bufs[0], bits)
        except KiviHipError:
            continue
        torch.cuda.synchronize()
        if kind == "k":
            if ref_k is None:
                ref_k = out0.float()
            ref = ref_k
        else:
            if ref_v is None:
                ref_v = out0.float()
            ref = ref_v
        rms = ref.pow(2).mean(-1, keepdim=True).sqrt()
        err = ((out0.float() - ref).abs() / torch.maximum(ref.abs(), rms)).max().item() if "_m3_" not in name else -1.0
        med, mn_, mx_ = time_dispatches(lambda i: matmul.bmm_fA_qB_outer_variant(kind, vid, g, x, *bufs[i], bits),
                                        args.nbuf, args.iters)
        tbs = nbytes / med / 1e6
        print(f"{name:38s} median {med:8.1f} us  min {mn_:8.1f}  max {mx_:8.1f}  {tbs:6.2f} TB/s  "
              f"{100 * tbs / 8.0:5.1f}% of 8TB/s  relerr-vs-first {err:.2e}")
        res[key].append(dict(name=name, median_us=med, min_us=mn_, max_us=mx_, tbps=tbs, err_vs_first=err))
    # default dispatch
    for label, x, bufs, nbytes in (("default qK", q, Kbufs, kbytes), ("default sV", a, Vbufs, vbytes)):
        med, mn_, mx_ = time_dispatches(lambda i: matmul.cuda_bmm_fA_qB_outer(g, x, *bufs[i], bits), args.nbuf, args.iters)
        print(f"{label:38s} median {med:8.1f} us  min {mn_:8.1f}  {nbytes / med / 1e6:6.2f} TB/s")
        res[label.replace(" ", "_")] = dict(median_us=med, min_us=mn_, tbps=nbytes / med / 1e6)

    if not args.skip_pack:
        n = k.numel()
        for label, fn, bytes_ in (
            ("pack lastdim (V prefill)", lambda i: new_pack.triton_quantize_and_pack_along_last_dim(k, g, bits), n * 2.375),
            ("pack K t-major (fused transpose)", lambda i: new_pack.quantize_and_pack_k_tmajor(k, g, bits), n * 2.375),
            ("K transpose copy + lastdim (reference route)",
             lambda i: new_pack.triton_quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), g, bits), n * 6.375),
            ("unpack+dequant lastdim", lambda i: new_pack.unpack_and_dequant_vcache(vc, vs.unsqueeze(-1), vm.unsqueeze(-1), g, bits),
             n * 2.375),
        ):
            med, mn_, mx_ = time_launches(fn, 1, 10, warm=2)
            print(f"{label:46s} median {med:9.1f} us  {bytes_ / med / 1e6:6.2f} TB/s (algorithmic)")
            res["pack"].append(dict(name=label, median_us=med, tbps=bytes_ / med / 1e6))
        v1 = torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16)
        med, mn_, mx_ = time_launches(lambda i: new_pack.triton_quantize_and_pack_along_last_dim(v1, g, bits), 1, 50)
        print(f"{'pack one V token (B,nh,1,D)':46s} median {med:9.1f} us")
        res["pack"].append(dict(name="pack one V token", median_us=med))

    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w") as f:
        json.dump(res, f, indent=1)
    best = sorted(res["k"], key=lambda r: r["median_us"])[:5]
    print("best K variants:", [(r["name"], round(r["median_us"], 1)) for r in best])
    best = sorted(res["v"], key=lambda r: r["median_us"])[:5]
    print("best V variants:", [(r["name"], round(r["median_us"], 1)) for r in best])


if __name__ == "__main__":
    main()
