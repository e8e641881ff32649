#!/usr/bin/env python3
"""bench.py -- decode-step throughput of the KIVI KV-cache hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric: "decode-step tokens/sec + peak KV bytes, Llama-2-7B B=32 seq=4k, 2b/2b g=32"):
one "step" = one decode step of the KIVI attention hot path for ALL 32 layers of a Llama-2-7B-shaped model at
batch 32 per GPU over a 4096-token prompt cache, k_bits = v_bits = 2, group 32, residual 32 -- per layer:
fused qK^T GEMV over the packed per-channel K cache + fp16 residual scores, fp32 softmax, fused sV GEMV over the
packed per-token V cache + fp16 residual, and the in-place cache append / quantise (kivi_amd.attention
.kivi_attention_decode, the reference's llama_kivi.py:314-399).  The dense projections / MLP are outside the
hot path and are not run.  Inputs are synthetic (seeded randn), resident in HBM before the timed region.

The default prompt is 4080 tokens ("seq = 4k"): the fp16 K residual then starts at 16 tokens, so the K flush of R = 32
tokens (llama_kivi.py:343-356) happens INSIDE the timed region (warm-up 5 + timed step 11) instead of never.

N > 1: one process per GPU, batch-sharded replicas -- the path has no exchange step, so there is no data-path
collective; ranks only barrier and reduce the elapsed time (MAX = the job's time, all-gather = per-rank times).
`python bench.py --gpus N` starts its N ranks itself (re-executes this file once per GPU with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, backend nccl = RCCL); under `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N` it uses the ranks it was given.  Rank 0 prints ONE JSON line.
`--dry-run` replaces the GPU work by a sleep (CPU / gloo: exercises launcher, barriers, reductions and the JSON line).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0  # same guide: float4 copy ceiling


def kgemv_bytes(B, nh, nh_kv, D, T, g, bits):
    """ALGORITHMIC bytes of one fused qK^T launch: codes + scale + zero + q + out."""
    per_kv_head = D * T * bits // 8 + 2 * D * (T // g) * 2
    return B * nh_kv * per_kv_head + B * nh * (D * 2 + T * 2)


def vgemv_bytes(B, nh, nh_kv, D, Tv, g, bits):
    per_kv_head = Tv * D * bits // 8 + 2 * Tv * (D // g) * 2
    return B * nh_kv * per_kv_head + B * nh * (Tv * 2 + D * 2)


def row_bytes(info):
    """ALGORITHMIC bytes of ONE fused decode-row launch from the launch hook's description of the step:
    packed K + packed V (codes + scale + zero) + the fp16 residual / window + q + out; no score round trip."""
    Bq, nhq, nkv, Dq = info["B"], info["nh"], info["nh_kv"], info["K"]
    kb = kgemv_bytes(Bq, nhq, nkv, Dq, info["N"], info["group_size"], info["bits"])
    return (kb - Bq * nhq * info["N"] * 2
            + vgemv_bytes(Bq, nhq, nkv, Dq, info["Tv"], info["group_size"], info["v_bits"]) - Bq * nhq * info["Tv"] * 2
            + Bq * nkv * (info["k_res"] + info["v_res"]) * Dq * 2)


def pmc_traffic_entry(kname, want):
    """HBM bytes per launch of `kname` at configuration `want` from the newest tracked rocprofv3 --pmc passes (profiles/)."""
    profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    for pf in reversed(profs):
        try:
            pj = json.load(open(pf))
        except Exception:
            continue
        ent = next((e for key, e in pj.get("kernels", {}).items() if key.split("@")[0] == kname and e.get("config") == want), None)
        if ent:
            return ent["hbm_bytes_per_launch"], "profiles/" + os.path.basename(pf)
    return None, None


# The other BASELINE.json configurations, measured by the default invocation so that the driver's own run carries them
# (VERDICT r5 weak #9): a few layers of each shape, every launch of the fused kernel bracketed by HIP events.
EXTRA_CONFIGS = [
    ("roofline_config4", "BASELINE configs[3]: Llama-3-8B attention shape, B=64, 32 / 8 heads, 8k keys, 2b/2b g=32 R=128",
     dict(B=64, nh=32, nh_kv=8, T0=8064, R=128, bits=2)),
    ("roofline_config4_4bit", "the same geometry with 4-bit K / V (the reference's Mistral-7B + KIVI-4 shape, docs/long_bench.md:35-53)",
     dict(B=64, nh=32, nh_kv=8, T0=8064, R=128, bits=4)),
    ("roofline_config5_slice", "BASELINE configs[4], one GPU's share of the batch-sharded job: Mistral-7B attention shape, B=16, 32 / 8 heads, "
     "32k keys, 2b/2b g=32 R=128", dict(B=16, nh=32, nh_kv=8, T0=32640, R=128, bits=2)),
]


def extra_config_roofline(label, c, dev, klib, D=128, g=32, n_layers=8, steps=42, warmup=2, event_every=3):
    """`n_layers` layer caches of the shape (each ~0.4-1 GB: together far beyond the 256 MiB Infinity Cache), `warmup` + `steps`
    decode steps over them, every `event_every`-th fused launch of the timed steps bracketed by a HIP event pair (>= 100 timed
    launches; as for the headline the launches in between run unbracketed, so a timed kernel follows an ordinary one)."""
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from kivi_amd.quant import matmul
    B, nh, nh_kv, T0, R, bits = c["B"], c["nh"], c["nh_kv"], c["T0"], c["R"], c["bits"]
    cfg = KiviConfig(bits, bits, g, R)
    layers = []
    for _ in range(n_layers):
        lc = make_layer_cache(cfg, B, nh_kv, D, T0 + warmup + steps + 1, dev, num_heads=nh)
        k = torch.randn((B, nh_kv, T0, D), device=dev, dtype=torch.float16)
        v = torch.randn((B, nh_kv, T0, D), device=dev, dtype=torch.float16)
        lc.prefill(k, v)
        del k, v
        layers.append(lc)
    q = torch.randn((B, nh, 1, D), device=dev, dtype=torch.float16)
    kn = torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16)
    vn = torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16)
    ev, timing, count = [], [False], [0]

    def hook(phase, kind, info):
        if kind == "k" and phase == "pre" and timing[0]:
            count[0] += 1
            if count[0] % event_every:
                return
            pair = (klib.kivi_event_create(), klib.kivi_event_create())
            klib.kivi_set_launch_events(*pair)
            ev.append((pair, row_bytes(info) if "Tv" in info else None))

    matmul.launch_hook = hook
    try:
        torch.cuda.synchronize()
        t0 = None
        for s in range(warmup + steps):
            if s == warmup:
                torch.cuda.synchronize()
                timing[0] = True
                t0 = time.perf_counter()
            for lc in layers:
                kivi_attention_decode(q, kn, vn, lc)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    finally:
        matmul.launch_hook = None
    kname = (klib.kivi_last_timed_kernel() or b"").decode().split("<")[0].strip("( ")
    us = sorted(klib.kivi_event_elapsed_us(a, b) for (a, b), _ in ev)
    for (a, b), _ in ev:
        klib.kivi_event_destroy(a)
        klib.kivi_event_destroy(b)
    nb = [r for _, r in ev if r is not None]
    if not us or len(nb) != len(us):
        return {"workload": label, "error": f"no fused launch was timed (kernel {kname!r})"}
    n = len(us)
    bytes_ = sum(nb) // n
    med, avg = us[n // 2], sum(us) / n
    fr = lambda t: round(bytes_ / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)       # noqa: E731
    traffic, src = pmc_traffic_entry(kname, {"B": B, "nh": nh, "nh_kv": nh_kv, "prompt": T0, "bits": bits, "group": g, "residual": R})
    return {"workload": label, "config": dict(c, head_dim=D, group_size=g, layers_timed=n_layers, steps=steps, warmup=warmup, event_every=event_every),
            "kernel": kname, "launches": n, "sampled": f"every {event_every}rd fused launch of the timed steps (HIP events on the launch stream)",
            "avg_launch_us": round(avg, 2), "median_launch_us": round(med, 2), "min_launch_us": round(us[0], 2),
            "p10_launch_us": round(us[n // 10], 2), "p90_launch_us": round(us[(9 * n) // 10], 2),
            "algorithmic_bytes_per_launch": bytes_, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "achieved": round(bytes_ / (med * 1e-6) / 1e9, 1), "frac": fr(med), "frac_at_avg": fr(avg), "frac_at_min": fr(us[0]),
            "traffic": traffic, "traffic_over_algorithmic": None if traffic is None else round(traffic / bytes_, 4), "traffic_source": src,
            "ms_per_32_layer_step_incl_event_overhead": round(wall * 1e3 / steps * 32 / n_layers, 4),
            "tokens_per_s_at_median_launch_32_layers": round(B / (med * 1e-6 * 32), 1)}


def launch_ranks(n_gpus: int, argv, timeout_s: float = 3600.0) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one process per GPU, the same
    environment contract as torch.distributed.run) and wait for them.  Rank 0 inherits stdout (the JSON line)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL over xGMI needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.time() + timeout_s
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
            if rc or time.time() > deadline:           # one rank died (or a hang): stop the others, by PID
                rc = rc or 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    return rc


def dist_setup(n_gpus: int):
    """Returns (rank, world, local_rank, dist-or-None).  backend nccl (= RCCL) on GPUs, gloo without."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if n_gpus > 1 and world != n_gpus:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: run `python bench.py --gpus {n_gpus}` (it starts its own "
                         f"ranks) or `python -m torch.distributed.run --nproc-per-node {n_gpus} bench.py --gpus {n_gpus}`")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        return rank, world, local, dist
    return 0, 1, local, None


def gather_over_ranks(seconds: float, dist, device):
    """Every rank's own elapsed seconds (list of length world), on every rank."""
    if dist is None:
        return [seconds]
    t = torch.tensor([seconds], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]


def max_over_ranks(seconds: float, dist, device) -> float:
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(dist):
    if dist is not None:
        dist.barrier()


def _pick_cpus(n, allowed):
    """n CPUs of ONE NUMA node, one per physical core (no SMT siblings), from the allowed set -- or the first n allowed ones when
    the topology files are not readable.  Returns (cpus, note)."""
    def parse(txt):
        out = []
        for part in txt.strip().split(","):
            if not part:
                continue
            a, _, b = part.partition("-")
            out.extend(range(int(a), int(b or a) + 1))
        return out
    try:
        best = None
        for node in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
            cpus = [c for c in parse(open(os.path.join(node, "cpulist")).read()) if c in allowed]
            cores, seen = [], set()
            for c in cpus:                                    # one hardware thread per core
                try:
                    sib = tuple(parse(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()))
                except OSError:
                    sib = (c,)
                if sib not in seen:
                    seen.add(sib)
                    cores.append(c)
            if best is None or len(cores) > len(best[1]):
                best = (os.path.basename(node), cores)
        if best and len(best[1]) >= n:
            return best[1][:n], f"{n} physical cores of NUMA {best[0]}"
    except (OSError, ValueError):
        pass
    return sorted(allowed)[:n], "first allowed CPUs (no NUMA topology readable)"


def cpu_baseline(B, nh, T, D, g, bits, layers, budget_s=25.0, threads=None):
    """The reference's pure-PyTorch fake-quant path (oracle/torch_fakequant.py port) on the host cores, on a bounded
    sample of the same workload: ONE batch row of ONE layer (nh heads x T tokens), decode-equivalent work =
    unpack+dequantise K and V (unpack -> fp16 mul -> fp16 add, un-fused as in quant/new_pack.py:51-83) + the two matmuls (the cache
    is already packed in a decode step; pack is timed and reported separately).  Extrapolated to tokens/s for `layers` layers.
    Round 6 (the round-5 line swung 2.3x between two boxes, p10-p90 34-173 ms): a SMALL fixed thread count (8) pinned to physical
    cores of one NUMA node -- 32 threads on a shared 256-CPU host mostly measured the neighbours --, >= 100 repetitions after 5
    warm-up ones, median with p10 / p90 and their spread reported."""
    from oracle import torch_fakequant as TF
    ncpu = os.cpu_count() or 1
    try:
        allowed = set(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = set(range(ncpu))
    nthr = threads or max(1, min(8, len(allowed)))
    cpus, pin_note = _pick_cpus(nthr, allowed)
    prev = torch.get_num_threads()
    torch.set_num_threads(nthr)
    prev_aff = None
    try:
        prev_aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(cpus))
    except (AttributeError, OSError):
        prev_aff = None
    torch.manual_seed(0)
    k = torch.randn((1, nh, T, D)).half()
    v = torch.randn((1, nh, T, D)).half()
    q = torch.randn((1, nh, 1, D)).half()
    a = torch.softmax(torch.randn((1, nh, 1, T)), -1).half()
    dec, pack, t_start = [], [], time.perf_counter()
    for _ in range(5):                              # the prompt's pack (its own figure; also warms the thread pool)
        t_p = time.perf_counter()
        TF.quant_pack_lastdim(k.transpose(2, 3).contiguous(), g, bits)
        TF.quant_pack_lastdim(v, g, bits)
        pack.append(time.perf_counter() - t_p)
    ws = {}                                         # the intermediates are written in place from the second call on: no fresh pages per repetition
    for _ in range(5):                              # warm-up (first touch of the intermediates)
        TF.fakequant_decode_layer(q, a, k, v, g, bits, ws)
    while True:
        _, _, st = TF.fakequant_decode_layer(q, a, k, v, g, bits, ws)
        dec.append(st["dequant_s"] + st["gemv_s"])
        el = time.perf_counter() - t_start
        if (len(dec) >= 100 and el > budget_s) or len(dec) >= 400 or el > 3 * budget_s:
            break
    torch.set_num_threads(prev)
    if prev_aff is not None:
        try:
            os.sched_setaffinity(0, prev_aff)
        except OSError:
            pass
    reps = len(dec)
    ds, ps = sorted(dec), sorted(pack)
    per_layer_row = ds[reps // 2]                   # median

    def pct(xs):
        n_ = len(xs)
        return {"median": round(xs[n_ // 2] * 1e3, 2), "min": round(xs[0] * 1e3, 2), "max": round(xs[-1] * 1e3, 2),
                "p10": round(xs[n_ // 10] * 1e3, 2), "p90": round(xs[(9 * n_) // 10] * 1e3, 2), "reps": n_}
    return {
        "value": 1.0 / (per_layer_row * layers), "unit": "tokens/s", "cores": nthr, "host_cpus": ncpu, "kind": "port",
        "pinned_to_cpus": cpus if prev_aff is not None else None, "pinning": pin_note, "reps": reps,
        "value_at_p10": 1.0 / (ds[reps // 10] * layers), "value_at_p90": 1.0 / (ds[(9 * reps) // 10] * layers),
        "p10_p90_spread_vs_median": [round(ds[reps // 10] / per_layer_row - 1, 3), round(ds[(9 * reps) // 10] / per_layer_row - 1, 3)],
        "ms_per_row_and_layer": pct(ds),
        "pack_ms_per_row_and_layer": pct(ps),
        "sample": f"ONE batch row (of the {B} the GPU step processes) x {nh} heads x T={T} x 1 layer, {reps} reps after 5 warm-up, "
                  f"torch.set_num_threads({nthr}) on {pin_note}: unpack -> fp16 mul -> fp16 add (K and V) + 2 matmuls, median "
                  f"{per_layer_row * 1e3:.0f} ms per row and layer (pack of a full {T}-token prompt {ps[len(ps) // 2] * 1e3:.0f} ms, reported "
                  f"separately, not in value); value = 1 / (median x {layers} layers) = tokens/s of one sequence, extrapolated linearly "
                  f"over the batch (a batch of {B} takes {B}x as long per step and yields {B} tokens: same tokens/s)",
        "port_of": "quant/new_pack.py:51-83 unpack_and_dequant_{k,v}cache + torch.matmul (procedure of quant/test.py:187-195), "
                   "vectorised (oracle/torch_fakequant.py, checked bit for bit against the C oracle)",
    }


def dry_run(args, rank, world, dist):
    """The multi-process skeleton of the benchmark without the GPU work: same barriers, same reductions, same JSON keys."""
    dev = torch.device("cpu")
    barrier(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))            # rank r is (r + 1) x slower: the slowest rank sets the job's time
    barrier(dist)
    own = time.perf_counter() - t0
    elapsed = max_over_ranks(own, dist, dev)
    per_rank = gather_over_ranks(own, dist, dev)
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU work)", "value": round(world * args.batch * args.steps / elapsed, 2),
                          "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
                          "per_rank_ms_per_step": [round(x * 1e3 / args.steps, 4) for x in per_rank],
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "dry_run": True, "config": {"workload": "sleep", "batch_per_gpu": args.batch,
                                                      "parallelism": f"batch-sharded replicas x{world}"}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=32)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--prompt", type=int, default=4080,
                    help="prompt tokens in the cache (4080 = seq 4k with the K flush of R tokens inside the timed region)")
    ap.add_argument("--bits", type=int, default=2)
    ap.add_argument("--group", type=int, default=32)
    ap.add_argument("--residual", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="reference-style composition (one launch per reference op)")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the K-GEMV launches with HIP events")
    ap.add_argument("--event-every", type=int, default=9, help="bracket the dominant kernel of every n-th layer step of the timed region")
    ap.add_argument("--no-hook-kgemv", action="store_true", help="skip the hook-state-layout K-GEMV comparison line")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step's launches from ONE hipGraph (device-resident lengths: kivi_amd/graph.py); matrix-pipe layout only")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: a sleep per step (launcher / reduction / JSON plumbing on CPU, gloo)")
    ap.add_argument("--form", default="auto", help="matrix-pipe layout, A/B: auto (the library's launch plan) | split (two launches) | "
                                                   "row (one launch, a block per row) | slicesN (one launch, N slices per row)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the roofline_config4 / _config4_4bit / _config5_slice objects")
    ap.add_argument("--kgemv-passes", type=int, default=10, help="timed passes over the rotating caches of the single-layer K-GEMV lines")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # plain `python bench.py --gpus N`: start the ranks ourselves
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank, world, local, dist = dist_setup(args.gpus)
    if args.dry_run:
        return dry_run(args, rank, world, dist)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from kivi_amd.quant import matmul

    B, L, nh, nh_kv, D = args.batch, args.layers, args.heads, args.kv_heads, args.head_dim
    T0, g, bits, R = args.prompt, args.group, args.bits, args.residual
    cfg = KiviConfig(bits, bits, g, R)
    total_steps = args.warmup + args.steps
    torch.manual_seed(1234 + rank)

    # ---- build the per-layer caches (prefill with synthetic K/V), inputs resident in HBM
    layers = []
    for _ in range(L):
        # hook-state layout for MHA, the matrix-pipe layout for the grouped-query shapes it covers (nh / nh_kv in {4, 8})
        lc = make_layer_cache(cfg, B, nh_kv, D, T0 + total_steps + 1, dev, num_heads=nh)
        k = torch.randn((B, nh_kv, T0, D), device=dev, dtype=torch.float16)
        v = torch.randn((B, nh_kv, T0, D), device=dev, dtype=torch.float16)
        lc.prefill(k, v)
        del k, v
        if args.form != "auto":
            from kivi_amd import _lib as _l
            assert getattr(lc, "layout", "hook") == "mfma", "--form applies to the matrix-pipe layout"
            lc.flags = (_l.GQA_FORCE_SPLIT if args.form == "split" else _l.GQA_FORCE_ROW if args.form == "row"
                        else _l.gqa_slices(int(args.form[6:])))
        layers.append(lc)
    qs = [torch.randn((B, nh, 1, D), device=dev, dtype=torch.float16) for _ in range(L)]
    ks = [torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16) for _ in range(L)]
    vs = [torch.randn((B, nh_kv, 1, D), device=dev, dtype=torch.float16) for _ in range(L)]
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)

    def step():
        for i in range(L):
            kivi_attention_decode(qs[i], ks[i], vs[i], layers[i], fused_kernels=not args.unfused)

    graphed = None
    if args.graph:
        # the whole step = ONE graph launch (+ a one-thread kernel that uploads the six lengths, + the K flushes every R steps)
        from kivi_amd.graph import GraphedDecode, MfStepDriver
        assert all(getattr(lc, "layout", "hook") == "mfma" for lc in layers), "--graph needs the matrix-pipe cache layout"
        outs = [torch.empty((B, nh, 1, D), device=dev, dtype=torch.float16) for _ in range(L)]
        drv = MfStepDriver(layers)

        def graph_body():
            for i in range(L):
                drv.enqueue(i, qs[i], ks[i], vs[i], outs[i])

        graphed = GraphedDecode(drv, graph_body)
        step = graphed.step          # noqa: F811
        args.no_kernel_events = True   # (event pairs cannot be captured)

    for _ in range(args.warmup):
        step()

    # ---- HIP events on every qK^T dispatch of the timed region: the library launches the kernel with
    # hipExtLaunchKernelGGL(start, stop) on torch's current stream, so the pair brackets exactly that dispatch
    # (what rocprofv3 reports as the kernel duration).
    from kivi_amd import _lib
    klib = _lib.load()
    kev = []

    launch_no = [0]

    def hook(phase, kind, info):
        if kind != "k" or phase != "pre":
            return
        launch_no[0] += 1
        if launch_no[0] % args.event_every:      # sample: the start/stop events cost ~5 us of stream time each
            return
        e0, e1 = klib.kivi_event_create(), klib.kivi_event_create()
        klib.kivi_set_launch_events(e0, e1)
        kb = kgemv_bytes(info["B"], info["nh"], info["nh_kv"], info["K"], info["N"], info["group_size"], info["bits"])
        rb = row_bytes(info) if "Tv" in info else None     # what the fused decode-row launch moves
        kev.append((e0, e1, kb, rb))

    if not args.no_kernel_events:
        matmul.launch_hook = hook

    torch.cuda.synchronize()
    barrier(dist)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue_s = time.perf_counter() - t0      # how long the host needed to enqueue the timed steps
    torch.cuda.synchronize()
    barrier(dist)
    own = time.perf_counter() - t0
    peak_timed = torch.cuda.max_memory_allocated(dev)     # caches + per-step scratch: before this script's own measurement harness allocates
    elapsed = max_over_ranks(own, dist, dev)
    per_rank = gather_over_ranks(own, dist, dev)
    matmul.launch_hook = None
    # K flushes (llama_kivi.py:343-356: R residual tokens quantised per channel, in place) that fell into the timed region
    R_ = cfg.residual_length
    k_res0 = T0 % R_
    flushes_timed = (k_res0 + total_steps) // R_ - (k_res0 + args.warmup) // R_

    ms_per_step = elapsed * 1e3 / args.steps
    tokens_per_s = world * B * args.steps / elapsed

    if rank == 0:
        roof = None
        if kev:
            us = [klib.kivi_event_elapsed_us(a, b) for a, b, _, _ in kev]
            timed = (klib.kivi_last_timed_kernel() or b"").decode()
            row_fused = any(kn in timed for kn in ("decode_row_kernel", "mf_row_kernel", "mf_row4_kernel")) and all(r is not None for _, _, _, r in kev)
            kname = timed.split("<")[0].strip("( ")
            tot_bytes = sum((r if row_fused else n) for _, _, n, r in kev)
            avg_us = sum(us) / len(us)
            achieved = tot_bytes / (sum(us) * 1e-6) / 1e9
            # HBM bytes per launch from the newest tracked rocprofv3 --pmc passes of the same command (profiles/)
            traffic, traffic_src = pmc_traffic_entry(kname, {"B": B, "nh": nh, "nh_kv": nh_kv, "prompt": T0, "bits": bits, "group": g, "residual": R})
            if traffic_src:
                traffic_src += (" (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this command, x2 gfx950 FETCH_SIZE correction; "
                                "a tracked measurement, not collected in this run)")
            mf = getattr(layers[0], "layout", "hook") == "mfma"
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": kname, "kernel_role": (
                        "one launch per layer: packed qK^T of the row on the matrix pipe -> LDS scores -> residual scores + softmax + "
                        "window + packed sV + cache update" if "mf_row_kernel" in timed else
                        "one launch per layer: the four query heads of a kv head in one block -- packed qK^T on the matrix pipe -> LDS "
                        "scores -> residual scores + softmax + window + packed sV + cache update" if "mf_row4_kernel" in timed else
                        "one launch per layer (VALU unpack): packed qK^T of the row -> LDS scores -> residual scores + softmax + window + "
                        "packed sV + cache update" if row_fused else
                        "packed qK^T on the matrix pipe + residual scores + softmax statistics; first of the two launches of a layer step"
                        if mf else "fused int2 qK^T over packed K"),
                    "launches": len(us),
                    "sampled": f"every {args.event_every}th layer step of the timed region (an event pair costs ~10 us of stream time)",
                    "avg_launch_us": round(avg_us, 2), "median_launch_us": round(sorted(us)[len(us) // 2], 2),
                    "min_launch_us": round(min(us), 2),
                    "algorithmic_bytes_per_launch": tot_bytes // len(us),
                    "frac_of_measured_copy_ceiling": round(achieved / HBM_MEASURED_COPY_GBS, 4)}
        # BASELINE configs[1] beside it: the same qK^T kernel launched back to back over the L layer caches (each launch
        # reads a different ~200 MiB cache, L x 200 MiB >> the 256 MiB Infinity Cache), every dispatch timed -- the
        # isolated single-layer K-GEMV number, without the decode loop's kernel alternation
        single = None
        single_mf = None

        def time_kgemv(launch, ncaches, nbytes, label):
            """`launch(layer_index)` enqueues one qK^T launch with a pending event pair; 1 warm-up + args.kgemv_passes timed passes
            over the caches (>= 100 timed launches: SURVEY section 8d)."""
            ev1 = []
            for rep in range(1 + max(1, args.kgemv_passes)):
                for i in range(ncaches):
                    pair = (klib.kivi_event_create(), klib.kivi_event_create())
                    klib.kivi_set_launch_events(*pair)
                    launch(i)
                    if rep:
                        ev1.append(pair)
            torch.cuda.synchronize()
            us1 = sorted(klib.kivi_event_elapsed_us(a, b) for a, b in ev1)
            med = us1[len(us1) // 2]
            avg = sum(us1) / len(us1)
            return {"workload": label, "kernel": (klib.kivi_last_timed_kernel() or b"").decode().split("<")[0].strip("( "),
                    "launches": len(us1), "median_launch_us": round(med, 2), "avg_launch_us": round(avg, 2), "min_launch_us": round(us1[0], 2),
                    "p10_launch_us": round(us1[len(us1) // 10], 2), "p90_launch_us": round(us1[(9 * len(us1)) // 10], 2),
                    "algorithmic_bytes_per_launch": nbytes,
                    "achieved": round(nbytes / (med * 1e-6) / 1e9, 1), "unit": "GB/s",
                    "frac": round(nbytes / (med * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_at_avg": round(nbytes / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_at_min": round(nbytes / (us1[0] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}

        if roof is not None and layers[0].k_quant_len:
            Tq = layers[0].k_quant_len
            scratch = torch.empty((B, nh, 1, Tq + 8), device=dev, dtype=torch.float16)
            nbytes = kgemv_bytes(B, nh, nh_kv, D, Tq, g, bits)
            label = ("BASELINE configs[1]: single-layer packed-K qK^T GEMV, back-to-back over the layer caches "
                     f"({L} x {nbytes / 2**20:.0f} MiB >> the 256 MiB Infinity Cache)")
            if hasattr(layers[0], "k_code"):
                single = time_kgemv(lambda i: matmul.gemv_k_paged(g, qs[0], layers[i].k_code, layers[i].k_scale, layers[i].k_mn,
                                                                  Tq, bits, out=scratch[..., :Tq]), L, nbytes, label + ", hook-state layout")
            else:
                from kivi_amd.quant import mfma
                single_mf = time_kgemv(lambda i: mfma.gqa_scores(qs[0], layers[i].kt, Tq, scratch, g, bits), L, nbytes,
                                       label + ", matrix-pipe layout (kivi_gqa_scores: raw fp16 scores to memory)")
                single = single_mf
                if not args.no_hook_kgemv and nh == nh_kv:
                    # the reference's operator for this GEMV, quant.matmul.cuda_bmm_fA_qB_outer on the hook-state layout
                    # (K_code_T (B,nh,D,T/16): kivi_gemv_k, the VALU kernel): 12 caches of ~200 MiB, rotating.  This is the
                    # drop-in the d2 row of SURVEY.md section 8 names, so it is the primary line; the matrix-pipe kernel on
                    # its own layout is reported beside it.
                    from kivi_amd.quant import new_pack
                    hk = []
                    for _ in range(12):
                        kk = torch.randn((B, nh_kv, Tq, D), device=dev, dtype=torch.float16)
                        hk.append(new_pack.quantize_and_pack_k_tmajor(kk, g, bits))
                        del kk
                    single = time_kgemv(lambda i: matmul.cuda_bmm_fA_qB_outer(g, qs[0], hk[i][0], hk[i][1], hk[i][2], bits), len(hk),
                                        nbytes, "BASELINE configs[1]: quant.matmul.cuda_bmm_fA_qB_outer (kivi_gemv_k) on the reference's "
                                        "hook-state layout, output allocated per call, 12 rotating caches of "
                                        f"{nbytes / 2**20:.0f} MiB")
                    del hk
        # cost of one K flush per layer (the launch of kivi_quant_pack_k_tmajor over the R residual tokens), timed apart
        flush_us = None
        try:
            from kivi_amd.quant import new_pack
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(L)]
            if hasattr(layers[0], "k_code"):
                scratch_page = [torch.empty_like(x[:, :, 0]) for x in (layers[0].k_code, layers[0].k_scale, layers[0].k_mn)]
                flush = lambda lc: new_pack.quantize_and_pack_k_tmajor(lc.k_res, g, bits, out=tuple(scratch_page), token_offset=0)
            else:
                from kivi_amd.quant import mfma
                spare = mfma.alloc_store(B, nh_kv, 1, dev, bits)      # kivi_kt_pack of the R residual tokens into a spare super-block
                flush = lambda lc: mfma.kt_pack(lc.k_res[:, :, :R], spare, 0, g, bits)
            for rep in range(2):
                for i, lc in enumerate(layers):
                    ev[i][0].record()
                    flush(lc)
                    ev[i][1].record()
            torch.cuda.synchronize()
            fl = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
            flush_us = round(fl[len(fl) // 2], 2)
        except Exception as e:    # instrumentation only
            flush_us = f"n/a ({type(e).__name__}: {e})"[:120]
        kv_bytes = sum(lc.nbytes() for lc in layers)
        fp16_bytes = 2 * L * B * nh_kv * layers[0].kv_seq_len * D * 2
        out = {
            "metric": ("decode-step tokens/sec (KIVI attention hot path, "
                       + ("Llama-2-7B shape" if (nh, nh_kv, D, L) == (32, 32, 128, 32) else f"{nh}/{nh_kv} heads x {D}, {L} layers")
                       + f", B={B}/GPU, seq={'4k' if T0 in (4080, 4096) else T0}, {bits}b/{bits}b g={g})"),
            "value": round(tokens_per_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "per_rank_ms_per_step": [round(x * 1e3 / args.steps, 4) for x in per_rank],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 accumulate over int2 codes, fp16 in/out", "data": "synthetic",
            "config": {"workload": "kivi_decode_attention_hotpath: per layer fused qK^T + residual + softmax + fused sV + "
                                   "residual + in-place KV append/quantise; 32 layers, no dense projections",
                       "launches_per_layer": "composed (~20)" if args.unfused else "fused (g=32 D=128, 2-bit or 4-bit with nh / nh_kv = 4, on the matrix-pipe layout: rows that fit the LDS 1 launch, else 2; other shapes on the hook-state layout: 1 decode-row launch or qK^T + [row softmax] + sV; +1 K flush every R steps)",
                       "layers": L, "batch_per_gpu": B, "heads": nh, "kv_heads": nh_kv, "head_dim": D, "prompt_len": T0,
                       "kv_len_end": layers[0].kv_seq_len, "k_bits": bits, "v_bits": bits, "group_size": g,
                       "residual_length": R, "parallelism": f"batch-sharded replicas x{world} (no data-path collective)",
                       "k_flushes_in_timed_region": flushes_timed, "k_flush_launch_us_per_layer": flush_us},
            "peak_kv_bytes": kv_bytes, "peak_kv_bytes_fp16_equivalent": fp16_bytes,
            "kv_compression": round(fp16_bytes / kv_bytes, 3),
            "allocator_peak_bytes": peak_timed,
            "allocator_peak_bytes_incl_bench_harness": torch.cuda.max_memory_allocated(dev),   # + the rotating K-GEMV caches etc. of the lines below
            "host_enqueue_ms_per_step": round(host_enqueue_s * 1e3 / args.steps, 4),
            "hipgraph": None if graphed is None else {"eager_steps": graphed.eager, "captures": graphed.captures, "replays": graphed.replays,
                                                      "note": "whole step replayed from one hipGraph; lengths device-resident (kivi_mf_decode_layer_dyn)"},
            "roofline": roof,
            "roofline_single_layer_kgemv": single,
            "roofline_single_layer_kgemv_mf_layout": single_mf,
        }
        default_workload = (B, nh, nh_kv, D, T0, bits, g, R, L) == (32, 32, 32, 128, 4080, 2, 32, 32, 32) and args.form == "auto"
        if world == 1 and default_workload and not args.no_extra_configs and not args.unfused:
            # the other BASELINE.json configurations (driver-visible: these keys ride on the one JSON line); the headline's caches are
            # released first
            del layers[:]
            qs.clear(); ks.clear(); vs.clear()
            torch.cuda.empty_cache()
            for key, label, c in EXTRA_CONFIGS:
                try:
                    out[key] = extra_config_roofline(label, c, dev, klib)
                except Exception as e:          # instrumentation must never cost the headline line
                    out[key] = {"workload": label, "error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(B, nh, (T0 // R) * R, D, g, bits, L)   # the packed K prefix holds floor(T0 / R) * R tokens
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
