#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/...) into the small, tracked files under profiles/.

    python tools/summarize_profiles.py r01

Inputs (produced on the GPU box, see profiles/README.md for the exact commands):
  gpurun_out/prof/bench_trace/bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats -- python bench.py ...
  gpurun_out/prof/pmc_fetch/k_counter_collection.csv     rocprofv3 --pmc FETCH_SIZE  -- python tools/gpu_sweep.py --only <default K>
  gpurun_out/prof/pmc_write/k_counter_collection.csv     rocprofv3 --pmc WRITE_SIZE  -- (same)
  gpurun_out/prof/pmc_calib/c_counter_collection.csv     rocprofv3 --pmc FETCH_SIZE  -- tools/hbm_read_bw.bin 2   (known byte count)
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def short(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"<.*", lambda m: m.group(0) if ("gemv_" in name or "quant_" in name or "unpack_" in name or "decode_row" in name or "row_softmax" in name) else "<...>", name)
    return name[:140]


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    os.makedirs(DST, exist_ok=True)
    # ---- kernel stats of the bench command
    stats = os.path.join(SRC, "bench_trace", "bench_kernel_stats.csv")
    rows = list(csv.DictReader(open(stats))) if os.path.exists(stats) else []    # absent in a PMC-only session
    if rows:
        with open(os.path.join(DST, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                            r["MaxNs"], r["StdDev"]])
    kg = next((r for r in rows if "gemv_k_kernel" in r["Name"]), None)
    vg = next((r for r in rows if "gemv_v_kernel" in r["Name"]), None)
    rg = next((r for r in rows if "decode_row_kernel" in r["Name"]), None)   # the fused MHA decode step (one launch)

    # ---- PMC
    def pmc(path, counter, match):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if r["Counter_Name"] == counter and match in r["Kernel_Name"]]
        return sum(vals) / len(vals), len(vals)

    fetch_k, n1 = pmc(os.path.join(SRC, "pmc_fetch", "k_counter_collection.csv"), "FETCH_SIZE", "gemv_k_kernel")
    write_k, n2 = pmc(os.path.join(SRC, "pmc_write", "k_counter_collection.csv"), "WRITE_SIZE", "gemv_k_kernel")
    fetch_v, n3 = pmc(os.path.join(SRC, "pmc_fetch", "k_counter_collection.csv"), "FETCH_SIZE", "gemv_v_kernel")
    write_v, n4 = pmc(os.path.join(SRC, "pmc_write", "k_counter_collection.csv"), "WRITE_SIZE", "gemv_v_kernel")
    # calibration: the read kernel streams exactly 2 GiB per launch
    cal = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(SRC, "pmc_calib", "c_counter_collection.csv"))):
        if "read_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            width = "dwordx4" if "__vector(4)" in r["Kernel_Name"] else "dwordx2"
            cal[width].append(float(r["Counter_Value"]))
    known = 2 * 2 ** 30
    calib = {w: {"launches": len(v), "median_FETCH_SIZE_KB": sorted(v)[len(v) // 2],
                 "bytes_per_FETCH_SIZE_unit": known / sorted(v)[len(v) // 2]} for w, v in cal.items()}
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE is in KB and counts 128-B requests as 64 B on gfx950 -> x2.
    unit = 1024 * 2
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/gpu_sweep.py, config C2 "
                  "(B=32,H=32,T=4096,D=128,g=32,2-bit); calibration on tools/hbm_read_bw.bin (2 GiB streaming reads)",
        "correction": "FETCH_SIZE[KB] x 1024 x 2 (gfx950 tallies 128-B fabric reads at 64 B: calibration below gives "
                      "~2048 bytes per unit for both 16-B and 8-B per-lane loads); WRITE_SIZE[KB] x 1024",
        "calibration": calib,
        "gemv_k": {"kernel": "gemv_k_kernel<2, 32, 2, 4, 1, 4, 2, true>(GemvKArgs)", "FETCH_SIZE_KB": fetch_k, "WRITE_SIZE_KB": write_k, "launches": [n1, n2],
                   "hbm_read_bytes": fetch_k * unit, "hbm_write_bytes": write_k * 1024,
                   "hbm_bytes_per_launch": fetch_k * unit + write_k * 1024,
                   "algorithmic_bytes_per_launch": 209977344},
        "gemv_v": {"kernel": "gemv_v_kernel<2, 32, 8, 4, 1, 1, 2, true, false>(GemvVArgs)", "FETCH_SIZE_KB": fetch_v, "WRITE_SIZE_KB": write_v, "launches": [n3, n4],
                   "hbm_bytes_per_launch": fetch_v * unit + write_v * 1024, "algorithmic_bytes_per_launch": 209977344},
        "hbm_bytes_per_launch": fetch_k * unit + write_k * 1024,
        "bench_kernel_trace": {(k + "_avg_us"): float(r["AverageNs"]) / 1e3 for k, r in
                               (("gemv_k", kg), ("gemv_v", vg), ("decode_row", rg)) if r is not None},
    }
    # the fused decode-row launch, measured on the bench command itself (separate --pmc passes)
    rf, rw = os.path.join(SRC, "pmc_fetch_row", "b_counter_collection.csv"), os.path.join(SRC, "pmc_write_row", "b_counter_collection.csv")
    if os.path.exists(rf) and os.path.exists(rw):
        fr, m1 = pmc(rf, "FETCH_SIZE", "decode_row_kernel")
        wr, m2 = pmc(rw, "WRITE_SIZE", "decode_row_kernel")
        out["decode_row"] = {"kernel": short(rg["Name"]) if rg is not None else "decode_row_kernel", "FETCH_SIZE_KB": fr, "WRITE_SIZE_KB": wr,
                             "launches": [m1, m2], "hbm_read_bytes": fr * unit, "hbm_write_bytes": wr * 1024,
                             "hbm_bytes_per_launch": fr * unit + wr * 1024}
        out["decode_row_hbm_bytes_per_launch"] = fr * unit + wr * 1024
    with open(os.path.join(DST, f"{tag}_kgemv_pmc.json"), "w") as f:   # bench.py reads the newest rNN_kgemv_pmc.json for roofline.traffic
        json.dump(out, f, indent=1)
    print(json.dumps(out["gemv_k"], indent=1))
    print(json.dumps(out["bench_kernel_trace"], indent=1))
    print(json.dumps(calib, indent=1))


if __name__ == "__main__":
    main()
