#!/usr/bin/env python3
"""HBM traffic per launch of the library's kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes,
kernel trace only) of `python bench.py ...`, as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:
    bytes = FETCH_SIZE[KB] x 1024 x 2  (128-byte fabric reads are tallied at 64 B; calibrated on tools/hbm_read_bw.bin)
          + WRITE_SIZE[KB] x 1024
    python tools/pmc_traffic.py <fetch counter csv> <write counter csv> [calibration counter csv] --config '{...}' --out X.json
Writes {"kernels": {name: {hbm_read_bytes, hbm_write_bytes, hbm_bytes_per_launch, launches, config}}} -- what bench.py reads as
roofline.traffic (profiles/rNN_pmc_traffic.json)."""
import argparse
import collections
import csv
import json
import re


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"[<(].*", "", name)
        agg[name].append(float(r["Counter_Value"]))
    return agg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch")
    ap.add_argument("write")
    ap.add_argument("calib", nargs="?")
    ap.add_argument("--config", default="{}")
    ap.add_argument("--skip", type=int, default=0, help="launches of every kernel to drop from the front (warm-up)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    f, w = per_kernel(a.fetch, "FETCH_SIZE"), per_kernel(a.write, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of the bench command",
           "correction": "FETCH_SIZE[KB] x 1024 x 2 (gfx950: 128-B fabric reads tallied at 64 B), WRITE_SIZE[KB] x 1024", "kernels": {}}
    if a.calib:
        c = per_kernel(a.calib, "FETCH_SIZE")
        cal = {}
        for k, v in c.items():
            if "read_kernel" in k:
                med = sorted(v)[len(v) // 2]
                cal[k] = {"launches": len(v), "median_FETCH_SIZE_KB": med, "bytes_per_FETCH_SIZE_unit_for_a_2GiB_stream": (2 << 30) / med if med else None}
        out["calibration"] = cal
    cfg = json.loads(a.config)
    for k in sorted(set(f) | set(w)):
        if not any(s in k for s in ("mf_", "decode_row", "gemv_", "gqa_", "kt_pack", "vt_pack", "quant_pack", "row_softmax")):
            continue
        fv, wv = f.get(k, [])[a.skip:], w.get(k, [])[a.skip:]
        if not fv or not wv:
            continue
        med = lambda x: sorted(x)[len(x) // 2]   # noqa: E731
        rb, wb = med(fv) * 2048.0, med(wv) * 1024.0
        out["kernels"][k] = {"launches": [len(fv), len(wv)], "median_FETCH_SIZE_KB": med(fv), "median_WRITE_SIZE_KB": med(wv),
                             "hbm_read_bytes": rb, "hbm_write_bytes": wb, "hbm_bytes_per_launch": rb + wb, "config": cfg}
    json.dump(out, open(a.out, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:28s} read {v['hbm_read_bytes'] / 1e6:9.2f} MB  write {v['hbm_write_bytes'] / 1e6:8.2f} MB  per launch {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()
