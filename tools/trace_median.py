#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace CSV, warm-up calls excluded.

    python tools/trace_median.py <x_kernel_trace.csv> [--skip N] [--match substr ...] [--json out.json]

rocprofv3's own --stats file reports the MEAN over all calls (warm-up, first-touch and the step that contains a K flush
included); the bench line's `roofline` is built from per-dispatch durations of the timed region.  This condenses the
raw trace the same way: for every kernel whose name contains one of the --match strings (default: the library's
kernels) drop the first N calls and report count / median / mean / min / p10 / p90 in microseconds, plus the
dispatch's register / LDS / grid footprint.  profiles/rNN_*_trace_summary.json are produced by this tool.
"""
import argparse
import csv
import json
import re
import sys

DEFAULT = ["decode_row_kernel", "gemv_k_kernel", "gemv_v_kernel", "gqa_", "row_softmax_kernel", "quant_pack", "softmax_scaled"]


def short(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\((GemvKArgs|GemvVArgs|RowSoftmaxArgs).*", "", name)[:120]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--skip", type=int, default=0, help="calls of every matched kernel to drop from the front (warm-up)")
    ap.add_argument("--skip-for", nargs="*", default=[], metavar="SUBSTR=N",
                    help="a different warm-up count for kernels whose name contains SUBSTR (e.g. gemv_k_kernel=12: the single-layer K-GEMV loop "
                         "of bench.py makes one warm-up pass over its 12 caches)")
    ap.add_argument("--match", nargs="*", default=DEFAULT)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    per = {}
    for r in csv.DictReader(open(args.trace)):
        n = r["Kernel_Name"]
        if not any(m in n for m in args.match):
            continue
        d = per.setdefault(short(n), {"us": [], "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"],
                                      "lds": r["LDS_Block_Size"], "scratch": r["Scratch_Size"], "grid": r["Grid_Size_X"],
                                      "wg": r["Workgroup_Size_X"]})
        d["us"].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    out = {}
    skip_for = [(x.split("=")[0], int(x.split("=")[1])) for x in args.skip_for]
    for k, d in per.items():
        skip = next((n for sub, n in skip_for if sub in k), args.skip)
        us = [u for _, u in sorted(d["us"])][skip:]
        if not us:
            continue
        s = sorted(us)
        q = lambda f: round(s[min(len(s) - 1, int(f * len(s)))], 2)   # noqa: E731
        out[k] = {"calls": len(us), "skipped_warmup_calls": min(skip, len(d["us"])), "median_us": q(0.5),
                  "mean_us": round(sum(us) / len(us), 2), "min_us": round(s[0], 2), "p10_us": q(0.1), "p90_us": q(0.9),
                  "vgpr": int(d["vgpr"]), "agpr": int(d["agpr"]), "sgpr": int(d["sgpr"]), "lds_bytes": int(d["lds"]),
                  "scratch_bytes": int(d["scratch"]), "grid_threads": int(d["grid"]), "workgroup": int(d["wg"])}
    txt = json.dumps(out, indent=1)
    if args.json:
        open(args.json, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    sys.exit(main())
