#!/usr/bin/env python3
"""Merge the per-configuration outputs of tools/pmc_traffic.py (gpurun_out/<tag>/pmc_traffic_<label>.json) into ONE tracked file
profiles/rNN_pmc_traffic.json: kernels of the headline configuration keep their names, the others become "<kernel>@<label>" (bench.py
looks an entry up by kernel name + configuration).
    python tools/merge_pmc.py profiles/r06_pmc_traffic.json gpurun_out/r6x/pmc_traffic_bench.json gpurun_out/r6x/pmc_traffic_config4.json ..."""
import json
import os
import sys

out_path, ins = sys.argv[1], sys.argv[2:]
merged = None
for i, p in enumerate(ins):
    j = json.load(open(p))
    label = os.path.basename(p)[len("pmc_traffic_"):-len(".json")]
    if merged is None:
        merged = {k: v for k, v in j.items() if k != "kernels"}
        merged["kernels"] = {}
        merged["session"] = [p]
    else:
        merged["session"].append(p)
    for k, v in j["kernels"].items():
        merged["kernels"][k if label == "bench" else f"{k}@{label}"] = v
json.dump(merged, open(out_path, "w"), indent=1)
print("wrote", out_path, len(merged["kernels"]), "entries")
