#!/usr/bin/env python3
"""Static instruction mix of every loop of one kernel in a hipcc -S listing:  tools/isa_loop_counts.py file.s kernel-substring"""
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if key in l and l.startswith("_Z") and ": ;" in l)
end = next(i for i in range(start, len(txt)) if "s_endpgm" in txt[i])
lines = txt[start:end]
is_v = re.compile(r"^\s+v_")
is_ds = re.compile(r"^\s+ds_")
is_s = re.compile(r"^\s+s_")
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        body = lines[a:i]
        valu = [x.split()[0] for x in body if is_v.match(x)]
        mf = sum(1 for x in valu if x.startswith("v_mfma"))
        vmem = sum(1 for x in body if "buffer_" in x or "global_" in x or "scratch_" in x)
        lds = sum(1 for x in body if is_ds.match(x))
        salu = sum(1 for x in body if is_s.match(x))
        print("loop %s lines %d-%d: VALU %d MFMA %d vmem %d lds %d salu %d" % (m.group(1), a, i, len(valu) - mf, mf, vmem, lds, salu))
print("total static VALU (incl. MFMA)", sum(1 for x in lines if is_v.match(x)), "lines", len(lines))
