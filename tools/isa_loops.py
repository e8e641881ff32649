#!/usr/bin/env python3
"""Instruction mix of the innermost hot loops of one kernel in a hipcc -S listing.
usage: isa_loops.py file.s <substring of the mangled kernel name> [min_instr]"""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
min_instr = int(sys.argv[3]) if len(sys.argv) > 3 else 60
text = open(src).read().split("\n")
start = next(i for i, l in enumerate(text) if key in l and re.match(r"^_Z\w+:", l) and not l.startswith("."))
end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
lines = text[start:end]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
print(lines[0], "total lines", len(lines))
for a, b in loops:
    ops = [x.split()[0] for x in lines[a:b + 1] if x.strip() and not x.strip().startswith((";", "."))]
    if len(ops) < min_instr:
        continue
    c = Counter(ops)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    salu = sum(v for k, v in c.items() if k.startswith("s_"))
    mem = sum(v for k, v in c.items() if k.startswith(("buffer_", "global_", "flat_", "ds_")))
    print(f"loop lines {a}-{b}: {len(ops)} instr  valu {valu} salu {salu} mem {mem}  branches {sum(v for k, v in c.items() if 'branch' in k)}")
    print("   ", dict(c.most_common(14)))
