#!/usr/bin/env python
"""Per basic block of one kernel in a hipcc -S listing: VALU / SALU / memory instruction counts (the big blocks
are the DP step loops).  python tools/asm_blocks.py eng.s <substring of the mangled kernel name> [min_valu]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.endswith(":") is False and re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), []]
        continue
    t = l.strip()
    if not t or t.startswith((";", ".")):
        continue
    cur[1].append(t.split()[0])
blocks.append(cur)
for name, ins in blocks:
    valu = [i for i in ins if i.startswith("v_")]
    if len(valu) < min_valu:
        continue
    hist = {}
    for i in valu:
        hist[i] = hist.get(i, 0) + 1
    top = sorted(hist.items(), key=lambda kv: -kv[1])[:14]
    print("%s: %d instr, %d VALU, %d SALU, %d ds, %d vmem, %d s_nop" % (
        name, len(ins), len(valu), sum(i.startswith("s_") and i != "s_nop" for i in ins), sum(i.startswith("ds_") for i in ins),
        sum(i.startswith(("global_", "buffer_", "scratch_")) for i in ins), ins.count("s_nop")))
    print("   " + ", ".join("%s x%d" % kv for kv in top))
