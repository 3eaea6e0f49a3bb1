"""Per inner loop of a kernel's assembly (hipcc -S --cuda-device-only): instruction counts by kind.
python tools/asm_loops.py file.s [min_instructions]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().splitlines()
min_n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
labels = {}
for k, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = k
# a loop = backward branch to a label
for k, l in enumerate(lines):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if not m or m.group(1) not in labels or labels[m.group(1)] > k:
        continue
    a = labels[m.group(1)]
    body = [x.split()[0] for x in lines[a:k + 1] if re.match(r"^\s+[vsdg][a-z_0-9]+", x) or re.match(r"^\s+(scratch|buffer|global|flat|ds)_", x)]
    if len(body) < min_n:
        continue
    c = Counter(body)
    valu = sum(v for n, v in c.items() if n.startswith("v_"))
    print("loop %s..line %d: %d instructions, %d VALU, %d SALU, %d LDS, %d vmem, %d scratch" % (
        m.group(1), k, len(body), valu, sum(v for n, v in c.items() if n.startswith("s_")),
        sum(v for n, v in c.items() if n.startswith("ds_")),
        sum(v for n, v in c.items() if n.startswith(("global_", "buffer_", "flat_"))),
        sum(v for n, v in c.items() if n.startswith("scratch_"))))
    print("   ", ", ".join("%s %d" % (n, v) for n, v in c.most_common(14)))
