#!/usr/bin/env python
"""Accuracy of the overlapper against simulator truth.

Own rewrite of what the reference's measure_sensitivity_PBSIM.py computes in its de-novo mode
(measure_sensitivity_PBSIM.py:84-106 theoretical overlaps, :129-175 filters, :183-214 matching, :265-270):

  * reads carry their genome interval in the header  >NAME<i>_<start>_<len>
  * theoretical overlap = ordered pair of different reads whose intervals share >= 1000 bases
  * heuristic overlap   = every output line and its mirror image (query/ref swapped), self pairs dropped,
                          kept if both aligned extents are >= 990 and score >= 600
  * FN = theoretical pairs no kept line reports, TP / FP = kept lines whose pair is / is not theoretical
  * sensitivity = TP / (TP + FN), specificity = TP / (TP + FP)

usage: measure_sensitivity.py reads.fasta darwin.0.out [darwin.1.out ...]
"""
import re
import sys

import numpy as np

SCORE_THRES, MIN_LENGTH, MIN_OVERLAP = 600, 990, 1000
LINE = re.compile(r"ref_id: (\S+), query_id: (\S+), ab: (-?\d+), ae: (-?\d+), bb: (-?\d+), be: (-?\d+), "
                  r"score: (-?\d+), comp: (\d)")


def read_headers(fasta):
    names, start, length = [], [], []
    for line in open(fasta):
        if line.startswith(">"):
            name = re.split(r"[^A-Za-z0-9_]", line[1:].strip())[0]      # fasta.cpp:19-33
            f = name.split("_")
            names.append(name); start.append(int(f[-2])); length.append(int(f[-1]))
    return names, np.array(start), np.array(length)


def theoretical_pairs(start, length):
    order = np.argsort(start, kind="stable")
    s, e = start[order], (start + length)[order]
    pairs = set()
    for a in range(len(order)):
        hi = np.searchsorted(s, e[a])            # reads starting before read a ends
        for b in range(a + 1, hi):
            if min(e[a], e[b]) - max(s[a], s[b]) >= MIN_OVERLAP:
                pairs.add((int(order[a]), int(order[b])))
                pairs.add((int(order[b]), int(order[a])))
    return pairs


def measure(fasta, outs):
    names, start, length = read_headers(fasta)
    index = {n: k for k, n in enumerate(names)}
    truth = theoretical_pairs(start, length)
    kept = []
    n_lines = 0
    for path in outs:
        for line in open(path):
            m = LINE.match(line)
            if not m:
                continue
            n_lines += 1
            r, q = index[m.group(1)], index[m.group(2)]
            ab, ae, bb, be, score = (int(m.group(k)) for k in range(3, 8))
            if r == q or ae - ab < MIN_LENGTH or be - bb < MIN_LENGTH or score < SCORE_THRES:
                continue
            kept += [(r, q), (q, r)]
    reported = set(kept)
    tp = sum(1 for p in kept if p in truth)
    fp = len(kept) - tp
    fn = sum(1 for p in truth if p not in reported)
    return {"reads": len(names), "lines": n_lines, "theoretical": len(truth), "kept": len(kept),
            "TP": tp, "FN": fn, "FP": fp,
            "sensitivity": tp / max(tp + fn, 1), "specificity": tp / max(tp + fp, 1)}


if __name__ == "__main__":
    res = measure(sys.argv[1], sys.argv[2:])
    for k, v in res.items():
        print("%s: %s" % (k, ("%.6f" % v) if isinstance(v, float) else v))
