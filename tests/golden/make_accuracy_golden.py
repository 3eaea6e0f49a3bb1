#!/usr/bin/env python3
"""Writes tests/golden/accuracy.json: TP / FN / FP, sensitivity and specificity as printed by THE REFERENCE'S OWN
accuracy script, /root/reference/measure_sensitivity_PBSIM.py, for committed inputs.  Build container only.

The script is Python 2 (print statements, list-returning map / zip); python2 is absent, so it is converted with
lib2to3 into a temporary directory, run there with python3 on `reference.fasta` + `out.darwin` (the file names it
hard-wires, measure_sensitivity_PBSIM.py:33-38,132), and deleted.  The converted text is never written under the
repository and never travels; only the numbers it printed do.

Inputs (all committed in the fixture itself, so the test needs nothing else):
  * case "e2e": the 16-read FASTA of tests/golden/dsoft.json and the 67 lines the reference's CPU program printed
    for it (tests/golden/e2e.json);
  * case "perturbed": the same lines plus edited copies that exercise every filter of the script -- scores under
    600, extents under 990, self pairs, pairs without a theoretical overlap, duplicated lines.
The script identifies a read by the FIRST INTEGER of its header and compares that with the read's position in the
FASTA (parse() at :11-12, `tovl[0] == hovl[0]` at :200), i.e. it expects PBSIM's numbering S<k> = k-th record
counted from 0; the golden FASTA counts from 1, so both cases are renumbered to S0.. before they are handed over
(the harness under test, tools/measure_sensitivity.py, keys reads by name and does not care).

usage: python3 tests/golden/make_accuracy_golden.py
"""
import json
import os
import random
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SCRIPT = "/root/reference/measure_sensitivity_PBSIM.py"
LINE = re.compile(r"ref_id: (\S+), query_id: (\S+), ab: (-?\d+), ae: (-?\d+), bb: (-?\d+), be: (-?\d+), score: (-?\d+), comp: (\d)")


def renumber(headers, lines):
    """S<i>_<start>_<len> with i counted from 1 -> counted from 0, in headers and lines alike"""
    new = {}
    for k, h in enumerate(headers):
        _, start, length = h.split("_")
        new[h] = "S%d_%s_%s" % (k, start, length)
    out = []
    for l in lines:
        m = LINE.match(l)
        out.append("ref_id: %s, query_id: %s, ab: %s, ae: %s, bb: %s, be: %s, score: %s, comp: %s" %
                   ((new[m.group(1)], new[m.group(2)]) + m.groups()[2:]))
    return [new[h] for h in headers], out


def perturb(headers, lines, seed=20261004):
    rng = random.Random(seed)
    out = list(lines)
    for l in lines:
        m = LINE.match(l)
        r, q, ab, ae, bb, be, score, comp = m.groups()
        ab, ae, bb, be, score = int(ab), int(ae), int(bb), int(be), int(score)
        roll = rng.random()
        if roll < 0.15:                          # score just under / at the threshold
            out.append("ref_id: %s, query_id: %s, ab: %d, ae: %d, bb: %d, be: %d, score: %d, comp: %s" %
                       (r, q, ab, ae, bb, be, rng.choice([599, 600, 601]), comp))
        elif roll < 0.30:                        # extent just under / at the minimum length
            ln = rng.choice([989, 990, 991])
            out.append("ref_id: %s, query_id: %s, ab: %d, ae: %d, bb: %d, be: %d, score: %d, comp: %s" %
                       (r, q, ab, ab + ln, bb, bb + rng.choice([989, 990, 2000]), max(score, 700), comp))
        elif roll < 0.40:                        # a self pair
            out.append("ref_id: %s, query_id: %s, ab: %d, ae: %d, bb: %d, be: %d, score: %d, comp: %s" %
                       (r, r, ab, ae, ab, ae, 5000, comp))
        elif roll < 0.55:                        # a pair picked at random: mostly no theoretical overlap
            a, b = rng.sample(headers, 2)
            out.append("ref_id: %s, query_id: %s, ab: %d, ae: %d, bb: %d, be: %d, score: %d, comp: %s" %
                       (a, b, 0, 1500, 10, 1400, 900, comp))
        elif roll < 0.60:
            out.append(l)                        # duplicate
    rng.shuffle(out)
    return out


def run_reference_script(headers, lines):
    """-> dict of what the reference script printed"""
    with tempfile.TemporaryDirectory(prefix="accuracy_golden.") as d:
        py3 = os.path.join(d, "measure_sensitivity_PBSIM.py")
        with open(SCRIPT) as f:
            src = f.read()
        with open(py3, "w") as f:
            f.write(src)
        subprocess.check_call([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", py3],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(os.path.join(d, "reference.fasta"), "w") as f:
            for h in headers:
                f.write(">%s\nACGT\n" % h)           # the script reads header lines only (:44-52)
        with open(os.path.join(d, "out.darwin"), "w") as f:
            f.write("\n".join(lines) + "\n")
        out = subprocess.run([sys.executable, "-W", "ignore", py3], cwd=d, capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            raise SystemExit("the converted reference script failed:\n" + out.stdout[-2000:] + out.stderr[-2000:])
    res = {}
    for key, pat in (("theoretical_with_self", r"Num theoretical ovls: (\d+)"),
                     ("theoretical", r"Num non-trivial theoretical ovls: (\d+)"),
                     ("heuristic_with_mirrors", r"Num heuristic overlaps: (\d+)"),
                     ("kept", r"Num heuristic overlaps after filtering: (\d+)"),
                     ("TP", r"^TP: (\d+)"), ("FN", r"^FN: (\d+)"), ("FP", r"^FP: (\d+)")):
        res[key] = int(re.search(pat, out.stdout, re.M).group(1))
    res["sensitivity"] = float(re.search(r"^sensitivity: ([0-9.]+)", out.stdout, re.M).group(1))
    res["specificity"] = float(re.search(r"^specificity: ([0-9.]+)", out.stdout, re.M).group(1))
    return res


def main():
    if not os.path.exists(SCRIPT):
        raise SystemExit("%s is not mounted: this generator runs in the build container only" % SCRIPT)
    fasta = json.load(open(os.path.join(HERE, "dsoft.json")))["fasta"]
    lines = json.load(open(os.path.join(HERE, "e2e.json")))["lines_sorted"]
    headers = [re.split(r"[^A-Za-z0-9_]", l[1:])[0] for l in fasta.splitlines() if l.startswith(">")]
    headers0, lines0 = renumber(headers, lines)
    cases = {"e2e": lines0, "perturbed": perturb(headers0, lines0)}
    out = {"source": "printed by the reference's own measure_sensitivity_PBSIM.py (Python 2; converted with lib2to3 into a "
                     "temporary directory, never committed) on the inputs below; generator tests/golden/make_accuracy_golden.py",
           "rules": "measure_sensitivity_PBSIM.py:20-22 (score >= 600, extents >= 990), :84-106 (theoretical overlap >= 1000), "
                    ":129-146 (every line and its mirror image), :160-175 (filters), :183-214 (matching), :265-270 (ratios)",
           "headers": headers0, "cases": {}}
    for name, ls in cases.items():
        out["cases"][name] = {"lines": ls, "expected": run_reference_script(headers0, ls)}
        print(name, out["cases"][name]["expected"])
    with open(os.path.join(HERE, "accuracy.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
