"""CPU suite: the accuracy harness (tools/measure_sensitivity.py, own rewrite of the reference's
measure_sensitivity_PBSIM.py in its de-novo mode) pinned on what THE REFERENCE SCRIPT ITSELF printed:
tests/golden/accuracy.json holds, for two sets of output lines over the 16 reads of tests/golden/dsoft.json (the
67 lines the reference's CPU program printed, and an edited set that trips every filter), the counts and ratios
measure_sensitivity_PBSIM.py printed when tests/golden/make_accuracy_golden.py ran it (lib2to3-converted, in a
temporary directory, build container only).  The brute-force restatement of the rules written out below
(theoretical overlaps :84-106, mirrored lines :129-146, filters :160-175, matching :183-214, ratios :265-270)
is kept as a second witness."""
import json
import os
import re
import sys

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def brute_force(fasta_text, lines, score_thres=600, min_length=990):
    reads = []                                                  # (name, start, length), file order
    for l in fasta_text.splitlines():
        if l.startswith(">"):
            name = re.split(r"[^A-Za-z0-9_]", l[1:])[0]
            _, st, ln = name.split("_")
            reads.append((name, int(st), int(ln)))
    idx = {r[0]: k for k, r in enumerate(reads)}
    tovl = []
    for i, (_, a1, la) in enumerate(reads):                     # :84-106, trivial pairs removed (:122-124)
        for j, (_, b1, lb) in enumerate(reads):
            a2, b2 = a1 + la, b1 + lb
            if a2 < b1 or b2 < a1:
                continue
            if min(a2, b2) - max(a1, b1) >= 1000 and i != j:
                tovl.append((i, j))
    hovl = []
    for l in lines:                                             # :129-146: each line and its mirror image
        m = re.match(r"ref_id: (\S+), query_id: (\S+), ab: (\d+), ae: (\d+), bb: (\d+), be: (\d+), score: (-?\d+), comp: (\d)", l)
        r, q = idx[m.group(1)], idx[m.group(2)]
        ab, ae, bb, be, score = (int(m.group(k)) for k in range(3, 8))
        hovl.append([r, q, ab, ae, bb, be, score, 0])
        hovl.append([q, r, bb, be, ab, ae, score, 0])
    hovl = [h for h in hovl if h[0] != h[1]]                    # :160-162
    hovl = [h for h in hovl if h[3] - h[2] >= min_length and h[5] - h[4] >= min_length and h[6] >= score_thres]
    fn = 0
    for t in tovl:                                              # :183-206
        hit = False
        for h in hovl:
            if (h[0], h[1]) == t:
                h[7] = 1
                hit = True
        fn += not hit
    tp = sum(h[7] for h in hovl)
    return {"theoretical": len(tovl), "kept": len(hovl), "TP": tp, "FN": fn, "FP": len(hovl) - tp}


def test_accuracy_harness_pinned(tmp_path):
    import measure_sensitivity
    fasta = json.load(open(os.path.join(GOLD, "dsoft.json")))["fasta"]
    lines = json.load(open(os.path.join(GOLD, "e2e.json")))["lines_sorted"]
    (tmp_path / "reads.fasta").write_text(fasta)
    # split over two files like darwin.0.out / darwin.1.out
    (tmp_path / "darwin.0.out").write_text("\n".join(lines[:30]) + "\n")
    (tmp_path / "darwin.1.out").write_text("\n".join(lines[30:]) + "\nnot an overlap line\n")
    got = measure_sensitivity.measure(str(tmp_path / "reads.fasta"),
                                      [str(tmp_path / "darwin.0.out"), str(tmp_path / "darwin.1.out")])
    want = brute_force(fasta, lines)
    for k in ("theoretical", "kept", "TP", "FN", "FP"):
        assert got[k] == want[k], (k, got, want)
    assert got["reads"] == 16 and got["lines"] == 67
    assert abs(got["sensitivity"] - want["TP"] / (want["TP"] + want["FN"])) < 1e-12
    assert abs(got["specificity"] - want["TP"] / (want["TP"] + want["FP"])) < 1e-12
    # the numbers themselves (reference output on the committed FASTA; any change here is a change of the rules)
    assert (got["theoretical"], got["kept"], got["TP"], got["FN"], got["FP"]) == PINNED, got


PINNED = (58, 118, 112, 4, 6)


def test_accuracy_harness_equals_the_reference_script(tmp_path):
    """every case of tests/golden/accuracy.json: the numbers the reference's own script printed"""
    import measure_sensitivity
    gold = json.load(open(os.path.join(GOLD, "accuracy.json")))
    assert set(gold["cases"]) >= {"e2e", "perturbed"}
    fasta = "".join(">%s\nACGT\n" % h for h in gold["headers"])
    (tmp_path / "reads.fasta").write_text(fasta)
    for name, case in gold["cases"].items():
        out = tmp_path / ("%s.out" % name)
        out.write_text("\n".join(case["lines"]) + "\n")
        got = measure_sensitivity.measure(str(tmp_path / "reads.fasta"), [str(out)])
        want = case["expected"]
        for k in ("theoretical", "kept", "TP", "FN", "FP"):
            assert got[k] == want[k], (name, k, got, want)
        assert got["lines"] * 2 == want["heuristic_with_mirrors"]
        # the script prints six decimals (:269-270)
        assert abs(got["sensitivity"] - want["sensitivity"]) < 1e-6 and abs(got["specificity"] - want["specificity"]) < 1e-6
        # ... and the brute-force restatement agrees with both
        bf = brute_force(fasta, case["lines"])
        assert {k: bf[k] for k in ("theoretical", "kept", "TP", "FN", "FP")} == {k: want[k] for k in ("theoretical", "kept", "TP", "FN", "FP")}
    e = gold["cases"]["e2e"]["expected"]
    assert (e["theoretical"], e["kept"], e["TP"], e["FN"], e["FP"]) == PINNED
