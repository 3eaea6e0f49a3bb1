"""GPU parity through the reference-shaped C++ surface (darwin-gpu_amd/host/gact.h, align.h):
AlignWithBT, Align_Batch_GPU, GACT, and GPU_init -> GACT_Batch -> GPU_close via the
darwin.cpp-shaped driver, all compared with the oracle."""
import struct
import subprocess

import pytest

from tilecases import random_tiles

pytestmark = pytest.mark.gpu


def _driver():
    from gact_amd import engine
    return engine.build_driver()


def test_alignwithbt_and_align_batch_gpu(oracle, tmp_path):
    cases = [c for c in random_tiles(404, 36) if len(c[0]) and len(c[1])]
    scorings = [(1, -1, -1, -1), (2, -3, -5, -2)]
    lines = []
    meta = []
    for k, (a, b, rev, first) in enumerate(cases):
        sc = scorings[k % 2]
        early = 200 if k % 3 else 64
        lines.append("T %s %s %d %d %d %d %d %d %d" % (a.decode(), b.decode(), *sc, rev, first, early))
        meta.append((a, b, sc, rev, first, early))
    f = tmp_path / "cases.txt"
    f.write_text("\n".join(lines) + "\n")
    out = subprocess.run([_driver(), "--selftest", str(f)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = [l for l in out.stdout.splitlines() if l.startswith(("AlignWithBT", "Align_Batch_GPU"))]
    it = iter(got)
    n_batch = 0
    for a, b, sc, rev, first, early in meta:
        want = oracle.align_with_bt(a, b, sc, rev, first, early)
        l = next(it)
        assert l.startswith("AlignWithBT") and [int(x) for x in l.split()[1:]] == want
        if sc == (1, -1, -1, -1) and early == 200:
            l = next(it)
            head, st = l[len("Align_Batch_GPU"):].split(":")
            score, rsteps, qsteps, mi, mj = [int(x) for x in head.split()]
            st = [int(x) for x in st.split()]
            ws = want[3:] if first else want[1:]
            assert score == want[0] and st == ws
            if first:
                assert [mi, mj] == want[1:3]
            assert rsteps == sum(1 for s in ws if s in (2, 3)) and qsteps == sum(1 for s in ws if s in (1, 3))
            n_batch += 1
    assert n_batch > 5


def test_gact_entry_point(oracle, tmp_path):
    from gact_amd import synth
    rs = synth.simulate_reads(8000, n_reads=6, seed=61, mean_len=2500, sd_len=500, min_len=800, max_len=4000)
    cf, cr = synth.synth_candidates(rs, seed=62, min_overlap=300, false_frac=0.2)
    lines, want = [], []
    for comp, cands in ((0, cf[:6]), (1, cr[:6])):
        for c in cands:
            r = rs.reads[c["ref_id"]].tobytes()
            q = (synth.revcomp(rs.reads[c["query_id"]]) if comp else rs.reads[c["query_id"]]).tobytes()
            lines.append("G %s %s %d %d 320 120 35 1 -1 -1 -1 %d" % (r.decode(), q.decode(), c["ref_pos"], c["query_pos"], comp))
            ov, _ = oracle.gact(r, q, int(c["ref_pos"]), int(c["query_pos"]), ref_id=0, query_id=1,
                                complement=bool(comp), same_file=False)
            want.append("GACT " + (oracle.format_line(ov, "refname", "queryname").strip() if ov.emitted else ""))
    f = tmp_path / "cases.txt"
    f.write_text("\n".join(lines) + "\n")
    out = subprocess.run([_driver(), "--selftest", str(f)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = [l.strip() for l in out.stdout.splitlines() if l.startswith("GACT ") or l.strip() == "GACT"]
    assert got == [w.strip() for w in want]


@pytest.mark.parametrize("threads", [1, 3])
def test_driver_end_to_end(oracle, tmp_path, threads):
    """FASTA + params.cfg + candidates -> darwin.<t>.out, `sort | uniq` equal to the CPU path (README:25)"""
    from gact_amd import synth
    rs = synth.simulate_reads(30000, n_reads=20, seed=71, mean_len=5000, sd_len=1200, min_len=800, max_len=9000)
    cf, cr = synth.synth_candidates(rs, seed=72, min_overlap=300)
    rs.write_fasta(str(tmp_path / "reads.fasta"))
    (tmp_path / "params.cfg").write_text(
        "[GACT_scoring]\nmatch = 1\nmismatch = -1\ngap_open = -1\ngap_extend = -1\n"
        "[GACT_first_tile]\nfirst_tile_size = 128\nfirst_tile_score_threshold = 35\n"
        "[GACT_extend]\ntile_size = 320\ntile_overlap = 120\n")
    with open(tmp_path / "cands.bin", "wb") as f:
        for comp, cands in ((0, cf), (1, cr)):
            for c in cands:
                f.write(struct.pack("<5i", c["ref_id"], c["query_id"], c["ref_pos"], c["query_pos"], comp))
    out = subprocess.run([_driver(), "reads.fasta", "reads.fasta", str(threads), "--candidates", "cands.bin"],
                         capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = []
    for t in range(threads):
        got += open(tmp_path / ("darwin.%d.out" % t)).read().splitlines()
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    want = []
    for comp, cands, qc, qo in ((False, cf, cat, offs), (True, cr, rcat, roffs)):
        recs, _ = oracle.gact_many(cat, offs, qc, qo, cands, complement=comp, same_file=True, n_threads=4)
        for r in recs:
            if r["emitted"]:
                want.append(oracle.format_line(r, rs.names[r["ref_id"]], rs.names[r["query_id"]]).rstrip("\n"))
    assert sorted(set(got)) == sorted(set(want))
    assert sorted(got) == sorted(want)
    assert len(want) > 50


def test_end_to_end_from_fasta_equals_reference_program(tmp_path):
    """FASTA -> D-SOFT restatement -> HIP GACT -> darwin.<t>.out, compared with the lines the REFERENCE's own
    CPU program (oracle/_ref/darwin_cpu) printed for the same FASTA and params.cfg (tests/golden/e2e.json):
    the reference's regression method, x_scalingrun.sh:5-22."""
    import json
    import os
    from test_dsoft import CFG
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fasta = json.load(open(os.path.join(gold, "dsoft.json")))["fasta"]
    e2e = json.load(open(os.path.join(gold, "e2e.json")))
    (tmp_path / "reads.fasta").write_text(fasta)
    (tmp_path / "params.cfg").write_text(CFG % e2e["seed_size"])
    # host filter with 1 and 3 feeder threads, then the whole pipeline on the device (--device-dsoft)
    for threads, extra in ((1, []), (3, []), (1, ["--device-dsoft"]), (3, ["--device-dsoft"])):
        out = subprocess.run([_driver(), "reads.fasta", "reads.fasta", str(threads)] + extra, capture_output=True,
                             text=True, cwd=tmp_path, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        got = []
        for t in range(threads):
            got += open(tmp_path / ("darwin.%d.out" % t)).read().splitlines()
        assert sorted(got) == e2e["lines_sorted"]
        assert len(got) > 40
