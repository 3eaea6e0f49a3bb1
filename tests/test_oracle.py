"""CPU suite: the oracle (oracle/gact_oracle.c) against
  * SURVEY.md Appendix B known-answer tiles (measured from the compiled reference),
  * tests/golden/*.json (generated from the reference by tests/golden/make_golden.py),
  * the reference itself when oracle/_ref is present (this container)."""
import json
import os

import numpy as np
import pytest

from tilecases import KAT, random_tiles

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_known_answer_tiles(oracle):
    for a, b, rev, first, header, counts in KAT:
        q = oracle.align_with_bt(a, b, (1, -1, -1, -1), rev, first, 200)
        assert q[:len(header)] == header
        st = q[len(header):]
        assert (st.count(1), st.count(2), st.count(3)) == counts


def test_identical_320mer_stops_at_early_terminate(oracle):
    rng = np.random.default_rng(1)
    s = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 320)])
    for rev, first, header in ((0, 0, [320]), (1, 0, [320]), (0, 1, [320, 320, 320])):
        q = oracle.align_with_bt(s, s, (1, -1, -1, -1), rev, first, 200)
        assert q[:len(header)] == header
        assert q[len(header):] == [3] * 200      # breaks before pushing the 201st (align.cpp:205)


def test_golden_tiles(oracle):
    g = json.load(open(os.path.join(GOLD, "tiles.json")))
    assert len(g["tiles"]) >= 200
    for k, t in enumerate(g["tiles"]):
        q = oracle.align_with_bt(t["ref"].encode("latin-1"), t["query"].encode("latin-1"), tuple(t["scoring"]),
                                 t["reverse"], t["first"], t["early"])
        assert q == t["queue"], k


def test_golden_chains(oracle):
    g = json.load(open(os.path.join(GOLD, "chains.json")))
    reads = [r.encode("latin-1") for r in g["reads"]]
    from gact_amd import synth
    rc = [synth.revcomp(np.frombuffer(r, dtype=np.uint8)).tobytes() for r in reads]
    n_lines = 0
    for k, c in enumerate(g["chains"]):
        st = g["settings"][c["setting"]]
        q = rc[c["query_id"]] if c["comp"] else reads[c["query_id"]]
        ov, _ = oracle.gact(reads[c["ref_id"]], q, c["ref_pos"], c["query_pos"], tile_size=st["tile_size"],
                            tile_overlap=st["tile_overlap"], threshold=st["threshold"], ref_id=c["ref_id"],
                            query_id=c["query_id"], complement=bool(c["comp"]), scoring=tuple(st["scoring"]),
                            same_file=True)
        line = oracle.format_line(ov, g["names"][c["ref_id"]], g["names"][c["query_id"]]) if ov.emitted else ""
        assert line == c["line"], k
        n_lines += bool(line)
    assert n_lines > 40


def test_oracle_equals_reference_tiles(oracle, reflib):
    scorings = [(1, -1, -1, -1), (3, -2, -4, -1)]
    for k, (a, b, rev, first) in enumerate(random_tiles(555, 60)):
        sc = scorings[k % 2]
        assert oracle.align_with_bt(a, b, sc, rev, first, 200) == reflib.align_with_bt(a, b, sc, rev, first, 200)


def test_oracle_equals_reference_start_positions(oracle, reflib):
    """(ref_pos, query_pos) inside, on the border of and beyond the tile (align.cpp:179-181,186; the matrix is
    zero-initialised, :85): pos_score and the traceback start follow the reference"""
    import random
    rng = random.Random(1)
    n = 0
    for a, q, rev, first in random_tiles(7, 40):
        if not len(a) or not len(q):
            continue
        for _ in range(5):
            rp, qp = rng.randint(0, len(a) + 2), rng.randint(0, len(q) + 2)
            assert oracle.align_with_bt(a, q, (1, -1, -1, -1), rev, first, 200, ref_pos=rp, query_pos=qp) == \
                reflib.align_with_bt(a, q, (1, -1, -1, -1), rev, first, 200, ref_pos=rp, query_pos=qp)
            n += 1
    assert n > 100


def test_oracle_equals_reference_chains(oracle, reflib):
    from gact_amd import synth
    rs = synth.simulate_reads(8000, n_reads=6, seed=91, mean_len=2500, sd_len=500, min_len=800, max_len=4000)
    cf, cr = synth.synth_candidates(rs, seed=92, min_overlap=300, false_frac=0.2)
    n = 0
    for comp, cands in ((False, cf[:12]), (True, cr[:12])):
        for c in cands:
            r = rs.reads[c["ref_id"]].tobytes()
            q = (synth.revcomp(rs.reads[c["query_id"]]) if comp else rs.reads[c["query_id"]]).tobytes()
            ov, _ = oracle.gact(r, q, int(c["ref_pos"]), int(c["query_pos"]), ref_id=int(c["ref_id"]),
                                query_id=int(c["query_id"]), complement=comp)
            line = oracle.format_line(ov, "r", "q") if ov.emitted else ""
            assert line == reflib.gact_line(r, q, int(c["ref_pos"]), int(c["query_pos"]), ref_id=int(c["ref_id"]),
                                            query_id=int(c["query_id"]), complement=comp)
            n += 1
    assert n > 10


def test_gact_many_threads_agree(oracle):
    from gact_amd import synth
    rs = synth.simulate_reads(9000, n_reads=8, seed=3, mean_len=2500, sd_len=500, min_len=800, max_len=4000)
    cf, _ = synth.synth_candidates(rs, seed=4, min_overlap=300)
    cat, offs = rs.concat()
    a, ca = oracle.gact_many(cat, offs, cat, offs, cf, n_threads=1)
    b, cb = oracle.gact_many(cat, offs, cat, offs, cf, n_threads=5)
    assert ca == cb and a.tobytes() == b.tobytes()
    # trace of one chain is consistent with its record
    c = cf[0]
    ov, tr = oracle.gact(rs.reads[c["ref_id"]].tobytes(), rs.reads[c["query_id"]].tobytes(), int(c["ref_pos"]),
                         int(c["query_pos"]), ref_id=int(c["ref_id"]), query_id=int(c["query_id"]), trace_cap=512)
    assert len(tr) == ov.n_tiles and sum(t.ref_len * t.query_len for t in tr) == ov.cells
