"""GPU parity against the committed golden vectors (outputs of the reference itself)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_hip_tiles_equal_reference_golden():
    from gact_amd import engine
    g = json.load(open(os.path.join(GOLD, "tiles.json")))["tiles"]
    groups = {}
    for k, t in enumerate(g):
        groups.setdefault((tuple(t["scoring"]), t["early"]), []).append(k)
    for (sc, early), idx in groups.items():
        eng = engine.Engine(tile_size=320, tile_overlap=320 - early, scoring=sc)
        res, states = eng.align_tiles_inline([g[k]["ref"].encode("latin-1") for k in idx],
                                             [g[k]["query"].encode("latin-1") for k in idx],
                                             [g[k]["reverse"] for k in idx], [g[k]["first"] for k in idx])
        for n, k in enumerate(idx):
            assert engine.queue_from_tile(res[n], states[n], g[k]["first"]) == g[k]["queue"], k
        eng.close()


def test_hip_chains_equal_reference_golden_lines():
    from gact_amd import engine, synth
    g = json.load(open(os.path.join(GOLD, "chains.json")))
    reads = [np.frombuffer(r.encode("latin-1"), dtype=np.uint8) for r in g["reads"]]
    for si, st in enumerate(g["settings"]):
        eng = engine.Engine(tile_size=st["tile_size"], tile_overlap=st["tile_overlap"],
                            scoring=tuple(st["scoring"]), threshold=st["threshold"])
        eng.upload_seqs(engine.SET_REF, reads)
        eng.upload_seqs(engine.SET_QUERY, reads)
        eng.upload_seqs(engine.SET_QUERY_RC, [synth.revcomp(r) for r in reads])
        for comp in (0, 1):
            cs = [c for c in g["chains"] if c["setting"] == si and c["comp"] == comp]
            cands = np.array([(c["ref_id"], c["query_id"], c["ref_pos"], c["query_pos"]) for c in cs],
                             dtype=engine.CAND_DTYPE)
            out = eng.extend(cands, complement=bool(comp), same_file=True)
            for c, o in zip(cs, out):
                line = eng.format_overlap(o, g["names"][c["ref_id"]], g["names"][c["query_id"]]) if o["emitted"] else ""
                assert line == c["line"], c
        eng.close()


def test_mixed_strand_launch_equals_two_launches():
    from gact_amd import engine, synth
    rs = synth.simulate_reads(20000, n_reads=14, seed=12, mean_len=4000, sd_len=900, min_len=800, max_len=7000)
    cf, cr = synth.synth_candidates(rs, seed=13, min_overlap=300)
    eng = engine.Engine()
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs)
    eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    a = eng.extend(cf, complement=False); b = eng.extend(cr, complement=True)
    eng.candidates_upload(np.concatenate([cf, cr]))
    eng.candidates_run_mixed(len(cf) + len(cr), rc_from=len(cf))
    m = eng.candidates_fetch(len(cf) + len(cr))
    assert m[:len(cf)].tobytes() == a.tobytes() and m[len(cf):].tobytes() == b.tobytes()
    # a sub-range, as a multi-GPU shard would run it
    eng.candidates_run_mixed(10, rc_from=len(cf), first=len(cf) - 4)
    s = eng.candidates_fetch(len(cf) + 6)
    assert s[len(cf) - 4:len(cf) + 6].tobytes() == m[len(cf) - 4:len(cf) + 6].tobytes()
    eng.close()
