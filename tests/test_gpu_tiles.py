"""GPU parity, tile level: HIP align_tiles (through the C-ABI) vs the oracle's
AlignWithBT restatement, bit-exact on score, arg-max and every traceback state."""
import numpy as np
import pytest

from tilecases import KAT, random_tiles

pytestmark = pytest.mark.gpu

SCORINGS = [(1, -1, -1, -1), (2, -3, -5, -2), (5, -4, -10, -1), (1, -1, -2, -1), (3, -2, -1, -4)]


def _check(eng, oracle, cases, scoring, early):
    from gact_amd import engine
    refs = [c[0] for c in cases]; qs = [c[1] for c in cases]
    res, states = eng.align_tiles_inline(refs, qs, [c[2] for c in cases], [c[3] for c in cases])
    for t, (a, b, rev, first) in enumerate(cases):
        want = oracle.align_with_bt(a, b, scoring, rev, first, early)
        got = engine.queue_from_tile(res[t], states[t], first)
        assert got == want, "tile %d R=%d Q=%d rev=%d first=%d scoring=%s\n got %s\nwant %s" % (
            t, len(a), len(b), rev, first, scoring, got[:12], want[:12])
        nst = len(want) - (3 if first else 1)
        st = want[(3 if first else 1):]
        assert res[t]["ref_steps"] == sum(1 for s in st if s in (2, 3))
        assert res[t]["query_steps"] == sum(1 for s in st if s in (1, 3))
        assert res[t]["n_states"] == nst


def test_known_answer_tiles(oracle):
    from gact_amd import engine
    eng = engine.Engine()
    cases = [(k[0], k[1], k[2], k[3]) for k in KAT]
    res, states = eng.align_tiles_inline([c[0] for c in cases], [c[1] for c in cases],
                                         [c[2] for c in cases], [c[3] for c in cases])
    for t, k in enumerate(KAT):
        got = engine.queue_from_tile(res[t], states[t], k[3])
        assert got[:len(k[4])] == k[4]
        st = got[len(k[4]):]
        assert (st.count(1), st.count(2), st.count(3)) == k[5]
    eng.close()


@pytest.mark.parametrize("scoring", SCORINGS)
def test_random_tiles_all_modes(oracle, scoring):
    from gact_amd import engine
    eng = engine.Engine(scoring=scoring)
    _check(eng, oracle, random_tiles(101 + scoring[0], 420), scoring, 200)
    eng.close()


def test_tiles_with_non_acgt(oracle):
    """N==N is a match, case matters (align.cpp:134): raw-byte mode"""
    from gact_amd import engine
    eng = engine.Engine()
    _check(eng, oracle, random_tiles(7, 200, with_n=True), (1, -1, -1, -1), 200)
    eng.close()


@pytest.mark.parametrize("tile,overlap", [(320, 120), (320, 0), (320, 319), (128, 32), (64, 8), (96, 48), (448, 128),
                                          (512, 200)])
def test_other_tile_geometries(oracle, tile, overlap):
    from gact_amd import engine
    eng = engine.Engine(tile_size=tile, tile_overlap=overlap)
    _check(eng, oracle, random_tiles(33 + tile + overlap, 160, max_len=tile), (1, -1, -1, -1), tile - overlap)
    eng.close()


def test_resident_tiles_both_strands(oracle):
    """tiles addressed as offsets into resident read sets (forward and rc)"""
    from gact_amd import engine, synth
    rs = synth.simulate_reads(5000, n_reads=6, seed=3, mean_len=1500, sd_len=300, min_len=700, max_len=2500)
    eng = engine.Engine()
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    eng.upload(engine.SET_REF, cat, offs)
    eng.upload(engine.SET_QUERY, cat, offs)
    eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    rng = np.random.default_rng(5)
    tiles = np.zeros(300, dtype=engine.TILE_DTYPE)
    want = []
    for t in range(300):
        ri, qi = int(rng.integers(0, rs.n)), int(rng.integers(0, rs.n))
        qset = engine.SET_QUERY_RC if t % 2 else engine.SET_QUERY
        rl = int(min(320, rng.integers(1, 400))); ql = int(min(320, rng.integers(1, 400)))
        rl = min(rl, len(rs.reads[ri])); ql = min(ql, len(rs.reads[qi]))
        ro = int(rng.integers(0, len(rs.reads[ri]) - rl + 1)); qo = int(rng.integers(0, len(rs.reads[qi]) - ql + 1))
        rev, first = int(t // 2 % 2), int(t // 4 % 2)
        tiles[t] = (ri, qi, ro, qo, rl, ql, rev, first, qset, 0)
        q = synth.revcomp(rs.reads[qi]) if qset == engine.SET_QUERY_RC else rs.reads[qi]
        want.append(oracle.align_with_bt(rs.reads[ri][ro:ro + rl].tobytes(), q[qo:qo + ql].tobytes(),
                                         (1, -1, -1, -1), rev, first, 200))
    tiles[17]["ref_len"] = -1     # idle slot (cuda_host.cu:70)
    res, states = eng.align_tiles(tiles)
    for t in range(300):
        if t == 17:
            assert res[t]["n_states"] == 0 and res[t]["score"] == 0
            continue
        assert engine.queue_from_tile(res[t], states[t], tiles[t]["first"]) == want[t], t
    eng.close()


def test_bad_descriptors_are_refused():
    from gact_amd import engine
    eng = engine.Engine()
    eng.upload_seqs(engine.SET_REF, [b"ACGT" * 100])
    eng.upload_seqs(engine.SET_QUERY, [b"ACGT" * 100])
    eng.upload_seqs(engine.SET_QUERY_RC, [b"ACGT" * 100])
    tiles = np.zeros(1, dtype=engine.TILE_DTYPE)
    tiles[0] = (0, 0, 390, 0, 20, 20, 0, 0, engine.SET_QUERY, 0)
    with pytest.raises(engine.GactHipError):
        eng.align_tiles(tiles)
    tiles[0] = (0, 0, 0, 0, 321, 20, 0, 0, engine.SET_QUERY, 0)
    with pytest.raises(engine.GactHipError):
        eng.align_tiles(tiles)
    eng.close()
    with pytest.raises(engine.GactHipError):
        engine.Engine(scoring=(1, 1, -1, -1))
    with pytest.raises(engine.GactHipError):
        engine.Engine(tile_size=9999)
