"""GPU: the two round-5 forms of the split linear-gap main launch -- the role launch (csrc/gact_roles.hpp: DP waves that carry
two banks of tiles, walker waves that take the traceback walk, align.cpp:185-230, off them) and the cooperative launch
(csrc/gact_coop.hpp: two banks per wave, the walks of a whole block batched on whichever wave needs a result first) -- give
the records of the one-wave-does-all launch and of the oracle.

The walk is walk_chain_lin's, move for move, on another wave; what can go wrong is the hand-over (jobs and results in
LDS, the staged bases of a tile kept for its walker, two banks sharing a wave's queues) and the grid (one block of
twelve waves per CU).  So: lists of every size around the grid's bank sizes, chains that end in every phase, a narrow band
(second runs of a tile through the full window), ragged reads, both strands, repeated runs, several slots in flight.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted", "first_tile_score", "n_tiles", "cells")


def _load(eng, rs):
    from gact_amd import engine
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)


def _oracle_records(oracle, rs, cf, cr):
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    wf, _ = oracle.gact_many(cat, offs, cat, offs, cf, complement=False, same_file=True, n_threads=8)
    wr, _ = oracle.gact_many(cat, offs, rcat, roffs, cr, complement=True, same_file=True, n_threads=8)
    return np.concatenate([wf, wr])


MODES = ("coop", "roles")


def _enable(monkeypatch, mode):
    monkeypatch.setenv("GACT_HIP_NO_WIDE", "1")              # (lists this small would take the wide layout)
    monkeypatch.setenv({"roles": "GACT_HIP_ROLES", "coop": "GACT_HIP_COOP"}[mode], "1")


def _ran(st, mode):
    return st["layout"] == "packed16-split" and st["linear_gap"] and (st["role_waves"] if mode == "roles" else st["coop_walks"])


def _same(got, want, what=""):
    for f in FIELDS:
        if not np.array_equal(got[f], want[f]):
            k = int(np.flatnonzero(got[f] != want[f])[0])
            raise AssertionError("%s: %s differs at candidate %d: hip %s, oracle %s" % (what, f, k, got[k], want[k]))


@pytest.fixture(scope="module")
def reads_and_records(oracle):
    from gact_amd import synth
    rs = synth.simulate_reads(260000, n_reads=220, seed=71, mean_len=7000, sd_len=2500, min_len=900, max_len=16000)
    cf, cr = synth.synth_candidates(rs, seed=72, min_overlap=300, false_frac=0.12)
    return rs, cf, cr, _oracle_records(oracle, rs, cf, cr)


@pytest.mark.parametrize("mode", MODES)
def test_role_launch_equals_the_oracle_and_the_old_launch(monkeypatch, reads_and_records, mode):
    from gact_amd import engine
    rs, cf, cr, want = reads_and_records
    cands = np.concatenate([cf, cr])
    _enable(monkeypatch, mode)
    eng = engine.Engine()
    _load(eng, rs)
    eng.candidates_upload(cands)
    for rep in range(3):
        eng.candidates_run_mixed(len(cands), rc_from=len(cf))
        got = eng.candidates_fetch(len(cands)).copy()
        st = eng.last_run_stats()
        assert _ran(st, mode), st
        _same(got, want, "%s launch, run %d" % (mode, rep))
    eng.close()
    monkeypatch.delenv({"roles": "GACT_HIP_ROLES", "coop": "GACT_HIP_COOP"}[mode])
    monkeypatch.setenv("GACT_HIP_COOP", "0")
    eng = engine.Engine()
    _load(eng, rs)
    eng.candidates_upload(cands)
    eng.candidates_run_mixed(len(cands), rc_from=len(cf))
    old = eng.candidates_fetch(len(cands)).copy()
    assert not eng.last_run_stats()["role_waves"] and not eng.last_run_stats()["coop_walks"]
    assert old.tobytes() == got.tobytes()
    eng.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n", [1, 7, 32, 33, 81, 161, 1000])
def test_role_launch_at_every_list_size_around_a_bank(monkeypatch, reads_and_records, n, mode):
    """a block's bank holds 80 chains (role launch: 10 DP waves x 4 groups x 2 slots) or 32 (cooperative launch: 4 waves): empty
    banks, half-filled groups, one chain alone"""
    from gact_amd import engine
    rs, cf, cr, want = reads_and_records
    nf = len(cf)
    monkeypatch.setenv("GACT_HIP_NO_WIDE", "1")
    eng = engine.Engine()
    eng.set_option(mode, 1)                                  # (the live switch)
    _load(eng, rs)
    cands = np.concatenate([cf, cr])
    eng.candidates_upload(cands)
    first = max(0, nf - n // 2)                              # a range across the strand boundary
    n = min(n, len(cands) - first)
    eng.candidates_run_mixed(n, rc_from=nf, first=first)
    got = eng.candidates_fetch(len(cands))[first:first + n].copy()
    assert _ran(eng.last_run_stats(), mode)
    _same(got, want[first:first + n], "range of %d" % n)
    eng.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("band", [24, 0])
def test_role_launch_second_runs_through_the_whole_window(monkeypatch, reads_and_records, band, mode):
    """GACT_HIP_BAND=24: some walks leave the stored band, the walker says so, the DP wave runs the tile again"""
    from gact_amd import engine
    rs, cf, cr, want = reads_and_records
    _enable(monkeypatch, mode)
    monkeypatch.setenv("GACT_HIP_BAND", str(band))
    eng = engine.Engine()
    _load(eng, rs)
    cands = np.concatenate([cf, cr])
    eng.candidates_upload(cands)
    eng.candidates_run_mixed(len(cands), rc_from=len(cf))
    got = eng.candidates_fetch(len(cands)).copy()
    st = eng.last_run_stats()
    assert _ran(st, mode) and (st["band_redos"] > 0) == (band == 24), st
    _same(got, want, "band %d" % band)
    eng.close()


@pytest.mark.parametrize("mode", MODES)
def test_role_launches_of_several_slots_in_flight(monkeypatch, reads_and_records, mode):
    from gact_amd import engine
    rs, cf, cr, want = reads_and_records
    _enable(monkeypatch, mode)
    S = 3
    eng = engine.Engine(n_slots=S)
    _load(eng, rs)
    cands = np.concatenate([cf, cr])
    for k in range(S):
        eng.candidates_upload(cands, slot=k)
    eng.set_option("runs_in_flight", 1)
    for rep in range(2):
        for k in range(S):
            eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=k)
        for k in range(S):
            got = eng.candidates_fetch(len(cands), slot=k).copy()
            assert _ran(eng.last_run_stats(k), mode)
            _same(got, want, "slot %d" % k)
    eng.close()
