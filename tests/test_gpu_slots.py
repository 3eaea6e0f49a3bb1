"""GPU: several runs in flight on the slots of one engine (what the reference's feeder threads do with a GPU_storage
each, darwin.cpp:619-629, and what bench.py does with its steps): the records of a run do not depend on what else is
running, and a launch made while another slot is busy takes the throughput layout instead of the wide one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(eng, rs):
    from gact_amd import engine
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)


def test_a_launch_beside_a_running_slot_takes_the_split_layout(monkeypatch):
    from gact_amd import engine, synth
    for var in ("GACT_HIP_NO_WIDE", "GACT_HIP_FORCE_WIDE", "GACT_HIP_NO_SHARED_HINT"):
        monkeypatch.delenv(var, raising=False)
    rs = synth.simulate_reads(200000, n_reads=160, seed=11, mean_len=9000, sd_len=2500, min_len=1500, max_len=16000)
    cf, cr = synth.synth_candidates(rs, seed=12, min_overlap=400, false_frac=0.1)
    cands = np.concatenate([cf, cr])
    assert 1500 < len(cands) < 20000                        # fewer chains than tile slots: the wide layout's case
    eng = engine.Engine(n_slots=3)
    _load(eng, rs)
    for k in range(3):
        eng.candidates_upload(cands, slot=k)
    eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=0)
    alone = eng.candidates_fetch(len(cands), slot=0).copy()
    assert eng.last_run_stats(0)["layout"] == "packed16-wide"
    # three runs back to back without waiting: the second and third are launched while the first is running
    for k in range(3):
        eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=k)
    recs = [eng.candidates_fetch(len(cands), slot=k).copy() for k in range(3)]
    layouts = [eng.last_run_stats(k)["layout"] for k in range(3)]
    assert layouts[0] == "packed16-wide" and layouts[1] == layouts[2] == "packed16-split", layouts
    for r in recs:
        assert r.tobytes() == alone.tobytes()
    # a caller that says it keeps runs in flight gets the throughput layout from the first launch on (on an idle machine too)
    eng.set_option("runs_in_flight", 1)
    eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=1)
    assert eng.candidates_fetch(len(cands), slot=1).tobytes() == alone.tobytes()
    assert eng.last_run_stats(1)["layout"] == "packed16-split"
    eng.set_option("runs_in_flight", 0)
    eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=1)
    assert eng.candidates_fetch(len(cands), slot=1).tobytes() == alone.tobytes()
    assert eng.last_run_stats(1)["layout"] == "packed16-wide"
    # ... and with the hint switched off every one of them is wide
    eng.close()
    monkeypatch.setenv("GACT_HIP_NO_SHARED_HINT", "1")
    eng = engine.Engine(n_slots=2)
    _load(eng, rs)
    for k in range(2):
        eng.candidates_upload(cands, slot=k)
        eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=k)
    for k in range(2):
        assert eng.candidates_fetch(len(cands), slot=k).tobytes() == alone.tobytes()
        assert eng.last_run_stats(k)["layout"] == "packed16-wide"
    eng.close()


def test_steps_in_flight_with_dirty_reads_keep_their_records(oracle):
    """four slots, each with routed runs (2-bit launches + the raw-byte side lane) in flight at once"""
    from gact_amd import engine, synth
    rs = synth.simulate_reads(60000, n_reads=48, seed=21, mean_len=6000, sd_len=1500, min_len=1200, max_len=10000)
    for k in (3, 17, 30):
        r = rs.reads[k]
        r[400:420] = ord("N")
        r[800:850] = np.frombuffer(bytes(r[800:850]).lower(), dtype=np.uint8)
    cf, cr = synth.synth_candidates(rs, seed=22, min_overlap=300, false_frac=0.1)
    cands = np.concatenate([cf, cr])
    eng = engine.Engine(n_slots=4)
    _load(eng, rs)
    for k in range(4):
        eng.candidates_upload(cands, slot=k)
    for rep in range(3):
        for k in range(4):
            eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=k)
        recs = [eng.candidates_fetch(len(cands), slot=k).copy() for k in range(4)]
        for r in recs[1:]:
            assert r.tobytes() == recs[0].tobytes()
    assert eng.last_run_stats(2)["raw_candidates"] > 0
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    wf, _ = oracle.gact_many(cat, offs, cat, offs, cf, complement=False, same_file=True, n_threads=8)
    wr, _ = oracle.gact_many(cat, offs, rcat, roffs, cr, complement=True, same_file=True, n_threads=8)
    want = np.concatenate([wf, wr])
    for f in ("ab", "ae", "bb", "be", "score", "emitted", "first_tile_score", "n_tiles", "cells"):
        assert np.array_equal(recs[0][f], want[f]), f
    eng.close()


def test_bus_copies_and_sdma_copies_move_the_same_bytes(monkeypatch):
    """candidate lists up and records down through the engine's bus_copy_kernel (default) or hipMemcpyAsync
    (GACT_HIP_SDMA_COPIES=1), into the engine's own pinned array or a registered output: the same records"""
    from gact_amd import engine, synth
    rs = synth.simulate_reads(60000, n_reads=40, seed=31, mean_len=6000, sd_len=1500, min_len=1200, max_len=10000)
    cf, cr = synth.synth_candidates(rs, seed=32, min_overlap=300, false_frac=0.1)
    cands = np.concatenate([cf, cr])
    got = {}
    for mode in ("kernel", "sdma"):
        if mode == "sdma":
            monkeypatch.setenv("GACT_HIP_SDMA_COPIES", "1")
        eng = engine.Engine(n_slots=2)
        _load(eng, rs)
        eng.candidates_upload(cands, slot=0)
        eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=0)
        plain = eng.candidates_fetch(len(cands), slot=0).copy()
        out = np.zeros(len(cands) + 7, dtype=engine.OVERLAP_DTYPE)             # registered, fetched into at an offset
        eng.register_output(out, slot=1)
        eng.candidates_upload(cands[::-1].copy(), slot=1)
        eng.candidates_upload(cands, slot=1)                                   # (a second upload reuses the staging array)
        eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=1)
        eng.candidates_fetch(len(cands), slot=1, out=out[7:])
        assert out[7:].tobytes() == plain.tobytes() and not out[:7].tobytes().strip(b"\0")
        got[mode] = plain
        eng.close()
    assert got["kernel"].tobytes() == got["sdma"].tobytes()
