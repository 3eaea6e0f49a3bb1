"""GPU: the three scheduling changes of round 4 leave every record as it was.

  * the call combiner: runs that several THREADS submit on several slots at about the same time are merged into one launch
    (csrc/gact_engine.hip Combiner; the reference's eight feeder threads behind their barrier, darwin.cpp:408-433,619-629)
  * ordered, overlapped seeding of a large run on an idle engine (run_overlapped: two seed launches, two main launches,
    two sets of queues)
  * banded pointer stores of the linear-gap main launch (csrc/gact_lin.hpp LinBand): a narrow band forces second runs
"""
import threading

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


def _same(got, want):
    for f in FIELDS:
        assert np.array_equal(got[f], want[f]), f


@pytest.mark.parametrize("dirty", [False, True], ids=["clean", "reads-with-N"])
def test_runs_of_several_threads_are_merged_and_keep_their_records(oracle, dirty):
    """six threads, six slots, six DIFFERENT lists (sizes, strand mixes, one forward-only, one a sub-range of its upload):
    merged into one launch they give what the oracle gives for each list"""
    from gact_amd import engine, synth
    rs = synth.simulate_reads(90000, n_reads=70, seed=31, mean_len=6000, sd_len=1500, min_len=1200, max_len=10000)
    if dirty:
        for k in (5, 22, 41):
            r = rs.reads[k]
            r[300:330] = ord("N")
            r[900:960] = np.frombuffer(bytes(r[900:960]).lower(), dtype=np.uint8)
    cf, cr = synth.synth_candidates(rs, seed=32, min_overlap=300, false_frac=0.15)
    want_all = _oracle_records(oracle, rs, cf, cr)
    T = 6
    eng = engine.Engine(n_slots=T)
    _load(eng, rs)
    lists = []
    for k in range(T):
        f, r = cf[k::T], cr[k::T]
        if k == 1:
            r = r[:0]                                        # forward strand only
        if k == 2:
            f = f[:0]                                        # reverse-complement only
        lists.append((f, r, np.concatenate([want_all[:len(cf)][k::T][:len(f)], want_all[len(cf):][k::T][:len(r)]])))
        eng.candidates_upload(np.concatenate([f, r]), slot=k)
    gate = threading.Barrier(T)
    got, merged, errors = [None] * T, [0] * T, []

    def feeder(k):
        try:
            f, r, _ = lists[k]
            n = len(f) + len(r)
            for rep in range(3):
                gate.wait()
                if k == 4:                                   # a sub-range of the uploaded list: records [5, n) only
                    eng.candidates_run_mixed(n - 5, rc_from=len(f), first=5, slot=k)
                else:
                    eng.candidates_run_mixed(n, rc_from=len(f), slot=k)
                got[k] = eng.candidates_fetch(n, slot=k).copy()
                merged[k] = max(merged[k], eng.last_run_stats(k)["merged_callers"])
        except Exception as err:
            errors.append(err)
            gate.abort()

    threads = [threading.Thread(target=feeder, args=(k,)) for k in range(T)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert max(merged) >= 2, merged                          # (behind a barrier all six normally end up in one launch)
    for k in range(T):
        f, r, want = lists[k]
        lo = 5 if k == 4 else 0
        _same(got[k][lo:], want[lo:])
    if dirty:
        assert eng.last_run_stats(0)["raw_candidates"] > 0
    # the same with the combiner switched off: every run its own launches
    eng.set_option("combine", 0)
    merged = [0] * T
    threads = [threading.Thread(target=feeder, args=(k,)) for k in range(T)]
    gate.reset()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors and max(merged) == 1
    for k in range(T):
        lo = 5 if k == 4 else 0
        _same(got[k][lo:], lists[k][2][lo:])
    eng.close()


def test_one_thread_with_several_slots_is_never_merged():
    """steps in flight from one thread (bench.py): no waiting, no merging"""
    from gact_amd import engine, synth
    rs = synth.simulate_reads(40000, n_reads=30, seed=41, mean_len=5000, sd_len=1000, min_len=1200, max_len=8000)
    cf, cr = synth.synth_candidates(rs, seed=42, min_overlap=300)
    cands = np.concatenate([cf, cr])
    eng = engine.Engine(n_slots=3)
    _load(eng, rs)
    for k in range(3):
        eng.candidates_upload(cands, slot=k)
    for k in range(3):
        eng.candidates_run_mixed(len(cands), rc_from=len(cf), slot=k)
    recs = [eng.candidates_fetch(len(cands), slot=k).copy() for k in range(3)]
    assert all(eng.last_run_stats(k)["merged_callers"] == 1 for k in range(3))
    assert recs[1].tobytes() == recs[0].tobytes() == recs[2].tobytes()
    eng.close()


def test_overlapped_seeding_equals_the_plain_sequence():
    """ecoli10x (65,766 candidates: large enough for the engine to seed in length order beside its main launch): the same
    records with the switch off, and run after run"""
    from conftest import workload_block
    from gact_amd import engine
    blk = workload_block("ecoli10x")
    eng = engine.Engine()
    _load(eng, blk.rs)
    cands = np.concatenate([blk.cf, blk.cr])
    nf = len(blk.cf)
    eng.candidates_upload(cands)
    recs = []
    for rep in range(3):
        eng.candidates_run_mixed(len(cands), rc_from=nf)
        recs.append(eng.candidates_fetch(len(cands)).copy())
        st = eng.last_run_stats()
        assert st["overlapped_seeding"] and st["handed_off"] > 60000
    eng.set_option("overlap_seed", 0)
    eng.candidates_run_mixed(len(cands), rc_from=nf)
    plain = eng.candidates_fetch(len(cands)).copy()
    assert not eng.last_run_stats()["overlapped_seeding"]
    for r in recs:
        assert r.tobytes() == plain.tobytes()
    # a range of the list (still large enough)
    eng.set_option("overlap_seed", 1)
    eng.candidates_run_mixed(len(cands) - 100, rc_from=nf, first=100)
    part = eng.candidates_fetch(len(cands))
    assert eng.last_run_stats()["overlapped_seeding"]
    assert part[100:].tobytes() == plain[100:].tobytes()
    eng.close()


def test_a_group_of_feeder_threads_that_came_apart_joins_again():
    """Eight feeder threads with an eighth of ecoli10x each, all runs in one launch; then half of them is held back for one
    step, so that the two halves launch apart -- and would go on taking turns, each launch with four runs, sharing the machine.
    The leader of one half waits for the other half's launch once (the call combiner's re-join rule) and from then on every
    launch carries all eight runs again; records as in the joint launches."""
    import time
    from conftest import workload_block
    from gact_amd import engine
    blk = workload_block("ecoli10x")
    T = 8
    eng = engine.Engine(n_slots=T)
    _load(eng, blk.rs)
    parts = []
    for k in range(T):
        f, r = blk.cf[k::T], blk.cr[k::T]
        eng.candidates_upload(np.concatenate([f, r]), slot=k)
        parts.append((len(f), len(r)))
    gate = threading.Barrier(T)
    first, last, merged_hist, errors = [None] * T, [None] * T, [[] for _ in range(T)], []

    def feeder(k):
        try:
            nf, nr = parts[k]
            def step():
                eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True, slot=k)
                rec = eng.candidates_fetch(nf + nr, slot=k).copy()
                merged_hist[k].append(eng.last_run_stats(k)["merged_callers"])
                return rec
            gate.wait()
            first[k] = step()                       # all together
            step()
            gate.wait()
            if k >= T // 2:
                time.sleep(0.012)                   # the second half misses this launch
            for _ in range(8):
                last[k] = step()
        except Exception as err:
            errors.append(err)
            try:
                gate.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=feeder, args=(k,)) for k in range(T)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # what must hold whatever the machine's timing: every run returned its own list's records, and the group that came
    # apart got larger again afterwards.  The exact shape -- T, T, < T, ..., T, T -- depends on the combiner's 1-3 ms collect
    # windows and on this process's thread scheduling (flaky on a loaded box or under a profiler): reported, and
    # fatal only with GACT_TEST_STRICT_TIMING=1
    for k in range(T):
        assert last[k].tobytes() == first[k].tobytes()
    split = min(min(h) for h in merged_hist)
    assert max(h[-1] for h in merged_hist) > split or split == T, merged_hist
    shape = (all(h[0] == T and h[1] == T for h in merged_hist) and any(h[2] < T for h in merged_hist) and
             all(h[-1] == T and h[-2] == T for h in merged_hist))
    if not shape:
        import os
        import warnings
        warnings.warn("feeder group shape not T, T, < T, ..., T, T on this run: %r" % (merged_hist,))
        assert not os.environ.get("GACT_TEST_STRICT_TIMING"), merged_hist
    eng.close()


def test_feeder_threads_with_jitter_keep_their_records(oracle):
    """seven threads, seven slots, fifteen runs each with random pauses of 0-6 ms in between (and one thread that leaves after
    five): whatever groups the combiner forms, splits and joins again, every run returns its own list's records, and nobody waits
    for long (each run of this size takes a few milliseconds)"""
    import time
    from gact_amd import engine, synth
    rs = synth.simulate_reads(90000, n_reads=70, seed=61, mean_len=6000, sd_len=1500, min_len=1200, max_len=10000)
    cf, cr = synth.synth_candidates(rs, seed=62, min_overlap=300, false_frac=0.15)
    want_all = _oracle_records(oracle, rs, cf, cr)
    T = 7
    eng = engine.Engine(n_slots=T)
    _load(eng, rs)
    lists = []
    for k in range(T):
        f, r = cf[k::T], cr[k::T]
        lists.append((f, r, np.concatenate([want_all[:len(cf)][k::T], want_all[len(cf):][k::T]])))
        eng.candidates_upload(np.concatenate([f, r]), slot=k)
    errors, slowest, merged_seen = [], [0.0] * T, [set() for _ in range(T)]

    def feeder(k):
        try:
            rng = np.random.default_rng(1000 + k)
            f, r, want = lists[k]
            n = len(f) + len(r)
            for rep in range(5 if k == 3 else 15):
                time.sleep(float(rng.uniform(0, 0.006)))
                t0 = time.perf_counter()
                eng.candidates_run_mixed(n, rc_from=len(f), slot=k)
                got = eng.candidates_fetch(n, slot=k)
                slowest[k] = max(slowest[k], time.perf_counter() - t0)
                merged_seen[k].add(eng.last_run_stats(k)["merged_callers"])
                for fld in ("ab", "ae", "bb", "be", "score", "emitted", "first_tile_score", "n_tiles", "cells"):
                    assert np.array_equal(got[fld], want[fld]), (k, rep, fld)
        except Exception as err:
            errors.append(err)

    threads = [threading.Thread(target=feeder, args=(k,)) for k in range(T)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]
    assert max(slowest) < 2.0, slowest                      # (no run sat in a collect window or a re-join wait for long; generous:
                                                            #  a wall-clock bound on a shared box)
    assert any(len(m) > 1 or max(m) > 1 for m in merged_seen), merged_seen      # (some runs were merged)
    eng.close()


def test_the_critical_lane_changes_no_record(monkeypatch):
    """A run of 1-1.5 chains per tile slot on an idle engine has a wide main launch beside its split one, on a third of the
    blocks, and the split launch leaves it the longest chains (ChainQueues::leave_longest): a 30,000-candidate range of
    ecoli10x (the lane beside the one main launch, as for the merged forward calls of the reference's feeder threads) and,
    with GACT_HIP_CRIT_LANE_ALWAYS=1, ecoli10x whole (the overlapped sequence: main launch 2 is the lane).  Same records as
    with GACT_HIP_NO_CRIT_LANE=1, run after run."""
    from conftest import workload_block
    from gact_amd import engine
    blk = workload_block("ecoli10x")
    cands = np.concatenate([blk.cf, blk.cr])
    nf = len(blk.cf)
    monkeypatch.setenv("GACT_HIP_NO_CRIT_LANE", "1")
    eng = engine.Engine()
    _load(eng, blk.rs)
    eng.candidates_upload(cands)
    eng.candidates_run_mixed(len(cands), rc_from=nf)
    whole = eng.candidates_fetch(len(cands)).copy()
    assert not eng.last_run_stats()["critical_lane"]
    eng.candidates_run_mixed(30000, rc_from=nf, first=20000)
    part = eng.candidates_fetch(len(cands))[20000:50000].copy()
    assert not eng.last_run_stats()["critical_lane"]
    eng.close()
    monkeypatch.delenv("GACT_HIP_NO_CRIT_LANE")
    eng = engine.Engine()
    _load(eng, blk.rs)
    eng.candidates_upload(cands)
    eng.candidates_run_mixed(len(cands), rc_from=nf)
    assert eng.candidates_fetch(len(cands)).tobytes() == whole.tobytes()
    st = eng.last_run_stats()
    assert st["overlapped_seeding"] and not st["critical_lane"], st          # (a run that size is bound by throughput)
    eng.close()
    monkeypatch.setenv("GACT_HIP_CRIT_LANE_ALWAYS", "1")
    eng = engine.Engine()
    _load(eng, blk.rs)
    eng.candidates_upload(cands)
    for rep in range(3):
        eng.candidates_run_mixed(len(cands), rc_from=nf)
        got = eng.candidates_fetch(len(cands))
        st = eng.last_run_stats()
        assert st["critical_lane"] and st["overlapped_seeding"], st
        assert got.tobytes() == whole.tobytes()
        eng.candidates_run_mixed(30000, rc_from=nf, first=20000)
        got = eng.candidates_fetch(len(cands))[20000:50000]
        st = eng.last_run_stats()
        assert st["critical_lane"] and not st["overlapped_seeding"], st
        assert got.tobytes() == part.tobytes()
    # random ranges of the size the lane is for (a candidate's record does not depend on what else is in the run)
    rng = np.random.default_rng(404)
    for _ in range(6):
        n = int(rng.integers(25000, 36000))
        first = int(rng.integers(0, len(cands) - n))
        eng.candidates_run_mixed(n, rc_from=nf, first=first)
        got = eng.candidates_fetch(len(cands))[first:first + n]
        assert eng.last_run_stats()["critical_lane"]
        assert got.tobytes() == whole[first:first + n].tobytes(), (first, n)
    # not for a run that shares the machine, nor for one small enough to be all wide
    eng.set_option("runs_in_flight", 1)
    eng.candidates_run_mixed(30000, rc_from=nf, first=20000)
    assert eng.candidates_fetch(len(cands))[20000:50000].tobytes() == part.tobytes()
    assert not eng.last_run_stats()["critical_lane"]
    eng.set_option("runs_in_flight", 0)
    eng.candidates_run_mixed(8000, rc_from=nf, first=20000)
    assert eng.candidates_fetch(len(cands))[20000:28000].tobytes() == part[:8000].tobytes()
    st = eng.last_run_stats()
    assert not st["critical_lane"] and st["layout"] == "packed16-wide"
    eng.close()


def test_the_lone_mix_changes_no_record():
    """A run of S/2 ... S chains alone on the machine (S = 24,576 resident tile slots) runs one block per CU of two kinds: 48
    wide blocks for its longest chains, split blocks with the look-ahead walker (SplitLayoutLinTeam) on the other CUs
    (gact_policy.hpp lone_lane).  Ranges of ecoli10x of that size: the same records as all wide ("lone_lane" 0), as without
    the look-ahead walker (-48), with other lane sizes, and as the same candidates inside the whole list's run."""
    from conftest import workload_block
    from gact_amd import engine
    blk = workload_block("ecoli10x")
    cands = np.concatenate([blk.cf, blk.cr])
    nf = len(blk.cf)
    eng = engine.Engine()
    _load(eng, blk.rs)
    eng.candidates_upload(cands)
    eng.candidates_run_mixed(len(cands), rc_from=nf)
    whole = eng.candidates_fetch(len(cands)).copy()
    rng = np.random.default_rng(505)
    for lane in (48, 0, -48, 16, 96, 48):
        eng.set_option("lone_lane", lane)
        for _ in range(3):
            n = int(rng.integers(12300, 24576))
            first = int(rng.integers(0, len(cands) - n))
            eng.candidates_run_mixed(n, rc_from=nf, first=first)
            got = eng.candidates_fetch(len(cands))[first:first + n]
            st = eng.last_run_stats()
            assert st["critical_lane"] == (lane != 0) and st["layout"] == ("packed16-split" if lane else "packed16-wide"), (lane, n, st)
            assert got.tobytes() == whole[first:first + n].tobytes(), (lane, first, n)
    # not below S/2 chains, not for a run that shares the machine
    eng.candidates_run_mixed(12000, rc_from=nf, first=30000)
    assert eng.candidates_fetch(len(cands))[30000:42000].tobytes() == whole[30000:42000].tobytes()
    assert eng.last_run_stats()["layout"] == "packed16-wide" and not eng.last_run_stats()["critical_lane"]
    eng.set_option("runs_in_flight", 1)
    eng.candidates_run_mixed(20000, rc_from=nf, first=30000)
    assert eng.candidates_fetch(len(cands))[30000:50000].tobytes() == whole[30000:50000].tobytes()
    assert not eng.last_run_stats()["critical_lane"]
    eng.close()


@pytest.mark.parametrize("scoring", [(1, -1, -1, -1), (2, -3, -5, -2), (3, -2, -4, -2)], ids=["linear", "affine", "affine-mismatch-is-extend"])
@pytest.mark.parametrize("band", [24, 0])
def test_a_narrow_band_runs_tiles_again_and_changes_nothing(monkeypatch, oracle, band, scoring):
    """GACT_HIP_BAND=24: walks leave the stored band in a per cent or two of the tiles, which are run again with their whole
    window stored; 0: no band at all.  Records against the oracle: both layouts of the linear-gap main launch, and the
    drifted affine pass (split layout; both forms of its diagonal constant)."""
    from gact_amd import engine, synth
    monkeypatch.setenv("GACT_HIP_BAND", str(band))
    rs = synth.simulate_reads(150000, n_reads=110, seed=51, mean_len=8000, sd_len=2500, min_len=1500, max_len=15000)
    cf, cr = synth.synth_candidates(rs, seed=52, min_overlap=400, false_frac=0.1)
    linear = scoring[1] == scoring[2] == scoring[3]
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    wf, _ = oracle.gact_many(cat, offs, cat, offs, cf, complement=False, same_file=True, scoring=scoring, n_threads=8)
    wr, _ = oracle.gact_many(cat, offs, rcat, roffs, cr, complement=True, same_file=True, scoring=scoring, n_threads=8)
    want = np.concatenate([wf, wr])
    cands = np.concatenate([cf, cr])
    for force in ("GACT_HIP_NO_WIDE", "GACT_HIP_FORCE_WIDE") if linear else ("GACT_HIP_NO_WIDE",):
        monkeypatch.delenv("GACT_HIP_NO_WIDE", raising=False)
        monkeypatch.delenv("GACT_HIP_FORCE_WIDE", raising=False)
        monkeypatch.setenv(force, "1")
        eng = engine.Engine(scoring=scoring)
        _load(eng, rs)
        eng.candidates_upload(cands)
        eng.candidates_run_mixed(len(cands), rc_from=len(cf))
        got = eng.candidates_fetch(len(cands)).copy()
        st = eng.last_run_stats()
        assert st["linear_gap"] == linear and st["affine_drift"] == (not linear)
        assert st["layout"] == ("packed16-split" if force == "GACT_HIP_NO_WIDE" else "packed16-wide")
        assert (st["band_redos"] > 0) == (band == 24), st
        _same(got, want)
        eng.close()
