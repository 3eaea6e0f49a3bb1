"""GPU parity of the routing by read content.  align.cpp:134 compares raw characters (N == N is a match, case
matters), which a 2-bit code cannot express, so a read set that holds such bytes is also kept as raw bytes.  The
engine decides per CANDIDATE: only candidates one of whose two reads holds a byte other than A/C/G/T are put off to a
second pair of launches on the raw bytes; everything else stays on the 2-bit image (and on the linear-gap pass where
the scoring allows it).  One soft-masked read in a set must not cost the whole launch its kernels."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted", "first_tile_score", "n_tiles",
          "cells")
MODES = {"auto": {"GACT_HIP_NO_WIDE": "1"}, "wide": {"GACT_HIP_FORCE_WIDE": "1"},
         "uniform": {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_FORCE_UNIFORM": "1"},
         "affine": {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_NO_LIN": "1"},
         "int32-seed": {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_FORCE_INT32_SEED": "1"},
         # one pass after the other on the slot's own stream instead of the raw-byte launches beside the 2-bit ones
         "one-after-the-other": {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_NO_SIDE_LANE": "1"}}
ALL_VARS = sorted({k for v in MODES.values() for k in v} | {"GACT_HIP_NO_ROUTING", "GACT_HIP_FORCE_INT32"})


def _dirty_reads(seed, dirty):
    """24 reads; those in `dirty` get a run of N and a soft-masked (lower-case) stretch"""
    from gact_amd import synth
    rs = synth.simulate_reads(30000, n_reads=24, seed=seed, mean_len=5000, sd_len=1500, min_len=900, max_len=9000)
    rng = np.random.default_rng(seed)
    for k in dirty:
        r = rs.reads[k]
        a = int(rng.integers(100, len(r) - 400))
        r[a:a + 25] = ord("N")
        b = int(rng.integers(100, len(r) - 400))
        r[b:b + 60] = np.frombuffer(bytes(r[b:b + 60]).lower(), dtype=np.uint8)
    return rs


def _run(rs, cf, cr, oracle, scoring=(1, -1, -1, -1)):
    from gact_amd import engine
    eng = engine.Engine(scoring=scoring)
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    cands = np.concatenate([cf, cr])
    eng.candidates_upload(cands)
    eng.candidates_run_mixed(len(cands), rc_from=len(cf))
    got = eng.candidates_fetch(len(cands)).copy()
    st = eng.last_run_stats()
    # a second run on the same engine: queues, counters and the list of the put-off start from scratch again
    eng.candidates_run_mixed(len(cands), rc_from=len(cf))
    assert eng.candidates_fetch(len(cands)).tobytes() == got.tobytes()
    eng.close()
    wf, _ = oracle.gact_many(cat, offs, cat, offs, cf, complement=False, same_file=True, scoring=scoring, n_threads=8)
    wr, _ = oracle.gact_many(cat, offs, rcat, roffs, cr, complement=True, same_file=True, scoring=scoring, n_threads=8)
    want = np.concatenate([wf, wr])
    for f in FIELDS:
        if not np.array_equal(got[f], want[f]):
            k = int(np.flatnonzero(got[f] != want[f])[0])
            raise AssertionError("field %s of candidate %d %s:\n hip    %s\n oracle %s\n stats %s" % (f, k, cands[k], got[k], want[k], st))
    return st


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("dirty", [(3,), (0, 7, 19), tuple(range(24))], ids=["one-read", "three-reads", "every-read"])
def test_only_candidates_with_a_dirty_read_leave_the_two_bit_kernels(oracle, monkeypatch, mode, dirty):
    from gact_amd import synth
    for var in ALL_VARS:
        monkeypatch.delenv(var, raising=False)
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    rs = _dirty_reads(501, dirty)
    cf, cr = synth.synth_candidates(rs, seed=502, min_overlap=300, false_frac=0.15)
    cands = np.concatenate([cf, cr])
    touched = int((np.isin(cands["ref_id"], dirty) | np.isin(cands["query_id"], dirty)).sum())
    assert 0 < touched <= len(cands) and len(cands) > 100
    for scoring in ((1, -1, -1, -1), (2, -3, -5, -2)):
        st = _run(rs, cf, cr, oracle, scoring=scoring)
        assert st["raw_candidates"] == touched
        # the launch of the clean candidates kept its kernels: the linear-gap pass where the scoring is linear
        linear = scoring[1] == scoring[2] == scoring[3]
        # (the uniform layout has no linear-gap pass; with every read dirty there is no 2-bit launch to report on)
        assert st["linear_gap"] == (linear and mode not in ("affine", "uniform") and touched < len(cands))
        assert st["layout"] == {"auto": "packed16-split", "wide": "packed16-wide", "uniform": "packed16-uniform",
                                "affine": "packed16-split", "int32-seed": "packed16-split",
                                "one-after-the-other": "packed16-split"}[mode]


def test_routing_switched_off_and_int32(oracle, monkeypatch):
    """GACT_HIP_NO_ROUTING: the whole launch on the raw-byte kernels as before; the int32 kernel compares raw bytes itself"""
    from gact_amd import synth
    rs = _dirty_reads(601, (2, 11))
    cf, cr = synth.synth_candidates(rs, seed=602, min_overlap=300)
    for var in ALL_VARS:
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("GACT_HIP_NO_WIDE", "1")
    monkeypatch.setenv("GACT_HIP_NO_ROUTING", "1")
    st = _run(rs, cf, cr, oracle)
    assert st["raw_candidates"] == 0 and not st["linear_gap"] and st["packed16"]
    monkeypatch.delenv("GACT_HIP_NO_ROUTING")
    monkeypatch.setenv("GACT_HIP_FORCE_INT32", "1")
    st = _run(rs, cf, cr, oracle)
    assert st["raw_candidates"] == 0 and not st["packed16"]


def test_clean_sets_take_no_second_pass(oracle, monkeypatch):
    from gact_amd import synth
    for var in ALL_VARS:
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("GACT_HIP_NO_WIDE", "1")
    rs = _dirty_reads(701, ())
    cf, cr = synth.synth_candidates(rs, seed=702, min_overlap=300)
    st = _run(rs, cf, cr, oracle)
    assert st["raw_candidates"] == 0 and st["linear_gap"]
