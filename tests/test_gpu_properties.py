"""GPU suite: size-independent properties of the chain engine on a workload too large to run through the oracle
in a unit test (1/20-scale E.coli-shape, ~5,000 D-SOFT candidates, ~1.6e10 cells): what the multi-GPU deal and
the persistent scheduling rely on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from gact_amd import engine, workload
    blk = workload.make_block("ecoli10x_small")
    eng = engine.Engine()
    cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs)
    eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    cands = np.concatenate([blk.cf, blk.cr])
    nf = len(blk.cf)
    eng.candidates_upload(cands)
    eng.candidates_run_mixed(len(cands), rc_from=nf)
    base = eng.candidates_fetch(len(cands))
    yield eng, blk, cands, nf, base
    eng.close()


def test_workload_is_substantial(setup):
    eng, blk, cands, nf, base = setup
    assert len(cands) > 2000 and base["cells"].sum() > 5e9
    assert base["emitted"].sum() > 0.5 * len(cands)
    # every executed tile is at most tile_size^2 cells and every chain ran at least one tile
    assert (base["cells"] <= base["n_tiles"].astype(np.int64) * 320 * 320).all()
    assert (base["n_tiles"] >= 1).all()
    # extents stay inside the reads and are ordered
    rl = np.array([len(r) for r in blk.rs.reads])
    assert (base["ab"] >= 0).all() and (base["ab"] <= base["ae"]).all() and (base["ae"] <= rl[base["ref_id"]]).all()
    assert (base["bb"] >= 0).all() and (base["bb"] <= base["be"]).all() and (base["be"] <= rl[base["query_id"]]).all()


def test_deterministic(setup):
    """persistent scheduling (atomic queues, longest-first buckets) must not leak into the results"""
    eng, blk, cands, nf, base = setup
    for _ in range(3):
        eng.candidates_run_mixed(len(cands), rc_from=nf)
        assert eng.candidates_fetch(len(cands)).tobytes() == base.tobytes()


def test_order_invariance(setup):
    """a candidate's record does not depend on where it sits in the list"""
    from gact_amd import engine
    eng, blk, cands, nf, base = setup
    rng = np.random.default_rng(3)
    for comp, sl in ((False, slice(0, nf)), (True, slice(nf, len(cands)))):
        part = cands[sl]
        perm = rng.permutation(len(part))
        got = eng.extend(part[perm], complement=comp, slot=0)
        assert got.tobytes() == base[sl][perm].tobytes()
    eng.candidates_upload(cands)     # restore the module fixture's upload


def test_shard_invariance(setup):
    """dealing the list round-robin over N ranks and re-interleaving the records equals the single run"""
    from gact_amd import dist as gdist
    eng, blk, cands, nf, base = setup
    for world in (2, 8):
        parts_f, parts_r = [], []
        for r in range(world):
            cf, cr = gdist.deal(blk.cf, r, world), gdist.deal(blk.cr, r, world)
            eng.candidates_upload(np.concatenate([cf, cr]))
            eng.candidates_run_mixed(len(cf) + len(cr), rc_from=len(cf))
            rec = eng.candidates_fetch(len(cf) + len(cr))
            parts_f.append(rec[:len(cf)]); parts_r.append(rec[len(cf):])
        assert gdist.undeal(parts_f, nf).tobytes() == base[:nf].tobytes()
        assert gdist.undeal(parts_r, len(cands) - nf).tobytes() == base[nf:].tobytes()
    eng.candidates_upload(cands)


def test_kernel_families_agree_at_size(setup, monkeypatch):
    """int32 kernel, packed-uniform and packed-split produce identical records on the whole workload"""
    from gact_amd import engine
    eng, blk, cands, nf, base = setup
    cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
    for var in ("GACT_HIP_FORCE_INT32", "GACT_HIP_FORCE_UNIFORM"):
        monkeypatch.setenv(var, "1")
        e2 = engine.Engine()
        monkeypatch.delenv(var)
        e2.upload(engine.SET_REF, cat, offs); e2.upload(engine.SET_QUERY, cat, offs)
        e2.upload(engine.SET_QUERY_RC, rcat, roffs)
        e2.candidates_upload(cands)
        e2.candidates_run_mixed(len(cands), rc_from=nf)
        got = e2.candidates_fetch(len(cands))
        assert e2.last_run_stats()["layout"] != "packed16-split"
        assert got.tobytes() == base.tobytes()
        e2.close()


def test_revcomp_on_device_equals_uploaded_set(oracle):
    """gact_hip_derive_revcomp (darwin.cpp:110-147 on the device): extending the reverse-complement candidates
    against the derived set gives the records of the uploaded host-made set, with N and lower case in the reads;
    a byte outside acgtnACGTN is refused like the reference's 'Bad Nt char'"""
    from gact_amd import engine, synth
    rs = synth.simulate_reads(20000, n_reads=16, seed=91, mean_len=4000, sd_len=1500, min_len=50, max_len=9000,
                              n_frac=0.003)
    reads = [np.array(r) for r in rs.reads]
    reads[2][10:40] = np.frombuffer(bytes(reads[2][10:40]).lower(), dtype=np.uint8)
    _, cr = synth.synth_candidates(rs, seed=92, min_overlap=300)
    assert len(cr) > 20
    recs, stats = [], []
    for derive in (False, True):
        eng = engine.Engine()
        eng.upload_seqs(engine.SET_REF, reads)
        eng.upload_seqs(engine.SET_QUERY, reads)
        if derive:
            eng.derive_revcomp()
        else:
            eng.upload_seqs(engine.SET_QUERY_RC, [synth.revcomp(r) for r in reads])
        recs.append(eng.extend(cr, complement=True))
        stats.append(eng.last_run_stats())
        eng.close()
    # ... and both are the oracle's records (so that a difference names the run that is wrong)
    cat = np.concatenate(reads)
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    want, _ = oracle.gact_many(cat, offs, np.concatenate([synth.revcomp(r) for r in reads]), offs, cr, complement=True,
                               same_file=True, n_threads=8)
    for name in ("ab", "ae", "bb", "be", "score", "emitted", "first_tile_score", "n_tiles", "cells"):
        for which, got in enumerate(recs):
            if not np.array_equal(got[name], want[name]):
                k = int(np.flatnonzero(got[name] != want[name])[0])
                raise AssertionError("%s set: field %s of candidate %d %s:\n hip    %s\n oracle %s\n other  %s\n stats %s" %
                                     ("derived" if which else "uploaded", name, k, cr[k], got[k], want[k],
                                      recs[1 - which][k], stats[which]))
    for name in recs[0].dtype.names:
        assert np.array_equal(recs[0][name], recs[1][name]), name
    eng = engine.Engine()
    eng.upload_seqs(engine.SET_QUERY, [np.frombuffer(b"ACGTXACGT", dtype=np.uint8)])
    with pytest.raises(engine.GactHipError):
        eng.derive_revcomp()
    eng.close()
