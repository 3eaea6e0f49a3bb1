"""GPU: argument checks of the C-ABI that guard device memory (no kernel may see an index the host did not
check): candidate ranges against what a slot really holds, candidate positions against the set their strand
reads, the seed filter's index against the reference set it was built from."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reads(seed, lens):
    rng = np.random.default_rng(seed)
    return [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)] for n in lens]


def test_run_and_fetch_are_bounded_by_the_uploaded_count():
    from gact_amd import engine
    eng = engine.Engine()
    reads = _reads(1, [3000, 3500])
    for which in (engine.SET_REF, engine.SET_QUERY, engine.SET_QUERY_RC):
        eng.upload_seqs(which, reads)
    c = np.zeros(10, dtype=engine.CAND_DTYPE)
    c["query_id"] = 1
    c["ref_pos"] = c["query_pos"] = 1500
    eng.candidates_upload(c)
    with pytest.raises(engine.GactHipError, match="outside the 10 candidates"):
        eng.candidates_run(1000)                     # the slot's buffers are larger than what was uploaded
    with pytest.raises(engine.GactHipError):
        eng.candidates_run(4, first=8)
    with pytest.raises(engine.GactHipError):
        eng.candidates_fetch(11)
    with pytest.raises(engine.GactHipError):
        eng.candidates_download(11)
    eng.candidates_run(10)
    assert len(eng.candidates_fetch(10)) == 10
    eng.close()


def test_positions_are_checked_against_the_strand_that_is_run(oracle):
    """darwin.cpp's CPU path calls GACT(read, rev_read, ...) for a read that has reverse-complement candidates only:
    the shim uploads QUERY_RC alone, and whatever GACT_SET_QUERY still holds must not decide the range check"""
    from gact_amd import engine, synth
    eng = engine.Engine()
    ref, = _reads(2, [6000])
    stale, = _reads(3, [900])                                  # left over from an earlier call
    q = synth.revcomp(ref)
    eng.upload_seqs(engine.SET_REF, [ref])
    eng.upload_seqs(engine.SET_QUERY, [stale])
    eng.upload_seqs(engine.SET_QUERY_RC, [q])
    c = np.zeros(1, dtype=engine.CAND_DTYPE)
    c["ref_pos"], c["query_pos"] = 3000, 3000                  # beyond `stale`, inside the read this strand uses
    got = eng.extend(c, complement=True, same_file=False)
    want, _ = oracle.gact_many(ref, [0, len(ref)], q, [0, len(q)], c, complement=True, same_file=False)
    for name in ("ab", "ae", "bb", "be", "score", "emitted", "n_tiles", "cells"):
        assert got[name][0] == want[name][0], name
    # the same candidate on the forward strand lies outside its read: refused at run time, nothing launched
    eng.candidates_upload(c)
    with pytest.raises(engine.GactHipError, match="position outside its read"):
        eng.candidates_run(1, complement=False)
    # ... and a set that shrinks after the check voids it
    eng.candidates_run(1, complement=True)
    eng.upload_seqs(engine.SET_QUERY_RC, [stale])
    with pytest.raises(engine.GactHipError, match="position outside its read"):
        eng.candidates_run(1, complement=True)
    eng.close()


def test_filter_index_dies_with_its_reference_set():
    from gact_amd import engine, synth
    eng = engine.Engine()
    reads = _reads(4, [5000, 4000, 6000])
    eng.upload_seqs(engine.SET_REF, reads)
    eng.upload_seqs(engine.SET_QUERY, reads)
    eng.upload_seqs(engine.SET_QUERY_RC, [synth.revcomp(r) for r in reads])
    eng.dsoft_build(engine.DsoftParams(seed_size=10))
    eng.dsoft_query(0, 3)
    eng.upload_seqs(engine.SET_REF, reads[:1])                 # fewer, shorter: the old index would read past it
    with pytest.raises(engine.GactHipError, match="no index"):
        eng.dsoft_query(0, 3)
    eng.dsoft_build(engine.DsoftParams(seed_size=10))
    nf, nr, _ = eng.dsoft_query(0, 3)
    c = eng.candidates_download(nf + nr)
    assert (c["ref_id"] == 0).all()
    eng.close()


def test_device_filter_list_dies_with_the_sets_it_was_made_from():
    """a list made by gact_hip_dsoft_query names reads and positions of the sets of that moment: after any upload
    candidates_run* refuses it instead of launching on unchecked ids (gact_engine.hip::check_candidate_range)"""
    from gact_amd import engine, synth
    eng = engine.Engine()
    reads = _reads(5, [5000, 4000, 6000, 4500])
    rc = [synth.revcomp(r) for r in reads]
    eng.upload_seqs(engine.SET_REF, reads); eng.upload_seqs(engine.SET_QUERY, reads); eng.upload_seqs(engine.SET_QUERY_RC, rc)
    eng.dsoft_build(engine.DsoftParams(seed_size=10))
    nf, nr, _ = eng.dsoft_query(0, 4)
    assert nf + nr > 0
    eng.candidates_run_mixed(nf + nr, rc_from=nf)
    first = eng.candidates_fetch(nf + nr).copy()
    eng.upload_seqs(engine.SET_QUERY, reads[:1])               # fewer, shorter queries: the old ids are out of range
    with pytest.raises(engine.GactHipError, match="dsoft_query again"):
        eng.candidates_run_mixed(nf + nr, rc_from=nf)
    eng.upload_seqs(engine.SET_QUERY, reads)                   # same bytes again: still a different epoch
    with pytest.raises(engine.GactHipError, match="dsoft_query again"):
        eng.candidates_run_mixed(nf + nr, rc_from=nf)
    nf2, nr2, _ = eng.dsoft_query(0, 4)
    assert (nf2, nr2) == (nf, nr)
    eng.candidates_run_mixed(nf + nr, rc_from=nf)
    assert eng.candidates_fetch(nf + nr).tobytes() == first.tobytes()
    eng.close()


def test_registered_output_buffer_is_opt_in():
    """fetches go through the engine's own pinned staging area unless the caller registered the destination
    (gact_hip_register_output); a buffer the caller frees and the allocator hands out again is never DMA'd into
    behind a stale registration, because nothing is registered on the engine's own initiative"""
    from gact_amd import engine
    eng = engine.Engine()
    reads = _reads(6, [4000, 4200, 3900])
    for which in (engine.SET_REF, engine.SET_QUERY, engine.SET_QUERY_RC):
        eng.upload_seqs(which, reads)
    c = np.zeros(64, dtype=engine.CAND_DTYPE)
    c["ref_id"] = np.arange(64) % 3
    c["query_id"] = (np.arange(64) + 1) % 3
    c["ref_pos"] = c["query_pos"] = 1000 + 7 * np.arange(64)
    eng.candidates_upload(c)
    eng.candidates_run(64)
    want = eng.candidates_fetch(64).copy()
    for _ in range(20):                                       # fresh arrays: same addresses come back from the allocator
        assert eng.candidates_fetch(64, out=np.empty(64, dtype=engine.OVERLAP_DTYPE)).tobytes() == want.tobytes()
    big = np.zeros(256, dtype=engine.OVERLAP_DTYPE)
    eng.register_output(big)
    for off in (0, 10, 192):                                  # anywhere inside the registered range: direct copy
        eng.candidates_fetch(64, out=big[off:off + 64])
        assert big[off:off + 64].tobytes() == want.tobytes()
    other = np.zeros(64, dtype=engine.OVERLAP_DTYPE)          # outside it: the staging path
    assert eng.candidates_fetch(64, out=other).tobytes() == want.tobytes()
    eng.unregister_output()
    eng.candidates_fetch(64, out=big[:64])
    assert big[:64].tobytes() == want.tobytes()
    eng.register_output(big); eng.register_output(other)      # a second registration replaces the first
    eng.candidates_fetch(64, out=other)
    assert other.tobytes() == want.tobytes()
    eng.close()                                               # destroys the registration with the engine
