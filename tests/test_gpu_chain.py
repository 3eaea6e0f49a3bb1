"""GPU parity, candidate level: the persistent HIP chain kernel vs the oracle's
GACT restatement -- offsets, score, emit flag, tile and cell counts, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted",
          "first_tile_score", "n_tiles", "cells")


def _compare(got, want, tag=""):
    for name in FIELDS:
        if not np.array_equal(got[name], want[name]):
            bad = int(np.flatnonzero(got[name] != want[name])[0])
            raise AssertionError("%s field %s differs at candidate %d:\n hip=%s\n ora=%s" %
                                 (tag, name, bad, got[bad], want[bad]))


def _run(rs, cf, cr, oracle, scoring=(1, -1, -1, -1), tile=320, overlap=120, thr=35, same_file=True):
    from gact_amd import engine
    eng = engine.Engine(tile_size=tile, tile_overlap=overlap, scoring=scoring, threshold=thr)
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    eng.upload(engine.SET_REF, cat, offs)
    eng.upload(engine.SET_QUERY, cat, offs)
    eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    total = 0
    for comp, cands, qcat, qoffs in ((False, cf, cat, offs), (True, cr, rcat, roffs)):
        if len(cands) == 0:
            continue
        got = eng.extend(cands, complement=comp, same_file=same_file)
        want, _ = oracle.gact_many(cat, offs, qcat, qoffs, cands, complement=comp, same_file=same_file,
                                   tile_size=tile, tile_overlap=overlap, threshold=thr, scoring=scoring,
                                   n_threads=8)
        _compare(got, want, "comp=%d" % comp)
        total += len(cands)
    eng.close()
    return total


@pytest.fixture(params=["packed16", "packed16-affine", "packed16-plain", "packed16-uniform", "packed16-wide",
                        "packed16-wide-affine", "int32-seed", "int32"], autouse=True)
def kernel_family(request, monkeypatch):
    """every chain test runs eight times: packed seed + packed main launch in its split layout (what many chains
    get where the geometry allows it; the linear-gap pass where open == extend == mismatch, else the tagged
    pointer scheme where the scoring allows it), the same with the affine pass whatever the scoring, the same with
    explicit pointer comparisons, the same in the uniform layout, the same in the wide layout (32 lanes per
    tile pair: what few chains get), the int32 seed launch in front of the packed main launch, and the int32
    kernel alone"""
    for var in ("GACT_HIP_FORCE_INT32", "GACT_HIP_FORCE_UNIFORM", "GACT_HIP_FORCE_INT32_SEED", "GACT_HIP_FORCE_WIDE",
                "GACT_HIP_NO_WIDE", "GACT_HIP_NO_TAGGED", "GACT_HIP_NO_LIN"):
        monkeypatch.delenv(var, raising=False)
    if request.param in ("packed16-affine", "packed16-wide-affine"):
        monkeypatch.setenv("GACT_HIP_NO_LIN", "1")           # tagged affine pass also for linear scorings
    if request.param == "packed16-plain":
        monkeypatch.setenv("GACT_HIP_NO_TAGGED", "1")        # split layout with explicit pointer comparisons
    if not request.param.startswith("packed16-wide"):
        monkeypatch.setenv("GACT_HIP_NO_WIDE", "1")          # the tests' candidate lists are short
    else:
        monkeypatch.setenv("GACT_HIP_FORCE_WIDE", "1")
    if request.param == "int32":
        monkeypatch.setenv("GACT_HIP_FORCE_INT32", "1")
    elif request.param == "packed16-uniform":
        monkeypatch.setenv("GACT_HIP_FORCE_UNIFORM", "1")
    elif request.param == "int32-seed":
        monkeypatch.setenv("GACT_HIP_FORCE_INT32_SEED", "1")
    return request.param


LIN_FAMILIES = ("packed16", "packed16-wide", "int32-seed")          # families whose main launch may take the linear-gap pass


def test_kernel_family_is_the_one_asked_for(kernel_family):
    from gact_amd import engine, synth
    rs = synth.simulate_reads(9000, n_reads=6, seed=2, mean_len=3000, sd_len=300, min_len=1500, max_len=4000)
    cf, _ = synth.synth_candidates(rs, seed=3, min_overlap=300)
    eng = engine.Engine()
    cat, offs = rs.concat()
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs)
    eng.extend(cf)
    st = eng.last_run_stats()
    assert st["layout"] == {"packed16": "packed16-split", "packed16-affine": "packed16-split",
                            "packed16-plain": "packed16-split", "packed16-uniform": "packed16-uniform",
                            "packed16-wide": "packed16-wide", "packed16-wide-affine": "packed16-wide",
                            "int32-seed": "packed16-split",
                            "int32": "int32"}[kernel_family]
    assert st["seed_layout"] == ("packed16" if kernel_family.startswith("packed16") else "int32")
    assert st["tagged_pointers"] == (kernel_family not in ("packed16-plain", "int32"))
    assert st["linear_gap"] == (kernel_family in LIN_FAMILIES)        # default scoring is linear: +1 / -1 / -1 / -1
    if st["packed16"]:
        assert 0 < st["handed_off"] <= len(cf) and st["seed_cells"] > 0
    eng.close()
    # scoring outside the int16-safe range must fall back to int32 by itself
    eng = engine.Engine(scoring=(100, -90, -200, -50))
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs)
    eng.extend(cf)
    assert eng.last_run_stats()["packed16"] is False
    eng.close()
    # arg-max keys of the packed seed kernel need match*(tile+2)*8 < 2^14: beyond that only the seed falls back
    eng = engine.Engine(scoring=(8, -8, -10, -4))
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs)
    eng.extend(cf)
    st = eng.last_run_stats()
    assert st["seed_layout"] == "int32" and st["packed16"] == (kernel_family != "int32")
    eng.close()


def test_chain_small(oracle):
    from gact_amd import synth
    rs = synth.simulate_reads(30000, n_reads=24, seed=5, mean_len=5000, sd_len=1500, min_len=800, max_len=9000)
    cf, cr = synth.synth_candidates(rs, seed=3, min_overlap=300)
    assert _run(rs, cf, cr, oracle) > 100


# (match, mismatch, open, extend); the last four are linear (open == extend == mismatch): the drifted pass
@pytest.mark.parametrize("scoring", [(2, -3, -5, -2), (5, -4, -10, -1), (1, -1, -2, -1), (30, -40, -70, -20),
                                     (6, -4, 0, 0), (3, 0, -2, 0),
                                     (2, -3, -3, -3), (5, -2, -2, -2), (1, 0, 0, 0), (3, -7, -7, -7)])
def test_chain_other_scoring(oracle, scoring, kernel_family):
    from gact_amd import engine, synth
    rs = synth.simulate_reads(20000, n_reads=16, seed=8, mean_len=4000, sd_len=1000, min_len=800, max_len=8000)
    cf, cr = synth.synth_candidates(rs, seed=9, min_overlap=300)
    _run(rs, cf, cr, oracle, scoring=scoring)
    if kernel_family in LIN_FAMILIES:
        # the engine took the linear pass exactly where the scoring is linear
        eng = engine.Engine(scoring=scoring)
        cat, offs = rs.concat()
        eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs)
        eng.extend(cf[:8])
        assert eng.last_run_stats()["linear_gap"] == (scoring[1] == scoring[2] == scoring[3])
        eng.close()


@pytest.mark.parametrize("tile,overlap,thr", [(320, 120, 35), (128, 32, 20), (200, 100, 35), (320, 200, 60),
                                              (320, 100, 35),     # early 220 > 208: uniform packed layout
                                              (400, 150, 35),     # tile > 320: the 32-column kernels
                                              (512, 256, 50)])
def test_chain_other_geometry(oracle, tile, overlap, thr):
    from gact_amd import synth
    rs = synth.simulate_reads(15000, n_reads=12, seed=21, mean_len=3000, sd_len=800, min_len=800, max_len=6000)
    cf, cr = synth.synth_candidates(rs, seed=22, min_overlap=300)
    _run(rs, cf, cr, oracle, tile=tile, overlap=overlap, thr=thr)


def test_chain_edge_candidates(oracle):
    """seeds at read starts/ends, pos 0, pos == len, false hits, N-containing reads"""
    from gact_amd import synth
    rs = synth.simulate_reads(9000, n_reads=10, seed=31, mean_len=2500, sd_len=700, min_len=400, max_len=5000,
                              n_frac=0.01)
    cf, cr = synth.synth_candidates(rs, seed=32, min_overlap=200, false_frac=0.3)
    extra = []
    for ri in range(rs.n):
        for qi in range(rs.n):
            L, M = len(rs.reads[ri]), len(rs.reads[qi])
            for rp, qp in ((0, 0), (L, M - 1), (L, 0), (0, M - 1), (L // 2, M // 2), (1, 1), (L - 1, M - 1)):
                extra.append((ri, qi, rp, max(qp, 0)))
    extra = np.array(extra, dtype=synth.CAND_DTYPE)
    _run(rs, np.concatenate([cf, extra]), np.concatenate([cr, extra]), oracle)
    _run(rs, np.concatenate([cf, extra]), cr, oracle, same_file=False)


def test_chain_pacbio_shape(oracle):
    """~10 kb PacBio-shape reads, the shape the metric is quoted on"""
    from gact_amd import synth
    rs = synth.simulate_reads(120000, coverage=6, seed=41)
    cf, cr = synth.synth_candidates(rs, seed=42)
    n = _run(rs, cf, cr, oracle)
    assert n > 300


def _edited_copy(rng, src, every=(120, 420), run=(8, 64), point=0.03, alphabet=b"ACGT"):
    """src with runs of bases deleted / inserted every few hundred bases (long INSERT and DELETE stretches in the
    traceback, crossing column octets, lanes and flush blocks of the pointer words) and a few point errors"""
    out, i = [], 0
    nxt = int(rng.integers(*every))
    while i < len(src):
        if i >= nxt:
            n = int(rng.integers(*run))
            if rng.random() < 0.5:
                i += n                                                   # deletion
            else:
                out.extend(rng.choice(list(alphabet), n).tolist())      # insertion
            nxt = i + int(rng.integers(*every))
            continue
        b = int(src[i])
        if rng.random() < point:
            b = int(rng.choice(list(alphabet)))
        out.append(b)
        i += 1
    return np.array(out, dtype=np.uint8)


@pytest.mark.parametrize("alphabet", [b"ACGT", b"AC"])
def test_chain_long_gap_runs(oracle, alphabet):
    """overlaps whose alignments hold insertions and deletions of 8-64 bases (and, on a two-letter alphabet, ties
    between the three moves nearly everywhere): what the flag-less walk of the linear-gap kernels has to get right
    over many consecutive INSERT / DELETE states, and the affine kernels' open/extend flags"""
    from gact_amd import synth
    rng = np.random.default_rng(20261004)
    genome = rng.choice(list(alphabet), 9000).astype(np.uint8)
    rs = synth.ReadSet()
    rs.genome = genome
    spans = [(0, 6000), (400, 6400), (1500, 8000), (2500, 9000), (100, 3000), (3000, 7000)]
    for n, (a, b) in enumerate(spans):
        rs.reads.append(_edited_copy(rng, genome[a:b], alphabet=alphabet) if n else genome[a:b].copy())
        rs.names.append("L%d_%d_%d" % (n, a, b - a))
    cands = []
    for r in range(len(spans)):
        for q in range(len(spans)):
            if r == q:
                continue
            lo, hi = max(spans[r][0], spans[q][0]), min(spans[r][1], spans[q][1])
            if hi - lo < 600:
                continue
            for g in (lo + 150, (lo + hi) // 2, hi - 150):
                # the same genome position in both reads, give or take what the edits moved
                rp = min(max(g - spans[r][0], 1), len(rs.reads[r]) - 1)
                qp = min(max(g - spans[q][0] + int(rng.integers(-40, 40)), 1), len(rs.reads[q]) - 1)
                cands.append((r, q, rp, qp))
    cf = np.array(cands, dtype=synth.CAND_DTYPE)
    n = _run(rs, cf, cf[:0], oracle)
    n += _run(rs, cf, cf[:0], oracle, scoring=(2, -3, -3, -3))
    n += _run(rs, cf, cf[:0], oracle, scoring=(1, -1, -2, -1))
    assert n == 3 * len(cf) and len(cf) > 60
