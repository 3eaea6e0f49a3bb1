"""GPU parity for tile sizes 513 .. 2048 (align.h:19 / align.cpp:66-67 accept tile_size < 2049; every caller of the
reference uses 320).  Beyond 512 the engine runs the one-wave-per-tile kernels of csrc/gact_big.hpp: same entry
points, same results -- tiles against the oracle's AlignWithBT and the compiled reference, chains against the
oracle's GACT and the reference's printed lines."""
import numpy as np
import pytest

from tilecases import random_tiles, related_pair

pytestmark = pytest.mark.gpu

FIELDS = ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted", "first_tile_score", "n_tiles",
          "cells")


def _check_tiles(eng, oracle, cases, scoring, early):
    from gact_amd import engine
    res, states = eng.align_tiles_inline([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases],
                                         [c[3] for c in cases])
    for t, (a, b, rev, first) in enumerate(cases):
        want = oracle.align_with_bt(a, b, scoring, rev, first, early)
        got = engine.queue_from_tile(res[t], states[t], first)
        assert got == want, "tile %d R=%d Q=%d rev=%d first=%d scoring=%s\n got %s\nwant %s" % (
            t, len(a), len(b), rev, first, scoring, got[:12], want[:12])


def _big_cases(seed, n, tile):
    """full tiles, ragged ones, lengths around the lane boundaries of both column widths, N and lower case"""
    rng = np.random.default_rng(seed)
    out = random_tiles(seed, 12, max_len=tile, with_n=True)
    special = [1, 15, 16, 17, 31, 32, 33, 511, 512, 513, 1023, 1024, 1025, 2047, 2048]
    for k in range(n):
        R = int(rng.choice(special)) if k % 3 == 0 else (tile if k % 3 == 1 else int(rng.integers(1, tile + 1)))
        Q = int(rng.choice(special)) if k % 4 == 0 else (tile if k % 3 == 1 else int(rng.integers(1, tile + 1)))
        R, Q = min(R, tile), min(Q, tile)
        if k % 5 == 4:
            a = b"A" * R; b = (b"A" * Q) if k % 2 else (b"C" * Q)
        else:
            a, b = related_pair(rng, R, Q, err=[0.15, 0.05, 0.3, 0.0][k % 4])
        out.append((a, b, int(k // 2 % 2), int(k % 2)))
    return out


@pytest.mark.parametrize("tile,overlap,scoring", [(513, 120, (1, -1, -1, -1)), (640, 0, (2, -3, -5, -2)),
                                                  (1024, 200, (1, -1, -1, -1)), (1025, 1024, (5, -4, -10, -1)),
                                                  (1500, 300, (1, -1, -2, -1)), (2048, 120, (1, -1, -1, -1)),
                                                  (2048, 1000, (3, -2, -1, -4))])
def test_big_tiles_equal_the_oracle(oracle, tile, overlap, scoring):
    from gact_amd import engine
    eng = engine.Engine(tile_size=tile, tile_overlap=overlap, scoring=scoring)
    _check_tiles(eng, oracle, _big_cases(1000 + tile + overlap, 28, tile), scoring, tile - overlap)
    eng.close()


def test_big_tiles_equal_the_compiled_reference(oracle):
    """a handful of tiles straight against align.cpp compiled unchanged (oracle/_ref), where it travelled along"""
    import oracle_py
    if not oracle_py.ref_available():
        pytest.skip("oracle/_ref/libdarwin_ref.so not present")
    from gact_amd import engine
    ref = oracle_py.RefLib()
    eng = engine.Engine(tile_size=2048, tile_overlap=128)
    cases = _big_cases(77, 8, 2048)
    res, states = eng.align_tiles_inline([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], [c[3] for c in cases])
    for t, (a, b, rev, first) in enumerate(cases):
        want = ref.align_with_bt(a, b, (1, -1, -1, -1), rev, first, 2048 - 128)
        assert engine.queue_from_tile(res[t], states[t], first) == want, t
    eng.close()


@pytest.mark.parametrize("tile,overlap,scoring", [(600, 120, (1, -1, -1, -1)), (1024, 128, (2, -3, -5, -2)),
                                                  (2048, 256, (1, -1, -1, -1))])
def test_big_tile_chains_equal_the_oracle(oracle, tile, overlap, scoring):
    """whole candidates (both strands, one read with N and lower case): gact.cpp:48-228 with big tiles"""
    from gact_amd import engine, synth
    rs = synth.simulate_reads(40000, n_reads=20, seed=tile, mean_len=9000, sd_len=2500, min_len=1500, max_len=16000)
    r = rs.reads[5]
    r[700:730] = ord("N")
    r[1500:1560] = np.frombuffer(bytes(r[1500:1560]).lower(), dtype=np.uint8)
    cf, cr = synth.synth_candidates(rs, seed=tile + 1, min_overlap=500, false_frac=0.2)
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    eng = engine.Engine(tile_size=tile, tile_overlap=overlap, scoring=scoring)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    cands = np.concatenate([cf, cr])
    assert len(cands) > 60
    eng.candidates_upload(cands)
    eng.candidates_run_mixed(len(cands), rc_from=len(cf))
    got = eng.candidates_fetch(len(cands)).copy()
    eng.candidates_run_mixed(len(cands), rc_from=len(cf))                 # again: same records
    assert eng.candidates_fetch(len(cands)).tobytes() == got.tobytes()
    eng.close()
    kw = dict(same_file=True, tile_size=tile, tile_overlap=overlap, scoring=scoring, n_threads=8)
    wf, _ = oracle.gact_many(cat, offs, cat, offs, cf, complement=False, **kw)
    wr, _ = oracle.gact_many(cat, offs, rcat, roffs, cr, complement=True, **kw)
    want = np.concatenate([wf, wr])
    for f in FIELDS:
        if not np.array_equal(got[f], want[f]):
            k = int(np.flatnonzero(got[f] != want[f])[0])
            raise AssertionError("field %s of candidate %d %s:\n hip    %s\n oracle %s" % (f, k, cands[k], got[k], want[k]))
    assert int(got["emitted"].sum()) > 20 and int(got["n_tiles"].max()) >= 3


def test_big_tile_chain_lines_equal_the_reference(oracle):
    """the printed line of the reference's own GACT (gact.cpp compiled unchanged) for a few candidates at tile 1024"""
    import oracle_py
    if not oracle_py.ref_available():
        pytest.skip("oracle/_ref/libdarwin_ref.so not present")
    from gact_amd import engine, synth
    rs = synth.simulate_reads(20000, n_reads=8, seed=99, mean_len=6000, sd_len=1500, min_len=2000, max_len=9000)
    cf, _ = synth.synth_candidates(rs, seed=100, min_overlap=800)
    cf = cf[:12]
    cat, offs = rs.concat()
    eng = engine.Engine(tile_size=1024, tile_overlap=128)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs)
    got = eng.extend(cf, complement=False, same_file=True)
    eng.close()
    ref = oracle_py.RefLib()
    orc = oracle_py.Oracle()
    n = 0
    for k, c in enumerate(cf):
        line = ref.gact_line(rs.reads[c["ref_id"]].tobytes(), rs.reads[c["query_id"]].tobytes(), int(c["ref_pos"]), int(c["query_pos"]),
                             tile_size=1024, tile_overlap=128, ref_id=int(c["ref_id"]), query_id=int(c["query_id"]), ref_name="r",
                             query_name="q")
        if got[k]["emitted"]:
            assert line == orc.format_line(got[k], "r", "q"), k
            n += 1
        else:
            assert not line
    assert n >= 4


def test_the_align_h_and_gact_h_shims_take_big_tiles(oracle, tmp_path):
    """AlignWithBT of host/align.h with tiles beyond 512 and GACT of host/gact.h at tile_size 1024, through
    darwin_hip --selftest (T and G lines, tests/test_gpu_shim.py)"""
    import os
    import subprocess
    from gact_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "darwin-gpu_amd", "host", "darwin_hip")
    rng = np.random.default_rng(4)
    lines, want_t = [], []
    for (R, Q, rev, first, early) in ((1500, 1400, 0, 1, 1300), (600, 2048, 1, 0, 1900), (2048, 2048, 0, 0, 1928), (513, 100, 1, 1, 400)):
        a, b = related_pair(rng, R, Q)
        lines.append("T %s %s 1 -1 -1 -1 %d %d %d" % (a.decode(), b.decode(), rev, first, early))
        want_t.append(oracle.align_with_bt(a, b, (1, -1, -1, -1), bool(rev), bool(first), early))
    rs = synth.simulate_reads(12000, n_reads=4, seed=5, mean_len=5000, sd_len=800, min_len=3000, max_len=7000)
    cf, _ = synth.synth_candidates(rs, seed=6, min_overlap=800)
    want_g = []
    for c in cf[:4]:
        a, b = rs.reads[c["ref_id"]].tobytes(), rs.reads[c["query_id"]].tobytes()
        lines.append("G %s %s %d %d 1024 128 35 1 -1 -1 -1 0" % (a.decode(), b.decode(), c["ref_pos"], c["query_pos"]))
        ov, _ = oracle.gact(a, b, int(c["ref_pos"]), int(c["query_pos"]), tile_size=1024, tile_overlap=128, ref_id=0, query_id=1,
                            same_file=False)
        want_g.append("GACT " + (oracle.format_line(ov, "refname", "queryname").strip() if ov.emitted else ""))
    f = tmp_path / "cases.txt"
    f.write_text("\n".join(lines) + "\n")
    out = subprocess.run([exe, "--selftest", str(f)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got_t = [[int(x) for x in l.split()[1:]] for l in out.stdout.splitlines() if l.startswith("AlignWithBT")]
    assert got_t == want_t
    got_g = [l.strip() for l in out.stdout.splitlines() if l.startswith("GACT ") or l.strip() == "GACT"]
    assert got_g == [w.strip() for w in want_g]
