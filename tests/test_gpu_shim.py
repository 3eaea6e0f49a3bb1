"""GPU parity through the reference-shaped C++ surface (darwin-gpu_amd/host/gact.h, align.h):
AlignWithBT, Align_Batch_GPU, GACT, and GPU_init -> GACT_Batch -> GPU_close via the
darwin.cpp-shaped driver, all compared with the oracle."""
import struct
import subprocess

import pytest

from tilecases import random_tiles

pytestmark = pytest.mark.gpu


def _driver():
    from gact_amd import engine
    return engine.build_driver()


def test_alignwithbt_and_align_batch_gpu(oracle, tmp_path):
    cases = [c for c in random_tiles(404, 36) if len(c[0]) and len(c[1])]
    scorings = [(1, -1, -1, -1), (2, -3, -5, -2)]
    lines = []
    meta = []
    for k, (a, b, rev, first) in enumerate(cases):
        sc = scorings[k % 2]
        early = 200 if k % 3 else 64
        lines.append("T %s %s %d %d %d %d %d %d %d" % (a.decode(), b.decode(), *sc, rev, first, early))
        meta.append((a, b, sc, rev, first, early))
    f = tmp_path / "cases.txt"
    f.write_text("\n".join(lines) + "\n")
    out = subprocess.run([_driver(), "--selftest", str(f)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = [l for l in out.stdout.splitlines() if l.startswith(("AlignWithBT", "Align_Batch_GPU"))]
    it = iter(got)
    n_batch = 0
    for a, b, sc, rev, first, early in meta:
        want = oracle.align_with_bt(a, b, sc, rev, first, early)
        l = next(it)
        assert l.startswith("AlignWithBT") and [int(x) for x in l.split()[1:]] == want
        if sc == (1, -1, -1, -1) and early == 200:
            l = next(it)
            head, st = l[len("Align_Batch_GPU"):].split(":")
            score, rsteps, qsteps, mi, mj = [int(x) for x in head.split()]
            st = [int(x) for x in st.split()]
            ws = want[3:] if first else want[1:]
            assert score == want[0] and st == ws
            if first:
                assert [mi, mj] == want[1:3]
            assert rsteps == sum(1 for s in ws if s in (2, 3)) and qsteps == sum(1 for s in ws if s in (1, 3))
            n_batch += 1
    assert n_batch > 5


def test_gact_entry_point(oracle, tmp_path):
    from gact_amd import synth
    rs = synth.simulate_reads(8000, n_reads=6, seed=61, mean_len=2500, sd_len=500, min_len=800, max_len=4000)
    cf, cr = synth.synth_candidates(rs, seed=62, min_overlap=300, false_frac=0.2)
    lines, want = [], []
    for comp, cands in ((0, cf[:6]), (1, cr[:6])):
        for c in cands:
            r = rs.reads[c["ref_id"]].tobytes()
            q = (synth.revcomp(rs.reads[c["query_id"]]) if comp else rs.reads[c["query_id"]]).tobytes()
            lines.append("G %s %s %d %d 320 120 35 1 -1 -1 -1 %d" % (r.decode(), q.decode(), c["ref_pos"], c["query_pos"], comp))
            ov, _ = oracle.gact(r, q, int(c["ref_pos"]), int(c["query_pos"]), ref_id=0, query_id=1,
                                complement=bool(comp), same_file=False)
            want.append("GACT " + (oracle.format_line(ov, "refname", "queryname").strip() if ov.emitted else ""))
    f = tmp_path / "cases.txt"
    f.write_text("\n".join(lines) + "\n")
    out = subprocess.run([_driver(), "--selftest", str(f)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = [l.strip() for l in out.stdout.splitlines() if l.startswith("GACT ") or l.strip() == "GACT"]
    assert got == [w.strip() for w in want]


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_driver_end_to_end(oracle, tmp_path, threads):
    """FASTA + params.cfg + candidates -> darwin.<t>.out, `sort | uniq` equal to the CPU path (README:25); 8 feeder
    threads is BASELINE config 2 as written (darwin.cpp:619-629: one GPU_storage / one engine slot per thread)"""
    from gact_amd import synth
    rs = synth.simulate_reads(30000, n_reads=20, seed=71, mean_len=5000, sd_len=1200, min_len=800, max_len=9000)
    cf, cr = synth.synth_candidates(rs, seed=72, min_overlap=300)
    rs.write_fasta(str(tmp_path / "reads.fasta"))
    (tmp_path / "params.cfg").write_text(
        "[GACT_scoring]\nmatch = 1\nmismatch = -1\ngap_open = -1\ngap_extend = -1\n"
        "[GACT_first_tile]\nfirst_tile_size = 128\nfirst_tile_score_threshold = 35\n"
        "[GACT_extend]\ntile_size = 320\ntile_overlap = 120\n")
    with open(tmp_path / "cands.bin", "wb") as f:
        for comp, cands in ((0, cf), (1, cr)):
            for c in cands:
                f.write(struct.pack("<5i", c["ref_id"], c["query_id"], c["ref_pos"], c["query_pos"], comp))
    out = subprocess.run([_driver(), "reads.fasta", "reads.fasta", str(threads), "--candidates", "cands.bin"],
                         capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = []
    for t in range(threads):
        got += open(tmp_path / ("darwin.%d.out" % t)).read().splitlines()
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    want = []
    for comp, cands, qc, qo in ((False, cf, cat, offs), (True, cr, rcat, roffs)):
        recs, _ = oracle.gact_many(cat, offs, qc, qo, cands, complement=comp, same_file=True, n_threads=4)
        for r in recs:
            if r["emitted"]:
                want.append(oracle.format_line(r, rs.names[r["ref_id"]], rs.names[r["query_id"]]).rstrip("\n"))
    assert sorted(set(got)) == sorted(set(want))
    assert sorted(got) == sorted(want)
    assert len(want) > 50


def test_end_to_end_from_fasta_equals_reference_program(tmp_path):
    """FASTA -> D-SOFT restatement -> HIP GACT -> darwin.<t>.out, compared with the lines the REFERENCE's own
    CPU program (oracle/_ref/darwin_cpu) printed for the same FASTA and params.cfg (tests/golden/e2e.json):
    the reference's regression method, x_scalingrun.sh:5-22."""
    import json
    import os
    from test_dsoft import CFG
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fasta = json.load(open(os.path.join(gold, "dsoft.json")))["fasta"]
    e2e = json.load(open(os.path.join(gold, "e2e.json")))
    (tmp_path / "reads.fasta").write_text(fasta)
    (tmp_path / "params.cfg").write_text(CFG % e2e["seed_size"])
    # host filter with 1 and 3 feeder threads, then the whole pipeline on the device (--device-dsoft)
    for threads, extra in ((1, []), (3, []), (1, ["--device-dsoft"]), (3, ["--device-dsoft"])):
        out = subprocess.run([_driver(), "reads.fasta", "reads.fasta", str(threads)] + extra, capture_output=True,
                             text=True, cwd=tmp_path, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        got = []
        for t in range(threads):
            got += open(tmp_path / ("darwin.%d.out" % t)).read().splitlines()
        assert sorted(got) == e2e["lines_sorted"]
        assert len(got) > 40


def test_align_batch_and_positions(oracle, tmp_path):
    """Align_Batch (align.cpp:17-54: idle entries give an empty queue) and AlignWithBT called with
    (ref_pos, query_pos) other than the tile's corner (align.cpp:179-181,186), both through the shim"""
    import random
    rng = random.Random(5150)
    cases = [c for c in random_tiles(405, 30) if len(c[0]) and len(c[1])]
    lines, want = [], []
    # B: three batches, idle entries in between
    for b in range(3):
        sc = [(1, -1, -1, -1), (2, -3, -5, -2), (1, -1, -2, -1)][b]
        early = (200, 64, 150)[b]
        entries = []
        for k in range(8):
            if k in (1 + b, 6):
                entries.append(None)
            else:
                entries.append(cases[(8 * b + k) % len(cases)])
        lines.append("B %d %d %d %d %d %d" % (len(entries), *sc, early))
        for e in entries:
            if e is None:
                lines.append("- - 0 0")
                want.append([])
            else:
                a, q, rev, first = e
                lines.append("%s %s %d %d" % (a.decode(), q.decode(), rev, first))
                want.append(oracle.align_with_bt(a, q, sc, rev, first, early))
    n_batch = len(want)
    # P: positions inside, on the border of, and outside the tile
    for k, (a, q, rev, first) in enumerate(cases):
        R, Q = len(a), len(q)
        for rp, qp in ((rng.randint(1, R), rng.randint(1, Q)), (R, rng.randint(1, Q)), (0, Q), (R, 0), (R + 1, Q),
                       (R, Q)):
            if k % 3 and (rp, qp) != (R, Q) and rp in (0, R + 1):
                continue
            sc = (1, -1, -1, -1) if k % 2 else (2, -3, -5, -2)
            lines.append("P %s %s %d %d %d %d %d %d %d %d %d" % (a.decode(), q.decode(), *sc, rev, first, 200, rp, qp))
            want.append(oracle.align_with_bt(a, q, sc, rev, first, 200, ref_pos=rp, query_pos=qp))
    f = tmp_path / "cases.txt"
    f.write_text("\n".join(lines) + "\n")
    out = subprocess.run([_driver(), "--selftest", str(f)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    got = [[int(x) for x in l.split()[1:]] for l in out.stdout.splitlines() if l.startswith(("Align_Batch ", "AlignWithBT"))
           or l.strip() == "Align_Batch"]
    assert len(got) == len(want) and n_batch == 24
    assert got == want


CFG_GACT = ("[GACT_scoring]\nmatch = 1\nmismatch = -1\ngap_open = -1\ngap_extend = -1\n"
            "[GACT_first_tile]\nfirst_tile_size = 128\nfirst_tile_score_threshold = 35\n"
            "[GACT_extend]\ntile_size = 320\ntile_overlap = 120\n")


def _write_cands(path, cf, cr):
    with open(path, "wb") as f:
        for comp, cands in ((0, cf), (1, cr)):
            for c in cands:
                f.write(struct.pack("<5i", c["ref_id"], c["query_id"], c["ref_pos"], c["query_pos"], comp))


def _lines(tmp_path, pattern):
    import glob
    got = []
    for p in sorted(glob.glob(str(tmp_path / pattern))):
        got += open(p).read().splitlines()
    return got


@pytest.mark.parametrize("with_n", [False, True])
def test_recoded_read_sets_through_gact_batch(tmp_path, with_n):
    """The reference's -DGPU build hands GACT_Batch read sets recoded in place to A0 C1 T2 G3, anything else
    left as it is (darwin.cpp:314-398).  Same lines as the ASCII run, with and without N / lower case left in."""
    import numpy as np
    from gact_amd import synth
    rs = synth.simulate_reads(30000, n_reads=20, seed=171, mean_len=5000, sd_len=1200, min_len=800, max_len=9000)
    if with_n:
        rng = np.random.default_rng(9)
        for r in rs.reads[::3]:
            r[rng.integers(0, len(r), 40)] = ord("N")
        rs.reads[1][100:140] = np.frombuffer(rs.reads[1][100:140].tobytes().lower(), dtype=np.uint8)
    cf, cr = synth.synth_candidates(rs, seed=172, min_overlap=300)
    rs.write_fasta(str(tmp_path / "reads.fasta"))
    (tmp_path / "params.cfg").write_text(CFG_GACT)
    _write_cands(tmp_path / "cands.bin", cf, cr)
    outs = []
    for extra in ([], ["--recode"]):
        out = subprocess.run([_driver(), "reads.fasta", "reads.fasta", "2", "--candidates", "cands.bin"] + extra,
                             capture_output=True, text=True, cwd=tmp_path, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        assert ("Time converting bases" in out.stdout) == bool(extra)
        for label in ("Time GACT calling", "Time elapsed (loading reads)",
                      "Time elapsed (seed table querying + aligning)"):       # darwin.cpp:441,574,639
            assert label in out.stdout
        outs.append(sorted(_lines(tmp_path, "darwin.[0-9].out")))
    assert outs[0] == outs[1] and len(outs[0]) > 40


def test_sharded_driver_union_equals_single_run(tmp_path):
    """darwin_hip --shard R/W --device 0: two ranks' files together are the single-process output"""
    from gact_amd import synth
    rs = synth.simulate_reads(30000, n_reads=20, seed=271, mean_len=5000, sd_len=1200, min_len=800, max_len=9000)
    cf, cr = synth.synth_candidates(rs, seed=272, min_overlap=300)
    rs.write_fasta(str(tmp_path / "reads.fasta"))
    (tmp_path / "params.cfg").write_text(CFG_GACT)
    _write_cands(tmp_path / "cands.bin", cf, cr)
    base = [_driver(), "reads.fasta", "reads.fasta", "2", "--candidates", "cands.bin", "--device", "0"]
    out = subprocess.run(base, capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    single = sorted(_lines(tmp_path, "darwin.[0-9].out"))
    for r in range(2):
        out = subprocess.run(base + ["--shard", "%d/2" % r], capture_output=True, text=True, cwd=tmp_path, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
    sharded = sorted(_lines(tmp_path, "darwin.[01].[0-9].out"))
    assert sharded == single and len(single) > 40


def test_rccl_gather_of_the_cpp_driver_prints_the_single_run_lines(tmp_path):
    """darwin_hip --shard 0/1 --rccl-gather ID: the rank's share as one run on the engine, then the C++ side's RCCL gather
    (gact_hip_comm_create / _gather_lines: unique id through a file, counts once around, the lines out of the engine's device
    array) and rank 0's darwin.gathered.out.  One GPU: a communicator of one rank (RCCL refuses two ranks on one device), so
    what runs here is RCCL's loading, the communicator, the all-gather of the counts, the device-side narrowing to 32-byte
    lines and the formatting; the send / receive pairs of ranks 1.. run on the driver's 8-GPU node only."""
    from gact_amd import synth
    rs = synth.simulate_reads(30000, n_reads=20, seed=271, mean_len=5000, sd_len=1200, min_len=800, max_len=9000)
    cf, cr = synth.synth_candidates(rs, seed=272, min_overlap=300)
    rs.write_fasta(str(tmp_path / "reads.fasta"))
    (tmp_path / "params.cfg").write_text(CFG_GACT)
    _write_cands(tmp_path / "cands.bin", cf, cr)
    base = [_driver(), "reads.fasta", "reads.fasta", "2", "--candidates", "cands.bin", "--device", "0"]
    out = subprocess.run(base, capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    single = sorted(_lines(tmp_path, "darwin.[0-9].out"))
    out = subprocess.run(base + ["--shard", "0/1", "--rccl-gather", "rccl.id"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gathered records per rank: %d" % (len(cf) + len(cr)) in out.stdout, out.stdout
    gathered = sorted(_lines(tmp_path, "darwin.gathered.out"))
    assert gathered == single and len(single) > 40
    assert not (tmp_path / "rccl.id").exists()                      # rank 0 takes the id file away again
    # a stale id file is somebody else's job: refused, loudly
    (tmp_path / "rccl.id").write_bytes(b"x" * 128)
    out = subprocess.run(base + ["--shard", "0/1", "--rccl-gather", "rccl.id"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert out.returncode != 0 and "exists already" in out.stdout + out.stderr
