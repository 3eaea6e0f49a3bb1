"""GPU parity of the device D-SOFT filter (gact_hip_dsoft_build / _query): candidate lists equal, element by
element and in order, to (1) the committed list the reference's own SeedPosTable::DSOFT produced
(tests/golden/dsoft.json), (2) the reference itself where oracle/_ref was built, (3) the host restatement
(host/dsoft.cpp through the driver's --dsoft-only mode) at the reference's default parameters, up to the full
bench workload; and the filter -> GACT hand-over on the device against the oracle."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def parse_fasta(text):
    names, reads = [], []
    for line in text.splitlines():
        if line.startswith(">"):
            names.append(line[1:])
            reads.append([])
        elif line:
            reads[-1].append(line)
    return names, [np.frombuffer("".join(r).encode(), dtype=np.uint8) for r in reads]


def device_candidates(reads, first=0, n=None, engine_kw=None, **dsoft_kw):
    """[(ref_id, query_id, ref_pos, query_pos, comp)], forward strand of all queries first"""
    from gact_amd import engine, synth
    eng = engine.Engine(**(engine_kw or {}))
    eng.upload_seqs(engine.SET_REF, reads)
    eng.upload_seqs(engine.SET_QUERY, reads)
    eng.upload_seqs(engine.SET_QUERY_RC, [synth.revcomp(r) for r in reads])
    info = eng.dsoft_build(engine.DsoftParams(**dsoft_kw))
    n = len(reads) - first if n is None else n
    nf, nr, ms = eng.dsoft_query(first, n)
    c = eng.candidates_download(nf + nr)
    eng.close()
    out = [(int(x["ref_id"]), int(x["query_id"]), int(x["ref_pos"]), int(x["query_pos"]), int(k >= nf))
           for k, x in enumerate(c)]
    return out, info


def strand_major(per_read_order):
    """the reference emits per read: forward, then reverse complement; the device lists all forward first"""
    return [c for c in per_read_order if c[4] == 0] + [c for c in per_read_order if c[4] == 1]


def host_candidates(tmp_path, reads, names, seed_size=14, threads=8, **cfg):
    from gact_amd import engine
    drv = engine.driver_path()
    lines = []
    for name, r in zip(names, reads):
        lines.append(">" + name)
        b = bytes(r).decode()
        lines += [b[k:k + 70] for k in range(0, len(b), 70)]
    (tmp_path / "reads.fasta").write_text("\n".join(lines) + "\n")
    p = dict(bin_size=64, window_size=4, threshold=21, num_seeds=800, seed_occurence_multiple=32)
    p.update(cfg)
    (tmp_path / "params.cfg").write_text(
        "[DSOFT_params]\nseed_size = %d\nbin_size = %d\nwindow_size = %d\nthreshold = %d\nnum_seeds = %d\n"
        "seed_occurence_multiple = %d\nmax_candidates = 1000000\n" %
        (seed_size, p["bin_size"], p["window_size"], p["threshold"], p["num_seeds"], p["seed_occurence_multiple"]))
    out = subprocess.run([drv, "reads.fasta", "reads.fasta", str(threads), "--dsoft-only", "--dump-candidates", "c.bin"],
                         capture_output=True, text=True, cwd=tmp_path, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = np.fromfile(tmp_path / "c.bin", dtype=np.int32).reshape(-1, 5)
    return [tuple(int(v) for v in row) for row in raw]


def test_dsoft_device_golden(monkeypatch):
    g = json.load(open(os.path.join(GOLD, "dsoft.json")))
    names, reads = parse_fasta(g["fasta"])
    got, info = device_candidates(reads, seed_size=g["seed_size"])
    assert got == strand_major([tuple(c) for c in g["candidates"]])
    assert info["n_minimizers"] > 0 and info["max_occurrence"] == 32
    # a staging area that is too small is re-sized from the exact counts and the filter runs again
    monkeypatch.setenv("GACT_HIP_DSOFT_TEMP_CAP", "8")
    again, _ = device_candidates(reads, seed_size=g["seed_size"])
    assert again == got


@pytest.mark.parametrize("seed_size", [11, 12])
def test_dsoft_device_equals_reference(reflib, seed_size):
    from gact_amd import synth
    rs = synth.simulate_reads(40000, n_reads=24, seed=5 + seed_size, mean_len=5000, sd_len=1500, min_len=900,
                              max_len=9000, n_frac=0.001)
    reads = [r.tobytes() for r in rs.reads]
    rc = [synth.revcomp(r).tobytes() for r in rs.reads]
    fw = reflib.dsoft_candidates(reads, reads, seed_size=seed_size)
    rv = reflib.dsoft_candidates(reads, rc, seed_size=seed_size)
    want = [(c[0], k, c[1], c[2], 0) for k in range(len(reads)) for c in fw[k]] + \
           [(c[0], k, c[1], c[2], 1) for k in range(len(reads)) for c in rv[k]]
    got, _ = device_candidates(rs.reads, seed_size=seed_size)
    assert len(want) > 100 and got == want


@pytest.mark.parametrize("cfg", [dict(), dict(seed_size=12, window_size=5, threshold=25, bin_size=128),
                                 dict(seed_size=13, window_size=1, num_seeds=100),
                                 dict(seed_size=10, window_size=9, seed_occurence_multiple=4, bin_size=48),
                                 dict(seed_size=15, window_size=6)])          # the largest table: 4 GiB
def test_dsoft_device_equals_host_restatement(tmp_path, cfg):
    """default parameters (k = 14: the 1 GiB direct table) and odd ones; reads with N runs, reads shorter than a
    window, homopolymer stretches (long runs of one window minimum)"""
    from gact_amd import synth
    rs = synth.simulate_reads(300000, coverage=8, seed=77, mean_len=6000, sd_len=2500, min_len=5, max_len=20000,
                              n_frac=0.002)
    reads = [np.array(r) for r in rs.reads]
    longish = [k for k, r in enumerate(reads) if len(r) > 2000]
    reads[longish[0]][100:700] = ord("A")             # one minimizer value for 600 positions
    reads[longish[1]][:40] = ord("N")
    # a soft-masked stretch: NtToTwoBit decodes lower case like upper case (ntcoding.cpp:57-70)
    reads[longish[2]][300:1500] = np.frombuffer(reads[longish[2]][300:1500].tobytes().lower(), dtype=np.uint8)
    reads[longish[3]][:] = np.frombuffer(reads[longish[3]].tobytes().lower(), dtype=np.uint8)
    reads.append(np.frombuffer(b"ACGTACGTAC", dtype=np.uint8))                 # shorter than k + w
    reads.append(np.frombuffer(b"ACGTTGCAAGGCTTAACGGATCCA", dtype=np.uint8))
    reads.append(np.full(3000, ord("T"), dtype=np.uint8))
    names = ["r%d" % k for k in range(len(reads))]
    kw = dict(cfg)
    seed_size = kw.pop("seed_size", 14)
    want = strand_major(host_candidates(tmp_path, reads, names, seed_size=seed_size, **kw))
    got, info = device_candidates(reads, seed_size=seed_size, **kw)
    assert len(want) > 500
    assert got == want
    # a sub-range of the queries gives the matching sub-list
    sub, _ = device_candidates(reads, first=7, n=20, seed_size=seed_size, **kw)
    assert sub == [c for c in want if 7 <= c[1] < 27]


def test_dsoft_device_full_workload_then_gact(oracle):
    """BASELINE configs[1] at full size: the device filter's list equals the host restatement's (what bench.py
    runs on), and a slice of it extended straight from the device array equals the oracle's GACT"""
    from gact_amd import engine, synth, workload
    from conftest import workload_block
    blk = workload_block("ecoli10x")
    reads = blk.rs.reads
    eng = engine.Engine()
    cat, offs = blk.rs.concat()
    rcat, _ = blk.rs.concat(rc=True)
    eng.upload(engine.SET_REF, cat, offs)
    eng.upload(engine.SET_QUERY, cat, offs)
    eng.upload(engine.SET_QUERY_RC, rcat, offs)
    info = eng.dsoft_build()
    nf, nr, ms = eng.dsoft_query(0, len(reads))
    got = eng.candidates_download(nf + nr)
    assert nf == len(blk.cf) and nr == len(blk.cr)
    assert np.array_equal(got[:nf], blk.cf) and np.array_equal(got[nf:], blk.cr)
    print("dsoft device: build %.1f ms (%d minimizers), query %.1f ms, %d candidates" %
          (info["build_ms"], info["n_minimizers"], ms, nf + nr))
    # hand-over without a host round trip: a sub-range of queries -> candidates on the device -> GACT
    nf, nr, _ = eng.dsoft_query(100, 60)
    eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True)
    rec = eng.candidates_fetch(nf + nr)
    cands = eng.candidates_download(nf + nr)
    for comp, sl, qcat in ((False, slice(0, nf), cat), (True, slice(nf, nf + nr), rcat)):
        want, _ = oracle.gact_many(cat, offs, qcat, offs, cands[sl], complement=comp, same_file=True, n_threads=8)
        for name in ("ab", "ae", "bb", "be", "score", "emitted", "n_tiles", "cells"):
            assert np.array_equal(rec[sl][name], want[name]), name
    eng.close()


def test_dsoft_device_max_candidates_truncates_in_emission_order():
    """seed_pos_table.cpp:141-143: a query strand keeps its first max_candidates threshold crossings"""
    from gact_amd import synth
    rs = synth.simulate_reads(60000, coverage=12, seed=123, mean_len=5000, sd_len=1500, min_len=900, max_len=9000)
    full, _ = device_candidates(rs.reads, seed_size=12)
    cut, _ = device_candidates(rs.reads, seed_size=12, max_candidates=3)
    want = []
    for comp in (0, 1):
        for q in range(len(rs.reads)):
            want += [c for c in full if c[1] == q and c[4] == comp][:3]
    assert cut == want and len(cut) < len(full) and max(sum(1 for c in full if c[1] == q and c[4] == 0)
                                                         for q in range(len(rs.reads))) > 3


def test_dsoft_device_argument_errors():
    from gact_amd import engine
    eng = engine.Engine()
    with pytest.raises(engine.GactHipError):
        eng.dsoft_build()                                   # nothing uploaded
    eng.upload_seqs(engine.SET_REF, [np.frombuffer(b"ACGT" * 200, dtype=np.uint8)])
    with pytest.raises(engine.GactHipError):
        eng.dsoft_query(0, 1)                               # not built
    with pytest.raises(engine.GactHipError):
        eng.dsoft_build(engine.DsoftParams(seed_size=16))
    with pytest.raises(engine.GactHipError):
        eng.dsoft_build(engine.DsoftParams(max_candidates=0))
    eng.dsoft_build(engine.DsoftParams(seed_size=8))
    with pytest.raises(engine.GactHipError):
        eng.dsoft_query(0, 1)                               # query sets missing
    eng.close()
