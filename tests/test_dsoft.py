"""CPU suite: the D-SOFT restatement (darwin-gpu_amd/host/dsoft.cpp, run through the driver with
--dsoft-only, no GPU touched) against the reference's own SeedPosTable::DSOFT (oracle/_ref), candidate
by candidate, both strands; and against the committed golden candidate list where oracle/_ref is absent."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG = ("[GACT_scoring]\nmatch = 1\nmismatch = -1\ngap_open = -1\ngap_extend = -1\n"
       "[DSOFT_params]\nseed_size = %d\nbin_size = 64\nwindow_size = 4\nthreshold = 21\nnum_seeds = 800\n"
       "seed_occurence_multiple = 32\nmax_candidates = 1000000\nnum_nz_bins = 2500000\n"
       "[GACT_first_tile]\nfirst_tile_size = 128\nfirst_tile_score_threshold = 35\n"
       "[GACT_extend]\ntile_size = 320\ntile_overlap = 120\n")


def run_dsoft(tmp_path, fasta_text, seed_size, threads=2):
    from gact_amd import engine
    drv = engine.build_driver()
    (tmp_path / "reads.fasta").write_text(fasta_text)
    (tmp_path / "params.cfg").write_text(CFG % seed_size)
    out = subprocess.run([drv, "reads.fasta", "reads.fasta", str(threads), "--dsoft-only", "--dump-candidates", "c.bin"],
                         capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    raw = open(tmp_path / "c.bin", "rb").read()
    return [struct.unpack_from("<5i", raw, k) for k in range(0, len(raw), 20)]


def fasta_of(rs):
    lines = []
    for name, r in zip(rs.names, rs.reads):
        lines.append(">" + name)
        b = r.tobytes().decode()
        lines += [b[k:k + 70] for k in range(0, len(b), 70)]
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed_size", [11, 12])
def test_dsoft_equals_reference(reflib, tmp_path, seed_size):
    from gact_amd import synth
    rs = synth.simulate_reads(40000, n_reads=24, seed=5 + seed_size, mean_len=5000, sd_len=1500, min_len=900,
                              max_len=9000, n_frac=0.001)
    got = run_dsoft(tmp_path, fasta_of(rs), seed_size)
    reads = [r.tobytes() for r in rs.reads]
    rc = [synth.revcomp(r).tobytes() for r in rs.reads]
    want = []
    fw = reflib.dsoft_candidates(reads, reads, seed_size=seed_size)
    rv = reflib.dsoft_candidates(reads, rc, seed_size=seed_size)
    for k in range(len(reads)):                      # darwin.cpp:209-288: per read, forward then reverse complement
        want += [(c[0], k, c[1], c[2], 0) for c in fw[k]]
        want += [(c[0], k, c[1], c[2], 1) for c in rv[k]]
    assert len(want) > 100
    assert got == want


def test_dsoft_golden(tmp_path):
    g = json.load(open(os.path.join(GOLD, "dsoft.json")))
    got = run_dsoft(tmp_path, g["fasta"], g["seed_size"], threads=3)
    assert [list(c) for c in got] == g["candidates"]
