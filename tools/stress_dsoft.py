"""Randomised parity sweep of the device D-SOFT filter against the host restatement (host/dsoft.cpp through the
driver's --dsoft-only mode): random filter parameters and read sets.  python tools/stress_dsoft.py [n] [seed]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, synth

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
drv = engine.driver_path()
t0 = time.time()
total = 0
for it in range(n_cfg):
    k = int(rng.integers(8, 16))
    w = int(rng.integers(1, min(k, 12)))
    p = dict(seed_size=k, window_size=w, bin_size=int(rng.choice([16, 32, 48, 64, 100, 128])),
             threshold=int(rng.integers(k, 3 * k)), num_seeds=int(rng.choice([20, 100, 400, 800, 2000])),
             seed_occurence_multiple=int(rng.choice([1, 4, 32, 64])))
    rs = synth.simulate_reads(int(rng.integers(20000, 150000)), coverage=int(rng.integers(3, 9)), seed=int(rng.integers(1 << 30)),
                              mean_len=int(rng.integers(1500, 7000)), sd_len=1500, min_len=5, max_len=15000,
                              n_frac=float(rng.choice([0.0, 0.003])))
    reads = [np.array(r) for r in rs.reads]
    if rng.random() < 0.5 and len(reads) > 2 and len(reads[1]) > 900:
        reads[1][200:800] = ord("ACGT"[int(rng.integers(4))])           # a homopolymer run
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "reads.fasta"), "w") as f:
            for i, r in enumerate(reads):
                f.write(">r%d\n%s\n" % (i, bytes(r).decode()))
        with open(os.path.join(d, "params.cfg"), "w") as f:
            f.write("[DSOFT_params]\nseed_size = %d\nbin_size = %d\nwindow_size = %d\nthreshold = %d\nnum_seeds = %d\n"
                    "seed_occurence_multiple = %d\nmax_candidates = 100000000\n" %
                    (p["seed_size"], p["bin_size"], p["window_size"], p["threshold"], p["num_seeds"],
                     p["seed_occurence_multiple"]))
        subprocess.check_call([drv, "reads.fasta", "reads.fasta", "8", "--dsoft-only", "--dump-candidates", "c.bin"],
                              cwd=d, stdout=subprocess.DEVNULL)
        raw = np.fromfile(os.path.join(d, "c.bin"), dtype=np.int32).reshape(-1, 5)
    want = np.concatenate([raw[raw[:, 4] == 0][:, :4], raw[raw[:, 4] == 1][:, :4]])
    eng = engine.Engine()
    eng.upload_seqs(engine.SET_REF, reads)
    eng.upload_seqs(engine.SET_QUERY, reads)
    eng.derive_revcomp()
    eng.dsoft_build(engine.DsoftParams(max_candidates=100000000, **p))
    nf, nr, _ = eng.dsoft_query(0, len(reads))
    got = eng.candidates_download(nf + nr)
    eng.close()
    g = np.stack([got["ref_id"], got["query_id"], got["ref_pos"], got["query_pos"]], axis=1)
    if nf != int((raw[:, 4] == 0).sum()) or not np.array_equal(g, want):
        print("MISMATCH config %d: %s, %d reads; device %d+%d candidates, host %d" % (it, p, len(reads), nf, nr, len(want)))
        sys.exit(1)
    total += len(want)
    if it % 10 == 9:
        print("config %d/%d ok, %d candidates so far, %.0f s" % (it + 1, n_cfg, total, time.time() - t0), flush=True)
print("stress dsoft: %d configurations, %d candidates, device lists equal to the host restatement's" % (n_cfg, total))
