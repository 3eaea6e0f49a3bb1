"""Diagnostic: the scenario of tests/test_gpu_properties.py::test_revcomp_on_device_equals_uploaded_set, repeated:
are two runs of the raw-byte kernels on the same input identical, and identical to the oracle?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_py
from gact_amd import engine, synth

rs = synth.simulate_reads(20000, n_reads=16, seed=91, mean_len=4000, sd_len=1500, min_len=50, max_len=9000, n_frac=0.003)
reads = [np.array(r) for r in rs.reads]
reads[2][10:40] = np.frombuffer(bytes(reads[2][10:40]).lower(), dtype=np.uint8)
_, cr = synth.synth_candidates(rs, seed=92, min_overlap=300)
rc = [synth.revcomp(r) for r in reads]
cat = np.concatenate(reads); offs = np.zeros(len(reads) + 1, np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
rcat = np.concatenate(rc)
orc = oracle_py.Oracle()
want, _ = orc.gact_many(cat, offs, rcat, offs, cr, complement=True, same_file=True, n_threads=8)
FIELDS = ("ab", "ae", "bb", "be", "score", "emitted", "first_tile_score", "n_tiles", "cells")
bad = 0
# other work in between, so that freshly allocated device memory holds what other kernels left there
prs = synth.simulate_reads(30000, n_reads=24, seed=5, mean_len=5000, sd_len=1500, min_len=500, max_len=9000)
pcf, pcr = synth.synth_candidates(prs, seed=6, min_overlap=300)
pcat, poffs = prs.concat(); prcat, _ = prs.concat(rc=True)
prng = np.random.default_rng(1)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    if len(sys.argv) > 2:
        tile = int(prng.choice([128, 256, 320, 320, 384]))
        pe = engine.Engine(tile_size=tile, tile_overlap=int(prng.integers(16, tile // 2)),
                           scoring=[(1, -1, -1, -1), (2, -3, -5, -2), (1, -1, -2, -1)][rep % 3])
        pe.upload(engine.SET_REF, pcat, poffs); pe.upload(engine.SET_QUERY, pcat, poffs); pe.upload(engine.SET_QUERY_RC, prcat, poffs)
        pe.extend(pcf); pe.extend(pcr, complement=True)
        pe.close()
    for derive in (False, True):
        eng = engine.Engine()
        eng.upload_seqs(engine.SET_REF, reads)
        eng.upload_seqs(engine.SET_QUERY, reads)
        if derive:
            eng.derive_revcomp()
        else:
            eng.upload_seqs(engine.SET_QUERY_RC, rc)
        got = eng.extend(cr, complement=True)
        st = eng.last_run_stats()
        eng.close()
        for f in FIELDS:
            if not np.array_equal(got[f], want[f]):
                k = int(np.flatnonzero(got[f] != want[f])[0])
                print("rep %d derive %d: field %s differs at candidate %d of %d: hip %s oracle %s (layout %s seed %s)" %
                      (rep, derive, f, k, len(cr), got[k], want[k], st["layout"], st["seed_layout"]))
                bad += 1
                break
print("runs with a difference: %d" % bad)
