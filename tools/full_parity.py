"""One-off: every candidate of a bench workload through the HIP engine and through the oracle,
all record fields compared.  python tools/full_parity.py [workload] [dsoft|synthetic]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_py
from gact_amd import engine, workload

name = sys.argv[1] if len(sys.argv) > 1 else "ecoli10x"
src = sys.argv[2] if len(sys.argv) > 2 else "dsoft"
blk = workload.make_block(name, candidates=src)
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
eng = engine.Engine()
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
nf = len(blk.cf)
eng.candidates_upload(np.concatenate([blk.cf, blk.cr]))
eng.candidates_run_mixed(nf + len(blk.cr), rc_from=nf)
rec = eng.candidates_fetch(nf + len(blk.cr))
print("engine layout:", eng.last_run_stats()["layout"])
orc = oracle_py.Oracle()
threads = min(16, os.cpu_count() or 1)
t = time.time()
wf, _ = orc.gact_many(cat, offs, cat, offs, blk.cf, complement=False, n_threads=threads)
wr, _ = orc.gact_many(cat, offs, rcat, roffs, blk.cr, complement=True, n_threads=threads)
print("oracle %.1f s on %d threads" % (time.time() - t, threads))
want = np.concatenate([wf, wr])
bad = 0
for f in ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted", "first_tile_score", "n_tiles", "cells"):
    d = int((rec[f] != want[f]).sum())
    bad += d
    if d:
        print("field %s differs on %d candidates" % (f, d))
print("%s/%s: %d candidates, %d tiles, %d cells, emitted %d: %s" % (
    name, src, len(rec), int(rec["n_tiles"].sum()), int(rec["cells"].sum()), int(rec["emitted"].sum()),
    "BIT-EXACT" if bad == 0 else "MISMATCH"))
sys.exit(1 if bad else 0)
