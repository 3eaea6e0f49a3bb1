"""Where one bench step's wall time goes on the host side: launch call, kernel wait, record D2H, stats query.
Run on the GPU box: python tools/step_breakdown.py [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "darwin-gpu_amd"))
import torch  # noqa: F401  (its HIP runtime first)
import numpy as np
from gact_amd import engine, synth, workload

name = sys.argv[1] if len(sys.argv) > 1 else "ecoli10x"
blk = workload.make_block(name, block=0, candidates="dsoft")
reads = blk.rs.reads
offs = np.zeros(len(reads) + 1, dtype=np.int64)
offs[1:] = np.cumsum([len(r) for r in reads])
cat = np.concatenate(reads)
rcat = np.concatenate([synth.revcomp(r) for r in reads])
eng = engine.Engine(n_slots=1)
nf, nr = len(blk.cf), len(blk.cr)
for rep in range(2):
    t0 = time.perf_counter()
    eng.upload(engine.SET_REF, cat, offs)
    eng.upload(engine.SET_QUERY, cat, offs)
    eng.upload(engine.SET_QUERY_RC, rcat, offs)
    t1 = time.perf_counter()
    eng.candidates_upload(np.concatenate([blk.cf, blk.cr]), slot=0)
    t2 = time.perf_counter()
    print("upload: 3 read sets (%.1f MB, incl. 2-bit packing on the device) %.2f ms, %d candidates %.2f ms"
          % (3 * len(cat) / 1e6, (t1 - t0) * 1e3, nf + nr, (t2 - t1) * 1e3))
for it in range(5):
    eng.sync(0)
    t0 = time.perf_counter()
    eng.candidates_run_mixed(nf + nr, rc_from=nf, same_file=True, slot=0)
    t1 = time.perf_counter()
    eng.sync(0)
    t2 = time.perf_counter()
    rec = eng.candidates_fetch(nf + nr, slot=0)
    t3 = time.perf_counter()
    st = eng.last_run_stats(0)
    t4 = time.perf_counter()
    print("launch %.3f ms  wait %.3f ms  fetch %.3f ms  stats %.3f ms | events: total %.3f seed %.3f main %.3f"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, st["total_ms"], st["seed_ms"],
             st["main_ms"]))
