"""Latency of one chain: runs the k longest chains of the bench workload alone (one wave, nothing else on the GPU)
and with the whole workload, and prints ms per tile of the longest chain."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "darwin-gpu_amd"))
import torch  # noqa: F401
import numpy as np
from gact_amd import engine, synth, workload

blk = workload.make_block(sys.argv[1] if len(sys.argv) > 1 else "ecoli10x", block=0, candidates="dsoft")
cat, offs = blk.rs.concat()
rcat, _ = blk.rs.concat(rc=True)
eng = engine.Engine(n_slots=1)
eng.upload(engine.SET_REF, cat, offs)
eng.upload(engine.SET_QUERY, cat, offs)
eng.upload(engine.SET_QUERY_RC, rcat, offs)
full = eng.extend(blk.cf, complement=False)
order = np.argsort(-full["n_tiles"])
print("forward candidates %d, tiles: max %d, p99 %d, p90 %d, mean %.1f" %
      (len(full), full["n_tiles"].max(), np.percentile(full["n_tiles"], 99), np.percentile(full["n_tiles"], 90),
       full["n_tiles"].mean()))
for k in (8, 64, 2048):
    sel = blk.cf[order[:k]]
    eng.candidates_upload(sel, slot=0)
    for it in range(2):
        eng.candidates_run(len(sel), complement=False, slot=0)
        eng.sync(0)
    st = eng.last_run_stats(0)
    nt = full["n_tiles"][order[:k]]
    print("%5d longest chains alone: seed %.2f ms, main %.2f ms; longest %d tiles -> %.3f ms per tile of the longest"
          % (k, st["seed_ms"], st["main_ms"], nt.max(), st["main_ms"] / (nt.max() - 1)))
