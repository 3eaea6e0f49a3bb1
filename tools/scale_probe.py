"""Is the chain launch throughput-bound or tail-bound?  Runs the bench workload's candidate list once, twice and
four times over in one launch: a throughput-bound launch scales linearly, a fixed tail shows as an offset."""
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
for mult in (1, 2, 4):
    cf = np.concatenate([blk.cf] * mult)
    cr = np.concatenate([blk.cr] * mult)
    eng.candidates_upload(np.concatenate([cf, cr]), slot=0)
    ms = []
    for it in range(3):
        eng.candidates_run_mixed(len(cf) + len(cr), rc_from=len(cf), same_file=True, slot=0)
        eng.sync(0)
        st = eng.last_run_stats(0)
        ms.append((st["seed_ms"], st["main_ms"]))
    print("x%d: seed %.2f ms, main %.2f ms  (per x1: %.2f ms)" % (mult, ms[-1][0], ms[-1][1], ms[-1][1] / mult))
