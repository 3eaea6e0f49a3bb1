"""Steps of one workload alternating over S engine slots (the same candidate list resident in each): the fetch of step
k is taken after step k+1 has been launched, so a step's tail (queues drained, last chains running) and its record
copy overlap the next step's seed launch.  python tools/pipeline_probe.py [workload] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, workload

name = sys.argv[1] if len(sys.argv) > 1 else "ecoli10x"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
blk = workload.make_block(name)
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
nf, nr = len(blk.cf), len(blk.cr)
cands = np.concatenate([blk.cf, blk.cr])
want = None
for S in [int(x) for x in os.environ.get("SLOTS", "1,2,3").split(",")]:
    eng = engine.Engine(n_slots=S)
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    recs = []
    for k in range(S):
        eng.candidates_upload(cands, slot=k)
        recs.append(np.zeros(nf + nr, dtype=engine.OVERLAP_DTYPE))
        eng.register_output(recs[k], slot=k)
    for k in range(S):                                    # warm-up
        eng.candidates_run_mixed(nf + nr, rc_from=nf, slot=k)
        eng.candidates_fetch(nf + nr, slot=k, out=recs[k])
    t0 = time.perf_counter()
    for k in range(steps):
        eng.candidates_run_mixed(nf + nr, rc_from=nf, slot=k % S)
        if k >= S - 1:
            j = (k - (S - 1)) % S
            eng.candidates_fetch(nf + nr, slot=j, out=recs[j])
    for k in range(steps - (S - 1), steps):
        eng.candidates_fetch(nf + nr, slot=k % S, out=recs[k % S])
    dt = (time.perf_counter() - t0) / steps
    cells = int(recs[0]["cells"].sum())
    if want is None:
        want = recs[0].copy()
    ok = all(r.tobytes() == want.tobytes() for r in recs)
    print("%s slots %d: %.2f ms per step, %.0f GCUPS, records equal %s" % (name, S, dt * 1e3, cells / dt / 1e9, ok))
    eng.close()
