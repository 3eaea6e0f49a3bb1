"""Would ONE big run be faster as K sub-runs kept in flight on the engine's slots?  The whole candidate list of a workload in K
parts (each with its share of both strands), `slots` of them in flight at a time, wall time from the first launch to the last
record on the host -- against the same list as one run on one slot.  python tools/subrun_probe.py [workload] [K ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, workload

name = sys.argv[1] if len(sys.argv) > 1 else "pacbio50mb"
Ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
SLOTS = 4
blk = workload.make_block(name)
eng = engine.Engine(n_slots=SLOTS)
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
nf, nr = len(blk.cf), len(blk.cr)
whole = None
for K in Ks:
    fb = np.linspace(0, nf, K + 1).astype(int); rb = np.linspace(0, nr, K + 1).astype(int)
    parts = [(np.concatenate([blk.cf[fb[k]:fb[k + 1]], blk.cr[rb[k]:rb[k + 1]]]), int(fb[k + 1] - fb[k])) for k in range(K)]
    eng.set_option("runs_in_flight", 1 if K > 1 else 0)
    walls = []
    for rep in range(5):
        recs = [None] * K
        # (uploads are outside the timed region, as in bench.py: a slot's list is resident when its run is launched -- so a part is
        #  uploaded when its slot is free, and with K > slots the later uploads ARE inside: they overlap the running parts)
        for k in range(min(K, SLOTS)):
            eng.candidates_upload(parts[k][0], slot=k)
        t0 = time.perf_counter()
        for k in range(min(K, SLOTS)):
            eng.candidates_run_mixed(len(parts[k][0]), rc_from=parts[k][1], slot=k)
        for k in range(K):
            recs[k] = eng.candidates_fetch(len(parts[k][0]), slot=k % SLOTS).copy()
            nxt = k + SLOTS
            if nxt < K:
                eng.candidates_upload(parts[nxt][0], slot=nxt % SLOTS)
                eng.candidates_run_mixed(len(parts[nxt][0]), rc_from=parts[nxt][1], slot=nxt % SLOTS)
        walls.append((time.perf_counter() - t0) * 1e3)
    cells = sum(int(r["cells"].sum()) for r in recs)
    crc = np.sort(np.concatenate([workload.record_crcs(r) for r in recs]))
    if whole is None:
        whole = crc
    w = sorted(walls[1:])
    print("%s as %d part(s), %d in flight: wall ms best %.2f med %.2f -> %.0f GCUPS; records %s" % (
        name, K, min(K, SLOTS), w[0], w[len(w) // 2], cells / (w[len(w) // 2] * 1e-3) / 1e9, "equal" if np.array_equal(crc, whole) else "DIFFER"), flush=True)
