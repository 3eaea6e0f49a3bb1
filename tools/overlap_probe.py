"""One run at a time on one slot, with and without the overlapped, ordered seeding of round 4 (GACT_HIP_NO_OVERLAP):
wall time per step (launch, wait, fetch), the engine's stats, records compared.  Engines alternate inside one process.
python tools/overlap_probe.py [workload] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, workload

name = sys.argv[1] if len(sys.argv) > 1 else "ecoli10x"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
blk = workload.make_block(name)
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
nf, nr = len(blk.cf), len(blk.cr)
cands = np.concatenate([blk.cf, blk.cr])
engs = {}
for label, env in (("sequential", {"GACT_HIP_NO_OVERLAP": "1"}), ("overlapped", {})):
    for k in ("GACT_HIP_NO_OVERLAP",):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = engine.Engine()
    e.upload(engine.SET_REF, cat, offs); e.upload(engine.SET_QUERY, cat, offs); e.upload(engine.SET_QUERY_RC, rcat, roffs)
    e.candidates_upload(cands)
    engs[label] = (e, np.zeros(nf + nr, dtype=engine.OVERLAP_DTYPE))
    e.register_output(engs[label][1])
os.environ.pop("GACT_HIP_NO_OVERLAP", None)
wall = {k: [] for k in engs}
stats = {}
ref = None
for rep in range(reps + 2):
    for label, (e, out) in engs.items():
        e.sync()
        t0 = time.perf_counter()
        e.candidates_run_mixed(nf + nr, rc_from=nf)
        rec = e.candidates_fetch(nf + nr, out=out)
        dt = time.perf_counter() - t0
        if rep >= 2:
            wall[label].append(dt * 1e3)
        stats[label] = e.last_run_stats()
        if ref is None:
            ref = rec.copy()
        elif rec.tobytes() != ref.tobytes():
            print("RECORDS DIFFER:", label, "rep", rep)
cells = float(ref["cells"].sum())
for label in engs:
    v = sorted(wall[label]); st = stats[label]
    print("%s %-10s: step ms min %.2f med %.2f max %.2f -> %.0f GCUPS (median) | events: total %.2f = seed %.2f + main %.2f | overlapped %s, handed off %d, second runs %d" % (
        name, label, v[0], v[len(v) // 2], v[-1], cells / (v[len(v) // 2] * 1e-3) / 1e9, st["total_ms"], st["seed_ms"], st["main_ms"],
        st["overlapped_seeding"], st["handed_off"], st["band_redos"]))
