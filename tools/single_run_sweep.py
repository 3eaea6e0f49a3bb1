"""One run at a time on an idle engine (the `single_slot` figure of bench.py) under several settings of the live options
(gact_hip_set_option), one engine, one process:

    python tools/single_run_sweep.py ecoli10x coop=0 coop=1,lean16=0 coop=1,lean16=11 ...

Per setting: 2 untimed + 6 timed runs (upload excluded: the list is resident; launch -> records on the host), wall ms
best / median, GCUPS of the median, the engine's own launch time, kernel layout.  Records are checked against the first
setting's (a CRC per record)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, workload

name = sys.argv[1] if len(sys.argv) > 1 else "ecoli10x"
settings = sys.argv[2:] or ["coop=2"]
blk = workload.make_block(name)
eng = engine.Engine()
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
nf, nr = len(blk.cf), len(blk.cr)
eng.candidates_upload(np.concatenate([blk.cf, blk.cr]))
want = None
for setting in settings:
    for kv in setting.split(","):
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    wall = []
    for rep in range(8):
        t0 = time.perf_counter()
        eng.candidates_run_mixed(nf + nr, rc_from=nf)
        rec = eng.candidates_fetch(nf + nr)
        wall.append((time.perf_counter() - t0) * 1e3)
    st = eng.last_run_stats()
    crc = workload.record_crcs(rec)
    if want is None:
        want = crc
    same = bool(np.array_equal(crc, want))
    w = sorted(wall[2:])
    cells = int(rec["cells"].sum())
    print("%s %-28s wall ms best %.2f med %.2f -> %.0f GCUPS; engine %.2f ms (seed %.2f); layout %s%s%s; records %s" % (
        name, setting, w[0], w[len(w) // 2], cells / (w[len(w) // 2] * 1e-3) / 1e9, st["total_ms"], st["seed_ms"], st["layout"],
        "-coop" if st.get("coop_walks") else "", "-overlapped" if st.get("overlapped_seeding") else "", "equal" if same else "DIFFER"), flush=True)
    if not same:
        sys.exit(1)
