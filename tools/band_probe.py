"""Banded pointer stores (gact_lin.hpp LinBand): main-launch time, second runs and parity against band 0 for several
band widths, engines alternating inside one process (run-to-run noise on one box is +-3 %).
python tools/band_probe.py [workload] [bands, comma separated] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, workload

name = sys.argv[1] if len(sys.argv) > 1 else "ecoli10x"
bands = (sys.argv[2] if len(sys.argv) > 2 else "0,32,40,48,40/4,40/8").split(",")        # band or band/quantum
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
blk = workload.make_block(name)
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
nf, nr = len(blk.cf), len(blk.cr)
cands = np.concatenate([blk.cf, blk.cr])
engs = {}
for b in bands:
    os.environ["GACT_HIP_BAND"] = b.split("/")[0]
    os.environ["GACT_HIP_BAND_QUANTUM"] = (b.split("/") + ["1"])[1]
    e = engine.Engine()
    e.upload(engine.SET_REF, cat, offs); e.upload(engine.SET_QUERY, cat, offs); e.upload(engine.SET_QUERY_RC, rcat, roffs)
    e.candidates_upload(cands)
    engs[b] = e
ms = {b: [] for b in bands}
redos, ref = {}, None
rng = np.random.default_rng(1)
for rep in range(reps + 1):
    for b in rng.permutation(bands):
        e = engs[b]
        e.candidates_run_mixed(nf + nr, rc_from=nf)
        rec = e.candidates_fetch(nf + nr)
        st = e.last_run_stats()
        if rep:
            ms[b].append(st["main_ms"])
        redos[b] = st["band_redos"]
        if ref is None:
            ref = rec.copy()
        elif rec.tobytes() != ref.tobytes():
            print("RECORDS DIFFER at band", b)
tiles = int(ref["n_tiles"].sum())
for b in bands:
    v = sorted(ms[b])
    print("%s band %6s: main ms min %.2f med %.2f max %.2f | second runs %d of %d tiles (%.3f %%) | layout %s" % (
        name, b, v[0], v[len(v) // 2], v[-1], redos[b], tiles, 100.0 * redos[b] / tiles, st["layout"]))
