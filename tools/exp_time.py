"""Timing experiments with builds whose results are WRONG on purpose (-DGACT_EXP_FAKE_WALK, -DGACT_EXP_NO_STORE): the
launch time and cell count of one workload, no parity check.  GACT_HIP_LIB_PATH=<lib> python tools/exp_time.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, workload

name = sys.argv[1] if len(sys.argv) > 1 else "ecoli10x"
blk = workload.make_block(name)
eng = engine.Engine()
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
nf, nr = len(blk.cf), len(blk.cr)
eng.candidates_upload(np.concatenate([blk.cf, blk.cr]))
ms = []
for rep in range(8):
    eng.candidates_run_mixed(nf + nr, rc_from=nf)
    rec = eng.candidates_fetch(nf + nr)
    st = eng.last_run_stats()
    ms.append((st["main_ms"], st["seed_ms"]))
main = sorted(m[0] for m in ms[2:])
cells = int(rec["cells"].sum()); tiles = int(rec["n_tiles"].sum())
print("%s %s: main ms min %.2f med %.2f; seed %.2f; tiles %d cells %.4e -> %.0f GCUPS of these cells (main launch), %.2f us per 1000 tiles" % (
    name, os.path.basename(os.environ.get("GACT_HIP_LIB_PATH", "default")), main[0], main[len(main) // 2], ms[-1][1], tiles, cells,
    cells / (main[len(main) // 2] * 1e-3) / 1e9, main[len(main) // 2] * 1e3 / (tiles / 1000.0)))
