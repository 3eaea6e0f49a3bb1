"""Device D-SOFT filter on a bench workload: index build and query times (HIP events inside the library), the
bytes the build has to move at least, and the host restatement timed beside it.
Run on the GPU box: python tools/bench_dsoft.py [workload] [--no-host]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "darwin-gpu_amd"))
import torch  # noqa: F401  (its HIP runtime first)
import numpy as np
from gact_amd import engine, synth, workload

name = next((a for a in sys.argv[1:] if not a.startswith("--")), "ecoli10x")
cfg = dict(workload.CONFIGS[name])
seed = cfg.pop("seed")
rs = synth.simulate_reads(seed=seed, **cfg)
cat, offs = rs.concat()
rcat, _ = rs.concat(rc=True)
eng = engine.Engine()
eng.upload(engine.SET_REF, cat, offs)
eng.upload(engine.SET_QUERY, cat, offs)
eng.upload(engine.SET_QUERY_RC, rcat, offs)
build_ms, query_ms = [], []
for it in range(4):
    info = eng.dsoft_build()
    nf, nr, ms = eng.dsoft_query(0, rs.n)
    if it:                                   # the first round pays the scratch allocation
        build_ms.append(info["build_ms"])
        query_ms.append(ms)
cands = eng.candidates_download(nf + nr)
# least traffic of the build: the 4^k table is cleared, counted into, scanned (read + write) and read by the
# segment sort; the 2-bit reference is read twice by the minimizer passes; positions written once
table = info["table_bytes"]
least = 4 * table + 2 * info["ref_length"] / 4 + 2 * info["pos_bytes"]
out = {"workload": name, "reads": rs.n, "bases": int(offs[-1]), "minimizers": info["n_minimizers"],
       "candidates": int(nf + nr), "build_ms": round(float(np.mean(build_ms)), 3),
       "query_ms": round(float(np.mean(query_ms)), 3), "table_GiB": round(table / 2**30, 3),
       "build_least_GB": round(least / 1e9, 3), "build_GBps": round(least / 1e9 / (np.mean(build_ms) * 1e-3), 1),
       "query_strands_per_s": round(2 * rs.n / (np.mean(query_ms) * 1e-3))}
if "--no-host" not in sys.argv:
    t = time.time()
    cf, cr = workload.dsoft_candidates(rs)
    out["host_restatement_s"] = round(time.time() - t, 2)
    out["host_threads"] = min(16, os.cpu_count() or 1)
    out["lists_equal"] = bool(np.array_equal(cands[:nf], cf) and np.array_equal(cands[nf:], cr))
print(json.dumps(out))
