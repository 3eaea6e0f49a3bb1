"""Scheduler robustness sweep: candidate count (1 k ... 1 M) x read length (2 k ... 100 k) -> what the
launch policy chose (csrc/gact_policy.hpp) and what it delivered, one run at a time on an idle engine.

    python tools/sweep_policy.py [--out profiles/r05/sweep_policy.json]

A base set is simulated per read length (reads of that length +- 20 %, 15 % error -- 12 % beyond 30 kb --, candidates from
simulator truth); the candidate list is the base list repeated and cut to the count (candidates are independent,
gact.cpp:48: a repeated candidate is simply another chain of the same length).  Per point: 1 warm-up + 3 runs, best time,
GCUPS, layout and sequence as gact_hip_run_stats reports them, the plan gact_hip_plan_describe gives for that count.
Flags neighbouring counts of one read length whose throughput per cell differs by more than 15 %."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, synth

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep_policy.json"))
ap.add_argument("--lengths", default="2000,5000,10000,30000,100000")
ap.add_argument("--counts", default="1000,3000,10000,20000,30000,45000,65000,100000,200000,400000,1000000")
ap.add_argument("--max-cells", type=float, default=6e12, help="points whose estimated cells exceed this are skipped")
ap.add_argument("--set", default="", help="live options for the engine, e.g. lone_lane=0,overlap_big=1 (gact_hip_set_option)")
args = ap.parse_args()
lengths = [int(x) for x in args.lengths.split(",")]
counts = [int(x) for x in args.counts.split(",")]

rows = []
for L in lengths:
    n_reads = max(60, min(600, int(4_000_000 // L)))
    err = 0.12 if L > 30000 else 0.15
    rs = synth.simulate_reads(int(L * n_reads / 10), n_reads=n_reads, seed=900 + L % 997, mean_len=L, sd_len=L // 5, min_len=L // 2,
                              max_len=2 * L, error=err)
    cf, cr = synth.synth_candidates(rs, seed=901 + L % 997, min_overlap=max(300, L // 10))
    base = np.concatenate([cf, cr])
    base_rc = np.concatenate([np.zeros(len(cf), bool), np.ones(len(cr), bool)])
    cat, offs = rs.concat(); rcat, roffs = rs.concat(rc=True)
    eng = engine.Engine()
    for kv in filter(None, args.set.split(",")):
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
    # cells of one pass over the base list
    eng.candidates_upload(base)
    eng.candidates_run_mixed(len(base), rc_from=len(cf))
    base_rec = eng.candidates_fetch(len(base)).copy()
    base_cells = float(base_rec["cells"].sum())
    for n in counts:
        est = base_cells * n / len(base)
        if est > args.max_cells:
            continue
        reps = -(-n // len(base))
        idx = np.tile(np.arange(len(base)), reps)[:n]
        order = np.argsort(base_rc[idx], kind="stable")                  # forward strand first, as the engine wants the list
        idx = idx[order]
        cands = base[idx]
        nf = int((~base_rc[idx]).sum())
        eng.candidates_upload(cands)
        best = None
        for rep in range(4):
            t0 = time.perf_counter()
            eng.candidates_run_mixed(n, rc_from=nf)
            rec = eng.candidates_fetch(n)
            dt = time.perf_counter() - t0
            if rep and (best is None or dt < best):
                best = dt
        st = eng.last_run_stats()
        if rec.tobytes() != base_rec[idx].tobytes():
            raise SystemExit("sweep: records of the repeated list differ from the base run's (L=%d, n=%d)" % (L, n))
        cells = float(rec["cells"].sum())
        plan = engine.plan(n)
        rows.append({"read_length": L, "candidates": n, "tiles": int(rec["n_tiles"].sum()), "cells": cells, "ms": round(best * 1e3, 3),
                     "gcups": round(cells / best / 1e9, 1), "layout": st["layout"] + ("-lin" if st["linear_gap"] else ""),
                     "overlapped_seeding": st["overlapped_seeding"], "critical_lane": st["critical_lane"], "main_ms": round(st["main_ms"], 3),
                     "seed_ms": round(st["seed_ms"], 3), "plan_sequence": plan["sequence"], "plan_main_kernel": plan["main_kernel"],
                     "plan_main_blocks": plan["main_blocks"], "tiles_per_chain": round(float(rec["n_tiles"].mean()), 1)})
        print(json.dumps(rows[-1]), flush=True)
    eng.close()

# neighbouring counts of one read length: throughput per cell more than 15 % apart?
flags = []
for L in lengths:
    pts = [r for r in rows if r["read_length"] == L]
    for a, b in zip(pts, pts[1:]):
        if min(a["gcups"], b["gcups"]) < 0.85 * max(a["gcups"], b["gcups"]):
            flags.append({"read_length": L, "from": a["candidates"], "to": b["candidates"], "gcups": [a["gcups"], b["gcups"]],
                          "sequences": [a["plan_sequence"], b["plan_sequence"]], "layouts": [a["layout"], b["layout"]]})
out = {"rows": rows, "neighbours_more_than_15_percent_apart": flags}
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(out, open(args.out, "w"), indent=1)
print("%d points, %d neighbour pairs more than 15 %% apart -> %s" % (len(rows), len(flags), args.out))
