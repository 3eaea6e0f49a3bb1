"""The reference's own caller at speed: oracle/_ref/darwin_on_hip (the reference's UNMODIFIED darwin.cpp compiled -DGPU
against host/gact.h + gact_shim.cpp, tests/test_reference_caller.py) on a workload's FASTA with N feeder threads; what
its threads print as "Time GACT calling" (darwin.cpp:424-441: both GACT_Batch calls of a thread) against the DP cells of
the workload.  Once as it is -- forward call, then reverse-complement call, each merged across the threads by the engine's
call combiner -- and once with GACT_HIP_PAIR_STRANDS=1 (the shim runs a thread's two calls as one).  Lines compared.
python tools/darwin_on_hip_timing.py [workload] [threads] [repetitions per mode, default 2]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
import numpy as np
from gact_amd import engine, workload

name = sys.argv[1] if len(sys.argv) > 1 else "ecoli10x"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
exe = os.path.join(ROOT, "oracle", "_ref", "darwin_on_hip")
if not os.path.exists(exe):
    raise SystemExit("oracle/_ref/darwin_on_hip not built (needs /root/reference at build time)")
blk = workload.make_block(name, candidates="synthetic")
# The reference's -DGPU path deals reads to its filter in ranges of ceil(N / T) but recodes them in place in ranges of
# floor(N / T) (darwin.cpp:304-312,340-347,619-621): where T does not divide N a thread that is done recodes reads another
# thread is still filtering and candidates are lost, differently from run to run (tests/test_reference_caller.py).  The
# read set is cut to a multiple of the thread count.
from gact_amd import synth
keep = (blk.rs.n // threads) * threads
blk.rs.reads = blk.rs.reads[:keep]
blk.rs.names = blk.rs.names[:keep]
blk.cf, blk.cr = workload.dsoft_candidates(blk.rs)
# the cells of the workload: the engine's own records for the filter's candidates
eng = engine.Engine()
cat, offs = blk.rs.concat(); rcat, roffs = blk.rs.concat(rc=True)
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, roffs)
cands = np.concatenate([blk.cf, blk.cr])
eng.candidates_upload(cands)
eng.candidates_run_mixed(len(cands), rc_from=len(blk.cf))
rec = eng.candidates_fetch(len(cands))
cells = int(rec["cells"].sum())
emitted = int(rec["emitted"].sum())
eng.close()
# the lines this repo's own pipeline prints for these candidates (gact.cpp:214-224; a name = the header's first field)
import re as _re
names = [_re.match(r"[A-Za-z0-9_]*", nm).group(0) for nm in blk.rs.names]
want_lines = sorted("ref_id: %s, query_id: %s, ab: %d, ae: %d, bb: %d, be: %d, score: %d, comp: %d" % (
    names[r["ref_id"]], names[r["query_id"]], r["ab"], r["ae"], r["bb"], r["be"], r["score"], r["comp"]) for r in rec if r["emitted"])
out = {"workload": name, "reads": keep, "feeder_threads": threads, "candidates": int(len(cands)), "cells": cells, "lines_expected": emitted, "runs": []}
base_lines = want_lines
for label, env in (("as it is (two calls per thread, one after the other)", {}),
                   ("GACT_HIP_PAIR_STRANDS=1 (a thread's two calls as one run)", {"GACT_HIP_PAIR_STRANDS": "1"})):
    with tempfile.TemporaryDirectory() as d:
        blk.rs.write_fasta(os.path.join(d, "reads.fasta"))
        with open(os.path.join(d, "params.cfg"), "w") as f:
            f.write(workload.PARAMS_CFG)
        e = dict(os.environ)
        e.update(env)
        e["GACT_HIP_TIME"] = "1"
        best = None
        for rep in range(reps):
            t0 = time.time()
            p = subprocess.run([exe, "reads.fasta", "reads.fasta", str(threads), "32", "64"], cwd=d, env=e, capture_output=True, text=True, timeout=900)
            wall = time.time() - t0
            if p.returncode != 0:
                raise SystemExit("darwin_on_hip failed: " + p.stdout[-1500:] + p.stderr[-1500:])
            gact_ms = [float(x) for x in re.findall(r"Time GACT calling: ([0-9.]+) msec", p.stdout)]
            lines = []
            for fn in sorted(os.listdir(d)):
                if fn.startswith("darwin.") and fn.endswith(".out"):
                    lines += open(os.path.join(d, fn)).read().splitlines()
            split = re.findall(r"time_gpu split, slot (\d+): upload (\d+) us, submit (\d+) us, wait \+ fetch (\d+) us \(launch: ([0-9.]+) ms on the device, (\d+) callers merged\)", p.stdout)
            row = {"shim_split_per_call_us": [{"slot": int(a), "upload": int(b), "submit": int(c), "wait_fetch": int(d2), "launch_ms": float(f), "merged": int(g)}
                                               for a, b, c, d2, f, g in split],
                   "gact_calling_ms_max_over_threads": max(gact_ms), "gact_calling_ms_per_thread": gact_ms, "lines": len(lines),
                   "wall_s": round(wall, 1), "gcups_of_the_gact_stage": round(cells / (max(gact_ms) * 1e-3) / 1e9, 1)}
            trace = [ln for ln in p.stderr.splitlines() if ln.startswith("[gact_hip] upload") or ln.startswith("[gact_hip] fetch") or ln.startswith("[gact_hip] combiner")]
            if trace:
                row["engine_trace"] = trace
            if best is None or row["gact_calling_ms_max_over_threads"] < best["gact_calling_ms_max_over_threads"]:
                best = row
            lines.sort()
            if base_lines is None:
                base_lines = lines
            elif lines != base_lines:
                a, b = set(base_lines), set(lines)
                raise SystemExit("darwin_on_hip: output lines differ between runs (%s): %d vs %d lines, %d only before, %d only now, e.g.\n %s\n %s"
                                 % (label, len(base_lines), len(lines), len(a - b), len(b - a), sorted(a - b)[:3], sorted(b - a)[:3]))
        best["mode"] = label
        out["runs"].append(best)
out["lines_equal_between_modes_and_to_this_repos_own_pipeline"] = True
out["lines"] = len(base_lines)
print(json.dumps(out))
