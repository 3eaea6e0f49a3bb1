"""Runs ONE rank's share of the N-rank weak-scaling bench workload on one GPU (no torch.distributed):
all N genome blocks are built and merged exactly as bench.py does after its exchange, the whole read set is made
resident, candidates are dealt round-robin and this rank's shard is extended and checked against the oracle on a
sample.  python tools/emulate_rank.py --world 8 --rank 0"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_py
from gact_amd import dist as gdist, engine, synth, workload

ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=0)
ap.add_argument("--workload", default="ecoli10x")
ap.add_argument("--check", type=int, default=3000)
a = ap.parse_args()

t = time.time()
blocks = []
for b in range(a.world):
    blk = workload.make_block(a.workload, block=b)
    blocks.append((blk.rs.reads, blk.cf, blk.cr))
reads, cf_all, cr_all = gdist.merge_blocks(blocks)
my_cf, my_cr = gdist.deal(cf_all, a.rank, a.world), gdist.deal(cr_all, a.rank, a.world)
offs = np.zeros(len(reads) + 1, dtype=np.int64); offs[1:] = np.cumsum([len(r) for r in reads])
cat = np.concatenate(reads); rcat = np.concatenate([synth.revcomp(r) for r in reads])
print("world %d: %d reads, %.1f Mb, %d + %d candidates, this rank %d + %d  (%.0f s to build)" % (
    a.world, len(reads), offs[-1] / 1e6, len(cf_all), len(cr_all), len(my_cf), len(my_cr), time.time() - t))
eng = engine.Engine()
eng.upload(engine.SET_REF, cat, offs); eng.upload(engine.SET_QUERY, cat, offs); eng.upload(engine.SET_QUERY_RC, rcat, offs)
nf, nr = len(my_cf), len(my_cr)
eng.candidates_upload(np.concatenate([my_cf, my_cr]))
for rep in range(3):
    t = time.perf_counter()
    eng.candidates_run_mixed(nf + nr, rc_from=nf)
    rec = eng.candidates_fetch(nf + nr)
    dt = time.perf_counter() - t
    st = eng.last_run_stats()
    print("step %.1f ms  (main %.1f, seed %.1f)  %.1f GCUPS for this rank's %d cells" % (
        dt * 1e3, st["main_ms"], st["seed_ms"], rec["cells"].sum() / dt / 1e9, rec["cells"].sum()))
orc = oracle_py.Oracle()
n = min(a.check, nf)
want, _ = orc.gact_many(cat, offs, cat, offs, my_cf[:n], complement=False, n_threads=min(16, os.cpu_count() or 1))
ok = all(np.array_equal(rec[f][:n], want[f]) for f in ("ab", "ae", "bb", "be", "score", "emitted", "n_tiles", "cells"))
m = min(a.check // 4, nr)
want_r, _ = orc.gact_many(cat, offs, rcat, offs, my_cr[:m], complement=True, n_threads=min(16, os.cpu_count() or 1))
ok = ok and all(np.array_equal(rec[f][nf:nf + m], want_r[f]) for f in ("ab", "ae", "bb", "be", "score", "emitted", "n_tiles", "cells"))
print("parity on %d candidates: %s" % (n + m, "BIT-EXACT" if ok else "MISMATCH"))
sys.exit(0 if ok else 1)
