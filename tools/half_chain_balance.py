"""Would splitting every candidate into a left and a right half-chain after the seed launch shorten the critical
path?  (VERDICT r02, next #2.)  The two directions are independent once the first tile is consumed
(gact.cpp:136-141), so the question is only how the tiles of a chain divide between them: tiles to the left of a
D-SOFT seed hit ~ min(ref_pos, query_pos) / early, to the right ~ min(ref_len - ref_pos, query_len - query_pos) / early.
Runs on the CPU (the filter's candidates of the bench workloads).  python tools/half_chain_balance.py [workload ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "darwin-gpu_amd")]
import numpy as np
from gact_amd import workload

EARLY = 200
for name in (sys.argv[1:] or ["ecoli10x", "pacbio50mb", "ont"]):
    blk = workload.make_block(name)
    rl = np.array([len(r) for r in blk.rs.reads])
    c = np.concatenate([blk.cf, blk.cr])
    left = np.minimum(c["ref_pos"], c["query_pos"]) / EARLY
    right = np.minimum(rl[c["ref_id"]] - c["ref_pos"], rl[c["query_id"]] - c["query_pos"]) / EARLY
    tot, longer = left + right, np.maximum(left, right)
    print("%-10s %7d candidates | tiles left of the seed: mean %.1f, max %.0f | right: mean %.1f, max %.0f | longest chain %.0f, "
          "longest half %.0f | longer half / chain: mean %.2f" %
          (name, len(c), left.mean(), left.max(), right.mean(), right.max(), tot.max(), longer.max(),
           (longer / np.maximum(tot, 1)).mean()))
