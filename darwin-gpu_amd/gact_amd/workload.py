"""Named synthetic workloads (BASELINE.json configs).

A workload is built from independent genome *blocks* (like chromosomes): block
b is simulated with seed+b, its reads get ids after those of block b-1, and its
candidates only pair reads of that block.  One block of `ecoli10x` is config
1/2; the 8-GPU run uses one block per rank so per-GPU work is fixed (weak
scaling) while every rank still holds the whole replicated read set.
"""
import numpy as np

from . import synth

CONFIGS = {
    # 10x E.coli-shape PBSIM reads, self-overlap (configs[0]/[1])
    "ecoli10x": dict(genome_len=4_641_652, coverage=10, seed=20260101),
    # 1/20-scale slice of the same (CI size)
    "ecoli10x_small": dict(genome_len=232_000, coverage=10, seed=20260101),
    # 50 MB PacBio-human-shape (configs[2])
    "pacbio50mb": dict(genome_len=1_000_000, n_reads=5000, seed=20260103),
    # ONT-shape ultra-long (configs[4])
    "ont": dict(genome_len=2_000_000, coverage=20, seed=20260105, uniform_len=(50_000, 100_000),
                error=0.12, split=(30, 30, 40)),
    "tiny": dict(genome_len=40_000, coverage=6, seed=7, mean_len=4000, sd_len=1000, min_len=800, max_len=8000),
}


class Block:
    def __init__(self, rs, cf, cr):
        self.rs, self.cf, self.cr = rs, cf, cr


def make_block(name, block=0):
    cfg = dict(CONFIGS[name])
    seed = cfg.pop("seed") + 1000 * block
    rs = synth.simulate_reads(seed=seed, **cfg)
    cf, cr = synth.synth_candidates(rs, seed=seed + 1)
    return Block(rs, cf, cr)


def shard(cands, rank, world):
    """round-robin deal of a candidate list"""
    return np.ascontiguousarray(cands[rank::world])
