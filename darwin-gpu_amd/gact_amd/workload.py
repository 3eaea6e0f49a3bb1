"""Named synthetic workloads (BASELINE.json configs).

A workload is built from independent genome *blocks* (like chromosomes): block
b is simulated with seed+b, its reads get ids after those of block b-1, and its
candidates only pair reads of that block.  One block of `ecoli10x` is config
1/2; the 8-GPU run uses one block per rank so per-GPU work is fixed (weak
scaling) while every rank still holds the whole replicated read set.
"""
import os
import subprocess
import tempfile

import numpy as np

from . import synth

PARAMS_CFG = """[GACT_scoring]
match = 1
mismatch = -1
gap_open = -1
gap_extend = -1

[DSOFT_params]
seed_size  = 14
bin_size   = 64
window_size= 4
threshold  = 21
num_seeds  = 800
seed_occurence_multiple = 32
max_candidates = 1000000
num_nz_bins    = 2500000

[GACT_first_tile]
first_tile_size = 128
first_tile_score_threshold = 35

[GACT_extend]
tile_size = 320
tile_overlap = 120
"""   # the reference's params.cfg values (params.cfg:1-23)

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


def dsoft_candidates(rs, threads=None):
    """Candidates of the D-SOFT filter itself (host/dsoft.cpp through the driver's --dsoft-only mode,
    reference parameters) for the self-overlap run of a read set: (forward, reverse-complement) arrays
    in the order darwin.cpp:209-288 produces them."""
    from . import engine
    drv = engine.driver_path()
    # (one filter per rank at N > 1: the ranks of a node share its cores, and every filter holds a 1 GiB index)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1)
    threads = threads or max(1, min(16, (os.cpu_count() or 1) // max(local_world, 1)))
    with tempfile.TemporaryDirectory() as d:
        rs.write_fasta(os.path.join(d, "reads.fasta"))
        with open(os.path.join(d, "params.cfg"), "w") as f:
            f.write(PARAMS_CFG)
        subprocess.check_call([drv, "reads.fasta", "reads.fasta", str(threads), "--dsoft-only",
                               "--dump-candidates", "cands.bin"], cwd=d, stdout=subprocess.DEVNULL)
        raw = np.fromfile(os.path.join(d, "cands.bin"), dtype=np.int32).reshape(-1, 5)
    out = []
    for comp in (0, 1):
        sel = raw[raw[:, 4] == comp]
        c = np.zeros(len(sel), dtype=synth.CAND_DTYPE)
        c["ref_id"], c["query_id"], c["ref_pos"], c["query_pos"] = sel[:, 0], sel[:, 1], sel[:, 2], sel[:, 3]
        out.append(c)
    return out[0], out[1]


def make_block(name, block=0, candidates="dsoft"):
    """candidates: "dsoft" = run the filter (needs the built driver), "synthetic" = place hits from simulator truth"""
    cfg = dict(CONFIGS[name])
    seed = cfg.pop("seed") + 1000 * block
    rs = synth.simulate_reads(seed=seed, **cfg)
    if candidates == "dsoft":
        cf, cr = dsoft_candidates(rs)
    else:
        cf, cr = synth.synth_candidates(rs, seed=seed + 1)
    return Block(rs, cf, cr)


def shard(cands, rank, world):
    """round-robin deal of a candidate list"""
    return np.ascontiguousarray(cands[rank::world])


# ---- golden records (tests/golden/config_*.npz, made by tests/golden/make_config_golden.py from the oracle): one CRC-32 per
#      record over the twelve fields the parity tests compare
RECORD_FIELDS = ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted", "first_tile_score", "n_tiles", "cells")


def record_crcs(rec):
    """one CRC-32 per record over RECORD_FIELDS as int64"""
    import zlib
    a = np.ascontiguousarray(np.stack([rec[f].astype(np.int64) for f in RECORD_FIELDS], axis=1))
    return np.fromiter((zlib.crc32(row.tobytes()) for row in a), dtype=np.uint32, count=len(a))


CONFIG4_BLOCKS = 8        # BASELINE config 4 in the form one node runs it: eight pacbio50mb genome blocks, 40,000 reads, 418 Mb


def config4_blocks(mine, candidates="dsoft", name="pacbio50mb"):
    """the genome blocks `mine` (indices) of the config-4 job: [(index, (reads, cands_fwd, cands_rc))]"""
    out = []
    for b in mine:
        blk = make_block(name, block=b, candidates=candidates)
        out.append((b, (blk.rs.reads, blk.cf, blk.cr)))
    return out
