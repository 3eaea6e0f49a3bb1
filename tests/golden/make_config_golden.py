"""Golden records of BASELINE configs 2, 3, 5 and two ranks of config 4 at full size: every candidate of the workload through the ORACLE
(oracle/gact_oracle.c, the CPU restatement pinned against the reference), one CRC-32 per record over the twelve fields the
GPU tests compare.  tests/test_gpu_configs.py checks EVERY record of the HIP engine against these (and a strided sample
against the oracle run live): the live oracle over all 65,766 + 14,501 candidates took 3.5 of the GPU suite's 8 minutes.

    python tests/golden/make_config_golden.py [workload ...]     # ~3 + 4 minutes on 8 cores; writes config_<workload>.npz
    python tests/golden/make_config_golden.py pacbio50mb config4:0 config4:5     # ~15 minutes each (1.0e12 cells)

`config4:<rank>` is rank <rank>'s share of the eight-block job of tests/test_gpu_configs.py::test_config4_one_rank_of_eight
(eight pacbio50mb genome blocks merged, the candidate list dealt round-robin over eight ranks); GOLDEN_THREADS=<n> caps the
oracle's threads.

The workloads are rebuilt from their seeds (gact_amd/workload.py; candidates from the D-SOFT restatement), so the file
also carries a checksum of the candidate list it was made for."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

from gact_amd.workload import RECORD_FIELDS as FIELDS, record_crcs        # (shared with bench.py's config-4 parity gate)


def _config4_block(b):
    from gact_amd import workload
    blk = workload.make_block("pacbio50mb", block=b, candidates="dsoft")
    return blk.rs.reads, blk.cf, blk.cr


def config4_rank(rank, world=8):
    """(cat, offs, rcat, cf, cr) of one rank of the eight-block job, built as the GPU test builds it"""
    from gact_amd import dist as gdist, synth
    reads, cf_all, cr_all = gdist.merge_blocks([_config4_block(b) for b in range(world)])
    offs = np.zeros(len(reads) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    cat = np.concatenate(reads)
    rcat = np.concatenate([synth.revcomp(r) for r in reads])
    return cat, offs, rcat, gdist.deal(cf_all, rank, world), gdist.deal(cr_all, rank, world)


def main():
    from gact_amd import workload
    import oracle_py
    orc = oracle_py.Oracle()
    threads = int(os.environ.get("GOLDEN_THREADS", len(os.sched_getaffinity(0))))
    for name in (sys.argv[1:] or ["ecoli10x", "ont"]):
        if name.startswith("config4:"):
            cat, offs, rcat, cf, cr = config4_rank(int(name.split(":")[1]))
            roffs = offs
            name = "config4_rank%d" % int(name.split(":")[1])
        else:
            blk = workload.make_block(name, candidates="dsoft")
            cat, offs = blk.rs.concat()
            rcat, roffs = blk.rs.concat(rc=True)
            cf, cr = blk.cf, blk.cr
        rf, cells_f = orc.gact_many(cat, offs, cat, offs, cf, complement=False, same_file=True, n_threads=threads)
        rr, cells_r = orc.gact_many(cat, offs, rcat, roffs, cr, complement=True, same_file=True, n_threads=threads)
        rec = np.concatenate([rf, rr])
        out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_%s.npz" % name)
        np.savez_compressed(out, crc=record_crcs(rec), n_forward=np.int64(len(cf)), n_reverse=np.int64(len(cr)),
                            candidates_crc=np.uint32(zlib.crc32(np.concatenate([cf, cr]).tobytes())),
                            cells=np.int64(rec["cells"].sum()), tiles=np.int64(rec["n_tiles"].sum()))
        print(name, len(rec), "records,", int(rec["cells"].sum()), "cells ->", out, os.path.getsize(out), "bytes", flush=True)


if __name__ == "__main__":
    main()
