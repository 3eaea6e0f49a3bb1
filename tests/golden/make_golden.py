#!/usr/bin/env python
"""Generates tests/golden/*.json from the REFERENCE ITSELF (oracle/_ref/libdarwin_ref.so =
/root/reference/align.cpp + gact.cpp compiled unchanged, see oracle/Makefile).

Run in the build container (where /root/reference is mounted):
    python tests/golden/make_golden.py
The fixtures are data only: inputs and the reference's outputs.
  tiles.json   AlignWithBT: (ref, query, scoring, reverse, first, early) -> the returned queue
  chains.json  GACT: reads + candidates -> the exact output line (or "" when nothing is printed)
  dsoft.json   SeedPosTable + SeedPosTable::DSOFT (seed_pos_table.cpp:46-167) on a 16-read FASTA, decoded as
               darwin.cpp:213-224 does: the candidate list, per read forward strand then reverse complement
  e2e.json     the reference's whole CPU program (oracle/_ref/darwin_cpu = its own sources through plain g++,
               oracle/Makefile) run on that FASTA: `cat darwin.*.out | sort` (README:25)
Re-running reproduces the committed files byte for byte (`git diff --stat tests/golden` stays empty).
"""
import glob
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402

import oracle_py  # noqa: E402
from gact_amd import synth  # noqa: E402
from tilecases import random_tiles  # noqa: E402


def main():
    ref = oracle_py.RefLib()
    scorings = [(1, -1, -1, -1), (2, -3, -5, -2), (5, -4, -10, -1), (1, -1, -2, -1)]
    tiles = []
    for k, (a, b, rev, first) in enumerate(random_tiles(2026, 200) + random_tiles(77, 40, with_n=True)):
        sc = scorings[k % len(scorings)]
        early = [200, 200, 320, 64][k % 4]
        q = ref.align_with_bt(a, b, sc, rev, first, early)
        tiles.append({"ref": a.decode("latin-1"), "query": b.decode("latin-1"), "scoring": list(sc),
                      "reverse": rev, "first": first, "early": early, "queue": q})
    json.dump({"source": "ref_align_with_bt -> AlignWithBT (reference align.cpp:60-233)", "tiles": tiles},
              open(os.path.join(HERE, "tiles.json"), "w"), separators=(",", ":"))

    rs = synth.simulate_reads(12000, n_reads=10, seed=17, mean_len=3000, sd_len=700, min_len=600, max_len=5000,
                              n_frac=0.002)
    cf, cr = synth.synth_candidates(rs, seed=18, min_overlap=300, false_frac=0.2)
    chains = []
    settings = [dict(tile_size=320, tile_overlap=120, threshold=35, scoring=(1, -1, -1, -1)),
                dict(tile_size=128, tile_overlap=32, threshold=20, scoring=(2, -3, -5, -2))]
    for comp, cands in ((False, cf), (True, cr)):
        for c in cands:
            r = rs.reads[c["ref_id"]].tobytes()
            q = (synth.revcomp(rs.reads[c["query_id"]]) if comp else rs.reads[c["query_id"]]).tobytes()
            for si, st in enumerate(settings):
                line = ref.gact_line(r, q, int(c["ref_pos"]), int(c["query_pos"]), ref_id=int(c["ref_id"]),
                                     query_id=int(c["query_id"]), complement=comp, same_file=True,
                                     ref_name=rs.names[c["ref_id"]], query_name=rs.names[c["query_id"]], **st)
                chains.append({"ref_id": int(c["ref_id"]), "query_id": int(c["query_id"]),
                               "ref_pos": int(c["ref_pos"]), "query_pos": int(c["query_pos"]),
                               "comp": int(comp), "setting": si, "line": line})
    json.dump({"source": "ref_gact -> GACT (reference gact.cpp:48-228)",
               "reads": [r.tobytes().decode("latin-1") for r in rs.reads], "names": rs.names,
               "settings": [dict(s, scoring=list(s["scoring"])) for s in settings], "chains": chains},
              open(os.path.join(HERE, "chains.json"), "w"), separators=(",", ":"))
    print("tiles:", len(tiles), "chains:", len(chains))
    dsoft_and_e2e(ref)


SEED_SIZE = 11          # a 4 MiB seed table instead of params.cfg's k = 14 (1 GiB): same code paths, CI-sized


def dsoft_and_e2e(ref):
    from test_dsoft import CFG, fasta_of
    rs = synth.simulate_reads(30000, n_reads=16, seed=77, mean_len=5000, sd_len=1200, min_len=800, max_len=9000)
    fasta = fasta_of(rs)
    reads = [r.tobytes() for r in rs.reads]
    rc = [synth.revcomp(r).tobytes() for r in rs.reads]
    fw = ref.dsoft_candidates(reads, reads, seed_size=SEED_SIZE)
    rv = ref.dsoft_candidates(reads, rc, seed_size=SEED_SIZE)
    cands = []
    for k in range(len(reads)):                          # darwin.cpp:209-288: per read, forward then reverse complement
        cands += [[c[0], k, c[1], c[2], 0] for c in fw[k]]
        cands += [[c[0], k, c[1], c[2], 1] for c in rv[k]]
    json.dump({"source": "SeedPosTable::DSOFT (reference seed_pos_table.cpp:100-167) via oracle/_ref, decode "
                         "darwin.cpp:213-224", "seed_size": SEED_SIZE, "fasta": fasta, "candidates": cands},
              open(os.path.join(HERE, "dsoft.json"), "w"), separators=(",", ":"))

    cpu = os.path.join(ROOT, "oracle", "_ref", "darwin_cpu")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "reads.fasta"), "w").write(fasta)
        open(os.path.join(d, "params.cfg"), "w").write(CFG % SEED_SIZE)
        subprocess.check_call([cpu, "reads.fasta", "reads.fasta", "2"], cwd=d, stdout=subprocess.DEVNULL)
        lines = []
        for p in glob.glob(os.path.join(d, "darwin.*.out")):
            lines += open(p).read().splitlines()
    json.dump({"source": "oracle/_ref/darwin_cpu (the reference's CPU build, g++ on its own sources) on "
                         "tests/golden/dsoft.json's FASTA, 2 threads", "seed_size": SEED_SIZE,
               "lines_sorted": sorted(lines)},
              open(os.path.join(HERE, "e2e.json"), "w"), separators=(",", ":"))
    print("dsoft candidates:", len(cands), "e2e lines:", len(lines))


if __name__ == "__main__":
    main()
