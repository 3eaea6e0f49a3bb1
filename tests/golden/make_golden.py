#!/usr/bin/env python
"""Generates tests/golden/*.json from the REFERENCE ITSELF (oracle/_ref/libdarwin_ref.so =
/root/reference/align.cpp + gact.cpp compiled unchanged, see oracle/Makefile).

Run in the build container (where /root/reference is mounted):
    python tests/golden/make_golden.py
The fixtures are data only: inputs and the reference's outputs.
  tiles.json   AlignWithBT: (ref, query, scoring, reverse, first, early) -> the returned queue
  chains.json  GACT: reads + candidates -> the exact output line (or "" when nothing is printed)
"""
import json
import os
import sys

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


if __name__ == "__main__":
    main()
