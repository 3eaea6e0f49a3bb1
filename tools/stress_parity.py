"""Randomised parity sweep: random tile geometry, thresholds, scorings, read sets with and without N, candidate
lists with false and edge hits -- every record field against the oracle, in whichever kernels the engine picks
plus the forced variants.  python tools/stress_parity.py [n_configs] [seed]
STRESS_RAW_ONLY=1: every read set with N / lower case (the raw-byte kernels only).  STRESS_MIXED=1: a quarter of the reads
dirty (per-candidate routing, side lane), tiles beyond 512 now and then (gact_big.hpp), and both strands launched on two
slots at once (runs in flight, the layout hint)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "darwin-gpu_amd"), os.path.join(ROOT, "oracle")]
import numpy as np
import oracle_py
from gact_amd import engine, synth

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
RAW_ONLY = os.environ.get("STRESS_RAW_ONLY")      # every read set with N (and lower case): the raw-byte kernels only
MIXED = os.environ.get("STRESS_MIXED")
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
orc = oracle_py.Oracle()
FIELDS = ("ref_id", "query_id", "ab", "ae", "bb", "be", "score", "comp", "emitted", "first_tile_score", "n_tiles", "cells")
MODES = [{}, {"GACT_HIP_FORCE_WIDE": "1"}, {"GACT_HIP_NO_WIDE": "1"}, {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_NO_TAGGED": "1"},
         {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_FORCE_UNIFORM": "1"},
         {"GACT_HIP_FORCE_INT32_SEED": "1"}, {"GACT_HIP_FORCE_INT32": "1"},
         {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_NO_LIN": "1"}, {"GACT_HIP_FORCE_WIDE": "1", "GACT_HIP_NO_LIN": "1"},
         # round 4: the tagged affine pass of round 1 instead of the drifted one; a narrow band (second runs) and none
         {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_NO_AFF": "1", "GACT_HIP_NO_LIN": "1"},
         {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_BAND": "24"}, {"GACT_HIP_BAND": "0"},
         {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_NO_LIN": "1", "GACT_HIP_BAND": "24"},
         # round 5: two banks of tiles per wave with cooperative, batched walks (gact_coop.hpp), also with a narrow band's second
         # runs; the role launch (gact_roles.hpp).  Both only where the split linear-gap launch would run
         {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_COOP": "1"}, {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_COOP": "1", "GACT_HIP_BAND": "24"},
         {"GACT_HIP_NO_WIDE": "1", "GACT_HIP_ROLES": "1"}]
if os.environ.get("STRESS_ROUND5_ONLY"):             # the three round-5 selections alone (and the automatic one)
    MODES = MODES[:1] + MODES[-3:]
ALL = sorted({k for m in MODES for k in m})
t0 = time.time()
total = 0
layouts = {}
for it in range(n_cfg):
    tile = int(rng.choice([64, 96, 128, 200, 256, 320, 320, 320, 384, 512]))
    overlap = int(rng.integers(0, max(1, tile - 16)))
    if rng.random() < 0.4:                       # the reference's geometry class: split layout, tagged pointers
        tile, overlap = 320, int(rng.integers(112, 300))
    big = bool(MIXED) and rng.random() < 0.15
    if big:
        tile = int(rng.choice([513, 600, 1024, 1500, 2048]))
        overlap = int(rng.integers(0, tile // 2))
    thr = int(rng.integers(1, 70))
    match = int(rng.integers(1, 7))
    scoring = (match, -int(rng.integers(0, 8)), -int(rng.integers(0, 12)), -int(rng.integers(0, 6)))
    if rng.random() < 0.6:                       # gap_open <= gap_extend, as scorings are in practice: the drifted affine pass
        ext = -int(rng.integers(0, 5))
        scoring = (match, -int(rng.integers(0, 8)), ext - int(rng.integers(0, 9)), ext)
    if rng.random() < 0.5:                       # linear gaps (open == extend == mismatch): the drifted pass
        g = -int(rng.choice([0, 1, 1, 1, 2, 3, 5, 9]))
        scoring = (match, g, g, g)
    n_frac = float(rng.choice([0.0, 0.0, 0.004])) if RAW_ONLY is None else 0.004
    if MIXED:
        n_frac = 0.0
    rs = synth.simulate_reads(int(rng.integers(6000, 20000)), n_reads=int(rng.integers(6, 16)), seed=int(rng.integers(1 << 30)),
                              mean_len=int(rng.integers(1500, 5000)), sd_len=900, min_len=200, max_len=9000, n_frac=n_frac)
    if n_frac > 0 and rng.random() < 0.5:       # soft-masked stretches: lower case compares unequal to upper (align.cpp:134)
        for r in rs.reads:
            a = int(rng.integers(0, max(1, len(r) - 40)))
            r[a:a + 30] = np.frombuffer(bytes(r[a:a + 30]).lower(), dtype=np.uint8)
    if MIXED:                                    # a few dirty reads among clean ones
        for k in range(rs.n):
            if rng.random() < 0.25 and len(rs.reads[k]) > 120:
                r = rs.reads[k]
                a = int(rng.integers(0, len(r) - 60))
                r[a:a + int(rng.integers(1, 30))] = ord("N")
                if rng.random() < 0.5:
                    b = int(rng.integers(0, len(r) - 60))
                    r[b:b + 40] = np.frombuffer(bytes(r[b:b + 40]).lower(), dtype=np.uint8)
    cf, cr = synth.synth_candidates(rs, seed=int(rng.integers(1 << 30)), min_overlap=150,
                                    false_frac=float(rng.choice([0.0, 0.3])))
    cat, offs = rs.concat()
    rcat, roffs = rs.concat(rc=True)
    want = {}
    for comp, cands, qcat, qoffs in ((False, cf, cat, offs), (True, cr, rcat, roffs)):
        if len(cands):
            want[comp], _ = orc.gact_many(cat, offs, qcat, qoffs, cands, complement=comp, same_file=True, tile_size=tile,
                                          tile_overlap=overlap, threshold=thr, scoring=scoring, n_threads=16)
    for mode in (MODES[:1] if big else MODES):
        for k in ALL:
            os.environ.pop(k, None)
        os.environ.update(mode)
        eng = engine.Engine(tile_size=tile, tile_overlap=overlap, scoring=scoring, threshold=thr, n_slots=2)
        eng.upload(engine.SET_REF, cat, offs)
        eng.upload(engine.SET_QUERY, cat, offs)
        eng.upload(engine.SET_QUERY_RC, rcat, roffs)
        in_flight = {}
        if MIXED:                                # both strands launched before either is fetched, a slot each
            for slot, (comp, cands) in enumerate(((False, cf), (True, cr))):
                if len(cands):
                    eng.candidates_upload(cands, slot=slot)
                    eng.candidates_run(len(cands), complement=comp, same_file=True, slot=slot)
            for slot, (comp, cands) in enumerate(((False, cf), (True, cr))):
                if len(cands):
                    in_flight[comp] = (eng.candidates_fetch(len(cands), slot=slot).copy(), eng.last_run_stats(slot))
        for comp, cands in ((False, cf), (True, cr)):
            if not len(cands):
                continue
            if MIXED:
                got, st = in_flight[comp]
            else:
                got = eng.extend(cands, complement=comp, same_file=True)
                st = eng.last_run_stats()
            key = ("big" if big else st["layout"] + ("-lin" if st["linear_gap"] else "-aff" if st.get("affine_drift") else "") +
                   ("-coop" if st.get("coop_walks") else "-roles" if st.get("role_waves") else "") + "/" + st["seed_layout"]) + (
                "+routed" if st["raw_candidates"] else "")
            layouts[key] = layouts.get(key, 0) + 1
            for f in FIELDS:
                if not np.array_equal(got[f], want[comp][f]):
                    bad = int(np.flatnonzero(got[f] != want[comp][f])[0])
                    print("MISMATCH config %d tile %d overlap %d thr %d scoring %s n_frac %g mode %s field %s cand %d"
                          % (it, tile, overlap, thr, scoring, n_frac, mode, f, bad))
                    print(" hip", got[bad], "\n ora", want[comp][bad])
                    sys.exit(1)
            total += len(cands)
        eng.close()
    if it % 10 == 9:
        print("config %d/%d ok, %d candidate runs so far, %.0f s" % (it + 1, n_cfg, total, time.time() - t0), flush=True)
print("stress parity: %d configurations x %d kernel selections, %d candidate runs, all BIT-EXACT; launches by kernels: %s"
      % (n_cfg, len(MODES), total, layouts))
