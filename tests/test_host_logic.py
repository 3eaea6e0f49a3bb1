"""CPU suite: host-side pieces of the engine headers that decide which kernels run and in which order chains are
popped (compiled as a host-only program with hipcc; no device code runs)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <cstdio>
#include "gact_lin.hpp"
int main()
{
    int prev = -1, ok = 1;
    for (int t = -5; t < 4000; t++) {
        const int c = gact::length_class(t);
        if (c < prev || c < 0 || c >= gact::kBuckets) ok = 0;
        prev = c;
    }
    printf("classes %d %d %d %d %d\n", ok, gact::length_class(0), gact::length_class(15), gact::length_class(16),
           gact::length_class(100000));
    // default scoring, a scoring too large for 16 bits, one too large for the tagged form only, one for the arg-max keys only
    printf("default %d %d %d\n", gact::p16_scoring_ok(320, 1, -1, -1, -1), gact::p16_argmax_ok(320, 1),
           gact::p16_tagged_ok(320, 1, -1, -1, -1));
    printf("large %d\n", gact::p16_scoring_ok(320, 100, -90, -200, -50));
    printf("mid %d %d %d\n", gact::p16_scoring_ok(320, 30, -40, -70, -20), gact::p16_argmax_ok(320, 30),
           gact::p16_tagged_ok(320, 30, -40, -70, -20));
    printf("tile512 %d %d\n", gact::p16_scoring_ok(512, 1, -1, -1, -1), gact::p16_tagged_ok(512, 1, -1, -1, -1));
    printf("pk2 %08x %08x\n", gact::pk2(-1), gact::pk2(3));
    // linear-gap pass: open == extend == mismatch, and the drift must fit next to the scaled scores
    printf("lin %d %d %d %d %d %d\n", gact::p16_lin_ok(320, 1, -1, -1, -1), gact::p16_lin_ok(320, 1, -1, -2, -1),
           gact::p16_lin_ok(320, 2, -1, -3, -3), gact::p16_lin_ok(320, 5, -2, -2, -2), gact::p16_lin_ok(320, 3, -20, -20, -20),
           gact::p16_lin_ok(512, 1, -1, -1, -1));
    return 0;
}
'''


def test_length_classes_and_kernel_predicates(tmp_path):
    (tmp_path / "t.hip").write_text(SRC)
    exe = str(tmp_path / "t")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "darwin-gpu_amd", "csrc"), "-o", exe, str(tmp_path / "t.hip")])
    out = subprocess.check_output([exe], text=True).split("\n")
    assert out[0] == "classes 1 0 15 16 63"         # monotone, one tile wide up to 15, everything very long in the last one
    assert out[1] == "default 1 1 1"
    assert out[2] == "large 0"
    assert out[3] == "mid 1 0 0"                    # packed main kernel yes; int32 seed kernel, explicit pointer comparisons
    assert out[4] == "tile512 1 1"
    assert out[5] == "pk2 ffffffff 00030003"
    assert out[6] == "lin 1 0 0 1 0 1"


# ---- the launch policy (csrc/gact_policy.hpp through gact_hip_plan_describe: no device) --------------------------------
def _plan(count, **kw):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "darwin-gpu_amd"))
    from gact_amd import engine
    return engine.plan(count, **kw)


def test_launch_policy_operating_points():
    """the four sequences of a pass, at the thresholds DESIGN 3.5 names (S = 24,576 resident tile slots on 256 CUs)"""
    S = 3 * 256 * 32
    for n, seq, main in ((1, "seed+main", "WideLayoutLin"), (S // 2, "seed+main", "WideLayoutLin"), (S, "seed+main", "WideLayoutLin"),
                         (S + 1, "seed+main+critical-lane", "SplitLayoutLin"), (S + S // 2 - 1, "seed+main+critical-lane", "SplitLayoutLin"),
                         (S + S // 2, "overlapped-seeding", "SplitLayoutLin"), (4 * S, "overlapped-seeding", "SplitLayoutLin"),
                         (4 * S + 1, "seed+main", "SplitLayoutLin"), (6 * S - 1, "seed+main", "SplitLayoutLin"),
                         (6 * S, "seed+main", "coop<SplitLayoutLin>"), (3_000_000, "seed+main", "coop<SplitLayoutLin>")):
        p = _plan(n)
        assert (p["sequence"], p["main_kernel"]) == (seq, main), (n, p)
        assert p["seed_kernel"] == "seed_p16<lin>" and p["linear"] and not p["roles"]
        # "overlap_big" 1 (flag 32): overlapped seeding beyond four chains per slot too, seed launch A = the longest eighth
        q = _plan(n, flags=32)
        assert q["sequence"] == ("overlapped-seeding" if n > 4 * S else seq) and q["main_kernel"] == main, (n, q)
        if main == "WideLayoutLin":         # two wide blocks per CU at most (16 tiles each)
            assert p["main_blocks"] == min(-(-n // 16), 512), (n, p)
        # "lone_lane" 48 (flag 64): S/2 ... S chains as the lone mix -- 256 blocks on 256 CUs, 48 of them wide, holding the 768 longest
        m = _plan(n, flags=64)
        if S // 2 < n <= S:
            assert (m["sequence"], m["main_kernel"]) == ("seed+main+critical-lane", "SplitLayoutLinTeam"), (n, m)
            assert m["main_blocks"] + m["second_main_blocks"] == 256 and m["second_main_blocks"] == 48 and m["leave_longest"] == 768
        else:
            assert (m["sequence"], m["main_kernel"]) == (seq, main), (n, m)
        if n > 4 * S:
            assert q["n_a"] == max(2 * S, n // 8) and q["main_blocks"] == 512 and q["second_main_blocks"] == 256
    # a launch that shares the machine: the throughput layout -- two banks per wave, cooperative walks -- on two thirds of the
    # blocks, no second stream, whatever the count; "coop" 0 keeps the one-wave-does-all launch
    for n in (100, S, 2 * S, 10 * S):
        p = _plan(n, flags=2)
        want = "coop<SplitLayoutLin>" if n >= S + S // 2 else "SplitLayoutLin"      # (few, long chains: one bank per wave)
        assert p["sequence"] == "seed+main" and p["main_kernel"] == want and p["main_blocks"] <= 512, (n, p)
        # (two thirds of the blocks while every chain gets a tile slot there, half of them beyond)
        assert p["main_blocks"] == (min(-(-n // 32), 512) if n <= 512 * 32 else 384), (n, p)
        assert _plan(n, flags=2 | 16)["main_kernel"] == "SplitLayoutLin"
    assert _plan(2 * S, flags=8)["main_kernel"] == "coop<SplitLayoutLin>" and _plan(2 * S, flags=8)["sequence"] == "overlapped-seeding"
    # the role launch: one block of twelve waves per CU, in the same sequences
    p = _plan(2 * S, flags=4)
    assert p["main_kernel"] == "roles<SplitLayoutLin>" and p["main_blocks"] + p["second_main_blocks"] == 256 and p["roles"]
    assert _plan(10 * S, flags=4)["main_blocks"] == 256 and _plan(10 * S, flags=4)["sequence"] == "seed+main"
    # other scorings and raw-byte sets: the drifted affine pass / the tagged raw kernels, never the linear-gap sequences
    p = _plan(2 * S, scoring=(2, -3, -5, -2))
    assert p["sequence"] == "seed+main" and p["main_kernel"] == "SplitLayoutAff<cbneg>" and p["affine_drift"]
    assert _plan(2 * S, scoring=(3, -2, -4, -2))["main_kernel"] == "SplitLayoutAff"
    assert _plan(2 * S, flags=1)["main_kernel"] == "SplitLayout<tag,raw>" and _plan(2 * S, flags=1)["seed_kernel"] == "seed_p16<raw>"
    assert _plan(2 * S, scoring=(100, -90, -200, -50))["sequence"] == "int32-one-launch"
    assert _plan(2 * S, tile_size=512, tile_overlap=128)["main_kernel"].startswith("UniformLayout")
    assert "gact_big" in _plan(10, tile_size=1024, tile_overlap=256)["sequence"]


def test_launch_policy_is_sane_over_a_sweep_of_counts():
    """1 ... 3 M candidates: grids never exceed the machine, never shrink as the list grows inside one sequence, every
    candidate of a list up to the resident tile slots has a slot of its own, and the two launches of a split sequence
    share the workspace without overlap"""
    import math
    S, cus = 3 * 256 * 32, 256
    ws_total = 3 * cus * 32 * 13760                     # words: what gact_hip_create allocates per slot for these grids
    prev = None
    n = 1
    while n <= 3_000_000:
        for flags in (0, 2, 4, 32, 64):
            p = _plan(n, flags=flags)
            assert 1 <= p["seed_blocks"] <= 3 * cus and 1 <= p["main_blocks"] <= 3 * cus and p["second_main_blocks"] <= cus
            per_block = 16 if p["wide"] else 160 if p["roles"] else 64 if p["coop"] else 32
            slots = (p["main_blocks"] + (p["second_main_blocks"] if not p["critical_lane"] else 0)) * per_block
            if p["critical_lane"]:
                slots += p["second_main_blocks"] * 16
            # (a role block spreads one bank's worth per block first; the wide launch deliberately takes ONE block per CU once
            #  it has more chains than two blocks per CU hold: bound by throughput either way)
            #  ... and a launch that shares the machine two blocks per CU of the three)
            if not p["wide"]:
                #  (the critical lane's third of the blocks holds 16 tiles a block, not 32)
                floor_slots = cus * 80 if p["roles"] else (S * 2 // 3 if n <= S * 2 // 3 else S // 2) if flags == 2 else S * 5 // 6 if p["critical_lane"] else S
                if p["main_kernel"] == "SplitLayoutLinTeam":     # (the lone mix is ONE block per CU by design)
                    floor_slots = (cus - 48) * 32 + 48 * 16
                assert slots >= min(n, floor_slots), (n, flags, p)
            if p["sequence"] in ("overlapped-seeding", "seed+main+critical-lane"):
                second = p["second_main_blocks"] * (16 if p["critical_lane"] else per_block) * 13760
                if p["coop"] and not p["critical_lane"]:     # (a cooperative block lays out its two banks' region-2 words only)
                    second = p["second_main_blocks"] * (p["ws_split_words"] // p["main_blocks"])
                assert 0 < p["ws_split_words"] and (p["roles"] or p["ws_split_words"] + second <= ws_total), (n, p)
            if p["sequence"] == "overlapped-seeding":
                assert 0 < p["n_a"] <= n and p["seed_b_blocks"] >= 1
        q = _plan(n)
        if prev is not None and prev["sequence"] == q["sequence"] and prev["main_kernel"] == q["main_kernel"] and not q["wide"]:
            assert q["main_blocks"] >= prev["main_blocks"], (n, prev, q)
        prev = q
        n = int(math.ceil(n * 1.37))
