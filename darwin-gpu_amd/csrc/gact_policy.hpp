// gact_policy.hpp -- WHAT a pass over `count` candidates runs as: which seed kernel, which main kernel, on how many blocks,
// in which sequence.  A pure function of counts and of the engine's capabilities -- no HIP call, no engine, no device --
// so that it can be read, swept and unit-tested on the CPU (tests/test_host_logic.py, gact_hip_plan_describe), apart from
// the code that launches (gact_engine.hip launch_extend).  VERDICT r04 #8: until round 4 the decisions were spread over
// 364 lines of launch_extend between the launches themselves.
//
// The operating points, in candidates per resident tile slot of the narrow layouts (S = lin_grid_blocks x 32; 24,576 on
// an MI355X), each from a measurement:
//   count <= S (and the engine idle)     all wide: every chain resident from the start, the launch lasts its longest chain
//   S < count < 1.5 S                     split launch + critical lane (a wide launch on a third of the blocks, longest chains)
//   1.5 S <= count <= 4 S                 ordered, overlapped seeding: seed A | main 1 (2/3) || seed B -> main 2 (1/3)
//   count > 4 S                           seed launch, then one main launch on the whole machine; from 6 S on in the cooperative
//                                         layout.  (overlap_big: overlapped seeding here too, seed A = the longest eighth of the
//                                         list.  pacbio50mb alone 141.1 -> 139.3 ms, +1.3 %: not the default -- seed launch B has
//                                         a third of the machine, and where chains are short it, not the DP, bounds the run)
// A launch that shares the machine (other slots running, or the caller says it keeps runs in flight) takes the plain
// sequence in the throughput layout on two thirds of the blocks -- on half of them when its list is longer than those hold.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <string>

namespace gact_policy {

constexpr int kWavesPerBlock = 4, kGroupsPerWave = 4, kSlots = 2;
constexpr int kNarrowTilesPerBlock = kWavesPerBlock * kGroupsPerWave * kSlots;       // 32
constexpr int kWideTilesPerBlock = kWavesPerBlock * 2 * kSlots;                       // 16

struct Caps {                    // fixed at gact_hip_create
    int C = 20;
    bool p16 = false, seed16 = false, lin = false, aff = false, aff_seed = true, split = false, tagged = false;
    bool mismatch_below_extend = false;
    bool roles = false, overlap_seed = true, crit_lane = true, crit_lane_always = false, lane_small = false, team_when_shared = false;
    int shared_twelfths = 6;     // a linear-gap launch that shares the machine takes this many twelfths of the resident blocks (see lin_cap)
    int lone_lane = 0;           // n != 0: a run of S/2 ... S chains alone on the machine runs as ONE block per CU of
                                 // two kinds: |n| wide blocks for its longest chains, split blocks on the other CUs (see plan_pass); n < 0: the
                                 // split blocks without the look-ahead walker; 0 (default): all wide
    bool overlap_big = false;    // overlapped seeding also for runs of more than four chains per tile slot (see plan_pass)
    int lane_small_factor = 3, lane_blocks = 0;
    int wide = 0;                // 0 auto, 1 always, -1 never
    int coop = 0;                // cooperative, batched walks (gact_coop.hpp): 0 auto -- where the launch is bound by throughput: it shares
                                 // the machine and has 1.5 chains and more per tile slot, or has six and more --, 1 always, -1 never
    int wide_blocks_per_cu = 0;
    int cus = 256;
    int grid_blocks = 0, seed_grid_blocks = 0, seed_lin_grid_blocks = 0, lin_grid_blocks = 0, aff_grid_blocks = 0,
        wide_lin_grid_blocks = 0, role_grid_blocks = 0;
    int role_dp_waves = 10;      // DP waves of a role block (gact_roles.hpp kRoleDp)
    size_t ws_words_per_tile = 0, role_ws_words_per_block = 0, coop_ws_words_per_block = 0;
};

struct Inputs {                  // of one pass
    int count = 0;
    bool raw = false;            // compared as raw bytes (sets with N / lower case)
    bool listed = false;         // the seed launch takes a list (routing)
    bool second_set = false, shared_machine = false, own_lane = true, trace = false, poison = false;
    int lane_max_blocks = 0;     // the side lane's cap (0: none)
};

enum class Seq { SingleInt32, Plain, Overlapped, CritLane };
enum class SeedK { Int32, P16Raw, P16, P16Lin, P16Aff, P16AffNeg };
enum class MainK { None, RolesLin, CoopLin, SplitLin, SplitLinTeam, WideLin, SplitAffNeg, SplitAff, WideTaggedRaw, WideTagged, WideRaw, Wide,
                   SplitTaggedRaw, SplitTagged, SplitRaw, Split, UniformTaggedRaw, UniformTagged, UniformRaw, Uniform };

struct Plan {
    Seq seq = Seq::Plain;
    SeedK seed = SeedK::Int32;
    MainK main = MainK::None;
    bool wide = false, lin = false, aff = false, roles = false, coop = false;
    int seed_blocks = 0, main_blocks = 0;
    // Overlapped: seed A takes the longest nA of the ordered list; main 1 / seed B + main 2 side by side
    int nA = 0, seedB_blocks = 0, main2_blocks = 0;
    bool lane = false;           // Overlapped: main 2 is the critical lane (wide); CritLane: always
    int leave_longest = 0;       // tiles the lane holds: the split launch leaves it that many of the longest chains
    size_t ws_split = 0;         // words of workspace in front of the second launch's share
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t ws_words_for(const Caps &c, int blocks) { return (size_t)blocks * kNarrowTilesPerBlock * c.ws_words_per_tile; }

inline Plan plan_pass(const Caps &c, const Inputs &in)
{
    Plan p;
    const int count = in.count;
    auto grid = [&](int needed, int cap) { return std::max(1, std::min(needed, in.lane_max_blocks ? std::min(cap, in.lane_max_blocks) : cap)); };
    const int groups_needed = c.p16 ? ceil_div(count, 2 * kGroupsPerWave) : ceil_div(count, kGroupsPerWave);
    const int seed_waves = ceil_div(count, kGroupsPerWave);
    const int seed_blocks_i32 = grid(ceil_div(seed_waves, 4), c.grid_blocks);
    const int main_blocks = grid(ceil_div(groups_needed, 4), c.grid_blocks);
    const int narrow_slots0 = c.lin_grid_blocks * kNarrowTilesPerBlock;
    const bool c20 = c.C == 20;

    // ---- ordered, overlapped seeding
    if (c20 && c.overlap_seed && !in.raw && !in.listed && !in.trace && !in.poison && c.seed16 && c.lin && c.split && c.wide <= 0 &&
        !in.second_set && !in.shared_machine && in.own_lane && in.lane_max_blocks == 0 && count >= narrow_slots0 + narrow_slots0 / 2 &&
        (count <= 4 * narrow_slots0 || c.overlap_big) && c.lin_grid_blocks >= 3) {
        p.seq = Seq::Overlapped;
        p.seed = SeedK::P16Lin;
        p.lin = true;
        p.roles = c.roles && c.role_grid_blocks > 0;
        const int main1 = c.lin_grid_blocks * 2 / 3, main2 = c.lin_grid_blocks - main1;
        const int role1 = c.role_grid_blocks * 2 / 3, role2 = c.role_grid_blocks - role1;
        // (a run of up to four chains per tile slot, one at a time on an idle engine: two banks per wave double a chain's time per
        //  tile, and the launch ends with its longest chains -- ecoli10x 37.3 ms against 30.6, profiles/r05/ab_coop_vs_old_first.txt)
        p.coop = !p.roles && (c.coop > 0 || (c.coop == 0 && count >= 6 * narrow_slots0));
        p.main = p.roles ? MainK::RolesLin : p.coop ? MainK::CoopLin : MainK::SplitLin;
        p.main_blocks = p.roles ? role1 : main1;
        p.main2_blocks = p.roles ? role2 : main2;
        p.ws_split = p.roles ? (size_t)role1 * c.role_ws_words_per_block : p.coop ? (size_t)main1 * c.coop_ws_words_per_block : ws_words_for(c, main1);
        // seed launch B runs on a third of the machine, and until it has ended main launch 1 gets no new chains: it is given
        // what it can seed in a few milliseconds, two candidates per resident tile slot; a larger run seeds the rest in A
        // (overlap_big, a run of more than four per slot: seed A takes two per slot or the longest eighth of the list, seed B the
        //  rest while main launch 1 works on A's -- B seeds ~9 candidates in the time main launch 1 finishes one chain of
        //  pacbio50mb's, and a wave of main launch 1 that finds nothing for 2 ms gives up)
        p.nA = count > 4 * narrow_slots0 ? std::max(2 * narrow_slots0, count / 8) : std::max(std::min(count, main1 * kNarrowTilesPerBlock), count - 2 * narrow_slots0);
        auto seed_blocks_for = [&](int cnt, int cap) { return std::max(1, std::min(ceil_div(ceil_div(cnt, 2 * kGroupsPerWave), 4), cap)); };
        p.seed_blocks = seed_blocks_for(p.nA, c.seed_lin_grid_blocks);
        p.seedB_blocks = seed_blocks_for(count - p.nA, main2);
        p.lane = !p.roles && c.crit_lane && c.crit_lane_always && c.wide == 0 && main2 <= c.wide_lin_grid_blocks && count > p.nA;
        p.leave_longest = p.lane ? main2 * kWideTilesPerBlock : 0;
        return p;
    }

    // ---- the seed launch
    if (c.seed16) {
        p.seed = in.raw ? SeedK::P16Raw : SeedK::P16;
        p.seed_blocks = grid(ceil_div(groups_needed, 4), c.seed_grid_blocks);
        if (c20 && !in.raw && c.lin) {
            p.seed = SeedK::P16Lin;
            p.seed_blocks = grid(ceil_div(groups_needed, 4), c.seed_lin_grid_blocks);
        } else if (c20 && !in.raw && c.aff && c.aff_seed) {
            p.seed = c.mismatch_below_extend ? SeedK::P16AffNeg : SeedK::P16Aff;
        }
    } else {
        p.seed = SeedK::Int32;
        p.seed_blocks = seed_blocks_i32;
    }
    if (!c.p16) { p.seq = Seq::SingleInt32; p.main = MainK::None; return p; }

    // ---- the main launch: layout
    const int narrow_slots = c.grid_blocks * kNarrowTilesPerBlock;
    const int lane_tiles_small = (c.lin_grid_blocks - c.lin_grid_blocks * 2 / 3) * kWideTilesPerBlock;
    const bool lane_small = c20 && c.lane_small && c.crit_lane && c.wide == 0 && c.lin && c.split && !in.raw && !in.trace && !in.second_set &&
                            !in.shared_machine && in.own_lane && in.lane_max_blocks == 0 && count <= narrow_slots &&
                            count >= c.lane_small_factor * lane_tiles_small / 2 && c.lin_grid_blocks >= 3;
    // fewer chains than the narrow layouts have tile slots: the launch lasts as long as its longest chain, so chains are made
    // faster (32 lanes per tile pair, 4 tiles per wave) instead of more numerous -- unless the launch shares the machine
    p.wide = c20 && c.wide >= 0 && !lane_small && (c.wide > 0 || (count <= narrow_slots && (!in.shared_machine || !in.own_lane)));
    p.lin = c.lin && !in.raw && (p.wide || c.split);
    p.aff = c.aff && !in.raw && !p.wide && c20;
    p.roles = c20 && c.roles && c.role_grid_blocks > 0 && p.lin && !p.wide && !(in.shared_machine && in.own_lane && c.team_when_shared);
    // cooperative walks: +5-8 % where throughput bounds the launch (ecoli10x with four runs in flight 7,681 -> 8,038 GCUPS,
    // pacbio50mb 7,833 -> 8,433 and its launch alone 140.1 -> 133.9 ms); a run of few chains per tile slot alone on the machine
    // ends with its longest chains and those advance half as fast with two banks per wave
    // (... and so does a launch of few, long chains even when it shares the machine: the ONT shape, 14.5 k chains of 200-500 tiles,
    //  with four runs in flight 5,055 GCUPS with two banks per wave against 6,890-7,470 with one)
    const bool coop_auto = (in.shared_machine && in.own_lane && count >= narrow_slots0 + narrow_slots0 / 2) || count >= 6 * narrow_slots0;
    p.coop = c20 && (c.coop > 0 || (c.coop == 0 && coop_auto)) && p.lin && !p.wide && !p.roles && c.split &&
             !(in.shared_machine && in.own_lane && c.team_when_shared);
    const bool tg = c.tagged, raw = in.raw;
    p.main = p.lin ? (p.wide ? MainK::WideLin
                             : p.roles ? MainK::RolesLin
                             : p.coop ? MainK::CoopLin
                                       : (in.shared_machine && in.own_lane && c.team_when_shared) ? MainK::SplitLinTeam : MainK::SplitLin)
           : p.aff ? (c.mismatch_below_extend ? MainK::SplitAffNeg : MainK::SplitAff)
           : p.wide ? (tg ? (raw ? MainK::WideTaggedRaw : MainK::WideTagged) : (raw ? MainK::WideRaw : MainK::Wide))
           : c.split ? (tg ? (raw ? MainK::SplitTaggedRaw : MainK::SplitTagged) : (raw ? MainK::SplitRaw : MainK::Split))
                     : (tg ? (raw ? MainK::UniformTaggedRaw : MainK::UniformTagged) : (raw ? MainK::UniformRaw : MainK::Uniform));
    // ---- grids
    // two waves per SIMD, not three, for the wide launch (it lasts its longest chain); ONE block per CU once the linear-gap
    // wide launch has more chains than two blocks per CU hold (bound by throughput either way)
    const int slots_at_two = 2 * c.cus * kWideTilesPerBlock;
    // (round 5, after the ranking's atomics were taken off every tile -- gact_chain.hpp ranked_longest --: TWO.  "One block per CU
    //  once the launch has more chains than two hold" was a measurement of that contention, which grew with the number of waves:
    //  ONT shape alone 72.9 / 58.8 / 74.4 ms at one / two / three blocks per CU now, 77.7 / 79.8 / 110 then; 13 k ... 24 k chains
    //  of 5 ... 100 kb reads 20-35 % faster at two than at one, profiles/r05/layout_calibration_after_the_ranking_fix.txt)
    const int per_cu = c.wide_blocks_per_cu > 0 ? c.wide_blocks_per_cu : 2;
    (void)slots_at_two;
    const int wide_cap = std::min(p.lin ? c.wide_lin_grid_blocks : c.grid_blocks, per_cu * c.cus);
    const int wide_blocks = grid(ceil_div(count, kWideTilesPerBlock), wide_cap);
    // two waves per SIMD already saturate the DP code: a launch that shares the machine takes two blocks per CU of the three
    // (... and HALF of them once the list is longer than two thirds hold: four runs in flight then put through 1.8 % more on
    //  ecoli10x, 0.5 % on pacbio50mb -- 6, 7, 8 twelfths, four interleaved runs each, profiles/r05/ab_shared_share_of_blocks.txt; a
    //  list that fits -- the ONT shape's 14.5 k long chains -- keeps a tile slot per chain)
    const int two_thirds = std::max(1, c.lin_grid_blocks * 2 / 3);
    const int lin_cap = !(in.shared_machine && in.own_lane) ? c.lin_grid_blocks
                        : count <= two_thirds * kNarrowTilesPerBlock ? two_thirds : std::max(1, c.lin_grid_blocks * c.shared_twelfths / 12);
    const int lin_blocks = grid(ceil_div(groups_needed, 4), lin_cap);
    const int main_blocks_now = p.aff ? grid(ceil_div(groups_needed, 4), c.aff_grid_blocks) : main_blocks;
    // ---- the critical lane beside ONE split main launch
    if (c20) {
        const int lane_blocks = c.lane_blocks > 0 ? std::min(c.lane_blocks, c.lin_grid_blocks / 2) : c.lin_grid_blocks - c.lin_grid_blocks * 2 / 3;
        if (!p.roles && c.crit_lane && c.wide == 0 && p.lin && !p.wide && !raw && !in.trace && !in.second_set && !in.shared_machine && in.own_lane &&
            in.lane_max_blocks == 0 && !c.team_when_shared && (count > narrow_slots || lane_small) &&
            (count < narrow_slots0 + narrow_slots0 / 2 || c.crit_lane_always) && count <= 4 * narrow_slots0 && c.lin_grid_blocks >= 3 &&
            lane_blocks <= c.wide_lin_grid_blocks) {
            p.seq = Seq::CritLane;
            p.lane = true;
            p.main2_blocks = lane_blocks;
            p.main_blocks = grid(ceil_div(groups_needed, 4), c.lin_grid_blocks - lane_blocks);
            p.ws_split = p.coop ? (size_t)(c.lin_grid_blocks - lane_blocks) * c.coop_ws_words_per_block : ws_words_for(c, c.lin_grid_blocks - lane_blocks);
            p.leave_longest = lane_blocks * kWideTilesPerBlock;
            return p;
        }
    }
    // ---- one wave per SIMD everywhere, two layouts (lone_lane).  The all-wide launch of a run with more chains than two wide blocks
    //      per CU hold runs one block per CU (a wave alone on its SIMD issues every 7 cycles, and the launch is bound by
    //      that).  The split layout executes fewer instructions per tile (7.9 k against 9.1 k), the wide one fewer per tile of ONE
    //      chain (0.108 ms against 0.185 alone on a SIMD): so the longest chains -- the ones a split wave could not finish in time --
    //      go to `lone` wide blocks and everything else to split blocks, every block alone on its CU (256 blocks on 256 CUs; the
    //      look-ahead walker pays for a wave that has its SIMD to itself).  ONT shape alone 78.6 -> 67.4 ms; 16 k ... 24 k chains of
    //      5 ... 100 kb reads +13 ... +23 %; below S/2 chains the all-wide launch is faster (profiles/r05/lone_lane_*.txt).
    //      NOT the default any more: all of that was measured against an all-wide launch at ONE block per CU, which is what the
    //      ranking's contention made the best wide launch; at two blocks per CU the all-wide launch beats the mix (ONT 58.8 against
    //      66.9 ms).  Kept as an option ("lone_lane")
    const int lone = c.lone_lane < 0 ? -c.lone_lane : c.lone_lane;
    if (c20 && lone > 0 && lone < c.cus && p.wide && p.lin && count > narrow_slots0 / 2 && c.wide == 0 && c.split && !raw && !in.trace &&
        !in.second_set && !in.shared_machine && in.own_lane && in.lane_max_blocks == 0 && lone <= c.wide_lin_grid_blocks && c.cus - lone <= c.lin_grid_blocks) {
        p.seq = Seq::CritLane;
        p.lane = true;
        p.wide = false; p.roles = false; p.coop = false;
        p.main = c.lone_lane > 0 ? MainK::SplitLinTeam : MainK::SplitLin;
        p.main2_blocks = lone;
        p.main_blocks = c.cus - lone;
        p.ws_split = ws_words_for(c, c.cus - lone);
        p.leave_longest = lone * kWideTilesPerBlock;
        return p;
    }
    p.seq = Seq::Plain;
    if (p.roles) {
        // a role block's bank holds role_dp_waves x 4 groups x 2 tiles; with fewer chains than two banks everywhere the blocks
        // are spread over the machine one bank full each before the second banks fill
        const int per_bank = c.role_dp_waves * kGroupsPerWave * kSlots;
        p.main_blocks = grid(ceil_div(count, per_bank), c.role_grid_blocks);
    } else {
        p.main_blocks = p.wide ? wide_blocks : (p.lin ? lin_blocks : main_blocks_now);
    }
    return p;
}

inline const char *name(Seq s)
{
    switch (s) { case Seq::SingleInt32: return "int32-one-launch"; case Seq::Plain: return "seed+main"; case Seq::Overlapped: return "overlapped-seeding";
                 case Seq::CritLane: return "seed+main+critical-lane"; }
    return "?";
}
inline const char *name(SeedK k)
{
    switch (k) { case SeedK::Int32: return "extend_kernel(seed)"; case SeedK::P16Raw: return "seed_p16<raw>"; case SeedK::P16: return "seed_p16";
                 case SeedK::P16Lin: return "seed_p16<lin>"; case SeedK::P16Aff: return "seed_p16<aff>"; case SeedK::P16AffNeg: return "seed_p16<aff,cbneg>"; }
    return "?";
}
inline const char *name(MainK k)
{
    switch (k) {
    case MainK::None: return "-"; case MainK::RolesLin: return "roles<SplitLayoutLin>"; case MainK::CoopLin: return "coop<SplitLayoutLin>";
    case MainK::SplitLin: return "SplitLayoutLin";
    case MainK::SplitLinTeam: return "SplitLayoutLinTeam"; case MainK::WideLin: return "WideLayoutLin"; case MainK::SplitAffNeg: return "SplitLayoutAff<cbneg>";
    case MainK::SplitAff: return "SplitLayoutAff"; case MainK::WideTaggedRaw: return "WideLayoutTagged<raw>"; case MainK::WideTagged: return "WideLayoutTagged";
    case MainK::WideRaw: return "WideLayout<raw>"; case MainK::Wide: return "WideLayout"; case MainK::SplitTaggedRaw: return "SplitLayout<tag,raw>";
    case MainK::SplitTagged: return "SplitLayout<tag>"; case MainK::SplitRaw: return "SplitLayout<raw>"; case MainK::Split: return "SplitLayout";
    case MainK::UniformTaggedRaw: return "UniformLayout<tag,raw>"; case MainK::UniformTagged: return "UniformLayout<tag>";
    case MainK::UniformRaw: return "UniformLayout<raw>"; case MainK::Uniform: return "UniformLayout";
    }
    return "?";
}
inline std::string describe(const Plan &p)
{
    char b[512];
    snprintf(b, sizeof b, "{\"sequence\": \"%s\", \"seed_kernel\": \"%s\", \"seed_blocks\": %d, \"main_kernel\": \"%s\", \"main_blocks\": %d, "
                          "\"second_main_blocks\": %d, \"seed_b_blocks\": %d, \"n_a\": %d, \"critical_lane\": %s, \"leave_longest\": %d, "
                          "\"ws_split_words\": %zu, \"wide\": %s, \"linear\": %s, \"affine_drift\": %s, \"roles\": %s, \"coop\": %s}",
             name(p.seq), name(p.seed), p.seed_blocks, name(p.main), p.main_blocks, p.main2_blocks, p.seedB_blocks, p.nA, p.lane ? "true" : "false",
             p.leave_longest, p.ws_split, p.wide ? "true" : "false", p.lin ? "true" : "false", p.aff ? "true" : "false", p.roles ? "true" : "false",
             p.coop ? "true" : "false");
    return b;
}

}  // namespace gact_policy
