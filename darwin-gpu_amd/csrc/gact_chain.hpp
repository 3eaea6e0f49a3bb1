// gact_chain.hpp -- the tile-chain state machine of GACT() (gact.cpp:48-228),
// shared by the int32 and the packed-int16 chain kernels.  One ChainState per
// candidate; it is also the hand-off record between the seed launch (first
// tiles) and the main launch.
#pragma once

#include "gact_device.hpp"

namespace gact {

struct ChainState {
    int cand;            // index into cands, -1 = none
    int ref_id, query_id;
    int ref_len, query_len;       // whole-read lengths
    int64_t rbase, qbase;         // concat offsets of the two reads
    int ref_pos, query_pos;
    int rev_ref_pos, rev_query_pos;
    int abpos, bbpos;
    int i, j;                     // gact.cpp's i (query steps) / j (ref steps) of the last tile
    int first_tile;               // gact.cpp:79
    int first_tile_score;
    int phase;                    // 0 left, 1 right, 2 done
    int brk;                      // threshold `break` pending (gact.cpp:107-109,168-170)
    // rescoring (gact.cpp:197-210) folded into the walk, see DESIGN.md 3.5
    int score;
    int pend_gap;                 // leftmost emitted column is a gap whose cost is not charged yet
    int have_left, left_first_gap;
    int open_flag;                // right phase: the reference's `open`
    int n_tiles;
    int comp;                     // candidate aligns against the reverse-complemented query set
    int full;                     // linear-gap passes: the next tile stores its whole pointer window (its banded run gave up)
    int64_t cells;
};

struct TilePick {
    bool have;
    int R, Q;
    bool reverse;                 // AlignWithBT's `reverse`: true in the right phase (gact.cpp:155)
    int64_t rp0, qp0;             // concat positions of the two tile slices
};

// A run's candidates [first, first + n) are forward-strand below index rc_from and reverse-complement from it on
// (GACT_calls_for / GACT_calls_rev, darwin.cpp:227-238,266-277).  A MERGED run (several callers' lists gathered into one
// array, gact_engine.hip) passes kCompInCand instead: every candidate carries its strand in bit 30 of query_id.
constexpr int kCompInCand = -2;
constexpr int kCompBit = 1 << 30;
__device__ __forceinline__ int cand_strand(const gact_candidate &c, int cand, int rc_from, int &query_id)
{
    const bool tagged = rc_from == kCompInCand;
    query_id = tagged ? (c.query_id & (kCompBit - 1)) : c.query_id;
    return tagged ? ((c.query_id >> 30) & 1) : (cand >= rc_from ? 1 : 0);
}

// darwin.cpp:227-238 + gact.cpp:57-79
__device__ __forceinline__ void chain_begin(ChainState &s, int cand, const gact_candidate &c,
                                            const SeqSetDev &refs, const SeqSetDev &qfwd, const SeqSetDev &qrc,
                                            int rc_from)
{
    s.cand = cand;
    int query_id;
    s.comp = cand_strand(c, cand, rc_from, query_id);          // darwin.cpp:279 passes rev_reads_char
    const SeqSetDev &cq = s.comp ? qrc : qfwd;
    s.ref_id = c.ref_id; s.query_id = query_id;
    s.rbase = refs.offsets[c.ref_id];
    s.qbase = cq.offsets[query_id];
    s.ref_len = (int)(refs.offsets[c.ref_id + 1] - s.rbase);
    s.query_len = (int)(cq.offsets[query_id + 1] - s.qbase);
    s.ref_pos = c.ref_pos; s.query_pos = c.query_pos;
    s.rev_ref_pos = c.ref_pos; s.rev_query_pos = c.query_pos;   // gact.cpp:72-73
    s.abpos = 0; s.bbpos = 0;
    s.i = 0; s.j = 0; s.first_tile = 1; s.first_tile_score = 0;
    s.phase = 0; s.brk = 0;
    s.score = 0; s.pend_gap = 0; s.have_left = 0; s.left_first_gap = 0; s.open_flag = 1;
    s.n_tiles = 0; s.cells = 0; s.full = 0;
}

__device__ __forceinline__ void chain_write_record(const ChainState &s, int same_file, gact_overlap *out)
{
    gact_overlap o;
    o.ref_id = s.ref_id; o.query_id = s.query_id;
    o.ab = s.abpos; o.ae = s.ref_pos; o.bb = s.bbpos; o.be = s.query_pos;       // gact.cpp:219-222
    o.score = s.score; o.comp = s.comp;
    o.emitted = (!(same_file && s.ref_id == s.query_id) && s.score > 0) ? 1 : 0;  // :213
    o.first_tile_score = s.first_tile_score;
    o.n_tiles = s.n_tiles; o.reserved = 0; o.cells = s.cells;
    out[s.cand] = o;
}

// Next tile of a candidate that is in phase 0 or 1; walks the left->right switch
// (gact.cpp:136-141) and the end of the chain.  On return either pick.have, or
// s.phase == 2 (finished: the record has been written by `writer` lanes).
__device__ __forceinline__ TilePick chain_pick(ChainState &s, const KParams &kp, int same_file,
                                               gact_overlap *out, bool writer)
{
    const int tile = kp.tile_size;
    TilePick p;
    p.have = false; p.R = 0; p.Q = 0; p.reverse = false; p.rp0 = 0; p.qp0 = 0;
    if (s.phase == 0) {
        // gact.cpp:82
        if (!s.brk && s.ref_pos > 0 && s.query_pos > 0 && ((s.i > 0 && s.j > 0) || s.first_tile)) {
            p.R = (s.ref_pos > tile) ? tile : s.ref_pos;           // :84-85
            p.Q = (s.query_pos > tile) ? tile : s.query_pos;
            p.reverse = false;
            p.rp0 = s.rbase + s.ref_pos - p.R;
            p.qp0 = s.qbase + s.query_pos - p.Q;
            p.have = true;
            return p;
        }
        // leftmost column has no predecessor: a gap there costs gap_open (open==true at :198)
        if (s.pend_gap) s.score += kp.open;
        s.pend_gap = 0;
        s.abpos = s.ref_pos; s.bbpos = s.query_pos;               // :136-141
        s.ref_pos = s.rev_ref_pos; s.query_pos = s.rev_query_pos;
        s.i = tile; s.j = tile;
        s.open_flag = !(s.have_left && s.left_first_gap);
        s.phase = 1; s.brk = 0;
    }
    if (s.phase == 1) {
        // gact.cpp:144
        if (!s.brk && s.ref_pos < s.ref_len && s.query_pos < s.query_len &&
            ((s.i > 0 && s.j > 0) || s.first_tile)) {
            p.R = (s.ref_pos + tile < s.ref_len) ? tile : s.ref_len - s.ref_pos;       // :146-147
            p.Q = (s.query_pos + tile < s.query_len) ? tile : s.query_len - s.query_pos;
            p.reverse = true;
            p.rp0 = s.rbase + s.ref_pos;
            p.qp0 = s.qbase + s.query_pos;
            p.have = true;
            return p;
        }
        if (writer) chain_write_record(s, same_file, out);
        s.phase = 2;
    }
    return p;
}

// first-tile bookkeeping (gact.cpp:99-110 / :162-171); returns true when the
// tile scored under the threshold (the reference `break`s)
__device__ __forceinline__ bool chain_first_tile(ChainState &s, const KParams &kp, int R, int Q,
                                                 int best, int bi, int bj)
{
    if (s.phase == 0) {
        s.ref_pos = s.ref_pos - R + bi;                    // :100-105
        s.query_pos = s.query_pos - Q + bj;
        s.rev_ref_pos = s.ref_pos; s.rev_query_pos = s.query_pos;
    } else {
        s.ref_pos = s.ref_pos + R - bi;                    // :163-166
        s.query_pos = s.query_pos + Q - bj;
    }
    s.first_tile_score = best;
    if (best < kp.thr) { s.brk = 1; return true; }          // :107-109 / :168-170
    return false;
}

#ifdef GACT_STAMPS
// diagnostic build, with -DGACT_STAMPS_REFILL on top: shader clocks spent in the walker's region refills.  Each
// reading of the clock is a scalar memory round trip of its own (~1 k clocks): the sum is an upper bound that is
// mostly the measurement, and it lengthens the walk it is taken in -- off unless asked for.
__device__ unsigned long long g_refill_clocks;
// look-ahead walker: walks, team iterations, team refills, loop trips of the wave (= its slowest team), columns
__device__ unsigned long long g_walk_counts[5];
#endif

// running rescoring state of one candidate while its states stream by
struct ScoreWalk {
    int score, pend_gap, open_flag, have_left, left_first_gap;
    __device__ __forceinline__ void load(const ChainState &s)
    {
        score = s.score; pend_gap = s.pend_gap; open_flag = s.open_flag;
        have_left = s.have_left; left_first_gap = s.left_first_gap;
    }
    // one alignment column; sub = substitution score of an M column
    __device__ __forceinline__ void column(int phase, bool gap, int sub, const KParams &kp)
    {
        if (phase == 0) {
            // columns arrive right-to-left; the previously emitted one now learns its left neighbour
            if (pend_gap) score += gap ? kp.ext : kp.open;
            if (!have_left) { have_left = 1; left_first_gap = gap; }
            if (gap) pend_gap = 1; else { score += sub; pend_gap = 0; }
        } else {
            if (gap) { score += open_flag ? kp.open : kp.ext; open_flag = 0; }
            else { score += sub; open_flag = 1; }
        }
    }
};

// The walker of the linear-gap pass (FMT 3 words, gact_lin.hpp: gap_open == gap_extend == mismatch =: g): the pointer is the op code alone,
// op = which of M, H_up + g, H_left + g made H (ties M, then I, then D -- the order of align.cpp:162-164).
//   * No flags.  align.cpp:218-229 leaves INSERT at (i, j) for MATCH iff M[i-1][j] >= I[i-1][j].  The walk is in
//     INSERT at (i, j) only when I[i][j] > M[i][j]; were D the strict maximum at (i-1, j), then
//     I[i][j] < D[i-1][j] + g <= H[i-1][j-1] + 2g <= H[i-1][j-1] + mismatch <= M[i][j].  So the op of (i-1, j) is
//     MATCH or INSERT there, and says exactly what the flag says; DELETE likewise with (i, j-1).  The next state is
//     the op of the cell the walk enters, whatever the move.  (tools/lin_walk_model.py checks the rule against
//     the oracle on random tiles; tests/test_gpu_chain.py the kernels.)
//   * No ZERO: H == 0 is the one case the tagged max cannot tell from M == H.  The walker carries the score of the
//     cell it stands on, v (H of the start cell comes from the pass: v0), and every move takes the move's own
//     score off it -- a MATCH step the substitution score of its cell, an INSERT / DELETE step g; after a diagonal
//     move v is H of the new cell, and ZERO means v == 0 (align.cpp:166-168: M <= 0, I <= 0, D <= 0, i.e. H == 0).
// Position arithmetic, region cache and arguments as in walk_chain below.
#ifndef GACT_LIN_WALK_SPAN
#define GACT_LIN_WALK_SPAN 8
#endif
// moves between two refills of the one-lane linear-gap walker: 8, or 16 with tb_refill_oct16 (-DGACT_LIN_WALK_SPAN=16: measured
// SLOWER, 7,218-7,321 GCUPS against 7,599-7,646 on ecoli10x with four steps in flight, main launch 31.7-32.4 ms against 30.2-30.9
// -- eighteen 16-byte loads per walker and refill cost more than the round trips they save; profiles/r04/walk_span16_ab.txt)
constexpr int kLinWalkSpan = GACT_LIN_WALK_SPAN;
static_assert(kLinWalkSpan == 8 || kLinWalkSpan == 16, "GACT_LIN_WALK_SPAN");
template <int CW, int QN, int ROW>
__device__ __forceinline__ void walk_chain_lin(const uint32_t *ws, uint32_t *scratch, int R, int Q, int l0, int c0, int k0,
                                               int early, const uint8_t *rrow, int rstride, const uint8_t *qrow,
                                               const KParams &kp, ScoreWalk &wk, int &ref_steps, int &query_steps,
                                               int &nst, int v0, const uint32_t *ws_all, const int band_lim, bool &redo)
{
    constexpr uint32_t kMagic = (65536u + CW - 1) / CW;         // p / CW == (p * kMagic) >> 16 for p < 6000
    constexpr uint32_t kM = 3u, kI = 2u, kD = 1u;               // align.h:23 numbering, as the pass tags them
    const int p0 = l0 * CW + c0, kA = k0 - l0;
    const int nlim_i = -imin(early, R), nlim_j = -imin(early, Q);
    typedef __attribute__((address_space(3))) const uint8_t LdsByte;
    typedef __attribute__((address_space(3))) const uint32_t LdsWord;
    LdsByte *ra = (LdsByte *)(rrow + (R - 1) * rstride);
    LdsByte *qa = (LdsByte *)(qrow + (Q - 1));
    LdsByte *cache = (LdsByte *)scratch;
    const uint32_t ws_off = (uint32_t)((const char *)ws - (const char *)ws_all);   // the workspace is under 4 GB (engine)
    int nis = 0, njs = 0;                                       // minus the ref / query steps taken
    uint32_t cur = 0;
    TbRegion<CW> rg;
    int off0 = 0, off1 = 0;                                     // byte offsets inside the region cache
#ifdef GACT_STAMPS_REFILL
    unsigned long long rf_clk = 0;
#endif
    // region cache: dword (c >> 1) + 8 * (k >> 3) of a lane's cached blocks x two cached octets
    // (kLinWalkSpan 16: three lanes x three blocks x both octets, one refill per sixteen moves -- tb_refill_oct16;
    //  8: round 2's two lanes x two blocks x two octets, one per eight)
    int off2 = 0;
    auto refill = [&](int l, int c, int k) {
#ifdef GACT_STAMPS_REFILL
        struct Acc { unsigned long long &sum, t0; __device__ ~Acc() { sum += __builtin_amdgcn_s_memtime() - t0; } } acc_{rf_clk, __builtin_amdgcn_s_memtime()};
#endif
        if constexpr (kLinWalkSpan == 16) {
            tb_refill_oct16<CW, QN, ROW>(ws_all, ws_off, scratch, l, k, rg);
            off0 = -32 * rg.fbase[0]; off1 = 96 - 32 * rg.fbase[1]; off2 = 192 - 32 * rg.fbase[2];
        } else {
            tb_refill_oct<CW, QN, ROW>(ws_all, ws_off, scratch, l, c, k, rg);
            off0 = 4 * (-8 * rg.fbase[0] - 4 * rg.qbase0);
            off1 = 4 * (16 - 8 * rg.fbase[1] - 4 * (QN - 2));
        }
    };
    auto fetch = [&](int l, int c, int k) {
        const int off = kLinWalkSpan == 16 ? (l == rg.l0 ? off0 : l == rg.l0 - 1 ? off1 : off2) : (l == rg.l0 ? off0 : off1);
        const uint32_t row = ((uint32_t)k >> 3 << 5) + (uint32_t)off;
        const uint32_t w = *(LdsWord *)(cache + (((uint32_t)c >> 1 << 2) + row));
        return __builtin_amdgcn_ubfe(w, (((uint32_t)c & 1u) << 4) + 14u - (((uint32_t)k & 7u) << 1), 2u);
    };
    // the three column scores in VGPRs (a select between kp's fields themselves is turned into an indexed load
    // from the kernel argument segment, one memory round trip per step).  s_nop: two wait states between the
    // v_readfirstlane_b32 that writes the SGPR and the VALU instruction that reads it -- inside an asm statement the
    // compiler does not place them
    int v = v0, v_gap, v_mism, v_match;
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_gap) : "s"(__builtin_amdgcn_readfirstlane(kp.ext)));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_mism) : "s"(__builtin_amdgcn_readfirstlane(kp.mismatch)));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_match) : "s"(__builtin_amdgcn_readfirstlane(kp.match)));
    uint32_t rbase = 0, qbase = 0;                              // bases of the current cell
    bool go = false;
    if (R >= 1 && Q >= 1 && early > 0) {
        refill(l0, c0, k0);
        cur = fetch(l0, c0, k0);
        go = v != 0;
        rbase = ra[0]; qbase = qa[0];
    }
    for (int it = 0; go; it++) {
        // ---- one alignment column (gact.cpp:115-130 / :176-191 and :202-209): its score comes off v.  With
        //      gap_open == gap_extend that running sum is also all the rescoring of gact.cpp:197-210 needs
        const bool diag = cur == kM;
        const int sub = rbase == qbase ? v_match : v_mism;
        v -= diag ? sub : v_gap;
        // ---- move (align.cpp:210-229)
        nis -= cur != kD;
        njs -= cur != kI;
        // (a walker that is about to stop may have left the tile: keep its addresses inside the stored window)
        const int p = imax(p0 + njs, 0);
        const int l = (int)(__umul24((uint32_t)p, kMagic) >> 16);       // 24-bit multiplies: full rate
        const int c = p + __mul24(l, -CW);
        const int k = imax(kA + l + nis, 0);
        if ((it & (kLinWalkSpan - 1)) == kLinWalkSpan - 1) {
            refill(l, c, k);
            // banded stores (gact_lin.hpp LinBand): words are there for |di - dj| <= band.  The moves up to the next refill
            // change di - dj by their number at most: further out than band - kLinWalkSpan here, the walk gives up and its
            // tile is run again with every block stored (band_lim = band - kLinWalkSpan; negative: the tile stored everything)
            redo = redo | ((band_lim >= 0) & ((unsigned)(nis - njs + band_lim) > (unsigned)(2 * band_lim)));
        }
        // the op and the bases of the cell just entered: three LDS reads in flight together.  The op is the next
        // state whatever the move was (see above); ZERO is only asked for after a diagonal move (align.cpp:211-212
        // against :219, :225); the step limit and the borders: align.cpp:205, :101-107
        cur = fetch(l, c, k);
        rbase = ra[nis * rstride]; qbase = qa[njs];
        go = !((diag && v == 0) || nis <= nlim_i || njs <= nlim_j || redo);
    }
#ifdef GACT_STAMPS_REFILL
    if ((threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_refill_clocks, rf_clk);
#endif
    ref_steps = -nis; query_steps = -njs;
    nst = -nis - njs;                                            // only "were there any columns" is asked (chain_advance)
    wk.score += v0 - v;                                          // what the columns of this tile scored
    // (open == extend: the gap bookkeeping of wk decides nothing)
}

// ---------------------------------------------------------------------------
// The look-ahead walker of the linear-gap pointer format: the same walk as walk_chain_lin, by a TEAM of eight lanes.
//
// A walker step is a whole wave instruction however few lanes walk, and a step is ~50 dependent instructions and an LDS
// round trip: ~700 clocks, 213 steps per tile -- 18 % of a wave's time in the split layout, 40 % in the wide one, with
// two lanes of sixteen (thirty-two) at work.  But the walk is mostly diagonal: a MATCH step is left only where the
// alignment has an indel (13 % of the columns of the PacBio-shape reads, 8 % of the ONT-shape ones), and along a
// diagonal nothing depends on the step before except the running score.  So lane m of the team looks at the cell m
// steps up the diagonal from the walk's cell -- its op, its two bases -- and the team settles how far the MATCH run
// goes in one iteration:
//   * cell m can be taken as a MATCH column iff it and every cell before it hold op M and the walk did not stop on
//     the way: after cell m the score is v - S_m (S_m = the substitution scores of cells 0..m: a count of equal-base
//     cells among them, from one ballot), the walk stops there iff that is 0 (ZERO, align.cpp:166-168 -- only asked
//     after a diagonal move) or a step limit / border is reached (align.cpp:205, :101-107);
//   * the first lane that is not M, or stops, ends the run: n = its index (+1 when
//     it stops: the stopping cell is processed) MATCH columns are consumed at once, every lane updates the same
//     state from the same ballots -- no lane-to-lane data movement at all;
//   * a cell that holds INSERT / DELETE at the head is one gap step, as in walk_chain_lin.
// Semantics are walk_chain_lin's, step for step (same cells, same order, same stop tests); only how many steps one
// trip through the loop takes differs.
// Region cache: rows i-SPAN..i x columns j-SPAN..j of the anchor cell (SPAN 32 in the main launch: up to four or five
// lanes x five flush blocks x all the lane's column octets = 40-50 uint4; SPAN 16 = 18 uint4 in the seed launch),
// loaded by the team together (a few 16-byte loads per lane, ONE memory round trip) whenever the head has moved more
// than SPAN - 7 rows or columns from the anchor, so eight cells of look-ahead are always inside.  The round trip is
// what a walk is made of now (the words of a 200 x 200 window are 14 KB per tile, 43 MB per XCD in flight: they come
// back from beyond the L2, ~3,500 clocks each time), and every team of the wave takes its refill at the same trip.
constexpr int kLaTeam = 8;
constexpr int kLaBandMargin = 12, kWalkBandMargin = 8;      // how far inside the stored band a walk must be when it refills (team / one lane;
                                                            // the one-lane linear-gap walker: kLinWalkSpan)
#ifndef GACT_WALK_SYNC_REFILL
#define GACT_WALK_SYNC_REFILL 1
#endif
// SPAN: the cached region is rows i-SPAN..i x columns j-SPAN..j of the anchor cell
template <int CW, int QN, int SPAN> struct LaRegion {
    static constexpr int NL = (SPAN + CW - 1) / CW + 1;    // lanes SPAN + 1 adjacent columns can touch
    static constexpr int NB = SPAN / 8 + 1;                // flush blocks SPAN + 1 adjacent stored steps can touch
    static constexpr int kUint4 = NL * NB * QN;
    static constexpr int kLoads = (kUint4 + kLaTeam - 1) / kLaTeam;    // 16-byte loads per lane and refill
    static constexpr int kWords = kUint4 * 4;              // dwords of LDS scratch per walker
    static constexpr int kRefill = SPAN - (kLaTeam - 1);   // the head may be this far from the anchor when a trip begins
    static_assert(SPAN % 8 == 0 && SPAN >= 16, "whole flush blocks");
};

template <int CW, int QN, int ROW, int SPAN>
__device__ __forceinline__ void walk_chain_lin_team(uint32_t *scratch, const bool active, int R, int Q, int l0, int c0, int k0,
                                                    int early, const uint8_t *rrow, int rstride, const uint8_t *qrow,
                                                    const KParams &kp, ScoreWalk &wk, int &ref_steps, int &query_steps,
                                                    int &nst, int v0, const uint32_t *ws, const uint32_t *ws_all,
                                                    const int band_lim, bool &redo)
{
    using RG = LaRegion<CW, QN, SPAN>;
    constexpr uint32_t kMagic = (65536u + CW - 1) / CW;         // p / CW == (p * kMagic) >> 16 for p < 6000
    constexpr uint32_t kM = 3u, kI = 2u;                        // align.h:23 numbering, as the pass tags them
    constexpr int kOct = 16 * ROW, kBlk = 16 * QN * ROW;        // byte strides of a column octet, of a flush block
    constexpr int kRow = 16 * QN;                               // bytes of one cached (lane, flush block)
    const int lane = threadIdx.x & 63, m = lane & (kLaTeam - 1);
    const uint32_t team_shift = (uint32_t)lane & 56u;
    auto team_bits = [team_shift](uint64_t mask) { return (uint32_t)(mask >> team_shift) & 0xffu; };
    const int p0 = l0 * CW + c0, kA = k0 - l0;
    const int nlim_i = -imin(early, R), nlim_j = -imin(early, Q);
    typedef __attribute__((address_space(3))) const uint8_t LdsByte;
    typedef __attribute__((address_space(3))) const uint32_t LdsWord;
    LdsByte *ra = (LdsByte *)(rrow + (R - 1) * rstride);
    LdsByte *qa = (LdsByte *)(qrow + (Q - 1));
    LdsByte *cache = (LdsByte *)scratch;
    const uint32_t ws_off = (uint32_t)((const char *)ws - (const char *)ws_all);   // the workspace is under 4 GB (engine)

    // this lane's loads of a refill: uint4 n = m, m + 8, ... of the region, n = (slot * NB + block) * QN + octet
    // (slot s = lane l_anchor - s; its first cached block is f(s) = max(((k_anchor - s) >> 3) - (NB - 1), 0))
    int ld_n[RG::kLoads], ld_slot[RG::kLoads], ld_stat[RG::kLoads];
#pragma unroll
    for (int r = 0; r < RG::kLoads; r++) {
        const int n = imin(m + kLaTeam * r, RG::kUint4 - 1);
        const int slot = n / (RG::NB * QN), rem = n - slot * (RG::NB * QN), blk = rem / QN, oct = rem - blk * QN;
        ld_n[r] = n; ld_slot[r] = slot; ld_stat[r] = blk * kBlk + oct * kOct;
    }
    int la = 0, ka = 0;                                         // anchor: lane and stored step (in that lane)
    auto first_block = [](int k_lane) { return imax((k_lane >> 3) - (RG::NB - 1), 0); };
    auto refill = [&](int l_a, int k_a) {
        u32x4 q[RG::kLoads];
#pragma unroll
        for (int r = 0; r < RG::kLoads; r++) {
            const int sl = ld_slot[r];
            const uint32_t a = ws_off + (uint32_t)(first_block(k_a - sl) * kBlk + ld_stat[r] + imax(l_a - sl, 0) * 16);
            // (s_nop: the base may have just been written by a VALU instruction -- a v_readlane_b32 out of a spill lane --
            // and a memory instruction must not read such an SGPR for five wait states)
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 sc1" : "=&v"(q[r]) : "v"(a), "s"(ws_all) : "memory");
        }
        if constexpr (RG::kLoads == 3)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]) :: "memory");
        else if constexpr (RG::kLoads == 5)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]) :: "memory");
        else {
            static_assert(RG::kLoads == 7, "refill sizes in use");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]) :: "memory");
        }
        u32x4 *dst = reinterpret_cast<u32x4 *>(scratch);
#pragma unroll
        for (int r = 0; r < RG::kLoads; r++) dst[ld_n[r]] = q[r];
        wave_sync();
        la = l_a; ka = k_a;
    };
    // the three column scores in VGPRs (see walk_chain_lin)
    int v = v0, v_gap, v_mism, v_dsub;
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_gap) : "s"(__builtin_amdgcn_readfirstlane(kp.ext)));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_mism) : "s"(__builtin_amdgcn_readfirstlane(kp.mismatch)));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_dsub) : "s"(__builtin_amdgcn_readfirstlane(kp.match - kp.mismatch)));
    int nis = 0, njs = 0;                                       // minus the ref / query steps taken (the head cell)
    int di = 0, dj = 0;                                         // rows / columns the head has moved from the anchor
    bool go = active && R >= 1 && Q >= 1 && early > 0 && v0 != 0;
#ifdef GACT_STAMPS
    int n_it = 0, n_rf = 0;
    const bool walked = go;
#endif
#ifdef GACT_STAMPS_REFILL
    unsigned long long rf_clk = 0;
#endif
    if (go) refill(l0, k0);
    const uint32_t below = (2u << m) - 1u;                      // cells 0..m of the diagonal
    while (go) {
#ifdef GACT_STAMPS
        n_it++;
#endif
        // ---- this lane's cell: m steps up the diagonal from the head.  The head is at most RG::kRefill rows and columns
        //      from the anchor here, so the cell is inside the cached region
        const int ci = nis - m, cj = njs - m;
        // (a cell beyond the end of the walk may lie outside the tile: keep its addresses inside the stored window)
        const int p = imax(p0 + cj, 0);
        const int l = (int)(__umul24((uint32_t)p, kMagic) >> 16);
        const int c = p + __mul24(l, -CW);
        const int k = imax(kA + l + ci, 0);
        const int sl = la - l;
        const uint32_t at = (uint32_t)(__mul24(__mul24(sl, RG::NB) - first_block(ka - sl) + (k >> 3), kRow) + ((c >> 1) << 2));
        const uint32_t w = *(LdsWord *)(cache + at);
        const uint32_t rb = ra[ci * rstride], qb = qa[cj];
        const uint32_t op = __builtin_amdgcn_ubfe(w, (((uint32_t)c & 1u) << 4) + 14u - (((uint32_t)k & 7u) << 1), 2u);
        const bool is_m = op == kM;
        // ---- the run of MATCH columns from the head (align.cpp:210-217, gact.cpp:115-130 / :176-191); written
        //      without branches: every lane evaluates both kinds of head and selects
        const uint32_t eq_bits = team_bits(lanes(rb == qb));
        const int s_m = __mul24(m + 1, v_mism) + __mul24((int)__builtin_popcount(eq_bits & below), v_dsub);
        const bool stops = is_m & ((v - s_m == 0) | (ci - 1 <= nlim_i) | (cj - 1 <= nlim_j));
        const uint32_t gap_bits = team_bits(lanes(!is_m)), stop_bits = team_bits(lanes(stops));
        const uint32_t i_bits = team_bits(lanes(op == kI));
        // first lane that is not M / that stops; a lane cannot be both, and 8 means none
        const int e_gap = __builtin_ctz(gap_bits | 0x100u), e_stop = __builtin_ctz(stop_bits | 0x100u);
        const bool halted = e_stop < e_gap;
        const int n_run = halted ? e_stop + 1 : e_gap;          // MATCH columns consumed (the stopping cell is one of them)
        const bool head_m = !(gap_bits & 1u), ins = i_bits & 1u;
        // one gap column (align.cpp:218-229) when the head is not M: its score comes off v, no ZERO test after it
        const int n_i = head_m ? n_run : (int)ins, n_j = head_m ? n_run : 1 - (int)ins;
        const int dv_run = __mul24(n_run, v_mism) + __mul24((int)__builtin_popcount(eq_bits & ((1u << n_run) - 1u)), v_dsub);
        v -= head_m ? dv_run : v_gap;
        nis -= n_i; njs -= n_j; di += n_i; dj += n_j;
        go = head_m ? !halted : !((nis <= nlim_i) | (njs <= nlim_j));
#if GACT_WALK_SYNC_REFILL
        // every team of the wave re-anchors when one has to: a refill is a memory round trip that the whole wave waits
        // for, so the teams take theirs together
        if (__any(go && imax(di, dj) > RG::kRefill) && go) {
#else
        if (go && imax(di, dj) > RG::kRefill) {
#endif
            const int ph = imax(p0 + njs, 0);
            const int lh = (int)(__umul24((uint32_t)ph, kMagic) >> 16);
#ifdef GACT_STAMPS_REFILL
            const unsigned long long rf_t0 = __builtin_amdgcn_s_memtime();
#endif
            // banded stores (see walk_chain_lin): between two re-anchorings the head moves RG::kRefill + 1 rows or columns
            // at most, so di - dj changes by no more than that (band_lim = band - kLaBandMargin)
            if ((band_lim >= 0) & ((unsigned)(nis - njs + band_lim) > (unsigned)(2 * band_lim))) { redo = true; go = false; }
            refill(lh, imax(kA + lh + nis, 0));
#ifdef GACT_STAMPS_REFILL
            rf_clk += __builtin_amdgcn_s_memtime() - rf_t0;
#endif
            di = 0; dj = 0;
#ifdef GACT_STAMPS
            n_rf++;
#endif
        }
    }
#ifdef GACT_STAMPS
    {
        int trips = n_it;
        for (int mm = 1; mm < 64; mm <<= 1) trips = imax(trips, __shfl_xor(trips, mm, 64));
        if (walked && m == 0) {
            atomicAdd(&g_walk_counts[0], 1ull); atomicAdd(&g_walk_counts[1], (unsigned long long)n_it);
            atomicAdd(&g_walk_counts[2], (unsigned long long)n_rf); atomicAdd(&g_walk_counts[4], (unsigned long long)(-nis - njs));
        }
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_walk_counts[3], (unsigned long long)trips);
#ifdef GACT_STAMPS_REFILL
        // (the longest-waiting lane of the wave: refills are taken by all teams together)
        for (int mm = 1; mm < 64; mm <<= 1) { const unsigned long long o = __shfl_xor(rf_clk, mm, 64); rf_clk = o > rf_clk ? o : rf_clk; }
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_refill_clocks, rf_clk);
#endif
    }
#endif
    ref_steps = -nis; query_steps = -njs;
    nst = -nis - njs;                                            // only "were there any columns" is asked (chain_advance)
    wk.score += v0 - v;                                          // what the columns of this tile scored
    // (open == extend: the gap bookkeeping of wk decides nothing)
}

// The chain kernels' walker: traceback (align.cpp:185-230) fused with the rescoring
// of gact.cpp:197-210, written for few instructions per step -- every step of a
// walker is a whole wave instruction however few lanes walk, and a tile's walk is
// ~200 steps.
//   * the only loop-carried position state is the two (negated) step counts; lane,
//     column-in-lane and stored step of the current cell follow from them by one
//     multiply-shift division, so a move is two conditional decrements;
//   * the state machine runs on the packed kernel's op code (0 ZERO 1 MATCH 2 INSERT
//     3 DELETE) and its inverted flags {ins_open<ins_extend, del_open<del_extend};
//     the step limit (align.cpp:205) and the borders (:101-107) fold into the state,
//     so the loop has one exit;
//   * rescoring counts events (gap-extend columns, gap-open columns, equal-base MATCH
//     columns, MATCH columns) and prices them once at the end.  With g(x) = "column x is a
//     gap", a and b two columns adjacent in emission order: the right phase charges b when
//     g(b), `extend` iff g(a); the left phase (columns arrive right to left) charges a when
//     g(a), `extend` iff g(b).  Either way: g(a)&g(b) -> extend, else X -> open with
//     X = g(a) in the left phase, g(b) in the right one;
//   * pointer words come from the region cache (gact_device.hpp); inside a region the
//     word of (lane, column, step) sits at c + 12*(k>>3) + off[lane == anchor ? 0 : 1].
// rrow/qrow point at the LDS byte of DP row 1 / column 1; rstride is the ref stream's
// byte stride.  (l0, c0, k0) = lane, column-in-lane and stored step of the start cell
// (R, Q) in the pass's layout; CW columns per lane, QN column quads stored per lane.
//
template <int CW, int FMT, int QN = CW / 4, int ROW = kGroup>
__device__ __forceinline__ void walk_chain(const uint32_t *ws, uint32_t *scratch, int R, int Q, int l0, int c0, int k0,
                                           int early, const uint8_t *rrow, int rstride, const uint8_t *qrow,
                                           int phase, const KParams &kp, ScoreWalk &wk, int &ref_steps,
                                           int &query_steps, int &nst, int v0 = 0, const uint32_t *ws_all = nullptr,
                                           const int band_lim = -1, bool *redo = nullptr)
{
    if constexpr (FMT == 3) {
        bool rd = false;
        walk_chain_lin<CW, QN, ROW>(ws, scratch, R, Q, l0, c0, k0, early, rrow, rstride, qrow, kp, wk, ref_steps,
                                      query_steps, nst, v0, ws_all, band_lim, rd);
        if (redo) *redo = rd;
        return;
    }
    // FMT 4 (the drifted affine pass, gact_aff.hpp): FMT 2's word layout and op numbering (3 MATCH 2 INSERT 1 DELETE), but
    //   * the two flags come as ONE code, (I'' + D'') & 3 of the pass: 2 = both gaps open here, 0 = the insertion only,
    //     1 = the deletion only, 3 = neither;
    //   * ZERO is not encoded (H == 0 reads as MATCH): the walker carries v, the score of the cell it stands on in the
    //     matrix its state names -- v0 = H[R][Q] from the pass; a MATCH column takes its substitution score off, a gap
    //     column gap_open where its cell's flag says the gap was opened there, gap_extend where not (align.cpp:149-156) --
    //     and after a diagonal move v is H of the cell entered: ZERO iff 0 (align.cpp:166-168), exactly as walk_chain_lin.
    constexpr bool AFF = FMT == 4;
    constexpr uint32_t kMagic = (65536u + CW - 1) / CW;         // p / CW == (p * kMagic) >> 16 for p < 6000
    const bool left = phase == 0;
    const int p0 = l0 * CW + c0, kA = k0 - l0;
    const int nlim_i = -imin(early, R), nlim_j = -imin(early, Q);
    const uint8_t *ra = rrow + (R - 1) * rstride;
    const uint8_t *qa = qrow + (Q - 1);
    int nis = 0, njs = 0;                                       // minus the ref / query steps taken
    int n_ext = 0, n_open = 0, n_eq = 0, n_m = 0;
    int v = v0;
    bool rd = false;
    uint32_t cur = 0, fl = 0;                                   // state (op-code numbering), flags of the current cell
    TbRegion<CW> rg;
    int off0 = 0, off1 = 0;                                     // byte offsets inside the region cache

    auto refill = [&](int l, int c, int k) {
        tb_refill_at<CW, QN, ROW>(ws, scratch, l, c, k, rg);
        off0 = 4 * (-12 * rg.fbase[0] - 4 * rg.qbase0);
        off1 = 4 * (24 - 12 * rg.fbase[1] - 4 * (QN - 3));
    };
    // 32-bit LDS addressing (through the generic pointer the index arithmetic is done in 64 bits)
    typedef __attribute__((address_space(3))) const uint8_t LdsByte;
    typedef __attribute__((address_space(3))) const uint32_t LdsWord;
    LdsByte *cache = (LdsByte *)scratch;
    auto fetch = [&](int l, int c, int k, uint32_t &code, uint32_t &flags) {
        const uint32_t at = (uint32_t)(4 * c + (int)__umul24((uint32_t)k >> 3, 48u) + (l == rg.l0 ? off0 : off1));
        const uint32_t w = *(LdsWord *)(cache + at);
        if (FMT == 1) {
            const uint32_t v = w >> ((~(uint32_t)k & 7u) * 2u);
            code = v & 3u;
            flags = (v >> 16) & 3u;
        } else if (FMT == 2 || AFF) {                            // taken as it is: the walk runs on this numbering
            const uint32_t v = w >> ((~(uint32_t)k & 7u) * 2u);
            code = v & 3u;
            flags = (v >> 16) & 3u;
        } else {
            const uint32_t nib = w >> ((~(uint32_t)k & 7u) * 4u);
            const uint32_t op = nib & 3u;                        // align.h:23 numbering Z0 D1 I2 M3
            code = op ? 4u - op : 0u;
            flags = (~nib >> 2) & 3u;
        }
    };

    if (R >= 1 && Q >= 1 && early > 0) {
        refill(l0, c0, k0);
        fetch(l0, c0, k0, cur, fl);
        if (AFF && v == 0) cur = 0;                              // H[R][Q] == 0: ZERO at the start cell
    }
    // state numbering of the walk: FMT 0 / 1 words are turned into the packed kernel's op codes (1 MATCH 2 INSERT
    // 3 DELETE, flag set = the gap goes on); FMT 2 words carry align.h:23 numbering (3 MATCH 2 INSERT 1 DELETE) and
    // flags that say the opposite (set = the gap was opened here), and the walk uses them as they are
    constexpr uint32_t kM = (FMT == 2 || AFF) ? 3u : 1u, kI = 2u, kD = (FMT == 2 || AFF) ? 1u : 3u;
    if (left && !wk.have_left && cur != 0) { wk.have_left = 1; wk.left_first_gap = cur != kM; }
    // conditions live as lane masks on the scalar unit; a counter takes one as the carry of a single VALU op
    const uint64_t left_m = lanes(left);
    uint64_t gprev = lanes(left ? wk.pend_gap != 0 : wk.open_flag == 0);
    for (int it = 0; cur != 0; it++) {
        // ---- one alignment column (gact.cpp:115-130 / :176-191 and :202-209)
        const uint64_t g = lanes(cur != kM);
        const uint64_t eq = lanes(ra[nis * rstride] == qa[njs]);
        n_ext = add_lane_bit(n_ext, gprev & g);
        n_open = add_lane_bit(n_open, (gprev ^ g) & ((left_m & gprev) | (~left_m & g)));
        n_m = add_lane_bit(n_m, ~g);
        n_eq = add_lane_bit(n_eq, ~g & eq);
        gprev = (gprev & ~lanes(true)) | g;                       // walkers that have stopped keep their last column
        // ---- move (align.cpp:210-229): INSERT / DELETE stay unless their flag says the gap was opened here
        nis = sub_lane_bit(nis, lanes(cur != kD));
        njs = sub_lane_bit(njs, lanes(cur != kI));
        uint32_t forced;
        if (AFF) {
            // the gap of the current cell was opened here: INSERT 2 and code 2 or 0, DELETE 1 and code 2 or 1
            const bool opened = cur == kI ? !(fl & 1u) : (((fl + 1u) & 2u) != 0);
            forced = opened ? kM : cur;
            const int sub = ((eq >> (threadIdx.x & 63)) & 1) ? kp.match : kp.mismatch;
            v -= cur == kM ? sub : (opened ? kp.open : kp.ext);
        } else {
            forced = FMT == 2 ? ((fl & cur) ? kM : cur) : ((fl & (4u - cur)) ? cur : kM);
        }
        // (a walker that is about to stop may have left the tile: keep its addresses inside the stored window)
        const int p = imax(p0 + njs, 0);
        const int l = (int)(__umul24((uint32_t)p, kMagic) >> 16);       // 24-bit multiplies: full rate
        const int c = p + __mul24(l, -CW);
        const int k = imax(kA + l + nis, 0);
        if ((it & 7) == 7) {
            refill(l, c, k);
            // banded stores (gact_lin.hpp LinBand), as in walk_chain_lin
            rd = rd | ((band_lim >= 0) & ((unsigned)(nis - njs + band_lim) > (unsigned)(2 * band_lim)));
        }
        uint32_t code;
        fetch(l, c, k, code, fl);
        const uint32_t nxt = cur == kM ? ((AFF && v == 0) ? 0u : code) : forced;
        cur = (nis <= nlim_i || njs <= nlim_j || rd) ? 0u : nxt;       // align.cpp:205, borders :101-107
    }
    if (redo) *redo = rd;
    ref_steps = -nis; query_steps = -njs;
    nst = -nis - njs - n_m;
    wk.score += n_ext * kp.ext + n_open * kp.open + n_eq * kp.match + (n_m - n_eq) * kp.mismatch;
    const bool last_gap = (gprev >> (threadIdx.x & 63)) & 1;
    wk.pend_gap = left ? (int)last_gap : wk.pend_gap;
    wk.open_flag = left ? wk.open_flag : (int)!last_gap;
}

// after the traceback (gact.cpp:111-133 / :172-194); src = lane holding the walk's results
template <int LANES = kGroup>
__device__ __forceinline__ void chain_advance(ChainState &s, bool stop, const ScoreWalk &wk, int ref_steps,
                                              int query_steps, int nst, int src)
{
    ref_steps = __shfl(ref_steps, src, LANES);
    query_steps = __shfl(query_steps, src, LANES);
    nst = __shfl(nst, src, LANES);
    s.score = __shfl(wk.score, src, LANES);
    s.pend_gap = __shfl(wk.pend_gap, src, LANES);
    s.open_flag = __shfl(wk.open_flag, src, LANES);
    s.have_left = __shfl(wk.have_left, src, LANES);
    s.left_first_gap = __shfl(wk.left_first_gap, src, LANES);
    if (nst > 0) s.first_tile = 0;                                   // :112 / :173
    s.i = query_steps; s.j = ref_steps;                              // gact.cpp's i counts query bases
    if (!stop) {
        if (s.phase == 0) { s.ref_pos -= ref_steps; s.query_pos -= query_steps; }   // :132-133
        else              { s.ref_pos += ref_steps; s.query_pos += query_steps; }   // :193-194
    } else {
        s.i = 0; s.j = 0;
    }
}

// device-side bookkeeping shared by the seed and the main launch.  The seed launch
// files every candidate it hands off under its estimated remaining chain length;
// the main launch pops the longest class first.  Once the queues are empty the
// launch lasts as long as the chains popped last, so the classes are one tile
// wide at the short end (measured with 16 classes of 8 tiles: queues empty at
// 61 ms, last wave done at 73 ms) and coarser where only the order matters:
// 0..15 tiles one class each, 16..47 in fours, 48..111 in eights, 112..607 in sixteens, longer in one.  (With
// everything beyond 112 tiles in one class, the 50-100 kb reads of the ONT-shape workload -- chains of up to 500
// tiles -- were popped in no particular order, and a 500-tile chain could start last.)
constexpr int kBuckets = 64;
__host__ __device__ constexpr int length_class(int tiles)      // ascending with the length, 0 .. kBuckets-1
{
    return tiles < 16 ? (tiles < 0 ? 0 : tiles) : tiles < 48 ? 16 + (tiles - 16) / 4 : tiles < 112 ? 24 + (tiles - 48) / 8
         : tiles < 608 ? 32 + (tiles - 112) / 16 : 63;
}
struct ChainQueues {
    int *pop_seed;               // next candidate index for the seed launch
    int *bucket_count;           // [kBuckets] candidates handed to the main launch, bucket 0 = longest
    int *bucket_pop;             // [kBuckets] next index to pop per bucket
    unsigned long long *seed_cells;   // DP cells executed by the seed launch
    int *live;                   // [kBuckets][live_stride] candidate ids handed off
    int live_stride;
    ChainState *states;          // their chain states
    int *longest_now;            // [kEpochs] longest remaining chain (bases) any wave reported, per time slice
    // routing by read content (align.cpp:134 compares raw bytes: N == N, case matters): route_kernel (gact_kernels.hpp)
    // sorts the candidates of a run into those whose two reads are plain A/C/G/T and the rest; a seed launch then takes
    // one of the two lists
    int *band_redos;             // tiles run a second time because their walk left the stored band (gact_lin.hpp LinBand)
    const int *list_count;       // null: the seed launch takes the candidates [first, first + n) themselves
    const int *list;             // candidate indices
    int list_n;                  // >= 0: length of `list`, known on the host (ordered seeding); -1: *list_count
    // Overlapped seeding (gact_engine.hip run_overlapped): a main launch that starts while a second seed launch is still
    // filing chains into a SECOND set of queues.  Once its own set is empty a wave looks at *more_flag (written in stream
    // order behind that seed launch), and when it is set goes on with the second set.
    const int *more_flag;        // null: this launch has one set of queues
    int *more_count, *more_pop;  // [kBuckets] each
    int *more_live;
    // The critical lane (gact_engine.hip run_pass): beside a split main launch a second, WIDE main launch on a third of the
    // blocks pops the same queues -- a wave of it carries four tiles instead of eight and advances its chains about twice
    // as fast.  Both pop longest-first; the split launch leaves the longest `leave_longest` chains of its (first) set to the
    // lane -- it starts popping behind their length classes and comes back to them when it has nothing else.  0: no lane.
    int leave_longest;
};

// next candidate of a seed launch into s (group-uniform; `leader` = the group's lane 0 does the atomic, bcast = a
// functor that broadcasts its value over the group).  Returns false when the list is exhausted.
template <class Bcast>
__device__ __forceinline__ bool seed_pop(ChainState &s, const ChainQueues &cq, bool leader, Bcast bcast, const gact_candidate *cands,
                                         int first_cand, int n, int rc_from, const SeqSetDev &refs, const SeqSetDev &qfwd,
                                         const SeqSetDev &qrc)
{
    const int total = cq.list_n >= 0 ? cq.list_n : cq.list_count ? *cq.list_count : n;
    int idx = 0;
    if (leader) idx = atomicAdd(cq.pop_seed, 1);
    idx = bcast(idx);
    if (idx >= total) return false;
    const int cand = cq.list ? cq.list[idx] : first_cand + idx;
    chain_begin(s, cand, cands[cand], refs, qfwd, qrc, rc_from);
    return true;
}

// The main launch's waves rank themselves against the longest chain still running anywhere: every wave posts
// its longest remaining chain into the slice of the 100 MHz clock it is in and reads the previous slice
// (complete) and the current one (so far).  Slices are 1.3 ms, the ring covers 84 ms and is recycled two
// slices ahead.
constexpr int kEpochs = 64;
__device__ __forceinline__ int longest_running(const ChainQueues &cq, int mine, bool writer)
{
    const unsigned e = (unsigned)(__builtin_amdgcn_s_memrealtime() >> 17);
    int ref = mine;
    if (writer) {
        atomicMax(&cq.longest_now[e % kEpochs], mine);
        cq.longest_now[(e + 2) % kEpochs] = 0;
        ref = imax(__hip_atomic_load(&cq.longest_now[(e + kEpochs - 1) % kEpochs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                   __hip_atomic_load(&cq.longest_now[e % kEpochs], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    return ref;
}

// ... every kRankEvery-th tile only, waves taking turns: longest_running is an agent-scope atomic max, a store and two loads on ONE
// line of memory; with every wave of a launch doing that at every tile (3,072 waves, a tile per 0.3 ms: ~40 operations per
// microsecond on one address) the waves queued for it -- pacbio50mb alone 139.7 ms with, 118.4 without (round 5,
// profiles/r05/ranking_atomics.txt; with several launches in flight each has its own ring, which is most of why runs in flight
// looked faster than one run alone).  The rank is a slow quantity: a slice of the ring is 1.3 ms.
constexpr int kRankEvery = 16;
__device__ __forceinline__ int ranked_longest(const ChainQueues &cq, const KParams &kp, int wave_longest, int &turn, int &cached)
{
    if (kp.prio_bases[0] != 0) return 0;                      // fixed thresholds or none: nobody reads the ring
    if ((turn & (kRankEvery - 1)) == 0) {
        cached = __builtin_amdgcn_readfirstlane(longest_running(cq, wave_longest, (threadIdx.x & 63) == 0));
        if (turn == 0) turn = (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (kRankEvery - 1));      // (from here on the waves take turns)
    }
    turn++;
    return imax(cached, wave_longest);
}

// bases this chain still has to cover, roughly (each tile advances ~early of them)
__device__ __forceinline__ int chain_remaining(const ChainState &s)
{
    if (s.phase == 0)
        return imax(0, imin(s.ref_pos, s.query_pos)) +
               imax(0, imin(s.ref_len - s.rev_ref_pos, s.query_len - s.rev_query_pos));
    return imax(0, imin(s.ref_len - s.ref_pos, s.query_len - s.query_pos));
}

__device__ __forceinline__ int chain_bucket(const ChainState &s, const KParams &kp)
{
    const int tiles = chain_remaining(s) / imax(kp.early, 1);
    return kBuckets - 1 - length_class(tiles);
}

}  // namespace gact
