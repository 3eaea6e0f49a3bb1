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
    int64_t cells;
};

struct TilePick {
    bool have;
    int R, Q;
    bool reverse;                 // AlignWithBT's `reverse`: true in the right phase (gact.cpp:155)
    int64_t rp0, qp0;             // concat positions of the two tile slices
};

// darwin.cpp:227-238 + gact.cpp:57-79
__device__ __forceinline__ void chain_begin(ChainState &s, int cand, const gact_candidate &c,
                                            const SeqSetDev &refs, const SeqSetDev &qfwd, const SeqSetDev &qrc,
                                            int rc_from)
{
    s.cand = cand;
    s.comp = (cand >= rc_from) ? 1 : 0;          // darwin.cpp:279 passes rev_reads_char
    const SeqSetDev &cq = s.comp ? qrc : qfwd;
    s.ref_id = c.ref_id; s.query_id = c.query_id;
    s.rbase = refs.offsets[c.ref_id];
    s.qbase = cq.offsets[c.query_id];
    s.ref_len = (int)(refs.offsets[c.ref_id + 1] - s.rbase);
    s.query_len = (int)(cq.offsets[c.query_id + 1] - s.qbase);
    s.ref_pos = c.ref_pos; s.query_pos = c.query_pos;
    s.rev_ref_pos = c.ref_pos; s.rev_query_pos = c.query_pos;   // gact.cpp:72-73
    s.abpos = 0; s.bbpos = 0;
    s.i = 0; s.j = 0; s.first_tile = 1; s.first_tile_score = 0;
    s.phase = 0; s.brk = 0;
    s.score = 0; s.pend_gap = 0; s.have_left = 0; s.left_first_gap = 0; s.open_flag = 1;
    s.n_tiles = 0; s.cells = 0;
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

// The chain kernels' walker: traceback (align.cpp:185-230) fused with the rescoring
// of gact.cpp:197-210, written for few instructions per step -- every step of a
// walker is a whole wave instruction however few lanes walk.  Lane / column /
// stored-step of the current cell are tracked incrementally (no division), the
// column bookkeeping is branch-free, pointer words come from the region cache.
// rrow/qrow point at the LDS byte of DP row 1 / column 1; rstride is the ref
// stream's byte stride.
// (l, c, k) = lane, column-in-lane and stored step of the start cell (R, Q) in the pass's layout;
// CW columns per lane, QN column quads stored per lane.
template <int CW, int FMT, int QN = CW / 4>
__device__ __forceinline__ void walk_chain(const uint32_t *ws, uint32_t *scratch, int R, int Q, int l, int c, int k,
                                           int early, const uint8_t *rrow, int rstride, const uint8_t *qrow,
                                           int phase, const KParams &kp, ScoreWalk &wk, int &ref_steps,
                                           int &query_steps, int &nst)
{
    int i = R, j = Q;
    int is = 0, js = 0, n = 0, since = 0;
    int state = GACT_STATE_Z;
    uint32_t nib = 0;
    TbRegion<CW> rg;
    if (R >= 1 && Q >= 1 && early > 0) {
        tb_refill_at<CW, QN>(ws, scratch, l, c, k, rg);
        nib = tb_lookup_at<CW, FMT, QN>(scratch, l, c, k, rg);
        state = nib & 3;
    }
    const bool left = phase == 0;
    int score = wk.score, pend = wk.pend_gap, openf = wk.open_flag, havel = wk.have_left, lfg = wk.left_first_gap;
    const uint8_t *ra = rrow + (R - 1) * rstride;
    const uint8_t *qa = qrow + (Q - 1);
    while (state != GACT_STATE_Z) {
        // ---- one alignment column (gact.cpp:115-130 / :176-191 and :202-209)
        const int gap = state != GACT_STATE_M;
        const int sub = (*ra == *qa) ? kp.match : kp.mismatch;
        // left phase: the previously emitted column, if a gap, now learns its left neighbour;
        // right phase: this column, if a gap, knows its left neighbour already
        const int charge = left ? pend : gap;
        const int nbr_gap = left ? gap : !openf;
        score += charge ? (nbr_gap ? kp.ext : kp.open) : 0;
        score += gap ? 0 : sub;
        lfg = (left && !havel) ? gap : lfg;
        havel = left ? 1 : havel;
        pend = left ? gap : pend;
        openf = left ? openf : !gap;
        n++;
        // ---- move (align.cpp:210-229)
        const int isM = state == GACT_STATE_M, isI = state == GACT_STATE_I;
        const int di = isM | isI, dj = isM | (state == GACT_STATE_D);
        const int next = isM ? -1 : (isI ? ((nib & 8) ? GACT_STATE_M : GACT_STATE_I)
                                         : ((nib & 4) ? GACT_STATE_M : GACT_STATE_D));
        i -= di; j -= dj; is += di; js += dj;
        ra -= di * rstride; qa -= dj;
        c -= dj;
        const int wrap = c < 0;
        c += wrap ? CW : 0; l -= wrap; k -= di + wrap;
        if (is >= early || js >= early || i < 1 || j < 1) break;       // align.cpp:205, borders :101-107
        if (++since == kTbSpan) {
            tb_refill_at<CW, QN>(ws, scratch, l, c, k, rg);
            since = 0;
        }
        nib = tb_lookup_at<CW, FMT, QN>(scratch, l, c, k, rg);
        state = (next < 0) ? (int)(nib & 3) : next;
    }
    wk.score = score; wk.pend_gap = pend; wk.open_flag = openf; wk.have_left = havel; wk.left_first_gap = lfg;
    ref_steps = is; query_steps = js; nst = n;
}

// after the traceback (gact.cpp:111-133 / :172-194); src = lane holding the walk's results
__device__ __forceinline__ void chain_advance(ChainState &s, bool stop, const ScoreWalk &wk, int ref_steps,
                                              int query_steps, int nst, int src)
{
    ref_steps = __shfl(ref_steps, src, kGroup);
    query_steps = __shfl(query_steps, src, kGroup);
    nst = __shfl(nst, src, kGroup);
    s.score = __shfl(wk.score, src, kGroup);
    s.pend_gap = __shfl(wk.pend_gap, src, kGroup);
    s.open_flag = __shfl(wk.open_flag, src, kGroup);
    s.have_left = __shfl(wk.have_left, src, kGroup);
    s.left_first_gap = __shfl(wk.left_first_gap, src, kGroup);
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
// files every candidate it hands off under its estimated remaining chain length
// (kBuckets classes of kBucketTiles tiles); the main launch pops the longest
// class first, so the tail of the launch consists of short chains only.
constexpr int kBuckets = 16;
constexpr int kBucketTiles = 8;
struct ChainQueues {
    int *pop_seed;               // next candidate index for the seed launch
    int *bucket_count;           // [kBuckets] candidates handed to the main launch, bucket 0 = longest
    int *bucket_pop;             // [kBuckets] next index to pop per bucket
    unsigned long long *seed_cells;   // DP cells executed by the seed launch
    int *live;                   // [kBuckets][live_stride] candidate ids handed off
    int live_stride;
    ChainState *states;          // their chain states
};

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
    return imax(0, kBuckets - 1 - tiles / kBucketTiles);
}

}  // namespace gact
