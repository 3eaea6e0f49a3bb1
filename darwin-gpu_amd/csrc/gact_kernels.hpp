// gact_kernels.hpp -- the two __global__ entry points built on gact_device.hpp.
//
//   align_tiles_kernel      one AlignWithBT per group per iteration
//                           (stands under Align_Batch_GPU, cuda_host.cu:23-190)
//   extend_kernel           persistent: every group owns one candidate and
//                           walks its whole tile chain (GACT, gact.cpp:48-228 /
//                           GACT_Batch, gact.cpp:231-560) without leaving the GPU
#pragma once

#include <type_traits>

#include "gact_device.hpp"

namespace gact {

struct WaveCtx {
    int gl;            // lane in group
    int g;             // group in wave
    int slot;          // global group slot (workspace index)
    int n_slots;       // total group slots of the grid
};

__device__ __forceinline__ WaveCtx wave_ctx()
{
    WaveCtx w;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    w.gl = lane & (kGroup - 1);
    w.g = lane >> 4;
    w.slot = wave * kGroupsPerWave + w.g;
    w.n_slots = ((gridDim.x * blockDim.x) >> 6) * kGroupsPerWave;
    return w;
}

// first step whose pointers the traceback can reach:
// non-first tiles start at (R,Q) and take < early steps in either dimension
// (align.cpp:205), so only rows > R-early and columns > Q-early are read.
template <int C>
__device__ __forceinline__ int first_pointer_step(int R, int Q, int early, bool first)
{
    if (first) return 1;
    const int r_first = imax(1, R - early + 1);
    const int l0 = (Q > early) ? (Q - early) / C : 0;
    return r_first + l0;
}

template <int C>
__device__ __forceinline__ int last_step(int R, int Q)
{
    return (R > 0 && Q > 0) ? R + (Q - 1) / C : 0;
}

// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(kBlockThreads, 3) void align_tiles_kernel(
    KParams kp, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc,
    const gact_tile *__restrict__ tiles, int n,
    gact_tile_result *__restrict__ results, uint8_t *__restrict__ states, int states_stride,
    uint32_t *__restrict__ ws_all)
{
    using G = Geometry<C>;
    __shared__ uint8_t lds[(kBlockThreads / 64) * kGroupsPerWave * G::kGroupLds];

    const WaveCtx w = wave_ctx();
    const int wave_in_block = threadIdx.x >> 6;
    uint8_t *ref_lds_g = lds + (wave_in_block * kGroupsPerWave + w.g) * G::kGroupLds;
    uint8_t *q_lds_g = ref_lds_g + G::kRefLds;
    const uint8_t *ref_lds_lane = ref_lds_g + (kGroup - 1 - w.gl);
    uint32_t *ws = ws_all + (size_t)w.slot * kp.ws_words;

    for (int base = (w.slot - w.g); base < n; base += w.n_slots) {
        const int ti = base + w.g;
        GroupTile gt{0, 0, 0, 0};
        bool reverse = false, raw = false;
        int64_t rp0 = 0, qp0 = 0;
        const SeqSetDev *qs = &qfwd;
        bool has = false;
        if (ti < n) {
            const gact_tile td = tiles[ti];
            if (td.ref_len >= 0) {
                has = true;
                gt.R = td.ref_len; gt.Q = td.query_len; gt.first = td.first;
                reverse = td.reverse != 0;
                qs = (td.query_set == GACT_SET_QUERY_RC) ? &qrc : &qfwd;
                rp0 = refs.offsets[td.ref_id] + td.ref_off;
                qp0 = qs->offsets[td.query_id] + td.query_off;
                raw = refs.use_raw | qs->use_raw;
            }
        }
        const bool active = gt.R > 0 && gt.Q > 0;
        const WavePlan wp = align_starts(last_step<C>(gt.R, gt.Q),
                                         first_pointer_step<C>(gt.R, gt.Q, kp.early, gt.first), active, gt.shift);
        uint32_t qb[C];
        load_tile<C>(refs, *qs, raw, rp0, qp0, gt.R, gt.Q, reverse, w.gl, ref_lds_g, q_lds_g, qb, gt.shift);
        wave_sync();
        const bool any_first = __any(gt.first != 0);

        PassOut po;
        if (any_first) dp_pass<C, true>(kp, w.gl, ref_lds_lane, qb, gt, wp.T_end, wp.tB, ws, po);
        else           dp_pass<C, false>(kp, w.gl, ref_lds_lane, qb, gt, wp.T_end, wp.tB, ws, po);
        po.tB -= gt.shift;      // the traceback indexes steps in the tile's own (undelayed) time

        // the pointer stores of all 16 lanes must have reached L2 before lane 0
        // reads them back (the loads bypass L1, load_ptr); same wave, same XCD,
        // so no agent-scope release (L2 write-back) is needed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        if (has && w.gl == 0) {
            gact_tile_result r;
            int i0 = gt.R, j0 = gt.Q;
            if (gt.first) {
                r.score = po.best; r.max_i = po.bi; r.max_j = po.bj;
                i0 = po.bi; j0 = po.bj;
            } else {
                r.score = po.pos_score; r.max_i = 0; r.max_j = 0;
            }
            uint8_t *sp = states + (size_t)ti * states_stride;
            int ns = 0;
            int rs = 0, qsn = 0;
            traceback<C>(ws, i0, j0, po.tB, kp.early, rs, qsn,
                         [&](int state, int, int) { sp[ns++] = (uint8_t)state; });
            r.ref_steps = rs; r.query_steps = qsn; r.n_states = ns;
            results[ti] = r;
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------
// Persistent chain kernel.  Per-group state mirrors the locals of GACT()
// (gact.cpp:57-79); every lane of the group carries an identical copy, only
// the traceback runs on one lane and its results are broadcast.

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

template <int C>
__global__ __launch_bounds__(kBlockThreads, 3) void extend_kernel(
    KParams kp, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc,
    const gact_candidate *__restrict__ cands, int first_cand, int n,
    int rc_from, int same_file,
    gact_overlap *__restrict__ out, int *__restrict__ counter,
    uint32_t *__restrict__ ws_all)
{
    using G = Geometry<C>;
    __shared__ uint8_t lds[(kBlockThreads / 64) * kGroupsPerWave * G::kGroupLds];

    const WaveCtx w = wave_ctx();
    const int wave_in_block = threadIdx.x >> 6;
    uint8_t *ref_lds_g = lds + (wave_in_block * kGroupsPerWave + w.g) * G::kGroupLds;
    uint8_t *q_lds_g = ref_lds_g + G::kRefLds;
    const uint8_t *ref_lds_lane = ref_lds_g + (kGroup - 1 - w.gl);
    uint32_t *ws = ws_all + (size_t)w.slot * kp.ws_words;
    const bool raw = refs.use_raw | qfwd.use_raw | qrc.use_raw;
    const int tile = kp.tile_size;

    ChainState s;
    s.comp = 0;
    s.cand = -1; s.phase = 2;
    bool exhausted = false;

    for (;;) {
        // ---- pick the next tile of this group, finishing / fetching candidates on the way
        GroupTile gt{0, 0, 0, 0};
        bool reverse = false;
        int64_t rp0 = 0, qp0 = 0;
        bool have_tile = false;
        for (int guard = 0; guard < 4 && !have_tile; guard++) {
            if (s.phase == 2) {
                if (exhausted) break;
                int idx = 0;
                if (w.gl == 0) idx = atomicAdd(counter, 1);
                idx = __shfl(idx, 0, kGroup);
                if (idx >= n) { exhausted = true; break; }
                const gact_candidate c = cands[first_cand + idx];
                s.cand = first_cand + idx;
                s.comp = (s.cand >= rc_from) ? 1 : 0;      // darwin.cpp:279 passes rev_reads_char
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
            if (s.phase == 0) {
                // gact.cpp:82
                if (!s.brk && s.ref_pos > 0 && s.query_pos > 0 && ((s.i > 0 && s.j > 0) || s.first_tile)) {
                    gt.R = (s.ref_pos > tile) ? tile : s.ref_pos;           // :84-85
                    gt.Q = (s.query_pos > tile) ? tile : s.query_pos;
                    gt.first = s.first_tile;
                    reverse = false;
                    rp0 = s.rbase + s.ref_pos - gt.R;
                    qp0 = s.qbase + s.query_pos - gt.Q;
                    have_tile = true;
                } else {
                    // leftmost column has no predecessor: a gap there costs gap_open (open==true at :198)
                    if (s.pend_gap) s.score += kp.open;
                    s.pend_gap = 0;
                    s.abpos = s.ref_pos; s.bbpos = s.query_pos;               // :136-141
                    s.ref_pos = s.rev_ref_pos; s.query_pos = s.rev_query_pos;
                    s.i = tile; s.j = tile;
                    s.open_flag = !(s.have_left && s.left_first_gap);
                    s.phase = 1; s.brk = 0;
                }
            }
            if (s.phase == 1 && !have_tile) {
                // gact.cpp:144
                if (!s.brk && s.ref_pos < s.ref_len && s.query_pos < s.query_len &&
                    ((s.i > 0 && s.j > 0) || s.first_tile)) {
                    gt.R = (s.ref_pos + tile < s.ref_len) ? tile : s.ref_len - s.ref_pos;       // :146-147
                    gt.Q = (s.query_pos + tile < s.query_len) ? tile : s.query_len - s.query_pos;
                    gt.first = s.first_tile;
                    reverse = true;
                    rp0 = s.rbase + s.ref_pos;
                    qp0 = s.qbase + s.query_pos;
                    have_tile = true;
                } else {
                    if (w.gl == 0) {
                        gact_overlap o;
                        o.ref_id = s.ref_id; o.query_id = s.query_id;
                        o.ab = s.abpos; o.ae = s.ref_pos; o.bb = s.bbpos; o.be = s.query_pos;
                        o.score = s.score; o.comp = s.comp;
                        o.emitted = (!(same_file && s.ref_id == s.query_id) && s.score > 0) ? 1 : 0;  // :213
                        o.first_tile_score = s.first_tile_score;
                        o.n_tiles = s.n_tiles; o.reserved = 0; o.cells = s.cells;
                        out[s.cand] = o;
                    }
                    s.phase = 2; s.cand = -1;
                }
            }
        }
        if (!__any(have_tile)) {
            // nobody in this wave has a tile: either all exhausted, or some group
            // still has transitions pending (guard ran out) -- loop again for those
            if (__all(exhausted && s.phase == 2)) break;
            continue;
        }
        if (!have_tile) { gt.R = 0; gt.Q = 0; gt.first = 0; gt.shift = 0; }

        const bool active = gt.R > 0 && gt.Q > 0;
        const WavePlan wp = align_starts(last_step<C>(gt.R, gt.Q),
                                         first_pointer_step<C>(gt.R, gt.Q, kp.early, gt.first), active, gt.shift);
        uint32_t qb[C];
        load_tile<C>(refs, s.comp ? qrc : qfwd, raw, rp0, qp0, gt.R, gt.Q, reverse, w.gl, ref_lds_g, q_lds_g, qb,
                     gt.shift);
        wave_sync();
        const bool any_first = __any(gt.first != 0);

        PassOut po;
        if (any_first) dp_pass<C, true>(kp, w.gl, ref_lds_lane, qb, gt, wp.T_end, wp.tB, ws, po);
        else           dp_pass<C, false>(kp, w.gl, ref_lds_lane, qb, gt, wp.T_end, wp.tB, ws, po);
        po.tB -= gt.shift;      // the traceback indexes steps in the tile's own (undelayed) time

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // see align_tiles_kernel

        // ---- consume the tile exactly as gact.cpp:95-133 / :158-194 do
        if (have_tile) {
            s.n_tiles++;
            s.cells += (int64_t)gt.R * gt.Q;
            int i0 = gt.R, j0 = gt.Q;
            bool stop = false;
            if (s.first_tile) {
                i0 = po.bi; j0 = po.bj;
                if (s.phase == 0) {
                    s.ref_pos = s.ref_pos - gt.R + po.bi;                    // :100-105
                    s.query_pos = s.query_pos - gt.Q + po.bj;
                    s.rev_ref_pos = s.ref_pos; s.rev_query_pos = s.query_pos;
                } else {
                    s.ref_pos = s.ref_pos + gt.R - po.bi;                    // :163-166
                    s.query_pos = s.query_pos + gt.Q - po.bj;
                }
                s.first_tile_score = po.best;
                if (po.best < kp.thr) { stop = true; s.brk = 1; }            // :107-109 / :168-170
            }
            int ref_steps = 0, query_steps = 0, nst = 0;
            int score = s.score, pend_gap = s.pend_gap, open_flag = s.open_flag;
            int have_left = s.have_left, left_first_gap = s.left_first_gap;
            if (!stop && w.gl == 0) {
                const int phase = s.phase;
                traceback<C>(ws, i0, j0, po.tB, kp.early, ref_steps, query_steps,
                    [&](int state, int ci, int cj) {
                        const bool gap = (state != GACT_STATE_M);
                        int sub = 0;
                        if (!gap) {
                            const uint32_t rbv = ref_lds_g[kGroup + gt.shift + ci - 1];
                            const uint32_t qbv = q_lds_g[cj - 1];
                            sub = (rbv == qbv) ? kp.match : kp.mismatch;     // gact.cpp:207
                        }
                        if (phase == 0) {
                            // columns arrive right-to-left; the previously emitted one
                            // now learns its left neighbour
                            if (pend_gap) score += gap ? kp.ext : kp.open;
                            if (!have_left) { have_left = 1; left_first_gap = gap; }
                            if (gap) pend_gap = 1; else { score += sub; pend_gap = 0; }
                        } else {
                            if (gap) { score += open_flag ? kp.open : kp.ext; open_flag = 0; }
                            else { score += sub; open_flag = 1; }
                        }
                        nst++;
                    });
            }
            // broadcast lane 0's results to the group
            ref_steps = __shfl(ref_steps, 0, kGroup);
            query_steps = __shfl(query_steps, 0, kGroup);
            nst = __shfl(nst, 0, kGroup);
            s.score = __shfl(score, 0, kGroup);
            s.pend_gap = __shfl(pend_gap, 0, kGroup);
            s.open_flag = __shfl(open_flag, 0, kGroup);
            s.have_left = __shfl(have_left, 0, kGroup);
            s.left_first_gap = __shfl(left_first_gap, 0, kGroup);
            if (nst > 0) s.first_tile = 0;                                   // :112 / :173
            s.i = query_steps; s.j = ref_steps;                              // gact.cpp's i counts query bases
            if (!stop) {
                if (s.phase == 0) { s.ref_pos -= ref_steps; s.query_pos -= query_steps; }   // :132-133
                else              { s.ref_pos += ref_steps; s.query_pos += query_steps; }   // :193-194
            } else {
                s.i = 0; s.j = 0;
            }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------
// 2-bit packer: 16 bases -> one word; flags[0] |= 1 when a byte is not A/C/G/T
// (or the GPU build's 0..3 recode, darwin.cpp:320-332).  Code order A0 C1 G2 T3
// as ntcoding.h:25-28.
__device__ __forceinline__ uint32_t base_code(uint32_t b, bool &bad)
{
    switch (b) {
        case 'A': case 0: return 0;
        case 'C': case 1: return 1;
        case 'G': case 3: return 2;
        case 'T': case 2: return 3;
        default: bad = true; return 0;
    }
}

__global__ void pack_kernel(const uint8_t *__restrict__ raw, int64_t n_bases,
                            uint32_t *__restrict__ packed, int64_t n_words, int *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += stride) {
        uint32_t word = 0;
        const int64_t b0 = wi * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t p = b0 + k;
            if (p < n_bases) word |= base_code(raw[p], bad) << (2 * k);
        }
        packed[wi] = word;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}


// ---------------------------------------------------------------------------
// Integer-VALU issue-rate probe: 16 independent v_add_u32 / v_max_i32 chains
// per lane, nothing else.  Gives the measured int32 lane-op/s ceiling the
// roofline fraction in bench.py is priced against.
__global__ __launch_bounds__(kBlockThreads) void valu_probe_kernel(int iters, int seed, int *__restrict__ sink)
{
    int a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = seed + k + threadIdx.x;
    const int inc = seed | 1;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            asm volatile("v_add_u32 %0, %0, %1\n\tv_max_i32 %0, %0, %2" : "+v"(a[k]) : "v"(inc), "v"(seed));
        }
    }
    int r = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) r ^= a[k];
    if (r == 0x7fffffff) sink[0] = r;
}

}  // namespace gact
