// gact_big.hpp -- tiles of 513 .. 2048 bases.  align.h:19 / align.cpp:66-67 accept tile_size < 2049; every caller in the
// reference passes 320 (params.cfg:22), and the kernels of gact_p16*.hpp / gact_kernels.hpp hold a tile's columns in
// the registers of 16 or 32 lanes, which ends at 512.  This file is the same path for the rest of the interface: the
// same cells (align.cpp:60-233), the same chain (gact.cpp:48-228, through gact_chain.hpp), one WAVE per tile:
//   * lane l owns CB adjacent columns (CB = 16: tiles up to 1024, CB = 32: up to 2048), int32 scores, M / I / max(M, I, D)
//     of the previous row in registers; the wave sweeps anti-diagonals, lane l one row behind lane l - 1, and three
//     values cross a lane boundary per step (M and D of the row, max(M, I, D) of the row above);
//   * the two slices are staged in LDS in DP order (reverse tiles back to front, align.cpp:130-131) as RAW bytes --
//     align.cpp:134 compares characters, so N == N and case matters with no second code path;
//   * pointers: one byte per cell, align.h:23 op code + the two "gap opened here" flags of align.cpp:169-170, row-major
//     in the wave's share of the workspace (row stride 64 * CB, so a lane's CB bytes of a row are CB / 16 uint4 stores);
//   * traceback (align.cpp:185-230): uniform over the wave.  The 64 lanes load a block of 64 rows x 16 columns of
//     pointer bytes at once (one uint4 each, past L1) and the walk reads it with a lane shuffle until it leaves the
//     block: one memory round trip per 16 .. 64 steps instead of one per step.
// Built for function, not for speed: a 2048-tile is 4.2 M cells on one wave (lone waves issue every ~8 cycles), and no
// bench line of BASELINE.json uses it.  tests/test_gpu_big_tiles.py holds it to the oracle and the compiled reference.
#pragma once

#include "gact_chain.hpp"

namespace gact {

constexpr int kBigInf = 1 << 28;                     // -kBigInf + gap scores stays far from wrapping (create checks the scores)
constexpr int kBigMaxTile = 2048;

template <int CB> struct BigGeom {
    static constexpr int kTileMax = 64 * CB;
    static constexpr int kStride = 64 * CB;          // pointer bytes per row
    static constexpr size_t kWsBytes = (size_t)kTileMax * kStride;
    static constexpr int kLds = 2 * kTileMax;        // ref + query bytes of one tile
};

struct BigOut {
    int best, bi, bj;        // arg-max of H, last in row-major order among equals (align.cpp:173-177: >=)
    int pos_score;           // H[R][Q]
};

__device__ __forceinline__ uint4 big_load_past_l1(const uint8_t *p)
{
    uint4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// the two slices into LDS, DP order
__device__ __forceinline__ void big_stage(const SeqSetDev &rs, const SeqSetDev &qs, int64_t rp0, int64_t qp0, int R, int Q,
                                          bool reverse, uint8_t *lr, uint8_t *lq, int tile_max)
{
    const int lane = threadIdx.x & 63;
    for (int d = lane; d < tile_max; d += 64) {
        lr[d] = d < R ? rs.raw[rp0 + (reverse ? R - 1 - d : d)] : (uint8_t)0xFF;      // pads never match (0xFF != 0xFE)
        lq[d] = d < Q ? qs.raw[qp0 + (reverse ? Q - 1 - d : d)] : (uint8_t)0xFE;
    }
}

template <int CB>
__device__ __forceinline__ void big_dp(const KParams &kp, int R, int Q, const uint8_t *lr, const uint8_t *lq, uint8_t *dir, BigOut &out)
{
    const int lane = threadIdx.x & 63;
    const int j0 = lane * CB;                         // my columns are j0 + 1 .. j0 + CB
    int Mp[CB], Ip[CB], Bp[CB];                       // row above: M, I, max(M, I, D)   (align.cpp:87-99: 0, -inf, so 0)
    uint32_t qw[CB / 4];
#pragma unroll
    for (int c = 0; c < CB; c++) { Mp[c] = 0; Ip[c] = -kBigInf; Bp[c] = 0; }
#pragma unroll
    for (int k = 0; k < CB / 4; k++) qw[k] = *reinterpret_cast<const uint32_t *>(lq + j0 + 4 * k);
    int best = 0, bi = 0, bj = 0, pos = 0;
    // from the lane to the left, for the row this lane takes next: M, D of that row and max(M, I, D) of the row above at
    // column j0; lane 0 reads the border column (align.cpp:101-107)
    int m_in = 0, d_in = -kBigInf, b_in = 0;
    const int n_cols = (Q + CB - 1) / CB;             // lanes that own a column
    const int steps = R + n_cols - 1;
    for (int t = 1; t <= steps; t++) {
        const int i = t - lane;
        const bool live = i >= 1 && i <= R && lane < n_cols;
        int m_left = m_in, d_left = d_in, diag = b_in;
        if (live) {
            const uint32_t ref_nt = lr[i - 1];
            uint32_t pw[CB / 4];
#pragma unroll
            for (int c = 0; c < CB; c++) {
                const uint32_t q_nt = (qw[c >> 2] >> ((c & 3) * 8)) & 0xffu;
                const int sub = (q_nt == ref_nt) ? kp.match : kp.mismatch;               // align.cpp:134
                int m = diag + sub;                                                        // :138-147
                m = m < 0 ? 0 : m;
                const int ins_open = Mp[c] + kp.open, ins_ext = Ip[c] + kp.ext;           // :149-156
                const int del_open = m_left + kp.open, del_ext = d_left + kp.ext;
                const int ins = ins_open > ins_ext ? ins_open : ins_ext;
                const int del = del_open > del_ext ? del_open : del_ext;
                const int max1 = m > ins ? m : ins, max2 = del > 0 ? del : 0;              // :158-160
                const int h = max1 > max2 ? max1 : max2;
                uint32_t p = (m >= ins) ? ((m >= del) ? GACT_STATE_M : GACT_STATE_D)      // :162-171
                                        : ((ins >= del) ? GACT_STATE_I : GACT_STATE_D);
                if (m <= 0 && ins <= 0 && del <= 0) p = GACT_STATE_Z;
                if (ins_open >= ins_ext) p += 8;
                if (del_open >= del_ext) p += 4;
                if ((c & 3) == 0) pw[c >> 2] = p; else pw[c >> 2] |= p << ((c & 3) * 8);
                const int j = j0 + c + 1;
                if (j <= Q) {
                    if (h >= best) { best = h; bi = i; bj = j; }                           // :173-177
                    if (i == R && j == Q) pos = h;                                         // :178-181
                }
                diag = Bp[c];
                Mp[c] = m; Ip[c] = ins;
                Bp[c] = max1 > del ? max1 : del;
                m_left = m; d_left = del;
            }
            uint4 *row = reinterpret_cast<uint4 *>(dir + (size_t)(i - 1) * BigGeom<CB>::kStride + j0);
#pragma unroll
            for (int k = 0; k < CB / 16; k++) row[k] = make_uint4(pw[4 * k], pw[4 * k + 1], pw[4 * k + 2], pw[4 * k + 3]);
        }
        // (after the last column: diag = max(M, I, D) of the row above at my last column)
        m_in = __shfl_up(m_left, 1);
        d_in = __shfl_up(d_left, 1);
        b_in = __shfl_up(diag, 1);
        if (lane == 0) { m_in = 0; d_in = -kBigInf; b_in = 0; }
    }
    // arg-max over the lanes: largest H, then largest row, then largest column
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int ob = __shfl_xor(best, d), oi = __shfl_xor(bi, d), oj = __shfl_xor(bj, d);
        const bool take = ob > best || (ob == best && (oi > bi || (oi == bi && oj > bj)));
        if (take) { best = ob; bi = oi; bj = oj; }
    }
    // H[R][Q] sits on the lane that owns column Q
    const int owner = Q > 0 ? (Q - 1) / CB : 0;
    out.best = best; out.bi = bi; out.bj = bj;
    out.pos_score = __shfl(pos, owner);
}

// Uniform walk from (i0, j0); emit(state, i, j) per pushed state, before the move (align.cpp:206).
template <int CB, class Emit>
__device__ __forceinline__ void big_traceback(const uint8_t *dir, int i0, int j0, int early, int &ref_steps, int &query_steps, Emit emit)
{
    const int lane = threadIdx.x & 63;
    int blk_i = -1, blk_jb = -1;                       // block held: rows blk_i - 63 .. blk_i, columns blk_jb + 1 .. blk_jb + 16
    uint4 held = make_uint4(0, 0, 0, 0);
    auto fetch = [&](int i, int j) -> uint32_t {
        const int jb = (j - 1) & ~15;
        if (blk_i < 0 || i > blk_i || i <= blk_i - 64 || jb != blk_jb) {
            blk_i = i; blk_jb = jb;
            const int row = i - lane;
            held = make_uint4(0, 0, 0, 0);
            if (row >= 1) held = big_load_past_l1(dir + (size_t)(row - 1) * BigGeom<CB>::kStride + jb);
        }
        const int c = (j - 1) - jb;
        uint32_t w = c < 4 ? held.x : c < 8 ? held.y : c < 12 ? held.z : held.w;
        w = __shfl(w, blk_i - i);
        return (w >> ((c & 3) * 8)) & 0xffu;
    };
    int i = i0, j = j0, is = 0, js = 0;
    uint32_t nib = 0;
    int state = GACT_STATE_Z;
    if (i >= 1 && j >= 1 && early > 0) {
        nib = fetch(i, j);
        state = nib & 3;
    }
    while (state != GACT_STATE_Z) {
        emit(state, i, j);
        int next;
        if (state == GACT_STATE_M) { next = -1; i--; j--; is++; js++; }
        else if (state == GACT_STATE_I) { next = (nib & 8) ? GACT_STATE_M : GACT_STATE_I; i--; is++; }
        else { next = (nib & 4) ? GACT_STATE_M : GACT_STATE_D; j--; js++; }
        if (is >= early || js >= early) break;       // align.cpp:205
        if (i < 1 || j < 1) break;                   // border pointers are ZERO (align.cpp:101-107)
        nib = fetch(i, j);
        state = (next < 0) ? (int)(nib & 3) : next;
    }
    ref_steps = is;
    query_steps = js;
}

// ---------------------------------------------------------------------------
// gact_hip_align_tiles for big tiles: one wave per tile
template <int CB>
__global__ __launch_bounds__(kBlockThreads) void big_tiles_kernel(
    KParams kp, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc, const gact_tile *__restrict__ tiles, int n,
    gact_tile_result *__restrict__ results, uint8_t *__restrict__ states, int states_stride, uint8_t *__restrict__ ws_all)
{
    using G = BigGeom<CB>;
    __shared__ __attribute__((aligned(16))) uint8_t lds[(kBlockThreads / 64) * G::kLds];
    const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (kBlockThreads / 64) + wave_in_block, n_waves = gridDim.x * (kBlockThreads / 64);
    uint8_t *lr = lds + wave_in_block * G::kLds, *lq = lr + G::kTileMax;
    uint8_t *dir = ws_all + (size_t)wave * G::kWsBytes;
    for (int ti = wave; ti < n; ti += n_waves) {
        const gact_tile td = tiles[ti];
        if (td.ref_len < 0) continue;                                    // idle entry (cuda_host.cu:70): its result stays zero
        const SeqSetDev &qs = (td.query_set == GACT_SET_QUERY_RC) ? qrc : qfwd;
        const int R = td.ref_len, Q = td.query_len;
        big_stage(refs, qs, refs.offsets[td.ref_id] + td.ref_off, qs.offsets[td.query_id] + td.query_off, R, Q, td.reverse != 0, lr, lq,
                  G::kTileMax);
        wave_sync();
        BigOut po{0, 0, 0, 0};
        if (R > 0 && Q > 0) big_dp<CB>(kp, R, Q, lr, lq, dir, po);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the pointer stores are in L2 before the walk reads them
        gact_tile_result r;
        int i0 = R, j0 = Q;
        if (td.first) { r.score = po.best; r.max_i = po.bi; r.max_j = po.bj; i0 = po.bi; j0 = po.bj; }
        else { r.score = po.pos_score; r.max_i = 0; r.max_j = 0; }
        uint8_t *sp = states + (size_t)ti * states_stride;
        int ns = 0, rs = 0, qsn = 0;
        big_traceback<CB>(dir, i0, j0, kp.early, rs, qsn, [&](int state, int, int) { if (lane == 0) sp[ns] = (uint8_t)state; ns++; });
        r.ref_steps = rs; r.query_steps = qsn; r.n_states = ns;
        if (lane == 0) results[ti] = r;
        wave_sync();
    }
}

// ---------------------------------------------------------------------------
// gact_hip_candidates_run* for big tiles: one wave per candidate, the whole chain (the consumption of a tile is
// extend_kernel's, gact_kernels.hpp, statement by statement)
template <int CB>
__global__ __launch_bounds__(kBlockThreads) void big_extend_kernel(
    KParams kp, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc, const gact_candidate *__restrict__ cands, int first_cand, int n,
    int rc_from, int same_file, gact_overlap *__restrict__ out, ChainQueues cq, uint8_t *__restrict__ ws_all)
{
    using G = BigGeom<CB>;
    __shared__ __attribute__((aligned(16))) uint8_t lds[(kBlockThreads / 64) * G::kLds];
    const int wave_in_block = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (kBlockThreads / 64) + wave_in_block;
    uint8_t *lr = lds + wave_in_block * G::kLds, *lq = lr + G::kTileMax;
    uint8_t *dir = ws_all + (size_t)wave * G::kWsBytes;
    ChainState s;
    for (;;) {
        if (!seed_pop(s, cq, lane == 0, [](int v) { return __shfl(v, 0); }, cands, first_cand, n, rc_from, refs, qfwd, qrc)) break;
        for (;;) {
            const TilePick pk = chain_pick(s, kp, same_file, out, lane == 0);
            if (!pk.have) break;
            big_stage(refs, s.comp ? qrc : qfwd, pk.rp0, pk.qp0, pk.R, pk.Q, pk.reverse, lr, lq, G::kTileMax);
            wave_sync();
            BigOut po{0, 0, 0, 0};
            big_dp<CB>(kp, pk.R, pk.Q, lr, lq, dir, po);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ---- gact.cpp:95-133 / :158-194
            s.n_tiles++;
            s.cells += (int64_t)pk.R * pk.Q;
            int i0 = pk.R, j0 = pk.Q;
            bool stop = false;
            if (s.first_tile) {
                i0 = po.bi; j0 = po.bj;
                stop = chain_first_tile(s, kp, pk.R, pk.Q, po.best, po.bi, po.bj);
            }
            int ref_steps = 0, query_steps = 0, nst = 0;
            ScoreWalk wk;
            wk.load(s);
            if (!stop) {
                const int phase = s.phase;
                big_traceback<CB>(dir, i0, j0, kp.early, ref_steps, query_steps, [&](int state, int i, int j) {
                    const bool gap = state != GACT_STATE_M;
                    wk.column(phase, gap, lr[i - 1] == lq[j - 1] ? kp.match : kp.mismatch, kp);     // gact.cpp:197-210
                    nst++;
                });
            }
            chain_advance<64>(s, stop, wk, ref_steps, query_steps, nst, 0);
            wave_sync();
        }
    }
}

}  // namespace gact
