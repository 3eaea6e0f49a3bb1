// gact_device.hpp -- device side of the MI355X (gfx950) GACT engine.
//
// Work decomposition:
//   * one GACT tile (<= C*16 x C*16 cells) per 16-lane group, four groups per
//     wave64.  Lane g owns C consecutive query columns; a step of the wave
//     advances every lane by one ref row, lane g being g rows behind lane g-1
//     (anti-diagonal wavefront).  The three values that cross a lane boundary
//     (M+open, D and H of the neighbour's last column) move by DPP row_shr:1,
//     which on CDNA is exactly a shift inside a 16-lane row.
//   * scores live in VGPRs (3*C per lane); nothing of the recurrence touches
//     LDS or HBM.  LDS holds only the group's unpacked bases.
//   * traceback pointers: 4 bits per cell, built from v_cmp lane masks that
//     are merged on the scalar unit and shifted into a per-column
//     accumulator with add-with-carry; 8 rows per dword, flushed every 8
//     steps as 16-byte stores into a per-group HBM workspace.  Only the
//     window the traceback can reach is produced.
//
// Semantics restated from the reference (bit-exact contract):
//   recurrence / pointer / arg-max rules   align.cpp:114-183
//   traceback                              align.cpp:185-230
//   tile chain, rescoring, emit            gact.cpp:82-225
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gact_hip.h"

namespace gact {

constexpr int kGroup = 16;                 // lanes per tile
constexpr int kGroupsPerWave = 4;
constexpr int kNegInf = -(1 << 30);        // align.h:18
constexpr uint32_t kRefPad = 0xFFu;        // never equals a query code
constexpr uint32_t kQueryPad = 0xFEu;      // never equals a ref code
constexpr int kBlockThreads = 256;

struct SeqSetDev {
    const uint32_t *packed;   // 2 bits/base, 16 bases per word, base k at bits 2*(k%16)
    const uint8_t *raw;       // raw bytes (always resident)
    const int64_t *offsets;   // n+1
    int32_t n;
    int32_t use_raw;          // 1: compare raw bytes (set holds non-ACGT)
    const int32_t *other;     // per sequence: 1 = holds a byte other than A/C/G/T (pack_kernel); the routing of gact_chain.hpp
};

struct KParams {
    int32_t tile_size, early;
    int32_t match, mismatch, open, ext;
    int32_t thr;
    int32_t ws_words;         // workspace dwords per group
    int32_t prio_bases[2];    // main launch: chains with more bases left than this run their DP at priority 1 / 2;
                              // {0, 0}: rank against the longest chain running right now instead (longest_running)
    int32_t band;             // linear-gap main launch: pointer words are stored within `band` columns of the diagonal through
                              // a tile's (R, Q) only (gact_lin.hpp LinBand); 0: the whole window
};

template <int C> struct Geometry {
    static constexpr int kTileMax = C * kGroup;
    static constexpr int kMaxSteps = kTileMax + kGroup;
    // front pad (skew + the largest start delay) + bases + tail pad
    static constexpr int kRefLds = kGroup + kMaxSteps + kTileMax + kGroup;
    static constexpr int kQueryLds = kTileMax;
    static constexpr int kGroupLds = kRefLds + kQueryLds;
    static constexpr int kMaxFlush = (kMaxSteps + 7) / 8 + 1;
    static constexpr int kWsWords = kMaxFlush * C * kGroup;
};

__device__ __forceinline__ int dpp_row_shr1(int v, int old)
{
    // lane n of each 16-lane row reads lane n-1; lane 0 keeps `old`
    return __builtin_amdgcn_update_dpp(old, v, 0x111, 0xF, 0xF, false);
}

// x*2 + bit(lane) in one VALU op: the lane mask of a compare is the carry-in
// of v_addc_co_u32 (hipcc does not form this from `x + x + cond`).  The mask may
// have been written by the VALU instruction right in front (v_cmp); gfx940+ wants
// two wait states between a VALU SGPR write and a VALU read of it, which the
// compiler cannot place for an operand of inline asm, hence the s_nop.
__device__ __forceinline__ uint32_t shl1_insert(uint32_t x, uint64_t lane_mask)
{
    uint32_t r;
    uint64_t carry_out;
    asm("s_nop 1\n\tv_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(x), "s"(lane_mask));
    return r;
}
__device__ __forceinline__ uint64_t lanes(bool cond) { return __builtin_amdgcn_ballot_w64(cond); }
// x + bit(lane) / x - bit(lane) in one VALU op (same wait states as shl1_insert)
__device__ __forceinline__ int add_lane_bit(int x, uint64_t lane_mask)
{
    int r;
    uint64_t carry_out;
    asm("s_nop 1\n\tv_addc_co_u32 %0, %1, 0, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(x), "s"(lane_mask));
    return r;
}
__device__ __forceinline__ int sub_lane_bit(int x, uint64_t lane_mask)
{
    int r;
    uint64_t carry_out;
    asm("s_nop 1\n\tv_subb_co_u32 %0, %1, %2, 0, %3" : "=v"(r), "=s"(carry_out) : "v"(x), "s"(lane_mask));
    return r;
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax3(int a, int b, int c) { return imax(imax(a, b), c); }

// value of `v` held by the first lane of group g, made wave-uniform
__device__ __forceinline__ int group_value(int v, int g)
{
    return __builtin_amdgcn_readlane(v, g * kGroup);
}
__device__ __forceinline__ int wave_max4(int v)
{
    return imax(imax(group_value(v, 0), group_value(v, 1)), imax(group_value(v, 2), group_value(v, 3)));
}
__device__ __forceinline__ int wave_min4(int v)
{
    return imin(imin(group_value(v, 0), group_value(v, 1)), imin(group_value(v, 2), group_value(v, 3)));
}
// the same for a wave cut into groups of LANES lanes (16: four groups, 32: two); v is group-uniform
template <int LANES> __device__ __forceinline__ int wave_max_groups(int v)
{
    return LANES == 32 ? imax(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32)) : wave_max4(v);
}
template <int LANES> __device__ __forceinline__ int wave_min_groups(int v)
{
    return LANES == 32 ? imin(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32)) : wave_min4(v);
}
// lane n reads lane n-1 across the two DPP rows of a 32-lane group: lane 16 (48) takes lane 15 (47) from a
// row_bcast:15 into the odd rows, then row_shr:1 overwrites every lane but the first of each row; lane 0 (32)
// keeps `old` (tools/dpp_probe.hip)
__device__ __forceinline__ int dpp_shr1_32(int v, int old)
{
    const int x = __builtin_amdgcn_update_dpp(old, v, 0x142, 0xA, 0xF, false);
    return __builtin_amdgcn_update_dpp(x, v, 0x111, 0xF, 0xF, false);
}

// one base of a resident set at concat position pos
template <bool RAW>
__device__ __forceinline__ uint32_t fetch_base(const SeqSetDev &s, int64_t pos)
{
    if (RAW) return s.raw[pos];
    const uint32_t w = s.packed[pos >> 4];
    return (w >> ((uint32_t)(pos & 15) * 2u)) & 3u;
}

// Position of DP index d (0-based) in a slice of `len` bases starting at p0: the
// slice is read back to front when `reverse` (align.cpp:130-131).  d is clamped
// into the slice (an empty slice reads its own first position, which always
// lies inside the set's padded allocation) so the load can be issued
// unpredicated; callers substitute the pad code for d >= len afterwards.
__device__ __forceinline__ int64_t slice_pos(int64_t p0, int len, bool reverse, int d)
{
    const int last = imax(len - 1, 0);
    const int dc = imin(d, last);
    return p0 + (reverse ? last - dc : dc);
}

// ---------------------------------------------------------------------------
// Tile descriptor of one group for one DP pass.
struct GroupTile {
    int R, Q;          // DP extent (0 => idle)
    int first;         // arg-max wanted (align.cpp:190)
    int shift;         // steps this tile starts late (virtual rows in front), see align_starts()
};

struct PassOut {
    int pos_score;     // H[R][Q]                       (align.cpp:179-181)
    int best, bi, bj;  // arg-max, group-reduced        (align.cpp:173-177)
    int tB;            // first step whose pointers were stored
};

// The four tiles of a wave run in lock step: the pass ends at the largest last
// step T* and pointer work starts at the smallest first-pointer step.  A short
// tile would drag that start forward for everybody, so it is started late
// instead (rows in front of row 1 are virtual and free): the latest common
// start is tB* = min_g(tB_g + T* - Tend_g), and tile g is delayed by
// max(0, tB* - tB_g), which never pushes it past T*.
struct WavePlan { int T_end, tB; };
__device__ __forceinline__ WavePlan align_starts(int Tend_g, int tB_g, bool active, int &shift)
{
    WavePlan wp;
    wp.T_end = wave_max4(active ? Tend_g : 0);
    const int reach = active ? tB_g + (wp.T_end - Tend_g) : 0x7fffffff;
    int tB = wave_min4(reach);
    if (tB == 0x7fffffff) tB = 1;
    wp.tB = tB;
    shift = active ? imax(0, tB - tB_g) : 0;
    return wp;
}

// ---------------------------------------------------------------------------
// The DP pass of one wave: four tiles, one per 16-lane group.
//
// ref_lds points at this lane's view of the group's ref bytes so that the base
// of DP row (t - gl) is ref_lds[t]; rows outside 1..R read the pad code.
template <int C, bool AMAX>
__device__ __forceinline__ void dp_pass(const KParams &kp, const int gl,
                                        const uint8_t *__restrict__ ref_lds,
                                        const uint32_t (&qb)[C],
                                        const GroupTile gt, const int T_end, const int tB,
                                        uint32_t *__restrict__ ws, PassOut &po)
{
    const int match = kp.match, mismatch = kp.mismatch, open = kp.open, ext = kp.ext;

    int Hup[C], Mo[C], Iup[C];
    uint32_t acc[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        Hup[c] = 0;            // H[0][j] = 0              (align.cpp:88)
        Mo[c] = open;          // M[0][j] + gap_open       (align.cpp:89,149)
        Iup[c] = kNegInf;      // I[0][j] = -INF           (align.cpp:90)
        acc[c] = 0;
    }
    // what lane gl-1 exposes to lane gl: its last column of the row it just did
    int Mo_last = open, D_last = kNegInf, H_last = 0;
    int H_left_prev = 0;       // H[row-1][j0-1]

    const int jbase = gl * C;                          // 0-based first column of this lane
    const int ncol = imin(imax(gt.Q - jbase, 0), C);   // valid columns in this lane
    const int lQ = (gt.Q > 0) ? (gt.Q - 1) / C : 0;
    const int cQ = (gt.Q > 0) ? (gt.Q - 1) - lQ * C : 0;
    int pos_score = 0;
    int best = 0, bi = 0, bj = 0;                      // align.cpp:109-112

    uint32_t rb = ref_lds[1];

    auto step = [&](const int t, auto ptr_tag) {
        constexpr bool PTR = decltype(ptr_tag)::value;
        const uint32_t rb_next = ref_lds[t + 1];
        const int row = t - gl - gt.shift;

        const int Ml0 = dpp_row_shr1(Mo_last, open);      // M[i][0] + open, M[i][0] = 0
        const int Dl0 = dpp_row_shr1(D_last, kNegInf);    // D[i][0] = -INF
        const int Hl = dpp_row_shr1(H_last, 0);           // H[i][0] = 0
        int Hd = H_left_prev;
        H_left_prev = Hl;

        int M[C];
        // pass 1: everything that depends only on the previous row
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int sub = (qb[c] == rb) ? match : mismatch;      // align.cpp:134
            const int Mx = Hd + sub;                                // :138-144 (max(M,I,D) == H)
            Hd = Hup[c];
            M[c] = imax(Mx, 0);                                     // :145-147
            const int Ie = Iup[c] + ext;                            // ins_extend :150
            if (PTR) acc[c] = shl1_insert(acc[c], lanes(Mo[c] >= Ie));   // :170
            Iup[c] = imax(Mo[c], Ie);                               // :154
            Mo[c] = M[c] + open;                                    // next row's ins_open, this row's del_open
        }
        // pass 2: the in-row chain through D
        int Ml = Ml0, Dl = Dl0;
        const bool rowvalid = (unsigned)(row - 1) < (unsigned)gt.R;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int De = Dl + ext;                                // del_extend :152
            const int D = imax(Ml, De);                             // :156
            const int H = imax3(M[c], Iup[c], D);                   // :158-160 (M >= 0)
            Hup[c] = H;
            if (PTR) {
                // :162-168 with M>=0: ZERO iff H==0, else MATCH iff M==H, else INSERT iff I==H
                // the three masks are merged on the scalar unit
                const uint64_t nz = lanes(H > 0);
                const uint64_t a = lanes(M[c] == H);
                const uint64_t b = lanes(Iup[c] == H);
                uint32_t x = shl1_insert(acc[c], lanes(Ml >= De));   // :171
                x = shl1_insert(x, nz & (a | b));
                x = shl1_insert(x, nz & (a | ~b));
                acc[c] = x;
            }
            if (AMAX) {
                // :173-177, `>=` in (i, then j) order; each lane sees its rows and
                // columns in that order, lanes are merged afterwards
                const bool take = (H >= best) & rowvalid & (c < ncol);
                best = take ? H : best;
                bi = take ? row : bi;
                bj = take ? (jbase + c + 1) : bj;
            }
            Ml = Mo[c];
            Dl = D;
        }
        Mo_last = Ml;
        D_last = Dl;
        H_last = Hup[C - 1];

        if (row == gt.R && gl == lQ && gt.Q > 0) {                  // :179-181
            int v = 0;
#pragma unroll
            for (int c = 0; c < C; c++) v = (c == cQ) ? Hup[c] : v;
            pos_score = v;
        }
        rb = rb_next;
    };

    int t = 1;
    for (; t < tB && t <= T_end; t++) step(t, std::false_type{});
    uint4 *wsq = reinterpret_cast<uint4 *>(ws) + gl;
    int k = 0;
    for (; t <= T_end; t++, k++) {
        step(t, std::true_type{});
        if ((k & 7) == 7) {
#pragma unroll
            for (int q = 0; q < C / 4; q++)
                wsq[q * kGroup] = make_uint4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
            wsq += (C / 4) * kGroup;
        }
    }
    if (k & 7) {
        const int sh = 4 * (8 - (k & 7));
#pragma unroll
        for (int q = 0; q < C / 4; q++)
            wsq[q * kGroup] = make_uint4(acc[4 * q] << sh, acc[4 * q + 1] << sh,
                                         acc[4 * q + 2] << sh, acc[4 * q + 3] << sh);
    }

    if (AMAX) {
        // merge the 16 lanes: largest H, then largest i, then largest j
#pragma unroll
        for (int m = 1; m < kGroup; m <<= 1) {
            const int ob = __shfl_xor(best, m, kGroup);
            const int oi = __shfl_xor(bi, m, kGroup);
            const int oj = __shfl_xor(bj, m, kGroup);
            const bool take = (ob > best) | ((ob == best) & ((oi > bi) | ((oi == bi) & (oj > bj))));
            best = take ? ob : best;
            bi = take ? oi : bi;
            bj = take ? oj : bj;
        }
    }
    po.pos_score = __shfl(pos_score, lQ, kGroup);
    po.best = best;
    po.bi = bi;
    po.bj = bj;
    po.tB = tB;
}

// ---------------------------------------------------------------------------
// Traceback (align.cpp:185-230), run by one lane per group.

// Pointer words are read back in 16-byte patches: one uint4 of the workspace
// holds 4 adjacent columns x 8 consecutive rows of one lane, and a fetch also
// brings the quad to its left, so a diagonal walk needs a new (HBM-latency)
// fetch only every ~6 steps instead of every step.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int C> struct PtrPatch {
    u32x4 cur, left;     // quads q and q-1 of one (flush block, lane)
    int key;             // uint4 index of `cur` in the workspace, -1 = nothing cached
    int has_left;
};

// The words were written by other lanes of this wave during the pass and the
// same addresses held the previous tile's pointers: read past the L1 (sc1).
__device__ __forceinline__ void fetch_patch(const u32x4 *p, bool with_left, u32x4 &cur, u32x4 &left)
{
    if (with_left) {
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                     "global_load_dwordx4 %1, %3, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(cur), "=&v"(left) : "v"(p), "v"(p - kGroup) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(cur) : "v"(p) : "memory");
    }
}

// tB here is the wave's first stored step minus the tile's start delay
template <int C>
__device__ __forceinline__ uint32_t load_ptr(const uint32_t *ws, PtrPatch<C> &pc, int i, int j, int tB)
{
    const int l = (j - 1) / C;
    const int c = (j - 1) - l * C;
    const int k = i + l - tB;
    const int q = c >> 2;
    const int key = ((k >> 3) * (C / 4) + q) * kGroup + l;
    u32x4 v;
    if (key == pc.key) {
        v = pc.cur;
    } else if (pc.has_left && key == pc.key - kGroup) {
        v = pc.left;
    } else {
        fetch_patch(reinterpret_cast<const u32x4 *>(ws) + key, q > 0, pc.cur, pc.left);
        pc.key = key;
        pc.has_left = q > 0;
        v = pc.cur;
    }
    const int wi = c & 3;
    const uint32_t w = wi == 0 ? v.x : (wi == 1 ? v.y : (wi == 2 ? v.z : v.w));
    return (w >> (28 - 4 * (k & 7))) & 15u;
}

// Emit is called once per state, in the order AlignWithBT pushes them, with
// the DP cell (i,j) the state was read at.
template <int C, class Emit>
__device__ __forceinline__ void traceback(const uint32_t *ws, int i, int j, int tB, int early,
                                          int &ref_steps, int &query_steps, Emit &&emit)
{
    int is = 0, js = 0;
    int state = GACT_STATE_Z;
    uint32_t nib = 0;
    PtrPatch<C> pc;
    pc.key = -1; pc.has_left = 0;
    if (i >= 1 && j >= 1 && early > 0) {
        nib = load_ptr<C>(ws, pc, i, j, tB);
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
        nib = load_ptr<C>(ws, pc, i, j, tB);
        state = (next < 0) ? (int)(nib & 3) : next;
    }
    ref_steps = is;
    query_steps = js;
}

// ---------------------------------------------------------------------------
// Region cache of the chain kernels' walker (walk_chain, gact_chain.hpp).  Several
// walkers (one lane each) run in lock step, so a per-step pointer load would cost every walker a full memory
// round trip whenever ANY of them misses.  Instead every kTbSpan steps each
// walker copies the whole region it can reach in the next kTbSpan steps
// (rows i-8..i, columns j-8..j: 2 lanes x 2 flush blocks x 3 column quads =
// 12 uint4) from the HBM workspace into its own LDS scratch -- all 12 loads in
// flight at once, one latency per 8 steps -- and the steps in between read LDS.
constexpr int kTbSpan = 8;
constexpr int kTbScratchWords = 18 * 4;      // dwords of LDS per walker (12 uint4 of tb_refill_at, 18 of the seed launch's look-ahead region)

template <int CW> struct TbRegion {
    int l0;              // lane of the anchor column
    int fbase[3];        // first cached flush block, for lane l0, lane l0-1 (and lane l0-2: tb_refill_oct16)
    int qbase0;          // first cached column quad of lane l0 (lane l0-1 always caches its last three)
};

// Workspace layout of one flush block (8 stored steps of a tile): [quad 0..QN-1][ROW uint4], a quad = the pointer
// words of four adjacent columns (eight in the linear-gap pass's format, below), ROW = how many uint4 lie between two
// quads of one lane.  The int32 kernels keep a tile's words to themselves (ROW = its 16 lanes).  The packed kernels
// interleave the eight tiles of a WAVE: a row is [tile A | tile B][the wave's 64 lanes] = kWsRow uint4, so that one
// store instruction of the pass writes ONE KB of adjacent bytes.  With a row per tile (four 256-byte pieces 55 KB
// apart per store instruction) the four stores of a flush cost a wave ~940 clocks, with adjacent bytes ~100
// (tools/store_probe.hip, profiles/r03/store_probe.json) -- 0.3 % of the pass's instructions were 17 % of its time.
constexpr int kWsRow = 2 * 64;
template <int QN, int ROW>
__device__ __forceinline__ const u32x4 *ws_quad_addr(const u32x4 *base, int blk, int q, int lane)
{
    return base + (blk * QN + q) * ROW + lane;
}

// anchor cell given as (lane l0, column-in-lane c0, stored step k0 = i + l0 - tB)
// CW = columns per lane, QN = 16-byte column quads stored per lane and flush block
template <int CW, int QN = CW / 4, int ROW = kGroup>
__device__ __forceinline__ void tb_refill_at(const uint32_t *ws, uint32_t *scratch, int l0, int c0, int k0,
                                             TbRegion<CW> &rg)
{
    rg.l0 = l0;
    rg.qbase0 = imax(c0 - kTbSpan, 0) >> 2;
    const u32x4 *base = reinterpret_cast<const u32x4 *>(ws);
    const u32x4 *addr[12];
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
        const int k_anchor = k0 - sl;
        const int fb = imax((k_anchor >> 3) - 1, 0);
        rg.fbase[sl] = fb;
        const int lane = imax(l0 - sl, 0);
        const int qb = sl ? QN - 3 : rg.qbase0;
#pragma unroll
        for (int lev = 0; lev < 2; lev++)
#pragma unroll
            for (int qq = 0; qq < 3; qq++)
                addr[(sl * 2 + lev) * 3 + qq] = ws_quad_addr<QN, ROW>(base, fb + lev, imin(qb + qq, QN - 1), lane);
    }
    u32x4 r[12];
#pragma unroll
    for (int g = 0; g < 3; g++)
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                     "global_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\t"
                     "global_load_dwordx4 %3, %7, off sc1"
                     : "=&v"(r[4 * g]), "=&v"(r[4 * g + 1]), "=&v"(r[4 * g + 2]), "=&v"(r[4 * g + 3])
                     : "v"(addr[4 * g]), "v"(addr[4 * g + 1]), "v"(addr[4 * g + 2]), "v"(addr[4 * g + 3])
                     : "memory");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                   "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11])
                 :: "memory");
    u32x4 *dst = reinterpret_cast<u32x4 *>(scratch);
#pragma unroll
    for (int n = 0; n < 12; n++) dst[n] = r[n];
}

// The same for the linear-gap pass's words (FMT 3, gact_lin.hpp: eight columns per uint4, QN uint4 per lane and
// flush block): 2 lanes x 2 flush blocks x 2 column octets = 8 uint4.  Nine adjacent columns always lie in two
// octets; lane l0-1 caches its last two.
// Written for few instructions (a refill runs at the walker's pace, one instruction per 8-10 clocks): the workspace
// is addressed as ws_all + a 32-bit byte offset (ws_off = the tile's own), a lane's four loads are one offset plus
// immediates (block and octet strides fit the instruction's offset field).
template <int CW, int QN, int ROW>
__device__ __forceinline__ void tb_refill_oct(const uint32_t *ws_all, uint32_t ws_off, uint32_t *scratch, int l0, int c0,
                                              int k0, TbRegion<CW> &rg)
{
    static_assert(QN >= 2 && CW > 8 && ((CW - 9) >> 3) + 1 <= QN - 1, "two column octets per lane at least; octet qbase0 + 1 exists");
    constexpr int kOct = 16 * ROW, kBlk = 16 * QN * ROW;           // byte strides of a column octet, of a flush block
    rg.l0 = l0;
    rg.qbase0 = imax(c0 - kTbSpan, 0) >> 3;                        // nine adjacent columns lie in octets qbase0, qbase0 + 1
    rg.fbase[0] = imax((k0 >> 3) - 1, 0);
    rg.fbase[1] = imax(((k0 - 1) >> 3) - 1, 0);
    const uint32_t a0 = ws_off + (uint32_t)(rg.fbase[0] * kBlk + rg.qbase0 * kOct + l0 * 16);
    const uint32_t a1 = ws_off + (uint32_t)(rg.fbase[1] * kBlk + (QN - 2) * kOct + imax(l0 - 1, 0) * 16);
    const uint32_t a0o = a0 + kOct, a0b = a0 + kBlk, a0bo = a0 + kBlk + kOct;
    const uint32_t a1o = a1 + kOct, a1b = a1 + kBlk, a1bo = a1 + kBlk + kOct;
    u32x4 r[8];
    // (s_nop: the base may have just been written by a VALU instruction -- a v_readlane_b32 out of a spill lane --
    // and a memory instruction must not read such an SGPR for five wait states; the compiler's hazard recogniser
    // does not look inside this statement)
    asm volatile("s_nop 4\n\t"
                 "global_load_dwordx4 %0, %8, %16 sc1\n\t"
                 "global_load_dwordx4 %1, %9, %16 sc1\n\t"
                 "global_load_dwordx4 %2, %10, %16 sc1\n\t"
                 "global_load_dwordx4 %3, %11, %16 sc1\n\t"
                 "global_load_dwordx4 %4, %12, %16 sc1\n\t"
                 "global_load_dwordx4 %5, %13, %16 sc1\n\t"
                 "global_load_dwordx4 %6, %14, %16 sc1\n\t"
                 "global_load_dwordx4 %7, %15, %16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                 : "v"(a0), "v"(a0o), "v"(a0b), "v"(a0bo), "v"(a1), "v"(a1o), "v"(a1b), "v"(a1bo), "s"(ws_all)
                 : "memory");
    u32x4 *dst = reinterpret_cast<u32x4 *>(scratch);
#pragma unroll
    for (int n = 0; n < 8; n++) dst[n] = r[n];
}

// A region of SIXTEEN steps for the same words (round 4): the one-lane walker's time is its refills -- 46 round trips of
// ~3,500 clocks per walk of 370 columns, 437 of the 478 clocks a step costs -- so it asks half as often for more.  Sixteen
// moves from the anchor reach at most 16 columns to the left (CW >= 8: three lanes, l0 .. l0-2, both column octets of each)
// and, in every one of those lanes, 16 stored steps back from the anchor's (three flush blocks): 3 x 3 x 2 = 18 uint4,
// all in flight together, the kTbScratchWords the scratch already has.  Cache layout: [lane slot s][block b][octet o] uint4.
template <int CW, int QN, int ROW>
__device__ __forceinline__ void tb_refill_oct16(const uint32_t *ws_all, uint32_t ws_off, uint32_t *scratch, int l0, int k0,
                                                TbRegion<CW> &rg)
{
    static_assert(QN == 2 && CW >= 8 && CW <= 16, "both column octets of three lanes cover sixteen columns to the left");
    static_assert(kTbScratchWords >= 18 * 4, "18 uint4 of scratch");
    constexpr int kOct = 16 * ROW, kBlk = 16 * QN * ROW;           // byte strides of a column octet, of a flush block
    static_assert(kOct < 4096, "the octet stride is the load's immediate offset");
    rg.l0 = l0;
    rg.qbase0 = 0;
    uint32_t a[9];
#pragma unroll
    for (int sl = 0; sl < 3; sl++) {
        rg.fbase[sl] = imax(((k0 - sl) >> 3) - 2, 0);
        const uint32_t base = ws_off + (uint32_t)(rg.fbase[sl] * kBlk + imax(l0 - sl, 0) * 16);
        a[3 * sl] = base; a[3 * sl + 1] = base + kBlk; a[3 * sl + 2] = base + 2 * kBlk;
    }
    u32x4 r[18];
    // (s_nop in every statement: the base may have just come out of a spill lane, see tb_refill_oct -- the compiler may put
    //  such a reload in front of any of them)
#pragma unroll
    for (int n = 0; n < 9; n++)
        asm volatile("s_nop 4\n\t"
                     "global_load_dwordx4 %0, %2, %3 sc1\n\t"
                     "global_load_dwordx4 %1, %2, %3 offset:%4 sc1"
                     : "=&v"(r[2 * n]), "=&v"(r[2 * n + 1]) : "v"(a[n]), "s"(ws_all), "n"(kOct) : "memory");
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                   "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(r[16]), "+v"(r[17])
                 :: "memory");
    u32x4 *dst = reinterpret_cast<u32x4 *>(scratch);
#pragma unroll
    for (int n = 0; n < 18; n++) dst[n] = r[n];
}

// Pointer word formats (read by walk_chain, gact_chain.hpp):
// FMT 0: the int32 kernels' word, 8 rows x 4 bits {ins_open>=ins_extend, del_open>=del_extend, op}, first row on top.
// FMT 1: the packed kernel's word, low half 8 rows x 2 bits op code (0 ZERO 1 MATCH 2 INSERT 3 DELETE),
//        high half 8 rows x 2 bits {ins_open<ins_extend, del_open<del_extend}.
// FMT 2: the tagged pass's word, low half 8 rows x 2 bits op in align.h:23 numbering (Z0 D1 I2 M3), high half
//        8 rows x 2 bits {ins_open>=ins_extend, del_open>=del_extend}.
// FMT 3: the linear-gap pass's word: two columns, each a half-word of 8 rows x 2 bits op (D1 I2 M3; ZERO is not
//        encoded), nothing else -- see walk_chain.

// ---------------------------------------------------------------------------
// Loads one group's tile into LDS + registers.  DP index d (0-based) of a
// sequence maps to slice position (reverse ? len-1-d : d)  (align.cpp:130-131).
template <int C>
__device__ __forceinline__ void load_tile(const SeqSetDev &rs, const SeqSetDev &qs, bool raw,
                                          int64_t rp0, int64_t qp0, int R, int Q, bool reverse,
                                          int gl, uint8_t *ref_lds_g, uint8_t *q_lds_g,
                                          uint32_t (&qb)[C], int shift)
{
    // unpredicated loads (addresses clamped into the slice) so that all of them are in flight together
    uint32_t rbv[C];
    if (raw) {
#pragma unroll
        for (int c = 0; c < C; c++) {
            rbv[c] = fetch_base<true>(rs, slice_pos(rp0, R, reverse, gl * C + c));
            qb[c] = fetch_base<true>(qs, slice_pos(qp0, Q, reverse, gl * C + c));
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; c++) {
            rbv[c] = fetch_base<false>(rs, slice_pos(rp0, R, reverse, gl * C + c));
            qb[c] = fetch_base<false>(qs, slice_pos(qp0, Q, reverse, gl * C + c));
        }
    }
#pragma unroll
    for (int c = 0; c < C; c++) {
        const int d = gl * C + c;
        rbv[c] = (d < R) ? rbv[c] : kRefPad;
        qb[c] = (d < Q) ? qb[c] : kQueryPad;
    }
    // row r (1-based) of this tile lives at ref_lds_g[kGroup + shift + r - 1]; everything in front
    // (skew + start delay) and behind reads as the pad base
    for (int k = gl; k < kGroup + shift; k += kGroup) ref_lds_g[k] = (uint8_t)kRefPad;
    uint8_t *rrow = ref_lds_g + kGroup + shift;
#pragma unroll
    for (int c = 0; c < C; c++) {
        rrow[gl * C + c] = (uint8_t)rbv[c];
        q_lds_g[gl * C + c] = (uint8_t)qb[c];
    }
    for (int k = Geometry<C>::kTileMax + gl; k < Geometry<C>::kRefLds - kGroup - shift; k += kGroup)
        rrow[k] = (uint8_t)kRefPad;
}

}  // namespace gact
