// gact_p16.hpp -- the packed-int16 chain kernels: DP pass, loaders, column layouts, main and seed launch.
//
// Same decomposition as gact_device.hpp (a tile per group of lanes, C columns per lane, anti-diagonal wavefront,
// DPP moves across lanes), but every 32-bit register carries TWO tiles: tile A in the low half-word, tile B in
// the high one, so a wave works on 8 tiles (4 in the wide layout) and each score instruction (v_pk_add_i16,
// v_pk_max_i16, v_perm_b32 ...) serves two cells.  Pointer bits are built arithmetically for both tiles at once
// (no lane masks); where the scoring allows it the pointer phase runs on tagged scores and the recurrence's own
// max operations deliver them (TAG).
//
//   dp_pass_p16        uniform column layout (16 or 32 lanes per tile pair), with the arg-max variant for first tiles
//   dp_pass_p16s       split two-region layout (gact_p16s.hpp): the one that runs at the reference's parameters
//   extend_p16_kernel  persistent main launch over a layout policy (UniformLayout / SplitLayout / WideLayout)
//   seed_p16_kernel    first tiles, then hand-off to the main launch
//
// Valid only while every intermediate fits int16 (p16_scoring_ok, p16_argmax_ok, p16_tagged_ok); the engine
// falls back step by step (explicit pointer comparisons, int32 seed kernel, int32 chain kernel) otherwise.
#pragma once

#include <type_traits>

#include "gact_chain.hpp"
#include "gact_kernels.hpp"      // WaveCtx, first_pointer_step, last_step

namespace gact {

// Diagnostic build only (-DGACT_STAMPS): per-phase shader-clock totals of the main kernel,
// summed over waves into g_stamps (never read by the kernel itself).
#ifdef GACT_STAMPS
__device__ unsigned long long g_stamps[8];
__device__ unsigned long long g_stamps2[8];
// per wave of the main launch: start, first time it found the queues empty, end (s_memrealtime, 100 MHz), iterations
__device__ unsigned long long g_timeline[4 * 4096];
// shader clocks (s_memtime) each of those waves lived: with the 100 MHz stamps above, the clock the chip held
__device__ unsigned long long g_wave_cycles[4096];
// cooperative launch (gact_coop.hpp): walk batches, their loop trips, lane-trips (a trip with n walking lanes counts n), jobs walked
__device__ unsigned long long g_coop_counts[4];
#define GACT_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define GACT_ACC(slot, t0, t1) stamp_acc[slot] += (t1) - (t0)
#else
#define GACT_STAMP(var)
#define GACT_ACC(slot, t0, t1)
#endif

// the seed launch's linear-gap walks by teams of eight lanes (1) or by one lane per tile (0)
#ifndef GACT_SEED_WALK_TEAM
#define GACT_SEED_WALK_TEAM 1
#endif

// timing experiment (results are WRONG by design): no traceback walk.  Only with -DGACT_EXPERIMENTS, and gact_hip_create
// says so on stderr (gact_engine.hip); the store experiments of rounds 2-4 are in the history of gact_lin.hpp and in
// DESIGN.md 5.0, not in the sources
#ifndef GACT_EXPERIMENTS
#undef GACT_EXP_FAKE_WALK
#endif
#ifndef GACT_EXP_FAKE_WALK
#define GACT_EXP_FAKE_WALK 0
#endif

constexpr int kNegInf16 = -16384;
constexpr int kSlots = 2;                  // tiles per 16-lane group (low / high half-word)

// scoring constants replicated into both half-words
struct P16Consts {
    uint32_t match, nd /* mismatch - match */, open, ext, ninf, one;
    uint32_t mism;      // mismatch in both half-words
    uint32_t dsub;      // (match - mismatch) << 24 (LUT form of the substitution score)
    // tagged form (dp_pass_p16s, TAG): scores times four, the two low bits of every value say where it came from
    uint32_t c3;        // 3 in both half-words
    uint32_t nmask;     // ~c3
    uint32_t tag1, tag2;
    uint32_t match4, mism4, nd4;          // 4 * (...)
    uint32_t open4m2;   // 4 * gap_open - 2: takes an M tagged 3 to M + gap_open tagged 1
    uint32_t ext4m2;    // 4 * gap_extend - 2: I tagged 2 -> ins_extend tagged 0
    uint32_t ext4m1;    // 4 * gap_extend - 1: D tagged 1 -> del_extend tagged 0
    uint32_t dsub4;     // 4 * (match - mismatch) << 24
    uint32_t floor4;    // -6000: what is still -INF when the scores are scaled
    // linear-gap pass (gact_lin.hpp): the row drift's step and the scaled gap score
    uint32_t next;      // -gap_extend
    uint32_t next4;     // -4 * gap_extend
    uint32_t ext4;      // 4 * gap_extend
    int32_t s_mismatch, s_open, s_ext;    // the scores themselves (the drifted affine pass derives its constants, gact_aff.hpp)
};

__host__ __device__ inline uint32_t pk2(int v) { return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16); }

// every value of the recurrence is in [gap_open, tile*match] once row 1 is
// reached; the -INF sentinel only ever gets one gap_extend added to it
__host__ inline bool p16_scoring_ok(int tile, int match, int mismatch, int open, int ext)
{
    return match >= 0 && (long long)match * (tile + 2) <= 12000 && mismatch >= -4000 && open >= -4000 &&
           ext >= -4000 && open - ext > kNegInf16 + 64 && match - mismatch <= 127;
}

// (the _s forms take a wave-uniform constant.  Its operand is a VGPR all the same: the compiler's hazard recogniser
// does not look inside an asm statement, and an SGPR that has just come out of a spill lane -- v_readlane_b32, a VALU
// write -- must not be read by the next two VALU instructions; the copy into the VGPR is the compiler's own
// instruction and gets its wait states)
#define GACT_PK2(name, op)                                                                        \
    __device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b)                               \
    {                                                                                             \
        uint32_t r;                                                                               \
        asm(op " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));                                         \
        return r;                                                                                 \
    }                                                                                             \
    __device__ __forceinline__ uint32_t name##_s(uint32_t a, uint32_t s)                           \
    {                                                                                             \
        uint32_t r;                                                                               \
        asm(op " %0, %1, %2" : "=v"(r) : "v"(a), "v"(s));                                         \
        return r;                                                                                 \
    }
GACT_PK2(pk_add, "v_pk_add_i16")
GACT_PK2(pk_max, "v_pk_max_i16")
GACT_PK2(pk_minu, "v_pk_min_u16")
#undef GACT_PK2

__device__ __forceinline__ uint32_t pk_max0(uint32_t a)
{
    uint32_t r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(a));
    return r;
}
// a*s + c per half-word
__device__ __forceinline__ uint32_t pk_mad_s(uint32_t a, uint32_t s, uint32_t c)
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(s), "v"(c));
    return r;
}

// per half-word: a - b, x >> 15 (logical), min(x, 1), a*K + c with a small constant K in both halves
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_sub_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_sign(uint32_t a)
{
    uint32_t r;
    asm("v_pk_lshrrev_b16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ uint32_t pk_min1(uint32_t a)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, 1 op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ uint32_t pk_mad_vvv(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_shl_add2(uint32_t acc, uint32_t bit)    // acc*2 + bit
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 2, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(acc), "v"(bit));
    return r;
}
__device__ __forceinline__ uint32_t pk_shl_add4(uint32_t acc, uint32_t code)   // acc*4 + code
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 4, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(acc), "v"(code));
    return r;
}

__device__ __forceinline__ uint32_t pk_mad8(uint32_t a, uint32_t c)              // a*8 + c (signed halves)
{
    uint32_t r;
    asm("v_pk_mad_i16 %0, %1, 8, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_mad_m1(uint32_t a, uint32_t b)             // a*b - 1 (wrapping halves)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, -1 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// tagged pointer scheme (split layout, pointer phase): 4 * score + 3 must fit, so must 4 * (match - mismatch) a byte
__host__ inline bool p16_tagged_ok(int tile, int match, int mismatch, int open, int ext)
{
    return (long long)match * (tile + 2) <= 7900 && match - mismatch <= 63 && mismatch >= -1000 && open >= -1000 &&
           ext >= -1000;
}

// arg-max of the packed pass (first tiles): the key H*8 + (step & 7) must stay under 2^14, so that a
// row outside the tile can be keyed negative by a -2^14 bias
constexpr int kKeyBias = -16384;
__host__ inline bool p16_argmax_ok(int tile, int match) { return (long long)match * (tile + 2) * 8 + 7 < 16384; }

struct P16Best { int best[2], bi[2], bj[2]; };

// LANES lanes per tile pair: 16 (one DPP row), or 32 for the latency-bound regime (two rows, half the columns
// per lane, half the instructions per step; see WideLayout)
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t vmask, uint32_t s_or)      // (a & vmask) | s_or
{
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(vmask), "v"(s_or));
    return r;
}
__device__ __forceinline__ uint32_t pk_lshr2(uint32_t a)
{
    uint32_t r;
    asm("v_pk_lshrrev_b16 %0, 2, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ uint32_t pk_mul(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_mad4(uint32_t a, uint32_t s_c)      // a * 4 + c (wrapping halves)
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 4, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(s_c));
    return r;
}

template <int C, int LANES = kGroup> struct GeometryP16 {
    static constexpr int kTileMax = C * LANES;
    static constexpr int kMaxSteps = kTileMax + LANES;
    static constexpr int kQuads = (C + 3) / 4;                     // 16-byte column quads stored per lane
    static constexpr int kWsWords = ((kMaxSteps + 7) / 8 + 1) * kQuads * 4 * LANES;
    // ref stream: one 16-bit entry per step-row, byte 0 = tile A's base, byte 1 = tile B's
    static constexpr int kRefEntries = LANES + kMaxSteps + kTileMax + LANES;
    static constexpr int kRefBytes = kRefEntries * 2;
    static constexpr int kQueryBytes = kTileMax * kSlots;
    static constexpr int kGroupLds = (kRefBytes + kQueryBytes + 15) & ~15;
};

// ---------------------------------------------------------------------------
// One pass of a wave: 4 groups x 2 tiles.  ref16 is this lane's view of the
// group's ref stream: entry [t] holds the bases of step-row (t - gl) of both
// tiles (each tile's own start delay is folded in when the stream is written).
//
// AMAX (first tiles, align.cpp:173-177): every column slot keeps the largest key H*8 + (step & 7) of the
// current 8-step block (a later row wins a tie; rows outside 1..R are keyed negative); at every flush the
// slots of a lane are folded in column order (a later column wins a tie, columns past Q masked) into one
// 32-bit (H, step, column) record per tile, and the 16 lanes are merged at the end.
//
// Substitution score, two forms.  RAW (sets holding bytes other than ACGT, compared as raw bytes like
// align.cpp:134): x = q ^ r, neq = min(x, 1), Mx = (H + match) + neq * (mismatch - match): 3 ops per cell pair.
// Otherwise (2-bit sets): every step builds one look-up word per tile, byte k = (k == r) ? match - mismatch : 0
// (one right shift of (match - mismatch) << 24 by the row's stream byte, 24 - 8 r; a pad row's byte is 31 and
// shifts everything out), and a slot's v_perm_b32 -- its selector bytes are the two query codes, fixed per
// tile -- picks both tiles' values at once; Mx = (H + mismatch) + that: 2 ops per cell pair.
// TAG: the pointer phase on tagged scores (see dp_pass_p16s); not for the arg-max pass, whose keys are built from H.
template <int C, bool AMAX = false, bool RAW = true, int LANES = kGroup, bool TAG = false>
__device__ __forceinline__ void dp_pass_p16(const P16Consts &kc, const int gl,
                                            const uint16_t *__restrict__ ref16,
                                            const uint32_t (&qb)[C],
                                            const int T_end, const int tB,
                                            uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB,
                                            const int (*RQ)[2] = nullptr, P16Best *pb = nullptr)
{
    const uint32_t hbias = RAW ? kc.match : kc.mism;
    uint32_t Hm[C], Mo[C], Iup[C];          // H+bias, M+open, I of the previous row (both tiles)
    // Pointer bits, packed for both tiles (tile A low half-word, tile B high), 8 rows per half-word:
    //   accO  2 bits/row  op code  0 ZERO, 1 MATCH, 2 INSERT, 3 DELETE   = nz * (1 + na * (1 + nb))
    //   accF  2 bits/row  {ins_open < ins_extend, del_open < del_extend}  (the complements of align.cpp:170-171)
    // built arithmetically: a comparison is the sign bit of a packed difference, or min(difference, 1)
    constexpr int QD = (C + 3) / 4;         // column quads per lane (the last one may be partly unused)
    static_assert(!AMAX || LANES == kGroup, "first tiles run on the 16-lane layout");
    static_assert(!(AMAX && TAG), "the arg-max pass keeps plain scores");
    uint32_t accO[QD * 4], accF[QD * 4];
#pragma unroll
    for (int c = 0; c < C; c++) {
        Hm[c] = hbias;                      // H[0][j] = 0
        Mo[c] = kc.open;                    // M[0][j] + gap_open
        Iup[c] = kc.ninf;                   // I[0][j] = -INF
    }
#pragma unroll
    for (int c = 0; c < QD * 4; c++) { accO[c] = 0; accF[c] = 0; }
    uint32_t Mo_last = kc.open, D_last = kc.ninf, Hm_last = hbias;
    uint32_t Hm_left_prev = hbias;
    uint32_t Ml0 = kc.open, Dl0 = kc.ninf, Hl = hbias;          // what lane gl-1 shows

    // arg-max state (AMAX only; RQ[h] = {R, Q} of tile h, no start delay: first tiles store from step 1)
    uint32_t bk[AMAX ? C : 1];
    int lane_best[2] = {-1, -1};
    uint32_t col_x0 = 0;                     // packed (0 - valid columns of this lane)
    int t_first = 0, rows[2] = {0, 0};
    if (AMAX) {
#pragma unroll
        for (int c = 0; c < C; c++) bk[c] = 0xffffffffu;
        const int ncA = imin(imax(RQ[0][1] - gl * C, 0), C), ncB = imin(imax(RQ[1][1] - gl * C, 0), C);
        col_x0 = ((uint32_t)(-ncA) & 0xffffu) | ((uint32_t)(-ncB) << 16);
        t_first = gl + 1;
        rows[0] = RQ[0][0]; rows[1] = RQ[1][0];
    }

    // RAW: the two ref bases as half-words.  LUT: entry bytes are 24 - 8 * code (31 for a pad row), so the
    // shift leaves match - mismatch in byte `code`, or nothing
    auto unpack = [](uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c010c00u); };
    auto lut = [&](uint32_t amount) { return kc.dsub >> (amount & 31u); };
    uint32_t rbp = 0, lutA = 0, lutB = 0;
    auto set_row = [&](uint32_t w) {
        if (RAW) rbp = unpack(w);
        else { lutA = lut(w & 0xffu); lutB = lut(w >> 8); }
    };
    set_row(ref16[1]);

    auto step = [&](const int t, auto ptr_tag) {
        constexpr bool PTR = decltype(ptr_tag)::value;
        const uint32_t w_next = ref16[t + 1];
        uint32_t key_c = 0;
        if (AMAX) {
            const uint32_t sidx = (uint32_t)(t - tB) & 7u, row0 = (uint32_t)(t - t_first);
            const uint32_t ka = row0 < (uint32_t)rows[0] ? sidx : ((uint32_t)kKeyBias & 0xffffu);
            const uint32_t kb = row0 < (uint32_t)rows[1] ? sidx : ((uint32_t)kKeyBias & 0xffffu);
            key_c = ka | (kb << 16);
        }

        // lane 0 of each destination keeps what it held: the j = 0 border it was initialised with
        if (LANES == 32) {
            Ml0 = (uint32_t)dpp_shr1_32((int)Mo_last, (int)Ml0);
            Dl0 = (uint32_t)dpp_shr1_32((int)D_last, (int)Dl0);
            Hl = (uint32_t)dpp_shr1_32((int)Hm_last, (int)Hl);
        } else {
            Ml0 = (uint32_t)dpp_row_shr1((int)Mo_last, (int)Ml0);
            Dl0 = (uint32_t)dpp_row_shr1((int)D_last, (int)Dl0);
            Hl = (uint32_t)dpp_row_shr1((int)Hm_last, (int)Hl);
        }
        uint32_t Hd = Hm_left_prev;
        Hm_left_prev = Hl;

        uint32_t M[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
            // sub = (q==r) ? match : mismatch   (align.cpp:134)
            uint32_t Mx;
            if (RAW) Mx = pk_mad_s(pk_min1(qb[c] ^ rbp), kc.nd, Hd);                 // (H[i-1][j-1] + match) + neq*nd
            else Mx = pk_add(Hd, __builtin_amdgcn_perm(lutB, lutA, qb[c]));         // (H[i-1][j-1] + mismatch) + eq*d
            Hd = Hm[c];
            M[c] = pk_max0(Mx);                                     // :145-147
            const uint32_t Ie = pk_add_s(Iup[c], kc.ext);           // ins_extend :150
            if (PTR) accF[c] = pk_shl_add2(accF[c], pk_sign(pk_sub(Mo[c], Ie)));   // ins_open < ins_extend  (:170)
            Iup[c] = pk_max(Mo[c], Ie);                             // :154
            Mo[c] = pk_add_s(M[c], kc.open);
        }
        uint32_t Ml = Ml0, Dl = Dl0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t De = pk_add_s(Dl, kc.ext);               // del_extend :152
            const uint32_t D = pk_max(Ml, De);                      // :156
            const uint32_t H = pk_max(pk_max(M[c], Iup[c]), D);     // :158-160 (M >= 0)
            if (PTR) {
                accF[c] = pk_shl_add2(accF[c], pk_sign(pk_sub(Ml, De)));       // del_open < del_extend  (:171)
                // :162-168 with M >= 0: ZERO iff H == 0, else MATCH iff M == H, else INSERT iff I == H, else DELETE
                const uint32_t nz = pk_min1(H);
                const uint32_t na = pk_min1(pk_sub(H, M[c]));       // M < H
                const uint32_t nb = pk_min1(pk_sub(H, Iup[c]));     // I < H
                const uint32_t o = pk_mad_vvv(na, nb, na);          // 0 MATCH, 1 INSERT, 2 DELETE
                accO[c] = pk_shl_add4(accO[c], pk_mad_vvv(nz, o, nz));
            }
            if (AMAX) bk[c] = pk_max(bk[c], pk_mad8(H, key_c));
            Hm[c] = pk_add_s(H, hbias);
            Ml = Mo[c];
            Dl = D;
        }
        Mo_last = Ml;
        D_last = Dl;
        Hm_last = Hm[C - 1];
        set_row(w_next);
    };

    // ---- TAG: the pointer-phase step on tagged scores (register contents as in dp_pass_p16s)
    const uint32_t hbias4 = RAW ? kc.match4 : kc.mism4;
    const uint32_t vmask = kc.nmask;
    auto lut4 = [&](uint32_t amount) { return kc.dsub4 >> (amount & 31u); };
    auto step_tagged = [&](const int t) {
        const uint32_t w_next = ref16[t + 1];
        if (LANES == 32) {
            Ml0 = (uint32_t)dpp_shr1_32((int)Mo_last, (int)Ml0);
            Dl0 = (uint32_t)dpp_shr1_32((int)D_last, (int)Dl0);
            Hl = (uint32_t)dpp_shr1_32((int)Hm_last, (int)Hl);
        } else {
            Ml0 = (uint32_t)dpp_row_shr1((int)Mo_last, (int)Ml0);
            Dl0 = (uint32_t)dpp_row_shr1((int)D_last, (int)Dl0);
            Hl = (uint32_t)dpp_row_shr1((int)Hm_last, (int)Hl);
        }
        uint32_t Hd = Hm_left_prev;
        Hm_left_prev = Hl;
        uint32_t M[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
            uint32_t Mx;                                                    // 4(H[i-1][j-1] + sub) + 3
            if (RAW) Mx = pk_mad_s(pk_min1(qb[c] ^ rbp), kc.nd4, Hd);
            else Mx = pk_add(Hd, __builtin_amdgcn_perm(lutB, lutA, qb[c]));
            Hd = Hm[c];
            M[c] = pk_max_s(Mx, kc.c3);                                     // 4M + 3, M >= 0
            Iup[c] = pk_max(Mo[c], Iup[c]);                                 // bit 0: ins_open >= ins_extend
            Mo[c] = pk_add_s(M[c], kc.open4m2);                             // 4(M + open) + 1
        }
        uint32_t Ml = Ml0, Dl = Dl0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t De = pk_add_s(Dl, kc.ext4m1);
            const uint32_t Dp = pk_max(Ml, De);                             // bit 0: del_open >= del_extend
            const uint32_t Dt = and_or(Dp, vmask, kc.tag1);
            const uint32_t It = and_or(Iup[c], vmask, kc.tag2);
            const uint32_t Hp = pk_max(pk_max(M[c], It), Dt);
            accF[c] = pk_shl_add4(accF[c], pk_shl_add2(Iup[c], Dp) & kc.c3);
            accO[c] = pk_shl_add4(accO[c], pk_mul(Hp & kc.c3, pk_min1(pk_lshr2(Hp))));       // H == 0: ZERO
            Hm[c] = pk_add_s(Hp | kc.c3, hbias4);
            Iup[c] = pk_add_s(It, kc.ext4m2);                               // the next row's ins_extend
            Ml = Mo[c];
            Dl = Dt;
        }
        Mo_last = Ml;
        D_last = Dl;
        Hm_last = Hm[C - 1];
        if (RAW) rbp = unpack(w_next);
        else { lutA = lut4(w_next & 0xffu); lutB = lut4(w_next >> 8); }
    };
    auto enter_tagged = [&]() {
#pragma unroll
        for (int c = 0; c < C; c++) {
            Hm[c] = pk_mad4(Hm[c], kc.c3);
            Mo[c] = pk_mad4(Mo[c], kc.tag1);
            Iup[c] = pk_add_s(pk_mad4(pk_max_s(Iup[c], kc.floor4), kc.tag2), kc.ext4m2);
        }
        Mo_last = pk_mad4(Mo_last, kc.tag1); Ml0 = pk_mad4(Ml0, kc.tag1);
        D_last = pk_mad4(pk_max_s(D_last, kc.floor4), kc.tag1); Dl0 = pk_mad4(pk_max_s(Dl0, kc.floor4), kc.tag1);
        Hm_last = pk_mad4(Hm_last, kc.c3); Hl = pk_mad4(Hl, kc.c3); Hm_left_prev = pk_mad4(Hm_left_prev, kc.c3);
        if (!RAW) { lutA <<= 2; lutB <<= 2; }
    };

    // tile A's word = {accF.lo, accO.lo}, tile B's = {accF.hi, accO.hi}: flags in the high half-word
    auto wordA = [](uint32_t o, uint32_t f) { return __builtin_amdgcn_perm(f, o, 0x05040100u); };
    auto wordB = [](uint32_t o, uint32_t f) { return __builtin_amdgcn_perm(f, o, 0x07060302u); };

    // fold the block keys of stored steps kblk..kblk+7 into lane_best
    auto fold = [&](const int kblk) {
        uint32_t m = 0xffffffffu, rel = 0, x = col_x0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t v = pk_sign(x);                              // column c exists in this tile
            x = pk_add(x, kc.one);
            const uint32_t key = pk_mad_m1(v, pk_add(bk[c], kc.one));   // v ? key : -1
            const uint32_t keep = pk_sign(pk_sub(key, m));              // key < m: the earlier column stays
            rel = pk_mad_m1(keep, rel);                                 // (best column) - (next c)
            m = pk_max(m, key);
            bk[c] = 0xffffffffu;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int m16 = (int)(m << (16 - 16 * h)) >> 16;
            const int col = ((int)(rel << (16 - 16 * h)) >> 16) + C;
            const int rec = ((m16 >> 3) << 15) | ((kblk + (m16 & 7)) << 5) | col;    // 11 + 10 + 5 bits
            lane_best[h] = imax(lane_best[h], m16 < 0 ? -1 : rec);
        }
    };

    int t = 1;
    for (; t < tB && t <= T_end; t++) step(t, std::false_type{});
    if (TAG) enter_tagged();
    uint4 *qA = reinterpret_cast<uint4 *>(wsA) + gl;
    uint4 *qB = reinterpret_cast<uint4 *>(wsB) + gl;
    int k = 0;
    for (; t <= T_end; t++, k++) {
        if (TAG) step_tagged(t); else step(t, std::true_type{});
        if ((k & 7) == 7) {
#pragma unroll
            for (int q = 0; q < QD; q++) {
                qA[q * kWsRow] = make_uint4(wordA(accO[4 * q], accF[4 * q]), wordA(accO[4 * q + 1], accF[4 * q + 1]),
                                            wordA(accO[4 * q + 2], accF[4 * q + 2]), wordA(accO[4 * q + 3], accF[4 * q + 3]));
                qB[q * kWsRow] = make_uint4(wordB(accO[4 * q], accF[4 * q]), wordB(accO[4 * q + 1], accF[4 * q + 1]),
                                           wordB(accO[4 * q + 2], accF[4 * q + 2]), wordB(accO[4 * q + 3], accF[4 * q + 3]));
            }
            qA += QD * kWsRow;
            qB += QD * kWsRow;
            if (AMAX) fold(k - 7);
        }
    }
    if (AMAX) {
        if (k & 7) fold(k & ~7);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int rec = lane_best[h];
            int best = 0, bi = 0, bj = 0;                               // align.cpp:109-112
            if (rec >= 0) {
                best = rec >> 15;
                bi = tB + ((rec >> 5) & 1023) - gl;
                bj = gl * C + (rec & 31) + 1;
            }
            // merge the 16 lanes: largest H, then largest i, then largest j
#pragma unroll
            for (int mm = 1; mm < kGroup; mm <<= 1) {
                const int ob = __shfl_xor(best, mm, kGroup);
                const int oi = __shfl_xor(bi, mm, kGroup);
                const int oj = __shfl_xor(bj, mm, kGroup);
                const bool take = (ob > best) | ((ob == best) & ((oi > bi) | ((oi == bi) & (oj > bj))));
                best = take ? ob : best;
                bi = take ? oi : bi;
                bj = take ? oj : bj;
            }
            pb->best[h] = best; pb->bi[h] = bi; pb->bj[h] = bj;
        }
    }
    if (k & 7) {
        // left-justify the partial block: each half-word holds 2 bits per stored row
        const int sh = 2 * (8 - (k & 7));
        auto just = [sh](uint32_t w) { return ((w & 0xffffu) << sh & 0xffffu) | ((w >> 16) << sh << 16); };
#pragma unroll
        for (int q = 0; q < QD; q++) {
            qA[q * kWsRow] = make_uint4(just(wordA(accO[4 * q], accF[4 * q])), just(wordA(accO[4 * q + 1], accF[4 * q + 1])),
                                        just(wordA(accO[4 * q + 2], accF[4 * q + 2])), just(wordA(accO[4 * q + 3], accF[4 * q + 3])));
            qB[q * kWsRow] = make_uint4(just(wordB(accO[4 * q], accF[4 * q])), just(wordB(accO[4 * q + 1], accF[4 * q + 1])),
                                        just(wordB(accO[4 * q + 2], accF[4 * q + 2])), just(wordB(accO[4 * q + 3], accF[4 * q + 3])));
        }
    }
}

// ---------------------------------------------------------------------------
// Loads the two tiles of a group: ref stream and query bytes into LDS, per-slot query operand into qb.
// RAW: bytes as they are (raw set bytes or 2-bit codes), pads 0xFF / 0xFE, qb = the two query bytes as half-words.
// Otherwise (2-bit sets, LUT form of dp_pass_p16): LDS bytes are 24 - 8 * code, a pad row is 31, a pad column
// 0xFE; qb = v_perm selector {code of tile A, zero, 4 + code of tile B, zero}, a pad column selects the constant 0.
constexpr uint32_t kLutPadRow = 31, kPermZero = 0x0c;
struct PairTile {
    int R[kSlots], Q[kSlots], shift[kSlots];
    bool reverse[kSlots];
    int64_t rp0[kSlots], qp0[kSlots];
    int comp[kSlots];
    int col_from;            // first DP column (1-based) a non-first tile's walk can reach, the smaller of the two tiles'
    int band;                // linear-gap passes: pointer words are stored within this many columns of the diagonal through (R, Q) ...
    bool full[kSlots];       // ... unless the tile stores its whole window (gact_lin.hpp LinBand; 0: every tile does)
};

template <int C, bool RAW, int LANES = kGroup>
__device__ __forceinline__ void load_pair(const SeqSetDev &rs, const SeqSetDev &qfwd, const SeqSetDev &qrc,
                                          const PairTile &pt, int gl, uint8_t *ref8, uint8_t *q8,
                                          uint32_t (&qb)[C])
{
    using G = GeometryP16<C, LANES>;
    // pad the whole stream: rows in front of row 1 (skew + start delay) and behind row R
    GACT_STAMP(l_a);
    uint32_t *ref32 = reinterpret_cast<uint32_t *>(ref8);
    for (int k = gl; k < G::kRefBytes / 4; k += LANES) ref32[k] = RAW ? 0xffffffffu : kLutPadRow * 0x01010101u;
    wave_sync();
    GACT_STAMP(l_b);
    // all loads first (addresses clamped into the slice, never predicated, so they are all in
    // flight together), pads substituted afterwards
    uint32_t rv[kSlots][C], qv[kSlots][C];
#pragma unroll
    for (int h = 0; h < kSlots; h++) {
        const SeqSetDev &qs = pt.comp[h] ? qrc : qfwd;
#pragma unroll
        for (int c = 0; c < C; c++) {
            rv[h][c] = fetch_base<RAW>(rs, slice_pos(pt.rp0[h], pt.R[h], pt.reverse[h], gl * C + c));
            qv[h][c] = fetch_base<RAW>(qs, slice_pos(pt.qp0[h], pt.Q[h], pt.reverse[h], gl * C + c));
        }
    }
#ifdef GACT_STAMPS
    { uint32_t x = 0;
#pragma unroll
      for (int h = 0; h < kSlots; h++)
#pragma unroll
        for (int c = 0; c < C; c++) x ^= rv[h][c] ^ qv[h][c];
      asm volatile("" :: "v"(x)); }
#endif
    GACT_STAMP(l_c);
#pragma unroll
    for (int c = 0; c < C; c++) qb[c] = RAW ? 0u : kPermZero * 0x01010101u;
#pragma unroll
    for (int h = 0; h < kSlots; h++) {
        const int R = pt.R[h], Q = pt.Q[h];
        uint8_t *rrow = ref8 + (LANES + pt.shift[h]) * 2 + h;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int d = gl * C + c;
            if (RAW) {
                const uint32_t qcode = (d < Q) ? qv[h][c] : kQueryPad;
                if (d < R) rrow[d * 2] = (uint8_t)rv[h][c];
                q8[h * G::kTileMax + d] = (uint8_t)qcode;
                qb[c] |= qcode << (16 * h);
            } else {
                if (d < R) rrow[d * 2] = (uint8_t)(24u - rv[h][c] * 8u);
                q8[h * G::kTileMax + d] = (uint8_t)((d < Q) ? 24u - qv[h][c] * 8u : kQueryPad);
                if (d < Q) qb[c] = (qb[c] & ~(0xffu << (16 * h))) | ((qv[h][c] + 4u * h) << (16 * h));
            }
        }
    }
#ifdef GACT_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GACT_STAMP(l_d);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_stamps2[0], l_b - l_a); atomicAdd(&g_stamps2[1], l_c - l_b); atomicAdd(&g_stamps2[2], l_d - l_c);
        atomicAdd(&g_stamps2[3], 1ull);
    }
#endif
}

// ---------------------------------------------------------------------------
// The same for 2-bit sets, without the per-base loads: a tile's ref and query slices are at most 21 packed words
// each, so the group's lanes copy those words into an LDS staging area (one or two coalesced loads per lane and
// slice, all in flight at once) and every lane then cuts its bases out of LDS.  load_pair / load_pair_split issue
// one global load per base -- 160 per lane -- into register arrays that do not fit next to the DP state of the
// chain kernels: the compiler spilled them, waiting for every load before storing it to scratch, ~70 serialised
// memory round trips per pass, a fifth of a wave's time in the main launch.
// COL(gl, s, Q) = 0-based DP column of slot s (negative: a pad column left of the tile); rows are dealt to the
// loader's slots as gl * CT + s.
// Staged slice of a tile: two spare words in front (a run read back to front may begin up to CT - 1 bases before
// the slice), the slice's own words (it starts anywhere inside its first word), three behind (a 32-base window
// may reach past its end).
template <int CT, int LANES> struct StageGeom {
    static constexpr int kFront = 2;
    static constexpr int kSeg = kFront + CT * LANES / 16 + 1 + 3;
    static constexpr int kWords = 4 * kSeg;         // {tile A, tile B} x {ref, query}
};

// N consecutive DP indices d0 .. d0+N-1 of a slice of `len` bases whose first base sits `bit0` bases into the
// staged words: their 2-bit codes, cut out of one 64-bit window (three LDS words, two funnel shifts, one
// v_bfe per code) instead of one LDS read per base.  Read back to front when `reverse` (align.cpp:130-131).
// Codes of indices outside 0..len-1 are garbage; the caller masks them.
template <int N, int KFRONT>
__device__ __forceinline__ void cut_run(const uint32_t *seg_words, int bit0, int len, bool reverse, int d0, uint32_t (&code)[N])
{
    static_assert(N <= 32, "one 64-bit window");
    typedef __attribute__((address_space(3))) const uint32_t LdsWord;
    LdsWord *st = (LdsWord *)seg_words;
    // lowest base of the run, in staged coordinates (front padding included)
    const int lo = imax(16 * KFRONT + bit0 + (reverse ? len - d0 - N : d0), 0);
    const int wi = lo >> 4;
    const uint32_t sh = 2u * (uint32_t)(lo & 15);
    const uint32_t w0 = st[wi], w1 = st[wi + 1], w2 = st[wi + 2];
    const uint32_t x_lo = __builtin_amdgcn_alignbit(w1, w0, sh), x_hi = __builtin_amdgcn_alignbit(w2, w1, sh);
#pragma unroll
    for (int k = 0; k < N; k++) {
        const int p = reverse ? N - 1 - k : k;          // position of index d0 + k inside the window
        // (a select between two compile-time positions, not a variable shift)
        const uint32_t fwd = k < 16 ? (x_lo >> (2 * k)) & 3u : (x_hi >> (2 * (k - 16))) & 3u;
        const int pr = N - 1 - k;
        const uint32_t rev = pr < 16 ? (x_lo >> (2 * pr)) & 3u : (x_hi >> (2 * (pr - 16))) & 3u;
        (void)p;
        code[k] = reverse ? rev : fwd;
    }
}

// COL::runs: the lane's slots as runs of consecutive DP columns -- first slot, slot count, and the column of the
// run's first slot (which may be negative: pad columns left of the tile).
// WRITE_Q8 = false: the caller's walker cuts its bases out of the staged words (gact_coop.hpp) and keeps no query bytes
template <int CT, int LANES, class COL, bool WRITE_Q8 = true>
__device__ __forceinline__ void load_pair_packed(const SeqSetDev &rs, const SeqSetDev &qfwd, const SeqSetDev &qrc,
                                                 const PairTile &pt, int gl, uint8_t *ref8, int ref_bytes, int row0,
                                                 uint8_t *q8, int q_stride, uint32_t (&qb)[CT], uint32_t *stage, COL)
{
    using SG = StageGeom<CT, LANES>;
    constexpr int kStageSeg = SG::kSeg;
    GACT_STAMP(l_a);
    uint32_t *ref32 = reinterpret_cast<uint32_t *>(ref8);
    for (int k = gl; k < ref_bytes / 4; k += LANES) ref32[k] = kLutPadRow * 0x01010101u;
#pragma unroll
    for (int seg = 0; seg < 4; seg++) {
        const int h = seg >> 1;
        const bool is_q = seg & 1;
        const uint32_t *words = is_q ? (pt.comp[h] ? qrc.packed : qfwd.packed) : rs.packed;
        const int64_t p0 = is_q ? pt.qp0[h] : pt.rp0[h];
        const int len = imax(is_q ? pt.Q[h] : pt.R[h], 1);
        const int64_t w0 = p0 >> 4;
        const int last = (int)(((p0 + len - 1) >> 4) - w0);
        for (int k = gl; k < kStageSeg - SG::kFront; k += LANES)
            stage[seg * kStageSeg + SG::kFront + k] = words[w0 + imin(k, last)];
    }
    wave_sync();
    GACT_STAMP(l_b);
#pragma unroll
    for (int h = 0; h < kSlots; h++) {
        const int R = pt.R[h], Q = pt.Q[h];
        const bool rev = pt.reverse[h];
        // ---- ref stream: DP rows gl * CT .. + CT - 1 of this tile.  Row R's entry is a pad row: rows past the
        //      tile write the pad there, so the stores need no predicate
        uint8_t *rrow = ref8 + (row0 + pt.shift[h]) * 2 + h;
        uint32_t rc[CT];
        cut_run<CT, SG::kFront>(stage + (2 * h) * kStageSeg, (int)(pt.rp0[h] & 15), R, rev, gl * CT, rc);
#pragma unroll
        for (int s = 0; s < CT; s++) {
            const int d = gl * CT + s;
            rrow[imin(d, R) * 2] = (uint8_t)(d < R ? 24u - rc[s] * 8u : kLutPadRow);
        }
        // ---- query: the walker's byte per column, and the slot's v_perm selector
        const uint32_t *qseg = stage + (2 * h + 1) * kStageSeg;
        const int qbit0 = (int)(pt.qp0[h] & 15);
        auto run = [&](auto first_tag, auto n_tag) {
            constexpr int S0 = decltype(first_tag)::value, N = decltype(n_tag)::value;
            const int c0 = COL::column(gl, S0, Q);
            uint32_t qc[N];
            cut_run<N, SG::kFront>(qseg, qbit0, Q, rev, c0, qc);
#pragma unroll
            for (int k = 0; k < N; k++) {
                const int dq = c0 + k;
                const bool real = (unsigned)dq < (unsigned)Q;
                if (WRITE_Q8 && real) q8[h * q_stride + dq] = (uint8_t)(24u - qc[k] * 8u);
                const uint32_t keep = h ? 0xff00ffffu : 0xffffff00u;
                const uint32_t mine = (h ? qb[S0 + k] : kPermZero * 0x01010101u);       // tile A initialises the selector
                qb[S0 + k] = real ? (mine & keep) | ((qc[k] + 4u * h) << (16 * h)) : mine;
            }
        };
        COL::for_each_run(run);
    }
#ifdef GACT_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GACT_STAMP(l_c);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_stamps2[4], l_b - l_a); atomicAdd(&g_stamps2[5], l_c - l_b); atomicAdd(&g_stamps2[6], 1ull);
    }
#endif
}

// the uniform layouts: one run of C consecutive columns per lane
template <int C> struct UniformCols {
    __device__ static int column(int gl, int slot, int Q) { (void)Q; return gl * C + slot; }
    template <class F> __device__ static void for_each_run(F &&f)
    { f(std::integral_constant<int, 0>{}, std::integral_constant<int, C>{}); }
};

// ---------------------------------------------------------------------------
// Column layout policy of the packed pass: how a tile's columns map to (lane, slot),
// which pass / loader go with it, and where the walker finds a cell's pointer word.
template <int C, int LANES = kGroup, bool TAG = false> struct UniformLayout {
    using G = GeometryP16<C, LANES>;
    static constexpr int kLanes = LANES;                          // lanes per tile pair
    static constexpr int kSlotsPerLane = C;
    static constexpr int kWalkCols = C, kWalkQuads = G::kQuads, kWalkFmt = TAG ? 2 : 1;   // for the walker
    static constexpr int kRow0 = LANES;                           // ref stream entry of (delay 0, row 1)
    static constexpr bool kEndAligned = false;                    // tiles are delayed no more than the pointer start asks
    __device__ static int fin_lane(int Q) { (void)Q; return 0; }
    // register budget: 32 columns per lane need a whole SIMD's file
    static constexpr int kBlocksPerCu = C <= 20 ? 3 : 1;
    __device__ static int last_step(int R, int Q) { return gact::last_step<C>(R, Q); }
    __device__ static int first_pointer_step(int R, int Q, int early) { return gact::first_pointer_step<C>(R, Q, early, false); }
    template <bool RAW>
    __device__ static void load(const SeqSetDev &rs, const SeqSetDev &qf, const SeqSetDev &qr,
                                const PairTile &pt, int gl, uint8_t *ref8, uint8_t *q8, uint32_t (&qb)[C], uint32_t *stage)
    {
        if (RAW) load_pair<C, RAW, LANES>(rs, qf, qr, pt, gl, ref8, q8, qb);
        else load_pair_packed<C, LANES>(rs, qf, qr, pt, gl, ref8, G::kRefBytes, kRow0, q8, G::kTileMax, qb, stage,
                                        UniformCols<C>{});
    }
    template <bool RAW>
    __device__ static uint32_t pass(const P16Consts &kc, int gl, const uint16_t *ref16, const uint32_t (&qb)[C], int T_end,
                                    int tB, uint32_t *wsA, uint32_t *wsB, const PairTile &)
    { dp_pass_p16<C, false, RAW, LANES, TAG>(kc, gl, ref16, qb, T_end, tB, wsA, wsB); return 0; }
    // start cell (R, Q) of the traceback: lane, column in lane, stored step (tB_tile = tile's own first stored step)
    __device__ static void walk_start(int R, int Q, int tB_tile, int &l, int &c, int &k)
    { l = (Q - 1) / C; c = (Q - 1) - l * C; k = R + l - tB_tile; }
    __device__ static int tile_tB(int tB, int shift) { return tB - shift; }
};

// The latency-bound regime (fewer chains than tile slots: the launch lasts as long as its longest chain): a tile
// pair on 32 lanes with 10 columns each -- half the instructions per step, so a chain advances about twice as
// fast, at 2 % more instructions per cell (both regions of a lane pay for pointers) and half as many tiles per wave.
using WideLayout = UniformLayout<10, 32>;
using WideLayoutTagged = UniformLayout<10, 32, true>;

// span of the look-ahead walker's region cache (layouts of the linear-gap format name theirs) and the LDS scratch a
// walker of layout L needs
template <class L> constexpr int walk_span()
{
    if constexpr (L::kWalkFmt == 3) return L::kWalkSpan; else return 16;
}
// which walker a layout of the linear-gap format gets: the team of eight lanes where a wave holds four tiles and the
// walk is 40 % of its time (wide layout: -4 % on the ONT-shape workload), one lane per tile where it holds eight and
// the walk hides behind the other waves' passes (split layout: the team was +4 % there); -DGACT_WALK_TEAM=0 / 1 forces
#ifndef GACT_WALK_TEAM
#define GACT_WALK_TEAM -1
#endif
template <class L, class = void> struct layout_team_walk : std::false_type {};
template <class L> struct layout_team_walk<L, std::enable_if_t<L::kTeamWalk>> : std::true_type {};
template <class L> constexpr bool walk_by_team()
{
    if constexpr (L::kWalkFmt != 3) return false;
    else if (GACT_WALK_TEAM >= 0) return GACT_WALK_TEAM != 0;
    else return L::kLanes == 32 || layout_team_walk<L>::value;       // (a layout may ask for the team: SplitLayoutLinTeam)
}
template <class L, bool TEAM> constexpr int walk_scratch_words()
{
    if constexpr (TEAM) return LaRegion<L::kWalkCols, L::kWalkQuads, walk_span<L>()>::kWords; else return kTbScratchWords;
}

// ---------------------------------------------------------------------------
// Persistent main kernel: every group carries two candidates (slot A / slot B).
// RAW: the sets hold bytes other than ACGT and are compared as raw bytes (see dp_pass_p16).
// TWO_SETS: the launch may go on with a second set of queues (overlapped seeding, ChainQueues::more_flag); a variant of its
// own so that the plain launch carries none of its state through the DP loop.
template <class L, bool RAW, bool TWO_SETS = false>
__global__ __launch_bounds__(kBlockThreads, L::kBlocksPerCu) void extend_p16_kernel(
    KParams kp, P16Consts kc, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc,
    int same_file, gact_overlap *__restrict__ out, ChainQueues cq,
    uint32_t *__restrict__ ws_all)
{
    using G = typename L::G;
    constexpr int LANES = L::kLanes;                       // lanes per tile pair: 16, or 32 in the wide layout
    constexpr int kGroupsOfWave = 64 / LANES;
    constexpr int kGroupsPerBlock = (kBlockThreads / 64) * kGroupsOfWave;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kGroupsPerBlock * G::kGroupLds];
    __shared__ ChainState chain_lds[kGroupsPerBlock][kSlots];
    // walker scratch (region cache) of the group's two tiles; the loader's staging area lives in the same bytes: a
    // group's loader has finished before its pass starts, its walkers start after it
    constexpr bool kTeamWalk = walk_by_team<L>();
    constexpr int kStageHalf = (StageGeom<L::kSlotsPerLane, LANES>::kWords / kSlots + 3) & ~3;
    constexpr int kScratchWords = walk_scratch_words<L, kTeamWalk>() > kStageHalf ? walk_scratch_words<L, kTeamWalk>() : kStageHalf;
    __shared__ __attribute__((aligned(16))) uint32_t tb_lds[kGroupsPerBlock][kSlots][kScratchWords];

    WaveCtx w;
    {
        const int lane = threadIdx.x & 63;
        const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        w.gl = lane & (LANES - 1);
        w.g = lane / LANES;
        w.slot = wave * kGroupsOfWave + w.g;
        w.n_slots = ((gridDim.x * blockDim.x) >> 6) * kGroupsOfWave;
    }
    const int group_in_block = (threadIdx.x >> 6) * kGroupsOfWave + w.g;
    uint8_t *ref8 = lds + group_in_block * G::kGroupLds;
    uint8_t *q8 = ref8 + G::kRefBytes;
    const uint16_t *ref16_lane = reinterpret_cast<const uint16_t *>(ref8) + (L::kRow0 - 1 - w.gl);
    // a wave owns 8 tile workspaces' worth of kp.ws_words, and its tiles' words are interleaved in it: row =
    // [tile A | tile B][the wave's 64 lanes] uint4 (kWsRow, gact_device.hpp), so a store instruction writes one KB
    constexpr int kWsPerTile = LANES / kGroup;
    static_assert(G::kWsWords <= kWsPerTile * Geometry<20>::kWsWords || L::kSlotsPerLane > 20, "workspace stride");
    uint32_t *wsA = ws_all + (size_t)(w.slot / kGroupsOfWave) * (8 * (size_t)kp.ws_words) + (w.g * LANES) * 4;
    uint32_t *wsB = wsA + 64 * 4;

    ChainState *st = chain_lds[group_in_block];
    if (w.gl < kSlots) { st[w.gl].phase = 2; st[w.gl].cand = -1; }
    wave_sync();
    bool exhausted = false;
    // longest chains first (ChainQueues) -- except that a launch with a critical lane beside it (cq.leave_longest, gact_chain.hpp)
    // starts behind the classes that hold the longest `leave_longest` chains of its own set and comes back to them last:
    // bucket = (bucket_first + my_bucket) mod kBuckets
    int my_bucket = 0;
    int bucket_first = 0;
    if (cq.leave_longest > 0) {
        int acc = 0;
        while (bucket_first < kBuckets && (acc += cq.bucket_count[bucket_first]) <= cq.leave_longest) bucket_first++;
        if (bucket_first >= kBuckets) bucket_first = 0;
    }
    // the set of queues this group pops from: the launch's own, later (overlapped seeding) the second one.  (One flag per
    // lane, the pointers are picked where they are used: the DP loop owns the register file)
    bool second_set = false;
    constexpr bool one_set = !TWO_SETS;              // nothing to switch to
    int idle_polls = 0;
    // (the ranking against the longest chain running: every kRankEvery-th tile, the waves of a launch taking turns; the first tile always)
    int rank_turn = 0, rank_cached = 0;
    __builtin_amdgcn_s_setprio(3);
#ifdef GACT_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tl_cyc0 = __builtin_amdgcn_s_memtime();
    unsigned long long tl_empty = 0;
#endif

    for (;;) {
        GACT_STAMP(t_a);
        // ---- control phase: both slots pick their next tile (chain state lives in LDS
        //      between passes so that the DP loop owns the register file)
        PairTile pt;
        bool have[kSlots];
        int Tend_h[kSlots], tB_h[kSlots];
        int longest = 0;             // bases the longest chain of this group still has to cover
#pragma unroll
        for (int h = 0; h < kSlots; h++) {
            ChainState s = st[h];
            TilePick pk;
            pk.have = false; pk.R = 0; pk.Q = 0; pk.reverse = false; pk.rp0 = 0; pk.qp0 = 0;
            for (int guard = 0; guard < 3 && !pk.have; guard++) {
                if (s.phase == 2) {
                    if (exhausted) break;
                    int cand = -1;
                    for (;;) {
                        const int *q_count = (TWO_SETS && second_set) ? cq.more_count : cq.bucket_count;
                        int *q_pop = (TWO_SETS && second_set) ? cq.more_pop : cq.bucket_pop;
                        const int *q_live = (TWO_SETS && second_set) ? cq.more_live : cq.live;
                        while (my_bucket < kBuckets) {
                            // look before popping: an atomic on a class that is empty or drained is one of ~12,000
                            // (every group comes by) serialised on one address -- 4 ms of a 55 ms launch with 32 empty classes
                            const int bkt = bucket_first + my_bucket - (bucket_first + my_bucket >= kBuckets ? kBuckets : 0);
                            const int cnt = q_count[bkt];
                            int idx = cnt;
                            if (w.gl == 0 && __hip_atomic_load(&q_pop[bkt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cnt)
                                idx = atomicAdd(&q_pop[bkt], 1);
                            idx = __shfl(idx, 0, LANES);
                            if (idx < cnt) {
                                cand = q_live[(size_t)bkt * cq.live_stride + idx];
                                break;
                            }
                            my_bucket++;
                        }
                        if (cand >= 0 || second_set || one_set) break;
                        // this launch's own queues are empty.  Overlapped seeding: a second set is being filled by a seed
                        // launch on another stream; *more_flag is written in stream order BEHIND that launch, so once it
                        // reads non-zero the launch has ended and its writes are in memory -- an acquire at agent scope
                        // (this XCD's caches drop what they hold of them) and the wave goes on with that set
                        if (__hip_atomic_load(cq.more_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) break;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        second_set = true;
                        my_bucket = 0;
                    }
                    if (cand < 0) {
                        // (no second set yet: not exhausted, the slot stays empty for this pass)
                        if (second_set || one_set) {
                            exhausted = true;
#ifdef GACT_STAMPS
                            if (!tl_empty) tl_empty = __builtin_amdgcn_s_memrealtime();
#endif
                        }
                        break;
                    }
                    s = cq.states[cand];
                }
                pk = chain_pick(s, kp, same_file, out, w.gl == 0);
            }
            have[h] = pk.have;
            pt.R[h] = pk.R; pt.Q[h] = pk.Q; pt.reverse[h] = pk.reverse;
            pt.rp0[h] = pk.rp0; pt.qp0[h] = pk.qp0; pt.comp[h] = s.comp; pt.shift[h] = 0;
            pt.full[h] = s.full != 0;
            if (pk.have) longest = imax(longest, chain_remaining(s));
            Tend_h[h] = L::last_step(pk.R, pk.Q);
            tB_h[h] = L::first_pointer_step(pk.R, pk.Q, kp.early);
            wave_sync();
            if (w.gl == 0) st[h] = s;
            wave_sync();
        }
        const bool any_here = have[0] | have[1];
        if (!__any(any_here)) {
            if (__all(exhausted && st[0].phase == 2 && st[1].phase == 2)) break;
            if (!second_set && !one_set) {
                // a wave with nothing to do while the second set of queues is not open yet: wait a little, but never
                // for ever -- the launch that opens it may be waiting for this wave's registers (several engine slots
                // at work); what is left over is taken by the main launch queued behind that seed launch
                __builtin_amdgcn_s_sleep(127);
                if (++idle_polls > 512) { second_set = true; exhausted = true; my_bucket = kBuckets; }
            }
            continue;
        }
        // common end / common pointer start over the wave's 8 tiles (align_starts)
        const int T_end = wave_max_groups<LANES>(imax(have[0] ? Tend_h[0] : 0, have[1] ? Tend_h[1] : 0));
        const int reach0 = have[0] ? tB_h[0] + (T_end - Tend_h[0]) : 0x7fffffff;
        const int reach1 = have[1] ? tB_h[1] + (T_end - Tend_h[1]) : 0x7fffffff;
        const int tB = wave_min_groups<LANES>(imin(reach0, reach1));
        // (a layout whose walker wants H[R][Q] from the pass delays every tile all the way: last row = last step)
        // the walk starts in column Q and stops after `early` query steps (align.cpp:205)
        pt.col_from = imax(imin(have[0] ? pt.Q[0] : 0x7fff, have[1] ? pt.Q[1] : 0x7fff) - kp.early, 0);
        pt.band = kp.band;
        pt.shift[0] = have[0] ? (L::kEndAligned ? T_end - Tend_h[0] : imax(0, tB - tB_h[0])) : 0;
        pt.shift[1] = have[1] ? (L::kEndAligned ? T_end - Tend_h[1] : imax(0, tB - tB_h[1])) : 0;

        GACT_STAMP(t_b);
        uint32_t qb[L::kSlotsPerLane];
        L::template load<RAW>(refs, qfwd, qrc, pt, w.gl, ref8, q8, qb, tb_lds[group_in_block][0]);
        wave_sync();
        GACT_STAMP(t_c);

        // the DP pass is throughput work; everything else in this loop is a short serial
        // chain (traceback, chain bookkeeping, loads) that must not queue behind other
        // waves' DP instructions: run it at raised issue priority
        // ... and among the DP passes, the waves that carry the longest chains go first: when there are
        // fewer chains than tile slots the launch lasts as long as its longest chain
        const int wave_longest = wave_max_groups<LANES>(longest);
        const int ref_longest = ranked_longest(cq, kp, wave_longest, rank_turn, rank_cached);
        // ranking mode (prio_bases[0] == 0): prio_bases[1] = thresholds in sixteenths of the longest, hi << 8 | mid
        const bool rank_hi = kp.prio_bases[0] == 0 ? 16 * wave_longest > (kp.prio_bases[1] >> 8) * ref_longest
                                                   : wave_longest > kp.prio_bases[1];
        const bool rank_mid = kp.prio_bases[0] == 0 ? 16 * wave_longest > (kp.prio_bases[1] & 255) * ref_longest
                                                    : wave_longest > kp.prio_bases[0];
        if (rank_hi) __builtin_amdgcn_s_setprio(2);
        else if (rank_mid) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        const uint32_t fin = L::template pass<RAW>(kc, w.gl, ref16_lane, qb, T_end, tB, wsA, wsB, pt);
        __builtin_amdgcn_s_setprio(3);
        // FMT 3 walkers start from H[R][Q]: held by the lane of column Q when the pass ends
        int v0_h[kSlots] = {0, 0};
        if (L::kWalkFmt == 3 || L::kWalkFmt == 4) {
            v0_h[0] = (int)(int16_t)(__shfl(fin, L::fin_lane(pt.Q[0]), LANES) & 0xffffu);
            v0_h[1] = (int)(int16_t)(__shfl(fin, L::fin_lane(pt.Q[1]), LANES) >> 16);
        }
        GACT_STAMP(t_d);

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // pointer stores -> L2 before the sc1 loads
        GACT_STAMP(t_e);

        // ---- traceback: lane h of the group walks slot h, all walkers of the wave in one loop
        int ref_steps = 0, query_steps = 0, nst = 0;
        bool redo = false;               // banded pointer stores (gact_lin.hpp LinBand): the walk left the band, the tile runs again
        ScoreWalk wk;
        wk.score = 0; wk.pend_gap = 0; wk.open_flag = 0; wk.have_left = 0; wk.left_first_gap = 0;
        // (the linear-gap format is walked by a team of eight lanes per tile: the first eight lanes of the tile's half
        // of the group, gact_chain.hpp walk_chain_lin_team; the other formats by one lane per tile, lanes 0 and 1)
        constexpr int kWalkLanes = kTeamWalk ? LANES / kSlots : 1;         // lane h * kWalkLanes holds the walk's results
        {
            const int h = kTeamWalk ? w.gl / kWalkLanes : (w.gl & 1);
            const bool mine = (kTeamWalk ? w.gl - h * kWalkLanes < kLaTeam : w.gl < kSlots) && (h ? have[1] : have[0]);
            const int sh = h ? pt.shift[1] : pt.shift[0];
            const int Rh = h ? pt.R[1] : pt.R[0], Qh = h ? pt.Q[1] : pt.Q[0];
            const uint8_t *rrow = ref8 + (L::kRow0 + sh) * 2 + h;
            const uint8_t *qrow = q8 + h * G::kTileMax;
            int l0, c0, k0;
            L::walk_start(Rh, Qh, L::tile_tB(tB, sh), l0, c0, k0);
            // how far from the diagonal through (R, Q) a walk of this tile may be when it refills; -1: every block is there
            const int band_lim = ((L::kWalkFmt == 3 || L::kWalkFmt == 4) && (kp.band & 0xffff) > 0 && !(h ? pt.full[1] : pt.full[0]))
                                     ? (kp.band & 0xffff) - (kTeamWalk ? kLaBandMargin : L::kWalkFmt == 3 ? kLinWalkSpan : kWalkBandMargin) : -1;
            if constexpr (kTeamWalk && GACT_EXP_FAKE_WALK) {
                // timing experiment only (results are wrong): no walk, every tile taken as a diagonal of `early` steps
                wk.load(st[h]);
                ref_steps = query_steps = imin(kp.early, imin(Rh, Qh));
                nst = mine ? 2 * ref_steps : 0;
                wk.score += ref_steps;
            } else if constexpr (kTeamWalk) {
                wk.load(st[h]);
                walk_chain_lin_team<L::kWalkCols, L::kWalkQuads, kWsRow, walk_span<L>()>(tb_lds[group_in_block][h], mine, Rh, Qh, l0, c0, k0, kp.early,
                                                                        rrow, 2, qrow, kp, wk, ref_steps, query_steps, nst,
                                                                        h ? v0_h[1] : v0_h[0], h ? wsB : wsA, ws_all, band_lim, redo);
            } else if (mine) {
                const ChainState &s = st[h];
                wk.load(s);
                walk_chain<L::kWalkCols, L::kWalkFmt, L::kWalkQuads, kWsRow>(h ? wsB : wsA, tb_lds[group_in_block][h], Rh, Qh, l0, c0, k0,
                                                           kp.early, rrow, 2, qrow, s.phase, kp, wk, ref_steps,
                                                           query_steps, nst, h ? v0_h[1] : v0_h[0], ws_all, band_lim, &redo);
            }
        }
        GACT_STAMP(t_f);
        // ---- consume (gact.cpp:111-133 / :172-194); non-first tiles only
#pragma unroll
        for (int h = 0; h < kSlots; h++) {
            if (have[h]) {
                ChainState s = st[h];
                if (__shfl((int)redo, h * kWalkLanes, LANES)) {
                    // the walk left the band its pass stored: nothing of it counts, the same tile comes again (chain_pick
                    // finds the state it found) and stores its whole window
                    s.full = 1;
                    if (w.gl == 0) atomicAdd(cq.band_redos, 1);
                } else {
                    s.full = 0;
                    s.n_tiles++;
                    s.cells += (int64_t)pt.R[h] * pt.Q[h];
                    chain_advance<LANES>(s, false, wk, ref_steps, query_steps, nst, h * kWalkLanes);
                }
                wave_sync();
                if (w.gl == 0) st[h] = s;
            }
            wave_sync();
        }
        GACT_STAMP(t_g);
        GACT_ACC(0, t_a, t_b); GACT_ACC(1, t_b, t_c); GACT_ACC(2, t_c, t_d); GACT_ACC(3, t_d, t_e);
        GACT_ACC(4, t_e, t_f); GACT_ACC(5, t_f, t_g);
#ifdef GACT_STAMPS
        stamp_acc[6] += 1; stamp_acc[7] += (unsigned long long)(T_end - tB + 1);
#endif
    }
#ifdef GACT_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 8; k++) atomicAdd(&g_stamps[k], stamp_acc[k]);
        const int wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (wv < 4096) {
            g_timeline[4 * wv] = tl_start; g_timeline[4 * wv + 1] = tl_empty;
            g_timeline[4 * wv + 2] = __builtin_amdgcn_s_memrealtime(); g_timeline[4 * wv + 3] = stamp_acc[6];
            g_wave_cycles[wv] = __builtin_amdgcn_s_memtime() - tl_cyc0;
        }
    }
#endif
}

// the linear-gap pass (gact_lin.hpp), uniform layout; LIN seed launch below
template <int C, int LANES, bool AMAX>
__device__ __forceinline__ uint32_t dp_pass_lin(const P16Consts &kc, const int gl, const uint16_t *__restrict__ ref16,
                                                const uint32_t (&qb)[C], const int T_end, const int tB,
                                                uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB,
                                                const int cqA, const int cqB, const int (*RQ)[2], P16Best *pb,
                                                const int col_from = 0, const int band = 0, const int QA = 0, const int QB = 0,
                                                const bool fullA = true, const bool fullB = true);

// the drifted affine pass for first tiles (gact_aff.hpp)
template <int C, bool CBNEG>
__device__ __forceinline__ void dp_pass_aff_seed(const P16Consts &kc, const int gl, const uint16_t *__restrict__ ref16,
                                                 const uint32_t (&qb)[C], const int T_end,
                                                 uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB,
                                                 const int (*RQ)[2], P16Best *pb);

// ---------------------------------------------------------------------------
// Packed seed launch: the first tile(s) of every candidate (arg-max, pointers of the
// whole tile), two candidates per group, then the chain is handed to the main launch
// (ChainQueues) exactly as the int32 seed launch does (extend_kernel, seed_mode).
// Needs p16_argmax_ok on top of p16_scoring_ok.  LIN: linear gap scoring on 2-bit sets (gact_lin.hpp).
// MODE: 0 round 1's affine pass (any scoring that fits int16, raw bytes or 2-bit), 1 the linear-gap pass, 2 / 3 the drifted
// affine pass (gact_aff.hpp; 3: mismatch < gap_extend) -- 1..3 on 2-bit sets only
template <int C, bool RAW, int MODE = 0>
__global__ __launch_bounds__(kBlockThreads, MODE == 1 ? 3 : 2) void seed_p16_kernel(
    KParams kp, P16Consts kc, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc,
    const gact_candidate *__restrict__ cands, int first_cand, int n, int rc_from,
    int same_file, gact_overlap *__restrict__ out, ChainQueues cq,
    uint32_t *__restrict__ ws_all)
{
    using L = UniformLayout<C>;
    using G = typename L::G;
    constexpr int kGroupsPerBlock = (kBlockThreads / 64) * kGroupsPerWave;
    __shared__ __attribute__((aligned(16))) uint8_t lds[kGroupsPerBlock * G::kGroupLds];
    __shared__ ChainState chain_lds[kGroupsPerBlock][kSlots];
    // (walker scratch and the loader's staging area in the same bytes, as in extend_p16_kernel)
    constexpr int kStageHalf = (StageGeom<C, kGroup>::kWords / kSlots + 3) & ~3;
    constexpr int kScratchWords = kTbScratchWords > kStageHalf ? kTbScratchWords : kStageHalf;
    __shared__ __attribute__((aligned(16))) uint32_t tb_lds[kGroupsPerBlock][kSlots][kScratchWords];

    constexpr bool LIN = MODE == 1, AFF = MODE >= 2;
    static_assert(!(RAW && MODE != 0), "the drifted passes read 2-bit sets");
    const WaveCtx w = wave_ctx();
    const int group_in_block = (threadIdx.x >> 6) * kGroupsPerWave + w.g;
    uint8_t *ref8 = lds + group_in_block * G::kGroupLds;
    uint8_t *q8 = ref8 + G::kRefBytes;
    const uint16_t *ref16_lane = reinterpret_cast<const uint16_t *>(ref8) + (L::kRow0 - 1 - w.gl);
    uint32_t *wsA = ws_all + (size_t)(w.slot / kGroupsPerWave) * (8 * (size_t)kp.ws_words) + (w.g * kGroup) * 4;     // (see extend_p16_kernel)
    uint32_t *wsB = wsA + 64 * 4;

    ChainState *st = chain_lds[group_in_block];
    if (w.gl < kSlots) { st[w.gl].phase = 2; st[w.gl].cand = -1; st[w.gl].comp = 0; }
    wave_sync();
    bool exhausted = false;
    // (DP cells of the candidates this group is done with: added to the launch's counter once, when the wave leaves -- an atomic
    //  per candidate on the line the candidates are popped from made every pop wait, profiles/r05/ranking_atomics.txt)
    unsigned long long cells_done = 0;
    __builtin_amdgcn_s_setprio(3);

    for (;;) {
        // ---- both slots pick their next first tile, finishing / fetching candidates on the way
        PairTile pt;
        bool have[kSlots];
        int RQ[kSlots][2];
#pragma unroll
        for (int h = 0; h < kSlots; h++) {
            ChainState s = st[h];
            TilePick pk;
            pk.have = false; pk.R = 0; pk.Q = 0; pk.reverse = false; pk.rp0 = 0; pk.qp0 = 0;
            for (int guard = 0; guard < 3 && !pk.have; guard++) {
                if (s.phase == 2) {
                    if (exhausted) break;
                    if (!seed_pop(s, cq, w.gl == 0, [](int v) { return __shfl(v, 0, kGroup); }, cands, first_cand, n, rc_from, refs,
                                  qfwd, qrc)) { exhausted = true; break; }
                }
                pk = chain_pick(s, kp, same_file, out, w.gl == 0);
                if (!pk.have && w.gl == 0) cells_done += (unsigned long long)s.cells;     // finished inside the seed launch
            }
            have[h] = pk.have;
            pt.R[h] = pk.R; pt.Q[h] = pk.Q; pt.reverse[h] = pk.reverse;
            pt.rp0[h] = pk.rp0; pt.qp0[h] = pk.qp0; pt.comp[h] = s.comp; pt.shift[h] = 0;
            RQ[h][0] = pk.R; RQ[h][1] = pk.Q;
            pt.full[h] = true;
            wave_sync();
            if (w.gl == 0) st[h] = s;
            wave_sync();
        }
        if (!__any(have[0] | have[1])) {
            if (__all(exhausted && st[0].phase == 2 && st[1].phase == 2)) break;
            continue;
        }
        // first tiles store pointers from step 1 on, so nobody is delayed; the pass ends with the longest tile
        const int T_end = wave_max4(imax(L::last_step(pt.R[0], pt.Q[0]), L::last_step(pt.R[1], pt.Q[1])));

        uint32_t qb[C];
        L::template load<RAW>(refs, qfwd, qrc, pt, w.gl, ref8, q8, qb, tb_lds[group_in_block][0]);
        wave_sync();

        P16Best pb;
        __builtin_amdgcn_s_setprio(0);
        if constexpr (LIN) dp_pass_lin<C, kGroup, true>(kc, w.gl, ref16_lane, qb, T_end, 1, wsA, wsB, 0, 0, RQ, &pb);
        else if constexpr (AFF) dp_pass_aff_seed<C, MODE == 3>(kc, w.gl, ref16_lane, qb, T_end, wsA, wsB, RQ, &pb);
        else dp_pass_p16<C, true, RAW>(kc, w.gl, ref16_lane, qb, T_end, 1, wsA, wsB, RQ, &pb);
        __builtin_amdgcn_s_setprio(3);

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // pointer stores -> L2 before the sc1 loads

        // ---- first-tile bookkeeping (gact.cpp:99-110 / :162-171), group-uniform
        bool stop[kSlots];
#pragma unroll
        for (int h = 0; h < kSlots; h++) {
            stop[h] = true;
            if (have[h]) {
                ChainState s = st[h];
                s.n_tiles++;
                s.cells += (int64_t)pt.R[h] * pt.Q[h];
                stop[h] = chain_first_tile(s, kp, pt.R[h], pt.Q[h], pb.best[h], pb.bi[h], pb.bj[h]);
                wave_sync();
                if (w.gl == 0) st[h] = s;
            }
            wave_sync();
        }
        // ---- traceback from the arg-max: lane h of the group walks slot h
        int ref_steps = 0, query_steps = 0, nst = 0;
        ScoreWalk wk;
        wk.score = 0; wk.pend_gap = 0; wk.open_flag = 0; wk.have_left = 0; wk.left_first_gap = 0;
        constexpr bool kTeamWalk = LIN && GACT_SEED_WALK_TEAM;              // (see extend_p16_kernel)
        constexpr int kWalkLanes = kTeamWalk ? kGroup / kSlots : 1;
        bool no_redo = false;                                              // (first tiles store every block: nothing to give up)
        {
            const int h = kTeamWalk ? w.gl / kWalkLanes : (w.gl & 1);
            const bool mine = (kTeamWalk || w.gl < kSlots) && !(h ? stop[1] : stop[0]);
            const ChainState &s = st[h];
            wk.load(s);
            const int i0 = h ? pb.bi[1] : pb.bi[0], j0 = h ? pb.bj[1] : pb.bj[0];
            const int l0 = (imax(j0, 1) - 1) / C;
            // (FMT 3 walkers start from the score of their cell: the arg-max)
            if constexpr (kTeamWalk) {
                walk_chain_lin_team<C, ((C + 1) / 2 + 3) / 4, kWsRow, 16>(tb_lds[group_in_block][h], mine, i0, j0, l0, (j0 - 1) - l0 * C,
                                                                      i0 + l0 - 1, kp.early, ref8 + L::kRow0 * 2 + h, 2, q8 + h * G::kTileMax,
                                                                      kp, wk, ref_steps, query_steps, nst, h ? pb.best[1] : pb.best[0],
                                                                      h ? wsB : wsA, ws_all, -1, no_redo);
            } else if (mine) {
                // (FMT 4 carries the score of its cell like FMT 3: the arg-max)
                walk_chain<C, LIN ? 3 : AFF ? 4 : 1, LIN ? ((C + 1) / 2 + 3) / 4 : C / 4, kWsRow>(h ? wsB : wsA, tb_lds[group_in_block][h], i0, j0, l0, (j0 - 1) - l0 * C,
                                                  i0 + l0 - 1, kp.early, ref8 + L::kRow0 * 2 + h, 2, q8 + h * G::kTileMax,
                                                  s.phase, kp, wk, ref_steps, query_steps, nst, h ? pb.best[1] : pb.best[0], ws_all);
            }
        }
        // ---- consume; a chain whose first tile is done belongs to the main launch
#pragma unroll
        for (int h = 0; h < kSlots; h++) {
            if (have[h]) {
                ChainState s = st[h];
                chain_advance(s, stop[h], wk, ref_steps, query_steps, nst, h * kWalkLanes);
                if (!s.first_tile) {
                    if (w.gl == 0) {
                        cq.states[s.cand] = s;
                        const int b = chain_bucket(s, kp);
                        const int slot = atomicAdd(&cq.bucket_count[b], 1);
                        cq.live[(size_t)b * cq.live_stride + slot] = s.cand;
                        cells_done += (unsigned long long)s.cells;
                    }
                    s.phase = 2;
                }
                wave_sync();
                if (w.gl == 0) st[h] = s;
            }
            wave_sync();
        }
    }
    if (w.gl == 0 && cells_done) atomicAdd(cq.seed_cells, cells_done);
}

}  // namespace gact
