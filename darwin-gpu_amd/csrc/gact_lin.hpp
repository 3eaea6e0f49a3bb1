// gact_lin.hpp -- the packed-int16 chain pass for LINEAR gap scoring: gap_open == gap_extend == mismatch =: g
// (the reference's own params.cfg: +1 / -1 / -1 / -1), 2-bit read sets.  Same cells, same results as
// dp_pass_p16s / dp_pass_p16 -- fewer instructions per cell: 6 instead of 11 per cell pair for the scores,
// 10 instead of 22 where pointers are made, and pointers of 2 bits per cell instead of 4.
//
// 1. H alone.  The reference keeps M, I, D (align.cpp:134-160): I[i][j] = max(M, I)[i-1][j] + g.  That differs from
//    H[i-1][j] + g only where D is the strict maximum at (i-1, j), and then
//      H[i-1][j] + g = D[i-1][j] + g <= H[i-1][j-1] + 2g <= H[i-1][j-1] + mismatch <= M[i][j]:
//    the difference never reaches H[i][j], nor the choice between M, I and D there (M wins a tie, align.cpp:162-164).
//    The same for D.  So one value per cell is carried,
//      H[i][j] = max(M, 0, H[i-1][j] + g, H[i][j-1] + g),        M = H[i-1][j-1] + sub,
//    and every H, every op and every arg-max is the reference's.
// 2. Row drift.  Every value of DP row i is kept as X + beta_i with beta_i = -i * g (it grows by |g| per row):
//    H[i-1][j] + g is then H'_up as it stands, the zero level is Z = beta_i, and M' = H'_diag + (sub - g) where
//    sub - g is 0 for a mismatch (mismatch == g) and match - g otherwise: the non-negative byte the look-up word of
//    dp_pass_p16 already holds.  Per cell pair: perm, add | max Z, max H'_up | sub, max = 6 instructions, one
//    register of state per column slot.  The drift is tied to the step, not to the tile's row number (rows in
//    front of row 1 are virtual and behave like row 0, gact_device.hpp), so two tiles with different start delays
//    share it.
// 3. Op-only pointers.  Inside the traceback window the scores are times four and the two low bits say where a value
//    came from -- M 3, H_up 2, H_left 1: the numbering of align.h:23, and the tie order of align.cpp:162-164 is the
//    order of the tags:
//      H'' = max(M'' | 3, Z'', H''_up - 1, H''_left - (4|g| + 2)),     op = H'' & 3,     stored: G = H'' | 3
//    (as of round 3 the stored form is G = (H'' & ~3) | 2 -- it is H''_up - 1 as it stands -- with the diagonal's tag
//    coming from a +1 in the look-up word and the left neighbour's from 4|g| + 1: see dp_pass_lin_split)
//    -- 10 instructions per pair with the two that shift the op into its column's half-word.  No open / extend flags
//    are made: the walker does not need them (walk_chain_lin, gact_chain.hpp: with these scorings the next state of
//    the traceback is the op of the cell it enters).  H == 0 shows as op 3 like MATCH (M'' is clamped to the zero
//    level tagged 3): ZERO is left to the walker too, which carries the score of the cell it stands on.  It needs H
//    of the start cell (R, Q): every tile of a wave is delayed so that its last row falls on the wave's last step,
//    and the value is simply what the lane of column Q holds when the loop ends.
// 4. Instruction classes.  tools/issue_probe.hip (profiles/r02/issue_rate_probe.json): with three waves on a SIMD
//    v_add_u32 / v_sub_u32 / v_and_b32 / v_or_b32 on VGPR operands issue every 1.9 cycles, v_pk_*, v_max_*,
//    v_perm_b32, v_mad_*, DPP moves and anything with an SGPR operand every 3.2-3.4.  So the frame is shifted up
//    (lin_base) until every value is a positive int16: the packed additions then cannot carry or borrow across the
//    half-words and run as plain 32-bit v_add_u32 / v_sub_u32, the re-taggings are a subtraction each, and the
//    constants sit in VGPRs.
#pragma once

#include "gact_p16s.hpp"

namespace gact {

// 5. One instruction for two of the three maxima.  Every value of the pass is a positive int16 below 0x7C00, and positive
//    IEEE half-precision numbers order exactly like their bit patterns: max(M', Z, H'_up) -- off the serial column chain --
//    is ONE v_pk_maximum3_f16 (gfx950) where the integer instruction set needs two v_pk_max_i16.  The maximum returns one
//    of its operands bit for bit (no NaN below 0x7C00; lin_base keeps every value at or above 0x0400, a normal number,
//    whatever the denormal mode).  5 instructions per cell pair for the scores, 9 where pointers are made; the slow
//    instruction class (3.2 cycles at three waves) goes from 4 to 3 / from 5 to 4 of them.
#ifndef GACT_LIN_MAX3
#define GACT_LIN_MAX3 1
#endif
constexpr int kLinFloor = GACT_LIN_MAX3 ? 1024 : 0;
// zero level of lane 0 before step 1: above 31 lanes' worth of drift plus one gap, so nothing ever goes below |g|
// (below kLinFloor + |g|)
__host__ __device__ constexpr int lin_base(int g) { return kLinFloor + 40 * (-g) + 8; }
// a wave-uniform constant the compiler must keep in a VGPR (an SGPR operand would put the instruction in the slow class)
__device__ __forceinline__ uint32_t vconst(uint32_t s)
{
    uint32_t r;
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(r) : "s"(s));     // (wait states: s may come out of a spill lane)
    return r;
}

// every score, times four, plus the drift of up to kMaxSteps + lanes + lag rows must fit int16
__host__ inline bool p16_lin_ok(int tile, int match, int mismatch, int open, int ext)
{
    const long long steps = (long long)tile + 4 * kGroup + 64 + 48;        // drift of the longest pass + lin_base
    return open == ext && mismatch == ext && ext <= 0 && match >= 0 &&
           p16_tagged_ok(tile, match, mismatch, open, ext) &&
           4 * ((long long)match * (tile + 2) + (long long)(-ext) * steps + kLinFloor) + 3 <= 30000 && match - ext <= 63;
}

__device__ __forceinline__ uint32_t pk_mad4v(uint32_t a, uint32_t v_c)     // a * 4 + c (wrapping halves), c in a VGPR
{
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, 4, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(v_c));
    return r;
}
// max of three positive int16 pairs as half-precision numbers (see 5. above)
__device__ __forceinline__ uint32_t pk_max3f(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// (a & ~b) | c in one fast-class instruction (v_bitop3_b32, gfx950; truth table over a = 0xF0, b = 0xCC, c = 0xAA)
__device__ __forceinline__ uint32_t andn_or(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xba" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t pk_ashr2(uint32_t a)
{
    uint32_t r;
    asm("v_pk_ashrrev_i16 %0, 2, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a));
    return r;
}

// Pointer words of this pass (walk_chain's FMT 3): two bits per cell, the op code alone.  A half-word holds eight
// stored steps of one column (first step on top), a dword two adjacent columns (the even one low), a uint4 eight
// columns; a flush block is [uint4 n][lane], n < kUint4 (ws_quad_addr, gact_device.hpp).
template <int CW> struct LinWords {
    static constexpr int kWords = (CW + 1) / 2;          // dwords per lane, tile and flush
    static constexpr int kUint4 = (kWords + 3) / 4;
};
// acc[c]: the codes of column c, tile A in the low half-word, tile B in the high one.  storeA / storeB: does this lane
// write its words of tile A / tile B for this flush block (LinBand below: the block meets the band a walk can reach, or
// the tile stores everything)
template <int NW, int LANES, class Fix>
__device__ __forceinline__ void lin_flush(const uint32_t (&acc)[2 * NW], uint4 *qA, uint4 *qB, Fix fix, const bool storeA, const bool storeB)
{
    constexpr int QD = (NW + 3) / 4;
    uint32_t wa[QD * 4], wb[QD * 4];
#pragma unroll
    for (int n = 0; n < QD * 4; n++) {
        wa[n] = n < NW ? fix(__builtin_amdgcn_perm(acc[n < NW ? 2 * n + 1 : 0], acc[n < NW ? 2 * n : 0], 0x05040100u)) : 0u;
        wb[n] = n < NW ? fix(__builtin_amdgcn_perm(acc[n < NW ? 2 * n + 1 : 0], acc[n < NW ? 2 * n : 0], 0x07060302u)) : 0u;
    }
    if (storeA) {
#pragma unroll
        for (int q = 0; q < QD; q++) qA[q * kWsRow] = make_uint4(wa[4 * q], wa[4 * q + 1], wa[4 * q + 2], wa[4 * q + 3]);
    }
    if (storeB) {
#pragma unroll
        for (int q = 0; q < QD; q++) qB[q * kWsRow] = make_uint4(wb[4 * q], wb[4 * q + 1], wb[4 * q + 2], wb[4 * q + 3]);
    }
}

// 6. Banded pointer stores (round 4).  A non-first tile's traceback starts at (R, Q) and moves up and to the left
//    (align.cpp:205-229); with di = R - i and dj = Q - j every move changes di - dj by at most one, and at the error rates
//    this aligner is made for the path stays within a few tens of columns of the diagonal di == dj for all of its <= 2 x early
//    steps.  The pass therefore stores, per lane, only the flush blocks that hold a cell with |di - dj| <= band -- about a
//    third of the window's 14 KB per tile -- and the walker checks at every region refill that it is still at least a
//    refill's worth of steps inside the band (walk_chain_lin, gact_chain.hpp).  A walk that is not gives up; its tile is run
//    again with every block stored (ChainState::full), the one case in some hundreds: exact by construction, whatever the
//    reads look like.  Every tile of a wave ends on the wave's last step (kEndAligned), so a cell's step is
//    T_end - di - (lanes between its lane and the lane of column Q): lane by lane a range of steps, [lo, hi].
constexpr int kLinBandDefault = 48, kLinBandMin = 24, kLinBandQuantum = 1;
struct LinBand {
    int lo[2], hi[2];          // steps whose cells of this lane lie inside the band, tile A / tile B (lo > hi: none)
    bool full[2];              // the tile stores every block of the window
    __device__ __forceinline__ bool store(int h, int first, int last) const { return full[h] | ((last >= lo[h]) & (first <= hi[h])); }
};
// lane's columns are dj_min .. dj_max away from column Q (dj_max < 0: pads right of the tile), `behind` lanes before Q's
__device__ __forceinline__ void lin_band_range(LinBand &b, int h, int T_end, int behind, int dj_min, int dj_max, int band, bool full)
{
    b.full[h] = full | (band <= 0);
    b.lo[h] = T_end - behind - dj_max - band;
    b.hi[h] = dj_max < 0 ? -0x40000000 : T_end - behind - imax(dj_min, 0) + band;
}

// ---------------------------------------------------------------------------
// Split layout (see gact_p16s.hpp for the column map).  Returns, in lane 15 of every group, H[R][Q] of both tiles
// (packed, plain scores) -- valid when every tile's last row is the wave's last step (shift = T_end - Tend).
template <int C1, int C2>
__device__ __forceinline__ uint32_t dp_pass_lin_split(const P16Consts &kc, const int gl,
                                                      const uint16_t *__restrict__ ref16,
                                                      const uint32_t (&qb)[C1 + C2],
                                                      const int T_end, const int tB,
                                                      uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB,
                                                      const int band, const bool fullA, const bool fullB)
{
    constexpr int CT = C1 + C2;
    constexpr int NW = LinWords<C2>::kWords, QD = LinWords<C2>::kUint4;
    constexpr int LAG = kGroup;
    const int g = (int)(int16_t)(kc.ext & 0xffffu);
    const uint32_t gv = vconst(kc.next), g4v = vconst(kc.next4), c3v = vconst(kc.c3), onev = vconst(kc.one),
                   dtv = vconst(kc.next4 + kc.tag1),               // 4|g| + 1: G'' (tagged 2) -> D'' tagged 1
                   c2v = vconst(kc.tag2), lut1v = vconst(0x01010101u);
    // zero level of the row a lane did "before step 1": region 1 is at row t - gl, region 2 at row t - gl - LAG
    uint32_t Z1 = pk2(lin_base(g) + gl * g), Z2 = pk2(lin_base(g) + (gl + LAG) * g);
    uint32_t G[CT];                         // H of the previous row (drifted)
    uint32_t acc[2 * NW];                   // op codes of the last (up to) eight steps, one column each
#pragma unroll
    for (int c = 0; c < CT; c++) G[c] = c < C1 ? Z1 : Z2;                          // row 0: H = 0
#pragma unroll
    for (int c = 0; c < 2 * NW; c++) acc[c] = 0;
    // last slot of each region as the neighbour lane will see it; on the j = 0 border it is the zero level
    uint32_t H1 = Z1, H2 = Z2;
    uint32_t Hdiag1 = Z1, Hdiag2 = Z2;

    auto lut = [&](uint32_t amount) { return kc.dsub >> (amount & 31u); };
    auto lut4 = [&](uint32_t amount) { return kc.dsub4 >> (amount & 31u); };
    uint32_t rb1 = 0, rb1b = 0, rb2 = 0, rb2b = 0;
    {
        const uint32_t w1 = ref16[1], w2 = ref16[1 - LAG];
        rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8); rb2 = lut(w2 & 0xffu); rb2b = lut(w2 >> 8);
    }

    // Instruction order.  With three waves on a SIMD a v_add / v_sub / v_and / v_or issues in 1.9 cycles when it does
    // not wait for the instruction in front of it, in 3.0 when it does (issue_rate_probe.json, dependent streams).
    // So the step is written in stages -- one kind of instruction for all column slots, then the next kind -- and the
    // two regions' column chains (H_left -> D -> H, two dependent instructions per slot) run side by side; the
    // scheduling barriers keep the compiler from folding the stages back into per-slot sequences.
#define GACT_SB() __builtin_amdgcn_sched_barrier(0)
    auto upper_all = [&](uint32_t (&U)[CT], const bool tag2, const uint32_t Zr2) {
        uint32_t P[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) P[c] = __builtin_amdgcn_perm(c < C1 ? rb1b : rb2b, c < C1 ? rb1 : rb2, qb[c]);
        GACT_SB();
#pragma unroll
        for (int c = 0; c < CT; c++) U[c] = (c == 0 ? Hdiag1 : c == C1 ? Hdiag2 : G[c - 1]) + P[c];   // align.cpp:134-144
        // (pointer phase: region 2's G is kept tagged 2, so it IS H_up'' as it stands; the look-up words of that phase
        //  carry a +1, so M'' = G_diag'' + 4 (sub - g) + 1 comes out tagged 3 with no instruction of its own)
        (void)tag2;
        GACT_SB();
        if (GACT_LIN_MAX3) {
#pragma unroll
            for (int c = 0; c < CT; c++)                                         // :145-147 and the insertion, :149-154
                U[c] = pk_max3f(U[c], c < C1 ? Z1 : Zr2, G[c]);
        } else {
#pragma unroll
            for (int c = 0; c < CT; c++) U[c] = pk_max(U[c], c < C1 ? Z1 : Zr2);     // :145-147
            GACT_SB();
#pragma unroll
            for (int c = 0; c < CT; c++) U[c] = pk_max(U[c], G[c]);              // the insertion, :149-154
        }
        GACT_SB();
    };

    auto step = [&](const int t) {
        const uint32_t w1 = ref16[t + 1], w2 = ref16[t + 1 - LAG];
        Z1 += gv; Z2 += gv;
        // lane 0 of region 1 sits on the j = 0 border: the zero level
        const uint32_t Hl1 = (uint32_t)dpp_row_shr1((int)H1, (int)Z1);
        // lane 0 of region 2 continues lane 15's region 1 (one step ago = same row, same zero level)
        const uint32_t Hl2 = (uint32_t)dpp_row_shr1((int)H2, dpp_row_ror1((int)H1));
        uint32_t U[CT];
        upper_all(U, false, Z2);
        Hdiag1 = Hl1; Hdiag2 = Hl2;
        uint32_t Ha = Hl1, Hb = Hl2;
        static_assert(C2 >= C1, "region 2 is the longer chain");
#pragma unroll
        for (int c = 0; c < C1; c++) {                                           // :151-160 (no borrow)
            const uint32_t Da = Ha - gv, Db = Hb - gv;
            GACT_SB();
            G[c] = pk_max(U[c], Da); G[C1 + c] = pk_max(U[C1 + c], Db);
            GACT_SB();
            Ha = G[c]; Hb = G[C1 + c];
        }
#pragma unroll
        for (int c = 2 * C1; c < CT; c++) { G[c] = pk_max(U[c], Hb - gv); Hb = G[c]; }
        H1 = Ha; H2 = Hb;
        rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8); rb2 = lut(w2 & 0xffu); rb2b = lut(w2 >> 8);
    };
    // a step in which region 2 is in front of its row 1 in every lane (see 7. below): region 1 alone, region 2's zero level
    auto step_r1 = [&](const int t) {
        const uint32_t w1 = ref16[t + 1];
        Z1 += gv; Z2 += gv;
        const uint32_t Hl1 = (uint32_t)dpp_row_shr1((int)H1, (int)Z1);
        uint32_t P[C1], U[C1];
#pragma unroll
        for (int c = 0; c < C1; c++) P[c] = __builtin_amdgcn_perm(rb1b, rb1, qb[c]);
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C1; c++) U[c] = (c == 0 ? Hdiag1 : G[c - 1]) + P[c];
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C1; c++) U[c] = GACT_LIN_MAX3 ? pk_max3f(U[c], Z1, G[c]) : pk_max(pk_max(U[c], Z1), G[c]);
        GACT_SB();
        Hdiag1 = Hl1;
        uint32_t Ha = Hl1;
#pragma unroll
        for (int c = 0; c < C1; c++) { G[c] = pk_max(U[c], Ha - gv); Ha = G[c]; }
        H1 = Ha;
        rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8);
    };

    // ---- pointer phase: region 2 on tagged scores; G = 4H + 2 there: H_up'' without an instruction, the diagonal gets
    //      its tag 3 from the look-up word (+1), the left neighbour its tag 1 from the gap subtraction (4|g| + 1), and
    //      "low bits := 2" is one fast-class v_bitop3_b32
    uint32_t Z24 = 0;
    auto step_tagged = [&](const int t) {
        const uint32_t w1 = ref16[t + 1], w2 = ref16[t + 1 - LAG];
        Z1 += gv; Z24 += g4v;
        const uint32_t Hl1 = (uint32_t)dpp_row_shr1((int)H1, (int)Z1);
        // lane 15's region-1 column enters region 2 scaled and tagged 2
        const uint32_t Hl2 = (uint32_t)dpp_row_shr1((int)H2, dpp_row_ror1((int)pk_mad4v(H1, c2v)));
        uint32_t U[CT];
        upper_all(U, true, Z24);
        Hdiag1 = Hl1; Hdiag2 = Hl2;
        uint32_t Ha = Hl1, Hb = Hl2;
        uint32_t tprev = 0;
#pragma unroll
        for (int c = 0; c < C2; c++) {
            const bool both = c < C1;
            uint32_t Da = 0;
            const uint32_t Db = Hb - dtv;                                        // G'' tagged 2 -> D'' tagged 1
            if (both) Da = Ha - gv;
            GACT_SB();
            const uint32_t Hp = pk_max(U[C1 + c], Db);                           // the low bits: the op (:162-164)
            if (both) { G[c] = pk_max(U[c], Da); Ha = G[c]; }
            if (c > 0) acc[c - 1] = pk_shl_add4(acc[c - 1], tprev);
            GACT_SB();
            G[C1 + c] = andn_or(Hp, c3v, c2v);                                   // low bits := 2
            tprev = Hp & c3v;
            GACT_SB();
            Hb = G[C1 + c];
        }
        acc[C2 - 1] = pk_shl_add4(acc[C2 - 1], tprev);
        H1 = Ha; H2 = Hb;
        rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8); rb2 = lut4(w2 & 0xffu) + lut1v; rb2b = lut4(w2 >> 8) + lut1v;
    };
    // a step of the pointer phase in which region 1 is past its last row in every lane (see 7. below): region 2 alone; H1
    // stays what lane 15 left at step T_end - LAG
    auto step_tagged_r2 = [&](const int t) {
        const uint32_t w2 = ref16[t + 1 - LAG];
        Z24 += g4v;
        const uint32_t Hl2 = (uint32_t)dpp_row_shr1((int)H2, dpp_row_ror1((int)pk_mad4v(H1, c2v)));
        uint32_t P[C2], U[C2];
#pragma unroll
        for (int c = 0; c < C2; c++) P[c] = __builtin_amdgcn_perm(rb2b, rb2, qb[C1 + c]);
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C2; c++) U[c] = (c == 0 ? Hdiag2 : G[C1 + c - 1]) + P[c];
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C2; c++) U[c] = GACT_LIN_MAX3 ? pk_max3f(U[c], Z24, G[C1 + c]) : pk_max(pk_max(U[c], Z24), G[C1 + c]);
        GACT_SB();
        Hdiag2 = Hl2;
        uint32_t Hb = Hl2;
        uint32_t tprev = 0;
#pragma unroll
        for (int c = 0; c < C2; c++) {
            const uint32_t Db = Hb - dtv;
            GACT_SB();
            const uint32_t Hp = pk_max(U[c], Db);
            if (c > 0) acc[c - 1] = pk_shl_add4(acc[c - 1], tprev);
            GACT_SB();
            G[C1 + c] = andn_or(Hp, c3v, c2v);
            tprev = Hp & c3v;
            GACT_SB();
            Hb = G[C1 + c];
        }
        acc[C2 - 1] = pk_shl_add4(acc[C2 - 1], tprev);
        H2 = Hb;
        rb2 = lut4(w2 & 0xffu) + lut1v; rb2b = lut4(w2 >> 8) + lut1v;
    };
#undef GACT_SB
    auto enter_tagged = [&]() {
#pragma unroll
        for (int c = C1; c < CT; c++) G[c] = pk_mad4v(G[c], c2v);
        H2 = pk_mad4v(H2, c2v);
        Hdiag2 = pk_mad4v(Hdiag2, c2v);
        Z24 = pk_mad4v(Z2, c3v);                        // the zero level stays tagged 3 (H == 0 reads as MATCH, see 3.)
        rb2 = (rb2 << 2) + lut1v; rb2b = (rb2b << 2) + lut1v;      // the row already fetched: bonus times four, + 1
    };

    int t = 1;
    // 7. Region 2 runs LAG steps behind region 1 and every tile ends on the wave's last step: for the first LAG steps region 2
    //    is in front of its row 1 in EVERY lane (its values are the zero level, whatever the reads hold), for the last LAG
    //    steps region 1 is past its last row in every lane (nothing reads what it would compute: lane 0 of region 2 takes
    //    lane 15's H of step T_end - LAG at step T_end - LAG + 1 and rows past R after that).  Those steps run without the
    //    idle region's column slots: 16 x 65 + up to 16 x 37 of a pass's ~53 k instructions.
    for (const int tP = imin(LAG, imin(tB - 1, T_end)); t <= tP; t++) step_r1(t);
    if (t > 1) {
#pragma unroll
        for (int c = C1; c < CT; c++) G[c] = Z2;
        H2 = Z2; Hdiag2 = Z2;
        const uint32_t w2 = ref16[t - LAG];
        rb2 = lut(w2 & 0xffu); rb2b = lut(w2 >> 8);
    }
    for (; t < tB && t <= T_end; t++) step(t);
    const bool tagged = t <= T_end;
    if (tagged) enter_tagged();
    uint4 *qA = reinterpret_cast<uint4 *>(wsA) + gl;
    uint4 *qB = reinterpret_cast<uint4 *>(wsB) + gl;
    // region 2 is right-aligned: lane gl's columns are 13 (15 - gl) .. 13 (15 - gl) + 12 away from column Q in EVERY tile
    // (quantum: lanes store in aligned groups of 1, 4 or 8 -- whole 64- or 128-byte pieces of a workspace row -- so that the
    //  walker's loads never meet a partly written cache line; both ends of a lane's range grow with the lane)
    LinBand bd;
    {
        const int q1 = (band >> 16) - 1, b = band & 0xffff;
        const int u_lo = kGroup - 1 - (gl & ~q1), u_hi = kGroup - 1 - (gl | q1);
        LinBand lo_, hi_;
        lin_band_range(lo_, 0, T_end, u_lo, C2 * u_lo, C2 * u_lo + C2 - 1, b, fullA);
        lin_band_range(hi_, 0, T_end, u_hi, C2 * u_hi, C2 * u_hi + C2 - 1, b, fullA);
        bd.lo[0] = bd.lo[1] = lo_.lo[0]; bd.hi[0] = bd.hi[1] = hi_.hi[0];
        bd.full[0] = fullA | (b <= 0); bd.full[1] = fullB | (b <= 0);
    }
    // whole blocks of eight steps, each followed by its flush (an `if ((k & 7) == 7)` inside one loop is
    // if-converted by the compiler: the re-pairing v_perm of the flush would then run at every step)
    int k = 0;
    // (two loops one behind the other, not one loop with a branch inside: the two kinds of step keep their registers
    //  differently, and a loop that holds both moves ~90 registers per block to reconcile them)
    while (t + 7 <= T_end && t <= T_end - LAG) {
        for (int s8 = 0; s8 < 8; s8++, t++) step_tagged(t);
        k += 8;
        lin_flush<NW, kGroup>(acc, qA, qB, [](uint32_t w) { return w; }, bd.store(0, t - 8, t - 1), bd.store(1, t - 8, t - 1));
        qA += QD * kWsRow;
        qB += QD * kWsRow;
    }
    while (t + 7 <= T_end) {
        for (int s8 = 0; s8 < 8; s8++, t++) step_tagged_r2(t);
        k += 8;
        lin_flush<NW, kGroup>(acc, qA, qB, [](uint32_t w) { return w; }, bd.store(0, t - 8, t - 1), bd.store(1, t - 8, t - 1));
        qA += QD * kWsRow;
        qB += QD * kWsRow;
    }
    // (what is left are the last seven steps at most: region 2 alone, unless the pointer phase began inside them)
    for (; t <= T_end - LAG; t++, k++) step_tagged(t);
    for (; t <= T_end; t++, k++) step_tagged_r2(t);
    if (k & 7) {
        const int sh = 2 * (8 - (k & 7));
        lin_flush<NW, kGroup>(acc, qA, qB, [sh](uint32_t w) { return ((w & 0xffffu) << sh & 0xffffu) | ((w >> 16) << sh << 16); },
                              bd.store(0, t - (k & 7), t - 1), bd.store(1, t - (k & 7), t - 1));
    }
    // H of the last column at the row of the last step, drift taken off
    return tagged ? pk_ashr2(pk_sub(H2 | kc.c3, Z24)) : pk_sub(H2, Z2);
}

// ---------------------------------------------------------------------------
// Uniform layout (lane gl owns columns gl*C .. gl*C + C-1; 16 or 32 lanes per tile pair), every slot tagged in the
// pointer phase.  Two users: the wide main launch (LANES = 32, few long chains) and, with AMAX, the seed launch
// (first tiles: pointers from step 1 on, arg-max of align.cpp:173-177 as in dp_pass_p16 -- the key 8H + (step & 7)
// is 2 G'' - 2 Z'' + (step & 7) on the scaled, drifted scores).
// Returns, in the lane of column Q_h, H[R][Q] of tile h in half-word h (cq[h] = that column's slot); valid when
// every tile's last row is the wave's last step.  AMAX: the return value is not used (the walk starts at the
// arg-max with its score).
template <int C, int LANES, bool AMAX>
__device__ __forceinline__ uint32_t dp_pass_lin(const P16Consts &kc, const int gl,
                                                const uint16_t *__restrict__ ref16,
                                                const uint32_t (&qb)[C],
                                                const int T_end, const int tB,
                                                uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB,
                                                const int cqA, const int cqB, const int (*RQ)[2], P16Best *pb,
                                                const int col_from, const int band, const int QA, const int QB,
                                                const bool fullA, const bool fullB)
{
    constexpr int NW = LinWords<C>::kWords, QD = LinWords<C>::kUint4;
    // col_from: first column (1-based) a walk can reach in either tile: lanes left of it keep their words
    const bool store = AMAX || gl * C + C >= col_from;
    // band (non-first tiles, see LinBand): the lane's columns gl C + 1 .. gl C + C are Q - gl C - C .. Q - gl C - 1 away from
    // column Q, whose lane is (Q - 1) / C
    LinBand bd;
    {
        const int q1 = AMAX ? 0 : (band >> 16) - 1, b = AMAX ? 0 : band & 0xffff;     // (quantum: see dp_pass_lin_split)
        const int g_lo = gl & ~q1, g_hi = gl | q1;
        LinBand lo_, hi_;
        lin_band_range(lo_, 0, T_end, (imax(QA, 1) - 1) / C - g_lo, QA - g_lo * C - C, QA - g_lo * C - 1, b, fullA);
        lin_band_range(lo_, 1, T_end, (imax(QB, 1) - 1) / C - g_lo, QB - g_lo * C - C, QB - g_lo * C - 1, b, fullB);
        lin_band_range(hi_, 0, T_end, (imax(QA, 1) - 1) / C - g_hi, QA - g_hi * C - C, QA - g_hi * C - 1, b, fullA);
        lin_band_range(hi_, 1, T_end, (imax(QB, 1) - 1) / C - g_hi, QB - g_hi * C - C, QB - g_hi * C - 1, b, fullB);
        bd = lo_;
        // (the quantum's highest lane may lie right of the tile -- dj_max < 0, nothing of its own to store: its hi is then the
        //  last step)
        bd.hi[0] = hi_.hi[0] < 0 ? (lo_.hi[0] < 0 ? lo_.hi[0] : T_end) : hi_.hi[0];
        bd.hi[1] = hi_.hi[1] < 0 ? (lo_.hi[1] < 0 ? lo_.hi[1] : T_end) : hi_.hi[1];
    }
    static_assert(!AMAX || LANES == kGroup, "first tiles run on the 16-lane layout");
    const int g = (int)(int16_t)(kc.ext & 0xffffu);
    const uint32_t gv = vconst(kc.next), g4v = vconst(kc.next4), c3v = vconst(kc.c3), onev = vconst(kc.one),
                   dtv = vconst(kc.next4 + kc.tag1),       // 4|g| + 1: G'' (tagged 2) -> D'' tagged 1
                   c2v = vconst(kc.tag2), lut1v = vconst(0x01010101u);
    uint32_t Z = pk2(lin_base(g) + gl * g);      // zero level of the row this lane did "before step 1"
    uint32_t G[C];                               // H of the previous row (see dp_pass_lin_split)
    uint32_t acc[2 * NW];
#pragma unroll
    for (int c = 0; c < C; c++) G[c] = Z;
#pragma unroll
    for (int c = 0; c < 2 * NW; c++) acc[c] = 0;
    uint32_t G_last = Z, Hdiag = Z;

    // arg-max state (see dp_pass_p16)
    uint32_t bk[AMAX ? C : 1];
    int lane_best[2] = {-1, -1};
    uint32_t col_x0 = 0;
    int t_first = 0, rows[2] = {0, 0};
    if (AMAX) {
#pragma unroll
        for (int c = 0; c < C; c++) bk[c] = 0xffffffffu;
        const int ncA = imin(imax(RQ[0][1] - gl * C, 0), C), ncB = imin(imax(RQ[1][1] - gl * C, 0), C);
        col_x0 = ((uint32_t)(-ncA) & 0xffffu) | ((uint32_t)(-ncB) << 16);
        t_first = gl + 1;
        rows[0] = RQ[0][0]; rows[1] = RQ[1][0];
    }

    auto lut = [&](uint32_t amount) { return kc.dsub >> (amount & 31u); };
    auto lut4 = [&](uint32_t amount) { return kc.dsub4 >> (amount & 31u); };
    uint32_t lutA = 0, lutB = 0;
    { const uint32_t w = ref16[1]; lutA = lut(w & 0xffu); lutB = lut(w >> 8); }

    auto shr1 = [](uint32_t v, uint32_t old) {
        return (uint32_t)(LANES == 32 ? dpp_shr1_32((int)v, (int)old) : dpp_row_shr1((int)v, (int)old));
    };

    // (stages of one instruction kind over all slots, as in dp_pass_lin_split)
#define GACT_SB() __builtin_amdgcn_sched_barrier(0)
    auto upper_all = [&](uint32_t (&U)[C], const bool tag2, const uint32_t Zr) {
        uint32_t P[C];
#pragma unroll
        for (int c = 0; c < C; c++) P[c] = __builtin_amdgcn_perm(lutB, lutA, qb[c]);
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C; c++) U[c] = (c == 0 ? Hdiag : G[c - 1]) + P[c];                 // align.cpp:134-144
        (void)tag2;                              // (pointer phase: G is kept tagged 2 = H_up'' as it stands, see dp_pass_lin_split)
        GACT_SB();
        if (GACT_LIN_MAX3) {
#pragma unroll
            for (int c = 0; c < C; c++) U[c] = pk_max3f(U[c], Zr, G[c]);                       // :145-147 and the insertion, :149-154
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) U[c] = pk_max(U[c], Zr);                               // :145-147
            GACT_SB();
#pragma unroll
            for (int c = 0; c < C; c++) U[c] = pk_max(U[c], G[c]);                             // the insertion, :149-154
        }
        GACT_SB();
    };

    auto step = [&](const int t) {
        const uint32_t w_next = ref16[t + 1];
        Z += gv;
        const uint32_t Hl0 = shr1(G_last, Z);            // j = 0 border: the zero level
        uint32_t U[C];
        upper_all(U, false, Z);
        Hdiag = Hl0;
        uint32_t Hl = Hl0;
#pragma unroll
        for (int c = 0; c < C; c++) { G[c] = pk_max(U[c], Hl - gv); Hl = G[c]; }               // :151-160
        G_last = Hl;
        lutA = lut(w_next & 0xffu); lutB = lut(w_next >> 8);
    };

    uint32_t Z4 = 0, Z8 = 0;
    auto step_tagged = [&](const int t) {
        const uint32_t w_next = ref16[t + 1];
        Z4 += g4v;
        uint32_t key_c = 0;
        if (AMAX) {
            Z8 += g4v + g4v;
            const uint32_t sidx = (uint32_t)(t - tB) & 7u, row0 = (uint32_t)(t - t_first);
            const uint32_t ka = row0 < (uint32_t)rows[0] ? sidx : ((uint32_t)kKeyBias & 0xffffu);
            const uint32_t kb = row0 < (uint32_t)rows[1] ? sidx : ((uint32_t)kKeyBias & 0xffffu);
            key_c = pk_sub(ka | (kb << 16), Z8);
        }
        const uint32_t Hl0 = shr1(G_last, Z4 - onev);       // j = 0 border: the zero level, tagged 2 like every G''
        uint32_t U[C];
        upper_all(U, true, Z4);
        Hdiag = Hl0;
        uint32_t Hl = Hl0, tprev = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t Db = Hl - dtv;                                                       // G'' tagged 2 -> D'' tagged 1
            GACT_SB();
            const uint32_t Hp = pk_max(U[c], Db);                                               // the low bits: the op (:162-164)
            if (c > 0) {
                acc[c - 1] = pk_shl_add4(acc[c - 1], tprev);
                if (AMAX) bk[c - 1] = pk_max(bk[c - 1], pk_mad_vvv(G[c - 1], kc.tag2, key_c)); // 2 G'' + (step & 7) - Z8
            }
            GACT_SB();
            G[c] = andn_or(Hp, c3v, c2v);                                                       // low bits := 2
            tprev = Hp & c3v;
            GACT_SB();
            Hl = G[c];
        }
        acc[C - 1] = pk_shl_add4(acc[C - 1], tprev);
        if (AMAX) bk[C - 1] = pk_max(bk[C - 1], pk_mad_vvv(G[C - 1], kc.tag2, key_c));
        G_last = Hl;
        lutA = lut4(w_next & 0xffu) + lut1v; lutB = lut4(w_next >> 8) + lut1v;
    };
#undef GACT_SB
    auto enter_tagged = [&]() {
#pragma unroll
        for (int c = 0; c < C; c++) G[c] = pk_mad4v(G[c], c2v);
        G_last = pk_mad4v(G_last, c2v);
        Hdiag = pk_mad4v(Hdiag, c2v);
        Z4 = pk_mad4v(Z, c3v);                          // the zero level itself stays tagged 3
        Z8 = Z4 + Z4 - c2v;                             // 8 Z + 4 = twice a zero-score G'': the keys are 8 (H - Z) + (step & 7)
        lutA = (lutA << 2) + lut1v; lutB = (lutB << 2) + lut1v;
    };

    // fold the block keys of stored steps kblk..kblk+7 into lane_best (as dp_pass_p16)
    auto fold = [&](const int kblk) {
        uint32_t m = 0xffffffffu, rel = 0, x = col_x0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t v = pk_sign(x);
            x = pk_add(x, kc.one);
            const uint32_t key = pk_mad_m1(v, pk_add(bk[AMAX ? c : 0], kc.one));
            const uint32_t keep = pk_sign(pk_sub(key, m));
            rel = pk_mad_m1(keep, rel);
            m = pk_max(m, key);
            bk[AMAX ? c : 0] = 0xffffffffu;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int m16 = (int)(m << (16 - 16 * h)) >> 16;
            const int col = ((int)(rel << (16 - 16 * h)) >> 16) + C;
            const int rec = ((m16 >> 3) << 15) | ((kblk + (m16 & 7)) << 5) | col;
            lane_best[h] = imax(lane_best[h], m16 < 0 ? -1 : rec);
        }
    };

    int t = 1;
    for (; t < tB && t <= T_end; t++) step(t);
    const bool tagged = t <= T_end;
    if (tagged) enter_tagged();
    uint4 *qA = reinterpret_cast<uint4 *>(wsA) + gl;
    uint4 *qB = reinterpret_cast<uint4 *>(wsB) + gl;
    int k = 0;
    while (t + 7 <= T_end) {                     // whole blocks of eight steps + flush (see dp_pass_lin_split)
        for (int s8 = 0; s8 < 8; s8++, t++) step_tagged(t);
        k += 8;
        lin_flush<NW, LANES>(acc, qA, qB, [](uint32_t w) { return w; }, store && bd.store(0, t - 8, t - 1), store && bd.store(1, t - 8, t - 1));
        qA += QD * kWsRow;
        qB += QD * kWsRow;
        if (AMAX) fold(k - 8);
    }
    for (; t <= T_end; t++, k++) step_tagged(t);
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
        const int sh = 2 * (8 - (k & 7));
        lin_flush<NW, LANES>(acc, qA, qB, [sh](uint32_t w) { return ((w & 0xffffu) << sh & 0xffffu) | ((w >> 16) << sh << 16); },
                             store && bd.store(0, t - (k & 7), t - 1), store && bd.store(1, t - (k & 7), t - 1));
    }
    if (AMAX) return 0;
    // H[R][Q]: slot cq of the lane that owns column Q, at the row of the last step; drift taken off
    uint32_t ga = G[0], gb = G[0];
#pragma unroll
    for (int c = 1; c < C; c++) { ga = (c == cqA) ? G[c] : ga; gb = (c == cqB) ? G[c] : gb; }
    const uint32_t pick = __builtin_amdgcn_perm(gb, ga, 0x07060100u);           // {tile B's half of gb, tile A's half of ga}
    return tagged ? pk_ashr2(pk_sub(pick | kc.c3, Z4)) : pk_sub(pick, Z);
}

// The wide main launch of linear scorings: UniformLayout<10, 32>'s column map, the pass above, FMT 3 words
struct WideLayoutLin : UniformLayout<10, 32, true> {
    static constexpr int kWalkFmt = 3, kWalkQuads = LinWords<10>::kUint4;
#ifndef GACT_WALK_SPAN
#define GACT_WALK_SPAN 16
#endif
    static constexpr int kWalkSpan = GACT_WALK_SPAN;          // region cache of the look-ahead walker (gact_chain.hpp)
#ifndef GACT_WIDE_LIN_BLOCKS
#define GACT_WIDE_LIN_BLOCKS 3
#endif
    static constexpr int kBlocksPerCu = GACT_WIDE_LIN_BLOCKS;  // launch bounds: waves per SIMD the register budget is cut for
    static constexpr bool kEndAligned = true;
    template <bool RAW>
    __device__ static uint32_t pass(const P16Consts &kc, int gl, const uint16_t *ref16, const uint32_t (&qb)[10], int T_end,
                                    int tB, uint32_t *wsA, uint32_t *wsB, const PairTile &pt)
    {
        static_assert(!RAW, "the linear-gap pass reads 2-bit sets");
        return dp_pass_lin<10, 32, false>(kc, gl, ref16, qb, T_end, tB, wsA, wsB, (imax(pt.Q[0], 1) - 1) % 10,
                                          (imax(pt.Q[1], 1) - 1) % 10, nullptr, nullptr, pt.col_from, pt.band, pt.Q[0], pt.Q[1],
                                          pt.full[0], pt.full[1]);
    }
    __device__ static int fin_lane(int Q) { return (imax(Q, 1) - 1) / 10; }
};

// Layout policy for extend_p16_kernel: SplitLayout's column map, the linear-gap pass, FMT 3 pointer words
#ifndef GACT_LIN_BLOCKS_PER_CU
#define GACT_LIN_BLOCKS_PER_CU 3
#endif
template <int C1, int C2> struct SplitLayoutLin : SplitLayout<C1, C2, true> {
    static constexpr int kBlocksPerCu = GACT_LIN_BLOCKS_PER_CU;
    static constexpr int kWalkFmt = 3, kWalkQuads = LinWords<C2>::kUint4;
    static constexpr int kWalkSpan = GACT_WALK_SPAN;
    static constexpr bool kEndAligned = true;       // every tile's last row on the wave's last step
    template <bool RAW>
    __device__ static uint32_t pass(const P16Consts &kc, int gl, const uint16_t *ref16, const uint32_t (&qb)[C1 + C2],
                                    int T_end, int tB, uint32_t *wsA, uint32_t *wsB, const PairTile &pt)
    {
        static_assert(!RAW, "the linear-gap pass reads 2-bit sets");
        return dp_pass_lin_split<C1, C2>(kc, gl, ref16, qb, T_end, tB, wsA, wsB, pt.band, pt.full[0], pt.full[1]);
    }
    // lane and half-word of pass()'s return value that hold H[R][Q] of slot h
    __device__ static int fin_lane(int Q) { (void)Q; return kGroup - 1; }
};

// The same launch with the look-ahead walker (teams of eight lanes, gact_chain.hpp).  Alone on the machine the team makes
// the split launch slower (+5 % on ecoli10x: a wave iteration lasts longer, and that is what a run waits for); with several
// runs in flight what counts is instructions and the team walks in a third of the loop trips: +1.6 % with the team in every
// launch, +0.3 % (noise) when only the launches that share the machine take it (round 4, profiles/r04/ab_team_walker.txt).
// Off by default; GACT_HIP_TEAM_WHEN_SHARED=1 makes the engine take this variant for a launch that shares the machine.
template <int C1, int C2> struct SplitLayoutLinTeam : SplitLayoutLin<C1, C2> {
    static constexpr bool kTeamWalk = true;
};

}  // namespace gact
