// gact_aff.hpp -- the packed-int16 chain pass for AFFINE gap scoring (gap_open != gap_extend, or mismatch != gap_extend:
// everything gact_lin.hpp does not take), split layout, 2-bit read sets.  Same cells and results as dp_pass_p16s<.., TAG>
// (gact_p16s.hpp), which it replaces in the main launch; round 4 carries over to it what rounds 2-3 found for the
// linear-gap pass:
//
// 1. Row drift.  Every value of DP row i is kept as X + Z_i, Z_i = base + i * |ext| (align.cpp:134-160 in that frame):
//      M[i][j] = max(Z_i, H[i-1][j-1] + (sub - ext))           sub - ext = (sub - mismatch) + (mismatch - ext): the look-up
//                                                              byte of the 2-bit passes + one constant
//      I[i][j] = max(M[i-1][j] + (open - ext), I[i-1][j])      the extension needs no instruction at all
//      D[i][j] = max(M[i][j-1] + open, D[i][j-1] + ext)        same row, same frame: = max(Mo[i][j-1], D[i][j-1]) + ext
//      H       = max(M, I, D)
//    with Mo = M + (open - ext) kept per column (the next row's insertion opens from it, the next column's deletion too).
//    -INF (align.cpp:87-97) is a floor below every value a cell can take; it never moves up (an extension adds nothing), so
//    it never wins.
// 2. Positive frame: base is chosen so that every value is a positive int16 -- additions and subtractions of constants
//    cannot carry across the half-words and run as v_add_u32 / v_sub_u32 / v_add3_u32 (fast class) -- and
//    above 0x0400, so that H = max(M, I, D) is ONE v_pk_maximum3_f16 (positive half-precision numbers order like their
//    bit patterns, gact_lin.hpp 5.).
// 2a. One clamp, and not on M.  M[i][j] = max(0, ...) is what keeps H >= 0; the pass clamps the INSERTION at the zero level
//    instead -- I = max(Mo_up, I_up, Z), one v_pk_maximum3_f16 where the insertion's own max was -- and carries M unclamped:
//    H = max(M, I, D) >= Z all the same.  Nothing the reference computes changes: a value of M, I or D that is positive in
//    align.cpp is produced exactly (an unclamped M differs from the clamped one only where that is 0; what opens from it is
//    then negative in both, and negative gap values only ever get more negative), a value that is <= 0 there is <= 0 here;
//    H, the op of every cell with H > 0 and the two flags of every cell a traceback can stand on in INSERT / DELETE (where
//    I resp. D is positive) are decided among positive values.  (The reference's own CUDA kernel carries its M unclamped
//    too, cuda_header.h:178; SURVEY a-6.)
//    Per cell pair: perm, add3 (add + sub when mismatch < ext) | maximum3, sub | max, sub | maximum3 = 7 (8) instructions
//    where dp_pass_p16s needs 11.
// 3. Pointer phase (region 2 inside the traceback window): scores times four, the two low bits are tags, as in the TAG
//    passes, but chosen so that every pointer bit falls out of a max the recurrence needs anyway:
//      I'' = max(Mo''_up | 3, I''_up | 2)        bit 0 = ins_open >= ins_extend (a tie goes to the open, align.cpp:169)
//      D'' = max(Mo''_left | 3, D''_left | 1) - 4|ext|    bit 1 of the max = del_open >= del_extend (:170)
//      H'' = max(M'' | 3, I'' | 2, D'' | 1)      low bits = the op, align.h:23 numbering, ties as align.cpp:162-164
//    Three re-taggings (v_bitop3_b32 / v_or_b32, fast class); the two flags travel as (I'' + D'') & 3 -- the sums of the low
//    bits {3,2} + {3,1} are all different -- which the walker decodes; and ZERO (H == 0 shows as MATCH: M'' is clamped to
//    the zero level tagged 3) is the walker's, which carries the score of the cell it stands on exactly as the linear-gap
//    walker does (walk_chain FMT 4, gact_chain.hpp).  15 (16) instructions per pointer cell pair, 22 + 3 before.
// 4. Banded pointer stores (gact_lin.hpp 6.) apply unchanged: four uint4 per lane, tile and flush block here.
#pragma once

#include "gact_lin.hpp"

namespace gact {

// zero level of lane 0 before step 1 (see lin_base): above the floor by 31 lanes' + 16 rows' worth of drift, one opening
__host__ __device__ constexpr int aff_floor() { return 1024; }
// (... and what an unclamped M can lie below the zero level: one mismatch)
__host__ __device__ constexpr int aff_base(int e, int moe, int mm) { return aff_floor() + 48 * e + 2 * moe + mm + 8; }

// open <= ext <= 0, every score times four plus the drift of the longest pass fits below 0x7C00 / 4
__host__ inline bool p16_aff_ok(int tile, int match, int mismatch, int open, int ext)
{
    const long long steps = (long long)tile + 4 * kGroup + 64 + 48;
    const int e = -ext, moe = ext - open;
    return ext <= 0 && open <= ext && mismatch <= 0 && match >= 0 && p16_tagged_ok(tile, match, mismatch, open, ext) &&
           4 * ((long long)match * (tile + 2) + (long long)e * steps + aff_base(e, moe, -mismatch)) + 3 <= 30000 && match - mismatch <= 63 &&
           (mismatch - ext > -1000);
}

__device__ __forceinline__ uint32_t add3u(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Returns, in lane 15 of every group, H[R][Q] of both tiles (packed, plain scores) -- every tile's last row is the wave's
// last step (kEndAligned).
// CBNEG: mismatch < gap_extend -- the diagonal's constant (mismatch - ext) is negative and is taken off in an instruction of
// its own (a packed add of a negative constant would borrow across the half-words)
template <int C1, int C2, bool CBNEG>
__device__ __forceinline__ uint32_t dp_pass_aff_split(const P16Consts &kc, const int gl,
                                                      const uint16_t *__restrict__ ref16,
                                                      const uint32_t (&qb)[C1 + C2],
                                                      const int T_end, const int tB,
                                                      uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB,
                                                      const int band, const bool fullA, const bool fullB)
{
    constexpr int CT = C1 + C2;
    constexpr int QD = (C2 + 3) / 4;
    constexpr int LAG = kGroup;
    const int e = -kc.s_ext, moe = kc.s_ext - kc.s_open, cb = kc.s_mismatch - kc.s_ext;
    constexpr bool cb_neg = CBNEG;
    const uint32_t ev = vconst(pk2(e)), e4v = vconst(pk2(4 * e)), moev = vconst(pk2(moe)), moe4v = vconst(pk2(4 * moe)),
                   cbv = vconst(pk2(cb_neg ? -cb : cb)), cb4v = vconst(pk2(4 * (cb_neg ? -cb : cb))),
                   c3v = vconst(kc.c3), c2v = vconst(kc.tag2), c1v = vconst(kc.tag1);
    const uint32_t floorv = pk2(aff_floor());
    // zero level of the row a lane did "before step 1": region 1 is at row t - gl, region 2 at row t - gl - LAG
    const int base = aff_base(e, moe, -kc.s_mismatch);
    uint32_t Z1 = pk2(base - gl * e), Z2 = pk2(base - (gl + LAG) * e);
    uint32_t G[CT], Mo[CT], Iu[CT];         // H, M + (open - ext), I of the previous row (drifted; region 2 tagged in the pointer phase)
    uint32_t accO[QD * 4], accF[QD * 4];    // op codes / flag codes of the last (up to) eight steps, one column each
#pragma unroll
    for (int c = 0; c < CT; c++) {
        const uint32_t z = c < C1 ? Z1 : Z2;
        G[c] = z; Mo[c] = z - moev; Iu[c] = floorv;                                // row 0: H = M = 0, I = -INF
    }
#pragma unroll
    for (int c = 0; c < QD * 4; c++) { accO[c] = 0; accF[c] = 0; }
    // last slot of each region as the neighbour lane will see it; on the j = 0 border: H = M = 0, D = -INF
    uint32_t H1 = Z1, Mo1 = Z1 - moev, D1 = floorv;
    uint32_t H2 = Z2, Mo2 = Z2 - moev, D2 = floorv;
    uint32_t Hdiag1 = Z1, Hdiag2 = Z2;

    auto lut = [&](uint32_t amount) { return kc.dsub >> (amount & 31u); };
    auto lut4 = [&](uint32_t amount) { return kc.dsub4 >> (amount & 31u); };
    uint32_t rb1 = 0, rb1b = 0, rb2 = 0, rb2b = 0;
    {
        const uint32_t w1 = ref16[1], w2 = ref16[1 - LAG];
        rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8); rb2 = lut(w2 & 0xffu); rb2b = lut(w2 >> 8);
    }
#define GACT_SB() __builtin_amdgcn_sched_barrier(0)
    // everything of a step that depends on the previous row only, in stages of one instruction kind (gact_lin.hpp):
    // M (clamped) into Mc, the insertion into Iu (tagged region 2: its low bits still say open / extend, the column
    // chain reads the flag and re-tags)
    auto upper_all = [&](uint32_t (&Mc)[CT], const bool tagged, const uint32_t Zr2) {
        uint32_t P[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) P[c] = __builtin_amdgcn_perm(c < C1 ? rb1b : rb2b, c < C1 ? rb1 : rb2, qb[c]);
        GACT_SB();
        if (!cb_neg) {
#pragma unroll
            for (int c = 0; c < CT; c++)                                               // align.cpp:134-144
                Mc[c] = add3u(c == 0 ? Hdiag1 : c == C1 ? Hdiag2 : G[c - 1], P[c], (tagged && c >= C1) ? cb4v : cbv);
        } else {
#pragma unroll
            for (int c = 0; c < CT; c++) Mc[c] = (c == 0 ? Hdiag1 : c == C1 ? Hdiag2 : G[c - 1]) + P[c];
            GACT_SB();
#pragma unroll
            for (int c = 0; c < CT; c++) Mc[c] -= (tagged && c >= C1) ? cb4v : cbv;
        }
        GACT_SB();
#pragma unroll
        for (int c = 0; c < CT; c++)                                                   // :149-154 (the extension: nothing to add), and
            Iu[c] = pk_max3f(Mo[c], Iu[c], c < C1 ? Z1 : Zr2);                         // the clamp of :145-147, see 2a.
        GACT_SB();
    };

    auto step = [&](const int t) {
        const uint32_t w1 = ref16[t + 1], w2 = ref16[t + 1 - LAG];
        Z1 += ev; Z2 += ev;
        // lane 0 of region 1 sits on the j = 0 border of its row
        const uint32_t Hl1 = (uint32_t)dpp_row_shr1((int)H1, (int)Z1);
        const uint32_t Ml1 = (uint32_t)dpp_row_shr1((int)Mo1, (int)(Z1 - moev));
        const uint32_t Dl1 = (uint32_t)dpp_row_shr1((int)D1, (int)floorv);
        // lane 0 of region 2 continues lane 15's region 1 (one step ago = same row)
        const uint32_t Hl2 = (uint32_t)dpp_row_shr1((int)H2, dpp_row_ror1((int)H1));
        const uint32_t Ml2 = (uint32_t)dpp_row_shr1((int)Mo2, dpp_row_ror1((int)Mo1));
        const uint32_t Dl2 = (uint32_t)dpp_row_shr1((int)D2, dpp_row_ror1((int)D1));
        uint32_t Mc[CT];
        upper_all(Mc, false, Z2);
        Hdiag1 = Hl1; Hdiag2 = Hl2;
        uint32_t Ma = Ml1, Da = Dl1, Mb = Ml2, Db = Dl2;
        static_assert(C2 >= C1, "region 2 is the longer chain");
#pragma unroll
        for (int c = 0; c < C2; c++) {                                               // :151-160, the two regions side by side
            const bool both = c < C1;
            uint32_t Dma = 0;
            const uint32_t Dmb = pk_max(Mb, Db);                                     // open from the left M, or go on
            if (both) Dma = pk_max(Ma, Da);
            GACT_SB();
            Db = Dmb - ev;
            if (both) Da = Dma - ev;
            GACT_SB();
            G[C1 + c] = pk_max3f(Mc[C1 + c], Iu[C1 + c], Db);
            Mb = Mc[C1 + c] - moev;
            if (both) { G[c] = pk_max3f(Mc[c], Iu[c], Da); Ma = Mc[c] - moev; Mo[c] = Ma; }
            Mo[C1 + c] = Mb;
            GACT_SB();
        }
        H1 = G[C1 - 1]; Mo1 = Ma; D1 = Da;
        H2 = G[CT - 1]; Mo2 = Mb; D2 = Db;
        rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8); rb2 = lut(w2 & 0xffu); rb2b = lut(w2 >> 8);
    };
    // a step in which region 2 is in front of its row 1 in every lane (t <= LAG; gact_lin.hpp 7.): region 1 alone
    auto step_r1 = [&](const int t) {
        const uint32_t w1 = ref16[t + 1];
        Z1 += ev; Z2 += ev;
        const uint32_t Hl1 = (uint32_t)dpp_row_shr1((int)H1, (int)Z1);
        const uint32_t Ml1 = (uint32_t)dpp_row_shr1((int)Mo1, (int)(Z1 - moev));
        const uint32_t Dl1 = (uint32_t)dpp_row_shr1((int)D1, (int)floorv);
        uint32_t P[C1], Mc[C1];
#pragma unroll
        for (int c = 0; c < C1; c++) P[c] = __builtin_amdgcn_perm(rb1b, rb1, qb[c]);
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C1; c++) {
            const uint32_t hd = c == 0 ? Hdiag1 : G[c - 1];
            Mc[c] = cb_neg ? (hd + P[c]) - cbv : add3u(hd, P[c], cbv);
        }
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C1; c++) Iu[c] = pk_max3f(Mo[c], Iu[c], Z1);
        GACT_SB();
        Hdiag1 = Hl1;
        uint32_t Ma = Ml1, Da = Dl1;
#pragma unroll
        for (int c = 0; c < C1; c++) {
            const uint32_t Dma = pk_max(Ma, Da);
            Da = Dma - ev;
            G[c] = pk_max3f(Mc[c], Iu[c], Da);
            Ma = Mc[c] - moev;
            Mo[c] = Ma;
        }
        H1 = G[C1 - 1]; Mo1 = Ma; D1 = Da;
        rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8);
    };

    // ---- pointer phase: region 2 on scores times four; G'' and Mo'' tagged 3, I'' tagged 2, D'' tagged 1
    uint32_t Z24 = 0;
    auto step_tagged = [&](const int t) {
        const uint32_t w1 = ref16[t + 1], w2 = ref16[t + 1 - LAG];
        Z1 += ev; Z24 += e4v;
        const uint32_t Hl1 = (uint32_t)dpp_row_shr1((int)H1, (int)Z1);
        const uint32_t Ml1 = (uint32_t)dpp_row_shr1((int)Mo1, (int)(Z1 - moev));
        const uint32_t Dl1 = (uint32_t)dpp_row_shr1((int)D1, (int)floorv);
        // lane 15's region-1 column enters region 2 scaled and tagged
        const uint32_t Hl2 = (uint32_t)dpp_row_shr1((int)H2, dpp_row_ror1((int)pk_mad4v(H1, c3v)));
        const uint32_t Ml2 = (uint32_t)dpp_row_shr1((int)Mo2, dpp_row_ror1((int)pk_mad4v(Mo1, c3v)));
        const uint32_t Dl2 = (uint32_t)dpp_row_shr1((int)D2, dpp_row_ror1((int)pk_mad4v(D1, c1v)));
        uint32_t Mc[CT];
        upper_all(Mc, true, Z24);
        Hdiag1 = Hl1; Hdiag2 = Hl2;
        uint32_t Ma = Ml1, Da = Dl1, Mb = Ml2, Db = Dl2;
        uint32_t tprev = 0, fprev = 0;
#pragma unroll
        for (int c = 0; c < C2; c++) {
            const bool both = c < C1;
            uint32_t Dma = 0;
            const uint32_t Dp = pk_max(Mb, Db);                                      // bit 1: del_open >= del_extend (:170)
            if (both) Dma = pk_max(Ma, Da);
            if (c > 0) { accO[c - 1] = pk_shl_add4(accO[c - 1], tprev); accF[c - 1] = pk_shl_add4(accF[c - 1], fprev); }
            GACT_SB();
            const uint32_t Ds = Dp - e4v;
            if (both) Da = Dma - ev;
            fprev = Iu[C1 + c] + Dp;
            GACT_SB();
            Db = andn_or(Ds, c3v, c1v);                                              // low bits := 1
            GACT_SB();
            Iu[C1 + c] = andn_or(Iu[C1 + c], c3v, c2v);                              // low bits := 2
            GACT_SB();
            const uint32_t Hp = pk_max3f(Mc[C1 + c], Iu[C1 + c], Db);                // the low bits: the op (:162-164)
            Mb = Mc[C1 + c] - moe4v;
            if (both) { G[c] = pk_max3f(Mc[c], Iu[c], Da); Ma = Mc[c] - moev; Mo[c] = Ma; }
            Mo[C1 + c] = Mb;
            fprev &= c3v;
            GACT_SB();
            G[C1 + c] = Hp | c3v;
            tprev = Hp & c3v;
            GACT_SB();
        }
        accO[C2 - 1] = pk_shl_add4(accO[C2 - 1], tprev);
        accF[C2 - 1] = pk_shl_add4(accF[C2 - 1], fprev);
        H1 = G[C1 - 1]; Mo1 = Ma; D1 = Da;
        H2 = G[CT - 1]; Mo2 = Mb; D2 = Db;
        rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8); rb2 = lut4(w2 & 0xffu); rb2b = lut4(w2 >> 8);
    };
    // a step of the pointer phase in which region 1 is past its last row in every lane (t > T_end - LAG; gact_lin.hpp 7.):
    // region 2 alone; H1, Mo1, D1 stay what lane 15 left at step T_end - LAG
    auto step_tagged_r2 = [&](const int t) {
        const uint32_t w2 = ref16[t + 1 - LAG];
        Z24 += e4v;
        const uint32_t Hl2 = (uint32_t)dpp_row_shr1((int)H2, dpp_row_ror1((int)pk_mad4v(H1, c3v)));
        const uint32_t Ml2 = (uint32_t)dpp_row_shr1((int)Mo2, dpp_row_ror1((int)pk_mad4v(Mo1, c3v)));
        const uint32_t Dl2 = (uint32_t)dpp_row_shr1((int)D2, dpp_row_ror1((int)pk_mad4v(D1, c1v)));
        uint32_t P[C2], Mc[C2];
#pragma unroll
        for (int c = 0; c < C2; c++) P[c] = __builtin_amdgcn_perm(rb2b, rb2, qb[C1 + c]);
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C2; c++) {
            const uint32_t hd = c == 0 ? Hdiag2 : G[C1 + c - 1];
            Mc[c] = cb_neg ? (hd + P[c]) - cb4v : add3u(hd, P[c], cb4v);
        }
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C2; c++) Iu[C1 + c] = pk_max3f(Mo[C1 + c], Iu[C1 + c], Z24);
        GACT_SB();
        Hdiag2 = Hl2;
        uint32_t Mb = Ml2, Db = Dl2;
        uint32_t tprev = 0, fprev = 0;
#pragma unroll
        for (int c = 0; c < C2; c++) {
            const uint32_t Dp = pk_max(Mb, Db);
            if (c > 0) { accO[c - 1] = pk_shl_add4(accO[c - 1], tprev); accF[c - 1] = pk_shl_add4(accF[c - 1], fprev); }
            GACT_SB();
            const uint32_t Ds = Dp - e4v;
            fprev = Iu[C1 + c] + Dp;
            GACT_SB();
            Db = andn_or(Ds, c3v, c1v);
            GACT_SB();
            Iu[C1 + c] = andn_or(Iu[C1 + c], c3v, c2v);
            GACT_SB();
            const uint32_t Hp = pk_max3f(Mc[c], Iu[C1 + c], Db);
            Mb = Mc[c] - moe4v;
            Mo[C1 + c] = Mb;
            fprev &= c3v;
            GACT_SB();
            G[C1 + c] = Hp | c3v;
            tprev = Hp & c3v;
            GACT_SB();
        }
        accO[C2 - 1] = pk_shl_add4(accO[C2 - 1], tprev);
        accF[C2 - 1] = pk_shl_add4(accF[C2 - 1], fprev);
        H2 = G[CT - 1]; Mo2 = Mb; D2 = Db;
        rb2 = lut4(w2 & 0xffu); rb2b = lut4(w2 >> 8);
    };
#undef GACT_SB
    auto enter_tagged = [&]() {
#pragma unroll
        for (int c = C1; c < CT; c++) {
            G[c] = pk_mad4v(G[c], c3v);
            Mo[c] = pk_mad4v(Mo[c], c3v);
            Iu[c] = pk_mad4v(Iu[c], c2v);
        }
        H2 = pk_mad4v(H2, c3v); Mo2 = pk_mad4v(Mo2, c3v); D2 = pk_mad4v(D2, c1v);
        Hdiag2 = pk_mad4v(Hdiag2, c3v);
        Z24 = pk_mad4v(Z2, c2v);                          // (tagged 2: an insertion clamped to it says "extended"; nobody asks)
        rb2 = rb2 << 2; rb2b = rb2b << 2;                 // the row already fetched: bonus times four
    };

    auto wordA = [](uint32_t o, uint32_t f) { return __builtin_amdgcn_perm(f, o, 0x05040100u); };
    auto wordB = [](uint32_t o, uint32_t f) { return __builtin_amdgcn_perm(f, o, 0x07060302u); };
    uint4 *qA = reinterpret_cast<uint4 *>(wsA) + gl;
    uint4 *qB = reinterpret_cast<uint4 *>(wsB) + gl;
    // region 2 is right-aligned: lane gl's columns are C2 (15 - gl) .. + C2 - 1 away from column Q in every tile (LinBand)
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
    auto flush = [&](const int first, const int last, auto fix) {
        const bool sa = bd.store(0, first, last), sb = bd.store(1, first, last);
        if (sa) {
#pragma unroll
            for (int q = 0; q < QD; q++)
                qA[q * kWsRow] = make_uint4(fix(wordA(accO[4 * q], accF[4 * q])), fix(wordA(accO[4 * q + 1], accF[4 * q + 1])),
                                            fix(wordA(accO[4 * q + 2], accF[4 * q + 2])), fix(wordA(accO[4 * q + 3], accF[4 * q + 3])));
        }
        if (sb) {
#pragma unroll
            for (int q = 0; q < QD; q++)
                qB[q * kWsRow] = make_uint4(fix(wordB(accO[4 * q], accF[4 * q])), fix(wordB(accO[4 * q + 1], accF[4 * q + 1])),
                                            fix(wordB(accO[4 * q + 2], accF[4 * q + 2])), fix(wordB(accO[4 * q + 3], accF[4 * q + 3])));
        }
    };

    int t = 1;
    // (gact_lin.hpp 7.: the first LAG steps without region 2 -- in front of its row 1 in every lane: H = I = the zero level, M =
    //  what a mismatch on the row before makes it, D of no consequence --, the last LAG steps without region 1)
    for (const int tP = imin(LAG, imin(tB - 1, T_end)); t <= tP; t++) step_r1(t);
    if (t > 1) {
        const uint32_t mo = (cb_neg ? (Z2 - ev) - cbv : (Z2 - ev) + cbv) - moev;
#pragma unroll
        for (int c = C1; c < CT; c++) { G[c] = Z2; Iu[c] = Z2; Mo[c] = mo; }
        H2 = Z2; Hdiag2 = Z2; Mo2 = mo; D2 = floorv;
        const uint32_t w2 = ref16[t - LAG];
        rb2 = lut(w2 & 0xffu); rb2b = lut(w2 >> 8);
    }
    for (; t < tB && t <= T_end; t++) step(t);
    const bool tagged = t <= T_end;
    if (tagged) enter_tagged();
    int k = 0;
    while (t + 7 <= T_end && t <= T_end - LAG) { // whole blocks of eight steps + flush (see dp_pass_lin_split)
        for (int s8 = 0; s8 < 8; s8++, t++) step_tagged(t);
        k += 8;
        flush(t - 8, t - 1, [](uint32_t w) { return w; });
        qA += QD * kWsRow;
        qB += QD * kWsRow;
    }
    while (t + 7 <= T_end) {
        for (int s8 = 0; s8 < 8; s8++, t++) step_tagged_r2(t);
        k += 8;
        flush(t - 8, t - 1, [](uint32_t w) { return w; });
        qA += QD * kWsRow;
        qB += QD * kWsRow;
    }
    for (; t <= T_end - LAG; t++, k++) step_tagged(t);
    for (; t <= T_end; t++, k++) step_tagged_r2(t);
    if (k & 7) {
        const int sh = 2 * (8 - (k & 7));
        flush(t - (k & 7), t - 1, [sh](uint32_t w) { return ((w & 0xffffu) << sh & 0xffffu) | ((w >> 16) << sh << 16); });
    }
    // H of the last column at the row of the last step, drift taken off
    return tagged ? pk_ashr2(pk_sub(H2, Z24)) : pk_sub(H2, Z2);          // (H2 tagged 3, Z24 tagged 2: the shift drops the 1)
}

// ---------------------------------------------------------------------------
// The same recurrence for FIRST tiles (seed launch): uniform layout, 16 lanes x C columns, every slot on tagged scores from
// step 1 on (a first tile's traceback starts at the arg-max, anywhere: the whole matrix is stored), and the packed arg-max of
// dp_pass_p16 / dp_pass_lin: per column slot the largest key 8H + (step & 7) of the current 8-step block -- on the scaled,
// drifted scores that is 2 G'' - (8 Z + 6) + (step & 7) with G'' = 4 H' + 3 -- folded into (H, step, column) records at every
// flush (a later row, then a later column wins a tie, align.cpp:173-177; rows outside 1..R are keyed negative).
template <int C, bool CBNEG>
__device__ __forceinline__ void dp_pass_aff_seed(const P16Consts &kc, const int gl, const uint16_t *__restrict__ ref16,
                                                 const uint32_t (&qb)[C], const int T_end,
                                                 uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB,
                                                 const int (*RQ)[2], P16Best *pb)
{
    constexpr int QD = (C + 3) / 4;
    constexpr int tB = 1;
    const int e = -kc.s_ext, moe = kc.s_ext - kc.s_open, cb = kc.s_mismatch - kc.s_ext;
    const uint32_t e4v = vconst(pk2(4 * e)), moe4v = vconst(pk2(4 * moe)), cb4v = vconst(pk2(4 * (CBNEG ? -cb : cb))),
                   c3v = vconst(kc.c3), c2v = vconst(kc.tag2), c1v = vconst(kc.tag1), onev = vconst(kc.one);
    const int base = aff_base(e, moe, -kc.s_mismatch);
    const uint32_t floor1 = pk2(4 * aff_floor() + 1);
    uint32_t Z4 = pk2(4 * (base - gl * e) + 2);              // zero level of the row this lane did "before step 1", tagged 2
    uint32_t Z8 = Z4 + Z4 + c2v;                             // 8 Z + 6 = twice a zero-score G''
    uint32_t G[C], Mo[C], Iu[C], accO[QD * 4], accF[QD * 4];
#pragma unroll
    for (int c = 0; c < C; c++) { G[c] = Z4 + onev; Mo[c] = Z4 + onev - moe4v; Iu[c] = pk2(4 * aff_floor() + 2); }
#pragma unroll
    for (int c = 0; c < QD * 4; c++) { accO[c] = 0; accF[c] = 0; }
    uint32_t G_last = Z4 + onev, Mo_last = Z4 + onev - moe4v, D_last = floor1, Hdiag = Z4 + onev;

    // arg-max state (see dp_pass_p16)
    uint32_t bk[C];
    int lane_best[2] = {-1, -1};
#pragma unroll
    for (int c = 0; c < C; c++) bk[c] = 0xffffffffu;
    const int ncA = imin(imax(RQ[0][1] - gl * C, 0), C), ncB = imin(imax(RQ[1][1] - gl * C, 0), C);
    const uint32_t col_x0 = ((uint32_t)(-ncA) & 0xffffu) | ((uint32_t)(-ncB) << 16);
    const int t_first = gl + 1;
    const int rows[2] = {RQ[0][0], RQ[1][0]};

    auto lut4 = [&](uint32_t amount) { return kc.dsub4 >> (amount & 31u); };
    uint32_t lutA = 0, lutB = 0;
    { const uint32_t w = ref16[1]; lutA = lut4(w & 0xffu); lutB = lut4(w >> 8); }
#define GACT_SB() __builtin_amdgcn_sched_barrier(0)
    auto step_tagged = [&](const int t) {
        const uint32_t w_next = ref16[t + 1];
        Z4 += e4v;
        Z8 += e4v + e4v;
        const uint32_t sidx = (uint32_t)(t - tB) & 7u, row0 = (uint32_t)(t - t_first);
        const uint32_t ka = row0 < (uint32_t)rows[0] ? sidx : ((uint32_t)kKeyBias & 0xffffu);
        const uint32_t kb = row0 < (uint32_t)rows[1] ? sidx : ((uint32_t)kKeyBias & 0xffffu);
        const uint32_t key_c = pk_sub(ka | (kb << 16), Z8);
        // lane 0 sits on the j = 0 border of its row: H = M = 0, D = -INF
        const uint32_t Hl0 = (uint32_t)dpp_row_shr1((int)G_last, (int)(Z4 + onev));
        const uint32_t Ml0 = (uint32_t)dpp_row_shr1((int)Mo_last, (int)(Z4 + onev - moe4v));
        const uint32_t Dl0 = (uint32_t)dpp_row_shr1((int)D_last, (int)floor1);
        uint32_t Mc[C], P[C];
#pragma unroll
        for (int c = 0; c < C; c++) P[c] = __builtin_amdgcn_perm(lutB, lutA, qb[c]);
        GACT_SB();
        if (!CBNEG) {
#pragma unroll
            for (int c = 0; c < C; c++) Mc[c] = add3u(c == 0 ? Hdiag : G[c - 1], P[c], cb4v);             // align.cpp:134-144
        } else {
#pragma unroll
            for (int c = 0; c < C; c++) Mc[c] = (c == 0 ? Hdiag : G[c - 1]) + P[c];
            GACT_SB();
#pragma unroll
            for (int c = 0; c < C; c++) Mc[c] -= cb4v;
        }
        GACT_SB();
#pragma unroll
        for (int c = 0; c < C; c++) Iu[c] = pk_max3f(Mo[c], Iu[c], Z4);       // :149-154 and the clamp of :145-147 (2a. above); bit 0: opened
        GACT_SB();
        Hdiag = Hl0;
        uint32_t Mb = Ml0, Db = Dl0, tprev = 0, fprev = 0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t Dp = pk_max(Mb, Db);                                // bit 1: del_open >= del_extend (:170)
            if (c > 0) {
                accO[c - 1] = pk_shl_add4(accO[c - 1], tprev);
                accF[c - 1] = pk_shl_add4(accF[c - 1], fprev);
                bk[c - 1] = pk_max(bk[c - 1], pk_mad_vvv(G[c - 1], kc.tag2, key_c));   // 2 G'' + (step & 7) - Z8
            }
            GACT_SB();
            const uint32_t Ds = Dp - e4v;
            fprev = Iu[c] + Dp;
            GACT_SB();
            Db = andn_or(Ds, c3v, c1v);                                        // low bits := 1
            Iu[c] = andn_or(Iu[c], c3v, c2v);                                  // low bits := 2
            GACT_SB();
            const uint32_t Hp = pk_max3f(Mc[c], Iu[c], Db);                    // the low bits: the op (:162-164)
            Mb = Mc[c] - moe4v;
            Mo[c] = Mb;
            fprev &= c3v;
            GACT_SB();
            G[c] = Hp | c3v;
            tprev = Hp & c3v;
            GACT_SB();
        }
        accO[C - 1] = pk_shl_add4(accO[C - 1], tprev);
        accF[C - 1] = pk_shl_add4(accF[C - 1], fprev);
        bk[C - 1] = pk_max(bk[C - 1], pk_mad_vvv(G[C - 1], kc.tag2, key_c));
        G_last = G[C - 1]; Mo_last = Mb; D_last = Db;
        lutA = lut4(w_next & 0xffu); lutB = lut4(w_next >> 8);
    };
#undef GACT_SB
    // fold the block keys of stored steps kblk..kblk+7 into lane_best (as dp_pass_p16 / dp_pass_lin)
    auto fold = [&](const int kblk) {
        uint32_t m = 0xffffffffu, rel = 0, x = col_x0;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const uint32_t v = pk_sign(x);
            x = pk_add(x, kc.one);
            const uint32_t key = pk_mad_m1(v, pk_add(bk[c], kc.one));
            const uint32_t keep = pk_sign(pk_sub(key, m));
            rel = pk_mad_m1(keep, rel);
            m = pk_max(m, key);
            bk[c] = 0xffffffffu;
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int m16 = (int)(m << (16 - 16 * h)) >> 16;
            const int col = ((int)(rel << (16 - 16 * h)) >> 16) + C;
            const int rec = ((m16 >> 3) << 15) | ((kblk + (m16 & 7)) << 5) | col;
            lane_best[h] = imax(lane_best[h], m16 < 0 ? -1 : rec);
        }
    };
    auto wordA = [](uint32_t o, uint32_t f) { return __builtin_amdgcn_perm(f, o, 0x05040100u); };
    auto wordB = [](uint32_t o, uint32_t f) { return __builtin_amdgcn_perm(f, o, 0x07060302u); };
    uint4 *qA = reinterpret_cast<uint4 *>(wsA) + gl;
    uint4 *qB = reinterpret_cast<uint4 *>(wsB) + gl;
    auto flush = [&](auto fix) {
#pragma unroll
        for (int q = 0; q < QD; q++) {
            qA[q * kWsRow] = make_uint4(fix(wordA(accO[4 * q], accF[4 * q])), fix(wordA(accO[4 * q + 1], accF[4 * q + 1])),
                                        fix(wordA(accO[4 * q + 2], accF[4 * q + 2])), fix(wordA(accO[4 * q + 3], accF[4 * q + 3])));
            qB[q * kWsRow] = make_uint4(fix(wordB(accO[4 * q], accF[4 * q])), fix(wordB(accO[4 * q + 1], accF[4 * q + 1])),
                                        fix(wordB(accO[4 * q + 2], accF[4 * q + 2])), fix(wordB(accO[4 * q + 3], accF[4 * q + 3])));
        }
    };
    int t = 1, k = 0;
    while (t + 7 <= T_end) {
        for (int s8 = 0; s8 < 8; s8++, t++) step_tagged(t);
        k += 8;
        flush([](uint32_t w) { return w; });
        qA += QD * kWsRow;
        qB += QD * kWsRow;
        fold(k - 8);
    }
    for (; t <= T_end; t++, k++) step_tagged(t);
    if (k & 7) {
        fold(k & ~7);
        const int sh = 2 * (8 - (k & 7));
        flush([sh](uint32_t w) { return ((w & 0xffffu) << sh & 0xffffu) | ((w >> 16) << sh << 16); });
    }
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

// Layout policy for extend_p16_kernel: SplitLayout's column map, the pass above, FMT 4 pointer words
#ifndef GACT_AFF_BLOCKS_PER_CU
#define GACT_AFF_BLOCKS_PER_CU 3
#endif
template <int C1, int C2, bool CBNEG> struct SplitLayoutAff : SplitLayout<C1, C2, true> {
    static constexpr int kWalkFmt = 4;
    // three waves per SIMD (168 registers): the pass holds three values per column slot plus two pointer accumulators and
    // does not fit -- ~70 registers are saved and restored around the phases of a wave iteration, none inside the step
    // loops -- and is still faster than two waves with all of it in registers (ecoli10x at +2/-3/-5/-2: main launch
    // 45.9 ms against 49.8)
    static constexpr int kBlocksPerCu = GACT_AFF_BLOCKS_PER_CU;
    static constexpr bool kEndAligned = true;       // every tile's last row on the wave's last step: H[R][Q] for the walker
    template <bool RAW>
    __device__ static uint32_t pass(const P16Consts &kc, int gl, const uint16_t *ref16, const uint32_t (&qb)[C1 + C2],
                                    int T_end, int tB, uint32_t *wsA, uint32_t *wsB, const PairTile &pt)
    {
        static_assert(!RAW, "the drifted affine pass reads 2-bit sets");
        return dp_pass_aff_split<C1, C2, CBNEG>(kc, gl, ref16, qb, T_end, tB, wsA, wsB, pt.band, pt.full[0], pt.full[1]);
    }
    __device__ static int fin_lane(int Q) { (void)Q; return kGroup - 1; }
};

}  // namespace gact
