// gact_p16s.hpp -- "split" layout of the packed-int16 chain pass.
//
// In the uniform layout (gact_p16.hpp) every lane owns C consecutive columns,
// and once the wave is inside the traceback window every column slot pays for
// pointer generation -- also the slots of lanes whose columns lie left of the
// window and can never be read (columns <= Q - early, align.cpp:205).  Here a
// lane's C1 + C2 slots come from two column regions instead:
//
//   region 2 = the LAST 16*C2 columns of the tile (>= early, so it contains the
//              whole window): lane l holds its columns l*C2+1 .. (l+1)*C2,
//   region 1 = the columns left of it, right-aligned into 16*C1 slots
//              (missing columns on the left are pads, which behave like the
//              j = 0 border),
//
// run as two fused wavefronts: region 1 is at row t - l, region 2 at row
// t - l - 16.  Lane 15 finishes a row of region 1 exactly one step before
// lane 0 starts that row of region 2, so region 1's last column reaches
// region 2's first one with the same one-step DPP hop as any other lane
// boundary (row_ror:1 instead of row_shr:1).  Pointer work is emitted for the
// C2 region-2 slots only: per step of the window phase 7 x 11 + 13 x 25 (22 on
// tagged scores) instead of 20 x 25 slot instructions, for 16 more steps per pass.
//
// Same cells, same arithmetic as dp_pass_p16 -- only the schedule differs.
#pragma once

#include "gact_p16.hpp"

namespace gact {

template <int C1, int C2> struct GeometrySplit {
    static constexpr int CT = C1 + C2;
    static constexpr int kTileMax = CT * kGroup;
    static constexpr int W1 = C1 * kGroup, W2 = C2 * kGroup;
    static constexpr int kLag = kGroup;
    static constexpr int kMaxSteps = kTileMax + kGroup + kLag;
    static constexpr int kQuads = (C2 + 3) / 4;
    static constexpr int kRow0 = kGroup + kLag;                   // stream entry of (delay 0, row 1)
    static constexpr int kRefEntries = kRow0 + kMaxSteps + kGroup + 8;
    static constexpr int kRefBytes = kRefEntries * 2;
    static constexpr int kQueryBytes = kTileMax * kSlots;
    static constexpr int kGroupLds = (kRefBytes + kQueryBytes + 15) & ~15;
    static constexpr int kMaxFlush = (kMaxSteps + 7) / 8 + 1;
    static constexpr int kWsWords = kMaxFlush * kQuads * 4 * kGroup;
};

__device__ __forceinline__ int dpp_row_ror1(int v)
{
    // lane n of each 16-lane row reads lane n-1, lane 0 reads lane 15
    return __builtin_amdgcn_mov_dpp(v, 0x121, 0xF, 0xF, false);      // every lane is written: no `old` operand
}

template <int C2> __device__ __forceinline__ int split_last_step(int R, int Q)
{
    return (R > 0 && Q > 0) ? R + (kGroup - 1) + kGroup : 0;      // lane 15, region 2 (16 steps behind), row R
}

// first step whose pointers the traceback can reach, in the split layout
template <int C2> __device__ __forceinline__ int split_first_pointer_step(int R, int Q, int early)
{
    const int r_first = imax(1, R - early + 1);
    const int j_first = imax(1, Q - early + 1);
    const int jj = j_first + (C2 * kGroup - Q);                   // 1-based index inside region 2
    return r_first + (jj - 1) / C2 + kGroup;
}

// ---------------------------------------------------------------------------
// ref16[t] / ref16[t - 16] hold the bases of region 1's / region 2's row at step t.
//
// TAG: in the pointer phase region 2 runs on scores times four whose two low bits say where a value came from, so
// that the `max` operations the recurrence needs anyway also produce the pointer bits:
//   ins: max(M+open tagged 1, ins_extend tagged 0) -> bit 0 of the result = (ins_open >= ins_extend), ties included
//   del: likewise
//   H:   max(M tagged 3, I tagged 2, D tagged 1) -> the low bits are the op in align.h:23 numbering (M3 I2 D1), with
//        the reference's priority on ties; M is kept as max(.., 3), so H == 0 shows as H' <= 3 -> ZERO
// Re-tagging costs three ops per cell pair, the explicit comparisons it replaces cost six more: 22 instead of 25.
// Region 1 and the score-only phase stay on plain scores; region 1's last column is converted as it crosses into
// region 2 (three ops per step), region 2's state once when the pointer phase begins.  Pointer words: FMT 2.
template <int C1, int C2, bool RAW = true, bool TAG = false>
__device__ __forceinline__ void dp_pass_p16s(const P16Consts &kc, const int gl,
                                             const uint16_t *__restrict__ ref16,
                                             const uint32_t (&qb)[C1 + C2],
                                             const int T_end, const int tB,
                                             uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB)
{
    constexpr int CT = C1 + C2;
    constexpr int QD = (C2 + 3) / 4;
    constexpr int LAG = kGroup;
    const uint32_t hbias = RAW ? kc.match : kc.mism;        // substitution score forms: see dp_pass_p16
    uint32_t Hm[CT], Mo[CT], Iup[CT];       // H+bias, M+open, I of the previous row (both tiles)
    uint32_t accO[QD * 4], accF[QD * 4];    // pointer bits of the region-2 slots (see dp_pass_p16)
#pragma unroll
    for (int c = 0; c < CT; c++) {
        Hm[c] = hbias; Mo[c] = kc.open; Iup[c] = kc.ninf;
    }
#pragma unroll
    for (int c = 0; c < QD * 4; c++) { accO[c] = 0; accF[c] = 0; }
    // last slot of each region as the neighbour lane will see it
    uint32_t Mo1 = kc.open, D1 = kc.ninf, H1 = hbias;
    uint32_t Mo2 = kc.open, D2 = kc.ninf, H2 = hbias;
    uint32_t Hdiag1 = hbias, Hdiag2 = hbias;
    uint32_t Ml1 = kc.open, Dl1 = kc.ninf, Hl1 = hbias;         // what lane gl-1 shows; lane 0 keeps the border

    auto unpack = [](uint32_t w) { return __builtin_amdgcn_perm(0u, w, 0x0c010c00u); };      // {byte 1, byte 0} -> two half-words
    auto lut = [&](uint32_t amount) { return kc.dsub >> (amount & 31u); };
    // per region: RAW the two ref bases as half-words (a), else the two tiles' look-up words (a, b)
    uint32_t rb1 = 0, rb1b = 0, rb2 = 0, rb2b = 0;
    auto set_rows = [&](uint32_t w1, uint32_t w2) {
        if (RAW) { rb1 = unpack(w1); rb2 = unpack(w2); }
        else { rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8); rb2 = lut(w2 & 0xffu); rb2b = lut(w2 >> 8); }
    };
    set_rows(ref16[1], ref16[1 - LAG]);

    auto step = [&](const int t, auto ptr_tag) {
        constexpr bool PTR = decltype(ptr_tag)::value;
        const uint32_t w1 = ref16[t + 1], w2 = ref16[t + 1 - LAG];

        // region 1: lane 0 sits on the j = 0 border (or on left pads, which behave like it).  The border value is
        // whatever lane 0 of the destination held before: the previous step's result, i.e. the border again --
        // no register has to be re-loaded with it
        Ml1 = (uint32_t)dpp_row_shr1((int)Mo1, (int)Ml1);
        Dl1 = (uint32_t)dpp_row_shr1((int)D1, (int)Dl1);
        Hl1 = (uint32_t)dpp_row_shr1((int)H1, (int)Hl1);
        // region 2: lane 0 continues lane 15's region 1 (computed one step ago = same row): rotate region 1's
        // values into place, then shift region 2's over every lane but lane 0
        const uint32_t Ml2 = (uint32_t)dpp_row_shr1((int)Mo2, dpp_row_ror1((int)Mo1));
        const uint32_t Dl2 = (uint32_t)dpp_row_shr1((int)D2, dpp_row_ror1((int)D1));
        const uint32_t Hl2 = (uint32_t)dpp_row_shr1((int)H2, dpp_row_ror1((int)H1));
        uint32_t Hd = Hdiag1;
        Hdiag1 = Hl1;

        uint32_t M[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            if (c == C1) { Hd = Hdiag2; Hdiag2 = Hl2; }
            uint32_t Mx;                                                        // align.cpp:134
            if (RAW) Mx = pk_mad_s(pk_min1(qb[c] ^ (c < C1 ? rb1 : rb2)), kc.nd, Hd);
            else Mx = pk_add(Hd, __builtin_amdgcn_perm(c < C1 ? rb1b : rb2b, c < C1 ? rb1 : rb2, qb[c]));
            Hd = Hm[c];
            M[c] = pk_max0(Mx);                                                 // :145-147
            const uint32_t Ie = pk_add_s(Iup[c], kc.ext);                       // :150
            if (PTR && c >= C1) accF[c - C1] = pk_shl_add2(accF[c - C1], pk_sign(pk_sub(Mo[c], Ie)));   // :170
            Iup[c] = pk_max(Mo[c], Ie);                                         // :154
            Mo[c] = pk_add_s(M[c], kc.open);
        }
        uint32_t Ml = Ml1, Dl = Dl1;
#pragma unroll
        for (int c = 0; c < CT; c++) {
            if (c == C1) {
                Mo1 = Ml; D1 = Dl; H1 = Hm[C1 - 1];
                Ml = Ml2; Dl = Dl2;
            }
            const uint32_t De = pk_add_s(Dl, kc.ext);                           // :152
            const uint32_t D = pk_max(Ml, De);                                  // :156
            const uint32_t H = pk_max(pk_max(M[c], Iup[c]), D);                 // :158-160
            if (PTR && c >= C1) {
                accF[c - C1] = pk_shl_add2(accF[c - C1], pk_sign(pk_sub(Ml, De)));                      // :171
                const uint32_t nz = pk_min1(H);                                 // :162-168, see dp_pass_p16
                const uint32_t na = pk_min1(pk_sub(H, M[c]));
                const uint32_t nb = pk_min1(pk_sub(H, Iup[c]));
                const uint32_t o = pk_mad_vvv(na, nb, na);
                accO[c - C1] = pk_shl_add4(accO[c - C1], pk_mad_vvv(nz, o, nz));
            }
            Hm[c] = pk_add_s(H, hbias);
            Ml = Mo[c];
            Dl = D;
        }
        Mo2 = Ml; D2 = Dl; H2 = Hm[CT - 1];
        set_rows(w1, w2);
    };

    // ---- TAG: the pointer-phase step.  Region-2 registers hold: Hm = 4(H+bias)+3, Mo = 4(M+open)+1, Iup = the
    //      next row's ins_extend 4(I+ext) (tag 0); the lane-boundary values Mo2 / D2 / H2 tagged 1 / 1 / 3.
    const uint32_t hbias4 = RAW ? kc.match4 : kc.mism4;
    const uint32_t vmask = kc.nmask;                    // in a VGPR: v_and_or_b32 takes one scalar operand
    auto lut4 = [&](uint32_t amount) { return kc.dsub4 >> (amount & 31u); };
    auto step_tagged = [&](const int t) {
        const uint32_t w1 = ref16[t + 1], w2 = ref16[t + 1 - LAG];
        Ml1 = (uint32_t)dpp_row_shr1((int)Mo1, (int)Ml1);
        Dl1 = (uint32_t)dpp_row_shr1((int)D1, (int)Dl1);
        Hl1 = (uint32_t)dpp_row_shr1((int)H1, (int)Hl1);
        // lane 15's region-1 column enters region 2: scaled and tagged like a region-2 column
        const uint32_t Ml2 = (uint32_t)dpp_row_shr1((int)Mo2, dpp_row_ror1((int)pk_mad4(Mo1, kc.tag1)));
        const uint32_t Dl2 = (uint32_t)dpp_row_shr1((int)D2, dpp_row_ror1((int)pk_mad4(D1, kc.tag1)));
        const uint32_t Hl2 = (uint32_t)dpp_row_shr1((int)H2, dpp_row_ror1((int)pk_mad4(H1, kc.c3)));
        uint32_t Hd = Hdiag1;
        Hdiag1 = Hl1;

        uint32_t M[CT];
#pragma unroll
        for (int c = 0; c < CT; c++) {
            if (c == C1) { Hd = Hdiag2; Hdiag2 = Hl2; }
            if (c < C1) {                                                       // plain, as in step()
                uint32_t Mx;
                if (RAW) Mx = pk_mad_s(pk_min1(qb[c] ^ rb1), kc.nd, Hd);
                else Mx = pk_add(Hd, __builtin_amdgcn_perm(rb1b, rb1, qb[c]));
                Hd = Hm[c];
                M[c] = pk_max0(Mx);
                const uint32_t Ie = pk_add_s(Iup[c], kc.ext);
                Iup[c] = pk_max(Mo[c], Ie);
                Mo[c] = pk_add_s(M[c], kc.open);
            } else {
                uint32_t Mx;                                                    // 4(H[i-1][j-1] + sub) + 3
                if (RAW) Mx = pk_mad_s(pk_min1(qb[c] ^ rb2), kc.nd4, Hd);
                else Mx = pk_add(Hd, __builtin_amdgcn_perm(rb2b, rb2, qb[c]));
                Hd = Hm[c];
                M[c] = pk_max_s(Mx, kc.c3);                                     // 4M + 3, M >= 0       :145-147
                Iup[c] = pk_max(Mo[c], Iup[c]);                                 // bit 0: ins_open >= ins_extend   :154,170
                Mo[c] = pk_add_s(M[c], kc.open4m2);                             // 4(M + open) + 1
            }
        }
        uint32_t Ml = Ml1, Dl = Dl1;
#pragma unroll
        for (int c = 0; c < CT; c++) {
            if (c == C1) {
                Mo1 = Ml; D1 = Dl; H1 = Hm[C1 - 1];
                Ml = Ml2; Dl = Dl2;
            }
            if (c < C1) {
                const uint32_t De = pk_add_s(Dl, kc.ext);
                const uint32_t D = pk_max(Ml, De);
                const uint32_t H = pk_max(pk_max(M[c], Iup[c]), D);
                Hm[c] = pk_add_s(H, hbias);
                Ml = Mo[c];
                Dl = D;
            } else {
                const uint32_t De = pk_add_s(Dl, kc.ext4m1);                    // 4 del_extend, tag 0   :152
                const uint32_t Dp = pk_max(Ml, De);                             // bit 0: del_open >= del_extend   :156,171
                const uint32_t Dt = and_or(Dp, vmask, kc.tag1);
                const uint32_t It = and_or(Iup[c], vmask, kc.tag2);
                const uint32_t Hp = pk_max(pk_max(M[c], It), Dt);               // :158-168
                accF[c - C1] = pk_shl_add4(accF[c - C1], pk_shl_add2(Iup[c], Dp) & kc.c3);
                const uint32_t op = pk_mul(Hp & kc.c3, pk_min1(pk_lshr2(Hp)));  // H == 0: ZERO
                accO[c - C1] = pk_shl_add4(accO[c - C1], op);
                Hm[c] = pk_add_s(Hp | kc.c3, hbias4);
                Iup[c] = pk_add_s(It, kc.ext4m2);                               // the next row's ins_extend   :150
                Ml = Mo[c];
                Dl = Dt;
            }
        }
        Mo2 = Ml; D2 = Dl; H2 = Hm[CT - 1];
        if (RAW) { rb1 = unpack(w1); rb2 = unpack(w2); }
        else { rb1 = lut(w1 & 0xffu); rb1b = lut(w1 >> 8); rb2 = lut4(w2 & 0xffu); rb2b = lut4(w2 >> 8); }
    };
    // plain -> tagged, once, when the pointer phase begins
    auto enter_tagged = [&]() {
#pragma unroll
        for (int c = C1; c < CT; c++) {
            Hm[c] = pk_mad4(Hm[c], kc.c3);
            Mo[c] = pk_mad4(Mo[c], kc.tag1);
            Iup[c] = pk_add_s(pk_mad4(pk_max_s(Iup[c], kc.floor4), kc.tag2), kc.ext4m2);
        }
        Mo2 = pk_mad4(Mo2, kc.tag1);
        D2 = pk_mad4(pk_max_s(D2, kc.floor4), kc.tag1);
        H2 = pk_mad4(H2, kc.c3);
        Hdiag2 = pk_mad4(Hdiag2, kc.c3);
        if (!RAW) { rb2 = rb2 << 2; rb2b = rb2b << 2; }     // the row already fetched: bonus times four
    };

    auto wordA = [](uint32_t o, uint32_t f) { return __builtin_amdgcn_perm(f, o, 0x05040100u); };
    auto wordB = [](uint32_t o, uint32_t f) { return __builtin_amdgcn_perm(f, o, 0x07060302u); };

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
        }
    }
    if (k & 7) {
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
// Column slot s of lane gl holds padded column p (1..16*CT, columns right-aligned:
// DP column j = p - (16*CT - Q), p <= 0 .. are pads).
template <int C1, int C2, bool RAW>
__device__ __forceinline__ void load_pair_split(const SeqSetDev &rs, const SeqSetDev &qfwd, const SeqSetDev &qrc,
                                                const PairTile &pt, int gl, uint8_t *ref8, uint8_t *q8,
                                                uint32_t (&qb)[C1 + C2])
{
    using G = GeometrySplit<C1, C2>;
    constexpr int CT = C1 + C2;
    uint32_t *ref32 = reinterpret_cast<uint32_t *>(ref8);
    for (int k = gl; k < G::kRefBytes / 4; k += kGroup) ref32[k] = RAW ? 0xffffffffu : kLutPadRow * 0x01010101u;
    wave_sync();
    uint32_t rv[kSlots][CT], qv[kSlots][CT];
    int dq[kSlots][CT];
#pragma unroll
    for (int h = 0; h < kSlots; h++) {
        const SeqSetDev &qs = pt.comp[h] ? qrc : qfwd;
        const int S = G::kTileMax - pt.Q[h];
#pragma unroll
        for (int s = 0; s < CT; s++) {
            const int p = (s < C1) ? gl * C1 + s + 1 : G::W1 + gl * C2 + (s - C1) + 1;
            dq[h][s] = p - S - 1;                                      // 0-based DP column, < 0 = left pad
            rv[h][s] = fetch_base<RAW>(rs, slice_pos(pt.rp0[h], pt.R[h], pt.reverse[h], gl * CT + s));
            qv[h][s] = fetch_base<RAW>(qs, slice_pos(pt.qp0[h], pt.Q[h], pt.reverse[h], imax(dq[h][s], 0)));
        }
    }
#pragma unroll
    for (int s = 0; s < CT; s++) qb[s] = RAW ? 0u : kPermZero * 0x01010101u;
#pragma unroll
    for (int h = 0; h < kSlots; h++) {
        const int R = pt.R[h];
        uint8_t *rrow = ref8 + (G::kRow0 + pt.shift[h]) * 2 + h;
#pragma unroll
        for (int s = 0; s < CT; s++) {
            const int d = gl * CT + s;                                     // DP row handled by this slot of the loader
            const bool real = dq[h][s] >= 0;
            if (RAW) {
                if (d < R) rrow[d * 2] = (uint8_t)rv[h][s];
                const uint32_t qcode = real ? qv[h][s] : kQueryPad;
                if (real) q8[h * G::kTileMax + dq[h][s]] = (uint8_t)qcode;
                qb[s] |= qcode << (16 * h);
            } else {                                                       // LUT form, see load_pair
                if (d < R) rrow[d * 2] = (uint8_t)(24u - rv[h][s] * 8u);
                if (real) {
                    q8[h * G::kTileMax + dq[h][s]] = (uint8_t)(24u - qv[h][s] * 8u);
                    qb[s] = (qb[s] & ~(0xffu << (16 * h))) | ((qv[h][s] + 4u * h) << (16 * h));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Layout policy for extend_p16_kernel (gact_p16.hpp).  Requires early <= 16*C2 (window inside region 2).
template <int C1, int C2, bool TAG = false> struct SplitLayout {
    using G = GeometrySplit<C1, C2>;
    static constexpr int kSlotsPerLane = C1 + C2;
    static constexpr int kWalkCols = C2, kWalkQuads = G::kQuads, kWalkFmt = TAG ? 2 : 1;
    static constexpr int kLanes = kGroup;
    static constexpr int kRow0 = G::kRow0;
    static constexpr int kBlocksPerCu = 3;
    static constexpr bool kEndAligned = false;
    __device__ static int fin_lane(int Q) { (void)Q; return 0; }
    __device__ static int last_step(int R, int Q) { return split_last_step<C2>(R, Q); }
    __device__ static int first_pointer_step(int R, int Q, int early) { return split_first_pointer_step<C2>(R, Q, early); }
    template <bool RAW>
    __device__ static void load(const SeqSetDev &rs, const SeqSetDev &qf, const SeqSetDev &qr,
                                const PairTile &pt, int gl, uint8_t *ref8, uint8_t *q8, uint32_t (&qb)[C1 + C2], uint32_t *stage)
    {
        if (RAW) load_pair_split<C1, C2, RAW>(rs, qf, qr, pt, gl, ref8, q8, qb);
        else load_pair_packed<C1 + C2, kGroup>(rs, qf, qr, pt, gl, ref8, G::kRefBytes, G::kRow0, q8, G::kTileMax, qb, stage,
                                               Cols{});
    }
    // a lane's slots are two runs of consecutive columns: C1 of region 1, C2 of region 2 (right-aligned tile)
    struct Cols {
        __device__ static int column(int gl, int slot, int Q)
        {
            const int p = (slot < C1) ? gl * C1 + slot + 1 : G::W1 + gl * C2 + (slot - C1) + 1;     // padded column
            return p - (G::kTileMax - Q) - 1;
        }
        template <class F> __device__ static void for_each_run(F &&f)
        {
            f(std::integral_constant<int, 0>{}, std::integral_constant<int, C1>{});
            f(std::integral_constant<int, C1>{}, std::integral_constant<int, C2>{});
        }
    };
    template <bool RAW>
    __device__ static uint32_t pass(const P16Consts &kc, int gl, const uint16_t *ref16, const uint32_t (&qb)[C1 + C2],
                                    int T_end, int tB, uint32_t *wsA, uint32_t *wsB, const PairTile &)
    { dp_pass_p16s<C1, C2, RAW, TAG>(kc, gl, ref16, qb, T_end, tB, wsA, wsB); return 0; }
    // the start cell (R, Q) is the last column of region 2: lane 15, slot C2-1
    __device__ static void walk_start(int R, int Q, int tB_tile, int &l, int &c, int &k)
    { (void)Q; l = kGroup - 1; c = C2 - 1; k = R + (kGroup - 1) - tB_tile; }
    __device__ static int tile_tB(int tB, int shift) { return tB - shift - G::kLag; }
};

}  // namespace gact
