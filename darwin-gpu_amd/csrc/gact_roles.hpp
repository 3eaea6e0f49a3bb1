// gact_roles.hpp -- the main launch of linear-gap scorings with the traceback walk OFF the DP waves' critical path
// (round 5).
//
// In extend_p16_kernel a wave does everything for its eight tiles in turn: pick, load, DP pass, walk, consume.  The walk
// (align.cpp:185-230) is a serial chain of LDS and memory round trips on two lanes of sixteen: 22-25 % of a wave's time
// during which it issues next to nothing, and the SIMD's other waves cannot make up for it.  Candidates are
// independent (gact.cpp:48); only tile k+1 of the SAME candidate needs tile k's walk (gact.cpp:82-134).  So here
//
//   * a block is kRoleDp DP waves and kRoleWalk walker waves (10 + 2: one block per CU, three waves per SIMD);
//   * a DP wave carries TWO banks of eight tiles (2 x 4 groups x {A, B}).  It runs bank 0's pass, posts the eight walks as
//     jobs in LDS, runs bank 1's pass while the walker wave walks bank 0, comes back to bank 0, finds the results, advances
//     its chains (gact.cpp:111-133), picks and loads their next tiles, and so on: a DP wave is inside a pass all the time;
//   * lane n of a walker wave serves one tile slot of its block (DP wave / group / slot): up to 40 walks side by side in
//     one instruction stream instead of 8 -- the walk's instructions per tile drop by the same factor.  Its region refills (tb_refill_oct: eight 16-byte loads past the L1, one round trip) are taken
//     by all walking lanes at the same trip, once per eight moves, exactly the cadence of walk_chain_lin.
//
// The walk itself is walk_chain_lin's, move for move (same cells, same order, same stop tests, same band rule): the two
// functions are kept side by side on purpose; extend_p16_kernel stays the default (this launch is taken with GACT_HIP_ROLES=1
// or gact_hip_set_option "roles": measured no faster) and the kernel of every other layout.  What changes is where the bases come from: the loader's staging words (the tile's two slices
// as they lie in the 2-bit image) stay in LDS until the bank is loaded again, and the walker cuts the two bases of a cell
// out of them -- the DP wave's unpacked base arrays belong to the other bank by then.
//
// Hand-over: LDS words only, no barrier inside the loop.  A job's fields are written before its sequence number, a
// result's before its sequence number (LDS operations of one wave are performed in order); the pointer words a walk
// reads were stored by the DP wave and waited for (s_waitcnt vmcnt(0)) before the job was posted, and are read past the
// L1 (sc1) from the XCD's L2 -- both waves sit on one CU.
#pragma once

#include "gact_lin.hpp"

namespace gact {

// A block is ONE PER CU: 10 DP waves + 2 walker waves = three waves on every SIMD.  (Blocks of 5 + 1 waves, two per CU, were
// what the occupancy calculator allowed and what round 5 tried first: only one of them became resident per CU -- six waves of
// 168 registers do not spread 2-2-1-1 over the SIMDs twice -- and the launch ran on half its DP waves.)
#ifndef GACT_ROLE_DP_WAVES
#define GACT_ROLE_DP_WAVES 10
#endif
constexpr int kRoleDp = GACT_ROLE_DP_WAVES;                 // DP waves per block
constexpr int kRoleWalk = kRoleDp > 7 ? 2 : 1;              // walker waves per block
constexpr int kRoleThreads = 64 * (kRoleDp + kRoleWalk);
constexpr int kRoleBanks = 2;
constexpr int kRoleJobs = kRoleDp * 8;                      // tile slots of a block
constexpr int kRoleJobsPerWalker = kRoleJobs / kRoleWalk;   // ... and of a walker wave: its lanes at work
static_assert(kRoleDp % kRoleWalk == 0 && kRoleJobsPerWalker <= 64, "a walker wave serves whole DP waves, a lane per tile slot");
constexpr int kRoleCacheStride = 36;                        // dwords of region cache per walker lane (32 used; 36: eight banks apart)

struct WalkJob {                 // DP wave -> walker lane
    uint32_t seq;                // written last; 0 = nothing posted yet
    uint32_t ws_off;             // the tile's pointer words: byte offset of its (wsA | wsB) from ws_all
    int R, Q;
    int k0;                      // stored step of the start cell (L::walk_start)
    int v0;                      // H[R][Q] from the pass
    int band_lim;                // see walk_chain_lin; -1: every block is there
    uint32_t where;              // bits 0-7: staged position of the ref slice's first base, 8-15: the query slice's,
                                 // 16: AlignWithBT's `reverse`, 17-31: dword index of the tile's ref segment in the stage array
};
struct WalkDone {                // walker lane -> DP wave
    uint32_t seq;                // written last
    int ref_steps, query_steps;
    int dv;                      // v0 - v: what the columns of the tile scored
    int redo;
};

// words the linear-gap pass writes per wave and bank: [flush block][uint4 n < QD][kWsRow]
template <class L> constexpr size_t role_bank_words() { return (size_t)L::G::kMaxFlush * L::kWalkQuads * kWsRow * 4; }
template <class L> __host__ constexpr size_t role_ws_words(int blocks) { return (size_t)blocks * kRoleDp * kRoleBanks * role_bank_words<L>(); }

// ---------------------------------------------------------------------------
// the walker wave
template <class L>
__device__ __forceinline__ void role_walker(const KParams &kp, const uint32_t *__restrict__ ws_all, WalkJob (*jobs)[kRoleBanks],
                                            WalkDone (*done)[kRoleBanks], const uint32_t *stage_all, uint32_t *cache_all,
                                            const int *dp_finished)
{
    constexpr int CW = L::kWalkCols, QN = L::kWalkQuads, ROW = kWsRow;
    constexpr int kSeg = StageGeom<L::kSlotsPerLane, L::kLanes>::kSeg;
    constexpr uint32_t kMagic = (65536u + CW - 1) / CW;
    constexpr uint32_t kM = 3u, kI = 2u, kD = 1u;               // align.h:23 numbering, as the pass tags them
    typedef __attribute__((address_space(3))) const uint8_t LdsByte;
    typedef __attribute__((address_space(3))) const uint32_t LdsWord;
    typedef __attribute__((address_space(3))) volatile uint32_t LdsFlag;
    const int wlane = threadIdx.x & 63;
    const bool serving = wlane < kRoleJobsPerWalker;
    // (the tile slot this lane serves; idle lanes look at the wave's first slot and never act)
    const int lane = ((threadIdx.x >> 6) - kRoleDp) * kRoleJobsPerWalker + (serving ? wlane : 0);
    uint32_t *scratch = cache_all + lane * kRoleCacheStride;
    LdsByte *cache = (LdsByte *)scratch;
    LdsWord *stage = (LdsWord *)stage_all;
    const int early = kp.early;
    int v_gap, v_mism, v_match;
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_gap) : "s"(__builtin_amdgcn_readfirstlane(kp.ext)));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_mism) : "s"(__builtin_amdgcn_readfirstlane(kp.mismatch)));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_match) : "s"(__builtin_amdgcn_readfirstlane(kp.match)));

    uint32_t done_seq0 = 0u, done_seq1 = 0u;                    // sequence number of the last job walked, per bank
    bool active = false, go = false, redo = false;
    int bank = 0;
    uint32_t seq = 0;
    int p0 = 0, kA = 0, nlim_i = 0, nlim_j = 0, nis = 0, njs = 0, v = 0, v0 = 0, band_lim = -1;
    uint32_t ws_off = 0;
    int rpos0 = 0, qpos0 = 0, dir = 0, seg = 0;                  // staged bit positions of the bases of cell (R, Q); +-1 per step
    int l = 0, c = 0, k = 0;
    uint32_t cur = 0, rcode = 0, qcode = 0;
    TbRegion<CW> rg;
    rg.l0 = 0; rg.fbase[0] = rg.fbase[1] = rg.fbase[2] = 0; rg.qbase0 = 0;
    int off0 = 0, off1 = 0;
    int idle = 0;

    auto fetch = [&](int fl, int fc, int fk) {
        const int off = fl == rg.l0 ? off0 : off1;
        const uint32_t row = ((uint32_t)fk >> 3 << 5) + (uint32_t)off;
        const uint32_t w = *(LdsWord *)(cache + (((uint32_t)fc >> 1 << 2) + row));
        return __builtin_amdgcn_ubfe(w, (((uint32_t)fc & 1u) << 4) + 14u - (((uint32_t)fk & 7u) << 1), 2u);
    };
    // the two bases of the cell the walk stands on, out of the tile's staged slices (load_pair_packed: base d of a slice
    // sits at bit 2 * (16 kFront + bit0 + d) of its segment)
    auto bases = [&]() {
        const int rp = rpos0 + dir * nis, qp = qpos0 + dir * njs;         // nis, njs <= 0; forward slices are walked downwards
        const uint32_t rw = stage[seg + (rp >> 4)], qw = stage[seg + kSeg + (qp >> 4)];
        rcode = __builtin_amdgcn_ubfe(rw, ((uint32_t)rp & 15u) << 1, 2u);
        qcode = __builtin_amdgcn_ubfe(qw, ((uint32_t)qp & 15u) << 1, 2u);
    };

    for (;;) {
        // ---- intake: an idle lane looks for a job in its tile slot's two banks (the older one first)
        bool fresh = false;
        if (serving && !active) {
            const uint32_t s0 = *(LdsFlag *)&jobs[lane][0].seq, s1 = *(LdsFlag *)&jobs[lane][1].seq;
            const bool n0 = s0 != done_seq0, n1 = s1 != done_seq1;
            if (n0 | n1) {
                bank = (n0 && n1) ? ((int)(s1 - s0) < 0 ? 1 : 0) : (n1 ? 1 : 0);
                seq = bank ? s1 : s0;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const WalkJob jb = jobs[lane][bank];
                const int R = jb.R, Q = jb.Q;
                ws_off = jb.ws_off; v0 = jb.v0; band_lim = jb.band_lim;
                int l0, c0, k0;
                L::walk_start(R, Q, 0, l0, c0, k0);                           // (lane and column of (R, Q); the step comes with the job)
                k0 = jb.k0;
                p0 = l0 * CW + c0; kA = k0 - l0;
                nlim_i = -imin(early, R); nlim_j = -imin(early, Q);
                const bool rev = (jb.where >> 16) & 1u;
                seg = (int)(jb.where >> 17);
                dir = rev ? -1 : 1;                                            // a step up the tile (nis - 1): slice index + 1 when reversed
                rpos0 = (int)(jb.where & 0xffu) + (rev ? 0 : R - 1);
                qpos0 = (int)((jb.where >> 8) & 0xffu) + (rev ? 0 : Q - 1);
                nis = 0; njs = 0; v = v0; redo = false;
                l = l0; c = c0; k = k0;
                go = (R >= 1) & (Q >= 1) & (early > 0) & (v != 0);
                fresh = true; active = true;
            }
        }
        if (!__any(active)) {
            // nothing to walk: the DP waves have all left (each waits for its last results before it goes), or wait a little
            if (*(__attribute__((address_space(3))) volatile const int *)dp_finished >= kRoleDp) break;
            if (idle < 64) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
            idle++;
            continue;
        }
        idle = 0;
        // ---- region refill at the cell the walk stands on, every walking lane at once (one memory round trip), the op
        //      and the bases of that cell
        if (active && go) {
            tb_refill_oct<CW, QN, ROW>(ws_all, ws_off, scratch, l, c, k, rg);
            off0 = 4 * (-8 * rg.fbase[0] - 4 * rg.qbase0);
            off1 = 4 * (16 - 8 * rg.fbase[1] - 4 * (QN - 2));
            cur = fetch(l, c, k);
            bases();
        }
        (void)fresh;
        // ---- eight moves (walk_chain_lin, statement for statement)
#pragma unroll 1
        for (int m = 0; m < 8; m++) {
            if (!__any(go)) break;
            if (go) {
                const bool diag = cur == kM;
                const int sub = rcode == qcode ? v_match : v_mism;
                v -= diag ? sub : v_gap;
                nis -= cur != kD;
                njs -= cur != kI;
                const int p = imax(p0 + njs, 0);
                l = (int)(__umul24((uint32_t)p, kMagic) >> 16);
                c = p + __mul24(l, -CW);
                k = imax(kA + l + nis, 0);
                if (m == 7) {
                    // (the next refill is due: still a refill's worth of moves inside the stored band?  gact_lin.hpp LinBand)
                    redo = redo | ((band_lim >= 0) & ((unsigned)(nis - njs + band_lim) > (unsigned)(2 * band_lim)));
                } else {
                    cur = fetch(l, c, k);
                    bases();
                }
                go = !((diag && v == 0) || nis <= nlim_i || njs <= nlim_j || redo);
            }
        }
        // ---- a walk that has ended: its result, then its sequence number
        if (active && !go) {
            WalkDone d;
            d.seq = seq; d.ref_steps = -nis; d.query_steps = -njs; d.dv = v0 - v; d.redo = redo ? 1 : 0;
            done[lane][bank].ref_steps = d.ref_steps; done[lane][bank].query_steps = d.query_steps;
            done[lane][bank].dv = d.dv; done[lane][bank].redo = d.redo;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            *(LdsFlag *)&done[lane][bank].seq = seq;
            if (bank) done_seq1 = seq; else done_seq0 = seq;
            active = false;
        }
    }
}

// ---------------------------------------------------------------------------
// L: SplitLayoutLin<7, 13> (2-bit sets, linear gaps).  TWO_SETS as in extend_p16_kernel (overlapped seeding).
template <class L, bool TWO_SETS = false>
__global__ __launch_bounds__(kRoleThreads, 3) void extend_roles_kernel(
    KParams kp, P16Consts kc, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc,
    int same_file, gact_overlap *__restrict__ out, ChainQueues cq,
    uint32_t *__restrict__ ws_all)
{
    using G = typename L::G;
    constexpr int LANES = L::kLanes;
    static_assert(LANES == kGroup && L::kWalkFmt == 3 && L::kEndAligned, "the split linear-gap layout");
    constexpr int kGroupsOfWave = 64 / LANES;
    constexpr int kDpGroups = kRoleDp * kGroupsOfWave;
    constexpr int kStageWords = StageGeom<L::kSlotsPerLane, LANES>::kWords;
    constexpr int kSeg = StageGeom<L::kSlotsPerLane, LANES>::kSeg;
    static_assert(((kDpGroups * kRoleBanks * kStageWords) >> 15) == 0, "stage index fits WalkJob::where");
    __shared__ __attribute__((aligned(16))) uint8_t lds[kDpGroups * G::kGroupLds];
    __shared__ ChainState chain_lds[kDpGroups][kRoleBanks][kSlots];
    // the loader's staging words of a group's two tiles, per bank: they stay until the bank is loaded again -- the walker
    // reads the bases of its cells from them
    __shared__ __attribute__((aligned(16))) uint32_t stage_lds[kDpGroups][kRoleBanks][kStageWords];
    __shared__ WalkJob jobs[kRoleJobs][kRoleBanks];
    __shared__ WalkDone done[kRoleJobs][kRoleBanks];
    __shared__ __attribute__((aligned(16))) uint32_t cache_lds[kRoleJobs * kRoleCacheStride];
    __shared__ int dp_finished;
    __shared__ int bank_lds[kDpGroups][kRoleBanks][6];       // {R_A, Q_A, R_B, Q_B, posted seq, -}

    for (int n = threadIdx.x; n < kRoleJobs * kRoleBanks; n += kRoleThreads) {
        (&jobs[0][0])[n].seq = 0;
        (&done[0][0])[n].seq = 0;
    }
    if (threadIdx.x == 0) dp_finished = 0;
    __syncthreads();

    const int wave_in_block = threadIdx.x >> 6;
    if (wave_in_block >= kRoleDp) {
        __builtin_amdgcn_s_setprio(3);
        role_walker<L>(kp, ws_all, jobs, done, &stage_lds[0][0][0], cache_lds, &dp_finished);
        return;
    }

    WaveCtx w;
    {
        const int lane = threadIdx.x & 63;
        w.gl = lane & (LANES - 1);
        w.g = lane / LANES;
        w.slot = (blockIdx.x * kRoleDp + wave_in_block) * kGroupsOfWave + w.g;
        w.n_slots = gridDim.x * kRoleDp * kGroupsOfWave;
    }
    const int group_in_block = wave_in_block * kGroupsOfWave + w.g;
    uint8_t *ref8 = lds + group_in_block * G::kGroupLds;
    uint8_t *q8 = ref8 + G::kRefBytes;
    const uint16_t *ref16_lane = reinterpret_cast<const uint16_t *>(ref8) + (L::kRow0 - 1 - w.gl);
    // a wave's pointer words: [bank][flush block][uint4 n][tile A | tile B][the wave's 64 lanes]
    uint32_t *ws_wave = ws_all + (size_t)(blockIdx.x * kRoleDp + wave_in_block) * (kRoleBanks * role_bank_words<L>());
    // the walker lane of this lane's tile slot (lanes 0 and 1 of a group: slots A and B)
    const int job_lane = wave_in_block * 8 + w.g * kSlots + (w.gl & 1);

    for (int b = 0; b < kRoleBanks; b++)
        if (w.gl < kSlots) { chain_lds[group_in_block][b][w.gl].phase = 2; chain_lds[group_in_block][b][w.gl].cand = -1; }
    wave_sync();
    bool exhausted = false;
    int my_bucket = 0;
    int bucket_first = 0;
    if (cq.leave_longest > 0) {
        int acc = 0;
        while (bucket_first < kBuckets && (acc += cq.bucket_count[bucket_first]) <= cq.leave_longest) bucket_first++;
        if (bucket_first >= kBuckets) bucket_first = 0;
    }
    bool second_set = false;
    constexpr bool one_set = !TWO_SETS;
    int idle_polls = 0;
    // (the ranking against the longest chain running: every kRankEvery-th tile, the waves of a launch taking turns; the first tile always)
    int rank_turn = 0, rank_cached = 0;
    // What a bank's pass leaves behind for its consume step lives in LDS, like the chain states: the DP loop owns the
    // register file.  bank_lds[group][bank] = {R, Q of slot A, of slot B (0: no tile), sequence number of the jobs in flight
    // (0: none)}; the wave's job counter sits beside it.
    int *bt = &bank_lds[group_in_block][0][0];
    if (w.gl == 0) {
        for (int n = 0; n < kRoleBanks * 6; n++) bt[n] = 0;
    }
    wave_sync();
    int bank = 0;
    __builtin_amdgcn_s_setprio(3);
#ifdef GACT_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tl_cyc0 = __builtin_amdgcn_s_memtime();
    unsigned long long tl_empty = 0;
#endif

    for (;;) {
        ChainState *st = chain_lds[group_in_block][bank];
        GACT_STAMP(t_w0);
        // ---- the results of this bank's walks (posted one pass of the other bank ago): gact.cpp:111-133 / :172-194
        const int *bk = &bank_lds[group_in_block][bank][0];
        const uint32_t want = (uint32_t)__builtin_amdgcn_readfirstlane(bk[4]);       // (the same for the wave's four groups)
        if (want != 0) {
            typedef __attribute__((address_space(3))) volatile uint32_t LdsFlag;
            const int sv_R[kSlots] = {bk[0], bk[2]}, sv_Q[kSlots] = {bk[1], bk[3]};
            const bool mine = w.gl < kSlots && ((w.gl & 1) ? sv_R[1] : sv_R[0]) > 0;
            // (a walker that never answers must not hang the machine: after two seconds the wave goes on without the results
            //  and says so -- bit 30 of the band_redos counter, which gact_hip_run_stats reports)
            const unsigned long long wd0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const bool ok = !mine || *(LdsFlag *)&done[job_lane][bank].seq == want;
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - wd0 > 200000000ull) {
                    if ((threadIdx.x & 63) == 0) atomicOr(cq.band_redos, 1 << 30);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int ref_steps = 0, query_steps = 0, nst = 0, redo = 0;
            ScoreWalk wk;
            wk.score = 0; wk.pend_gap = 0; wk.open_flag = 0; wk.have_left = 0; wk.left_first_gap = 0;
            if (mine) {
                const WalkDone d = done[job_lane][bank];
                wk.load(st[w.gl & 1]);
                wk.score += d.dv;                                        // (open == extend: the gap bookkeeping decides nothing)
                ref_steps = d.ref_steps; query_steps = d.query_steps; nst = d.ref_steps + d.query_steps; redo = d.redo;
            }
#pragma unroll
            for (int h = 0; h < kSlots; h++) {
                const bool had = sv_R[h] > 0;
                if (had) {
                    ChainState s = st[h];
                    if (__shfl(redo, h, LANES)) {
                        s.full = 1;
                        if (w.gl == 0) atomicAdd(cq.band_redos, 1);
                    } else {
                        s.full = 0;
                        s.n_tiles++;
                        s.cells += (int64_t)sv_R[h] * sv_Q[h];
                        chain_advance<LANES>(s, false, wk, ref_steps, query_steps, nst, h);
                    }
                    wave_sync();
                    if (w.gl == 0) st[h] = s;
                }
                wave_sync();
            }
            wave_sync();
            if (w.gl == 0) bank_lds[group_in_block][bank][4] = 0;
            wave_sync();
        }
        GACT_STAMP(t_a);
        // ---- control phase: both slots of the bank pick their next tile
        PairTile pt;
        bool have[kSlots];
        int Tend_h[kSlots], tB_h[kSlots];
        int longest = 0;
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
                        if (__hip_atomic_load(cq.more_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) break;
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        second_set = true;
                        my_bucket = 0;
                    }
                    if (cand < 0) {
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
        if (w.gl == 0) {
            int *bw = &bank_lds[group_in_block][bank][0];
            bw[0] = have[0] ? pt.R[0] : 0; bw[1] = pt.Q[0]; bw[2] = have[1] ? pt.R[1] : 0; bw[3] = pt.Q[1];
        }
        const bool any_here = have[0] | have[1];
        if (!__any(any_here)) {
            // nothing for this bank.  The other bank may have walks in flight or chains of its own: go there; with neither,
            // the wave is done once the queues are (or waits for the second set of overlapped seeding, as extend_p16_kernel)
            const bool other_busy = __builtin_amdgcn_readfirstlane(bank_lds[group_in_block][bank ^ 1][4]) != 0;
            if (!other_busy) {
                const ChainState *so = chain_lds[group_in_block][bank ^ 1];
                if (__all(exhausted && st[0].phase == 2 && st[1].phase == 2 && so[0].phase == 2 && so[1].phase == 2)) break;
                if (!second_set && !one_set) {
                    __builtin_amdgcn_s_sleep(127);
                    if (++idle_polls > 512) { second_set = true; exhausted = true; my_bucket = kBuckets; }
                }
            }
            bank ^= 1;
            continue;
        }
        const int T_end = wave_max_groups<LANES>(imax(have[0] ? Tend_h[0] : 0, have[1] ? Tend_h[1] : 0));
        const int reach0 = have[0] ? tB_h[0] + (T_end - Tend_h[0]) : 0x7fffffff;
        const int reach1 = have[1] ? tB_h[1] + (T_end - Tend_h[1]) : 0x7fffffff;
        const int tB = wave_min_groups<LANES>(imin(reach0, reach1));
        pt.col_from = imax(imin(have[0] ? pt.Q[0] : 0x7fff, have[1] ? pt.Q[1] : 0x7fff) - kp.early, 0);
        pt.band = kp.band;
        pt.shift[0] = have[0] ? T_end - Tend_h[0] : 0;
        pt.shift[1] = have[1] ? T_end - Tend_h[1] : 0;

        GACT_STAMP(t_b);
        uint32_t qb[L::kSlotsPerLane];
        uint32_t *stage = stage_lds[group_in_block][bank];
        L::template load<false>(refs, qfwd, qrc, pt, w.gl, ref8, q8, qb, stage);
        wave_sync();
        GACT_STAMP(t_c);

        const int wave_longest = wave_max_groups<LANES>(longest);
        const int ref_longest = ranked_longest(cq, kp, wave_longest, rank_turn, rank_cached);
        const bool rank_hi = kp.prio_bases[0] == 0 ? 16 * wave_longest > (kp.prio_bases[1] >> 8) * ref_longest
                                                   : wave_longest > kp.prio_bases[1];
        const bool rank_mid = kp.prio_bases[0] == 0 ? 16 * wave_longest > (kp.prio_bases[1] & 255) * ref_longest
                                                    : wave_longest > kp.prio_bases[0];
        if (rank_hi) __builtin_amdgcn_s_setprio(2);
        else if (rank_mid) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        uint32_t *wsA = ws_wave + (size_t)bank * role_bank_words<L>() + (w.g * LANES) * 4;
        uint32_t *wsB = wsA + 64 * 4;
        const uint32_t fin = L::template pass<false>(kc, w.gl, ref16_lane, qb, T_end, tB, wsA, wsB, pt);
        __builtin_amdgcn_s_setprio(3);
        const int v0A = (int)(int16_t)(__shfl(fin, L::fin_lane(pt.Q[0]), LANES) & 0xffffu);
        const int v0B = (int)(int16_t)(__shfl(fin, L::fin_lane(pt.Q[1]), LANES) >> 16);
        GACT_STAMP(t_d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pointer stores have reached the L2 before a walker is told
        GACT_STAMP(t_e);

        // ---- post the walks of this bank: lane h of the group writes slot h's job, its sequence number last
        // (jobs of this wave are numbered 1, 2, ... over both banks; the counter sits in the first group's spare LDS word)
        const uint32_t wave_seq = (uint32_t)__builtin_amdgcn_readfirstlane(bank_lds[wave_in_block * kGroupsOfWave][0][5]) + 1u;
        if (w.gl < kSlots) {
            const int h = w.gl;
            if (h ? have[1] : have[0]) {
                const int Rh = h ? pt.R[1] : pt.R[0], Qh = h ? pt.Q[1] : pt.Q[0], sh = h ? pt.shift[1] : pt.shift[0];
                int l0, c0, k0;
                L::walk_start(Rh, Qh, L::tile_tB(tB, sh), l0, c0, k0);
                WalkJob *jb = &jobs[job_lane][bank];
                jb->ws_off = (uint32_t)((const char *)(h ? wsB : wsA) - (const char *)ws_all);
                jb->R = Rh; jb->Q = Qh; jb->k0 = k0; jb->v0 = h ? v0B : v0A;
                jb->band_lim = ((kp.band & 0xffff) > 0 && !(h ? pt.full[1] : pt.full[0])) ? (kp.band & 0xffff) - kLinWalkSpan : -1;
                const uint32_t seg_index = (uint32_t)((group_in_block * kRoleBanks + bank) * kStageWords + (2 * h) * kSeg);
                constexpr uint32_t kFrontBits = 16u * StageGeom<L::kSlotsPerLane, LANES>::kFront;
                jb->where = (kFrontBits + (uint32_t)((h ? pt.rp0[1] : pt.rp0[0]) & 15)) |
                            ((kFrontBits + (uint32_t)((h ? pt.qp0[1] : pt.qp0[0]) & 15)) << 8) |
                            ((h ? pt.reverse[1] : pt.reverse[0]) ? 1u << 16 : 0u) | (seg_index << 17);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                *(__attribute__((address_space(3))) volatile uint32_t *)&jb->seq = wave_seq;
            }
        }
        wave_sync();
        if (w.gl == 0) bank_lds[group_in_block][bank][4] = (int)wave_seq;
        if ((threadIdx.x & 63) == 0) bank_lds[wave_in_block * kGroupsOfWave][0][5] = (int)wave_seq;
        wave_sync();
        GACT_STAMP(t_f);
        GACT_ACC(0, t_a, t_b); GACT_ACC(1, t_b, t_c); GACT_ACC(2, t_c, t_d); GACT_ACC(3, t_d, t_e);
        GACT_ACC(4, t_w0, t_a); GACT_ACC(5, t_e, t_f);
#ifdef GACT_STAMPS
        stamp_acc[6] += 1; stamp_acc[7] += (unsigned long long)(T_end - tB + 1);
#endif
        bank ^= 1;
    }
    if ((threadIdx.x & 63) == 0) atomicAdd(&dp_finished, 1);
#ifdef GACT_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 8; k++) atomicAdd(&g_stamps[k], stamp_acc[k]);
        const int wv = blockIdx.x * kRoleDp + wave_in_block;
        if (wv < 4096) {
            g_timeline[4 * wv] = tl_start; g_timeline[4 * wv + 1] = tl_empty;
            g_timeline[4 * wv + 2] = __builtin_amdgcn_s_memrealtime(); g_timeline[4 * wv + 3] = stamp_acc[6];
            g_wave_cycles[wv] = __builtin_amdgcn_s_memtime() - tl_cyc0;
        }
    }
#endif
}

}  // namespace gact
