// gact_coop.hpp -- the split linear-gap main launch with COOPERATIVE, BATCHED traceback walks (round 5).
//
// What round 5 measured first (gact_roles.hpp, profiles/r05/): taking the walk off the DP waves -- dedicated walker waves, two
// banks of tiles per DP wave -- does not make the launch faster.  The machine is bound by VALU issue (one wave64 instruction
// per ~4 cycles and SIMD, two waves inside a pass saturate it), so a wave that
// idles through its walk costs nothing the other waves of its SIMD do not make up, and a walker wave that runs all the time
// with a few lanes at work executes as many instructions as the 8-lane walks it replaced (SQ_INSTS_VALU 1.608e10 against
// 1.656e10).  What does help is FEWER instructions: the walk (align.cpp:185-230) is 213 dependent steps of ~45 VALU
// instructions per tile, executed in extend_p16_kernel with 8 of a wave's 64 lanes at work -- 16-18 % of the launch's
// instructions.  So here
//
//   * every wave of a block (4 waves, 3 blocks per CU: the shape of extend_p16_kernel) is a DP wave with TWO banks of
//     eight tiles; after a pass it posts the bank's eight walks as jobs in LDS and goes on with its other bank;
//   * nobody walks until somebody needs a result: a wave that comes back to a bank whose jobs are still unclaimed takes the
//     block's walk lock and walks EVERY posted job of the block -- its own and the other three waves', up to 64, one per
//     lane -- in one instruction stream, then goes on.  A walk batch is ~30 jobs in the steady state: the walk's
//     instructions per tile drop by that factor over 8;
//   * the walk itself is walk_chain_lin's, move for move (same cells, same order, same stop tests, same band rule); the
//     bases of a cell come out of the loader's staging words, which stay in LDS until their bank is loaded again.
//
// Hand-over: LDS words only.  A job's fields are written before its state word (posted), a lane claims a job with a
// compare-and-swap on the state word, a result is written before the state word says done; the pointer words a walk reads
// were stored by the posting wave and waited for (s_waitcnt vmcnt(0)) before it posted, and are read past the L1 (sc1).
// Progress: a wave that needs a result either finds it, or finds the job unclaimed and walks it itself (after the lock), or
// finds it claimed -- then the lock holder is walking it and finishes in bounded time.  A watchdog turns a hang into a
// reported error (bit 30 of the band_redos counter).
#pragma once

#include "gact_lin.hpp"

namespace gact {

constexpr int kCoopBanks = 2;
constexpr int kCoopJobs = (kBlockThreads / 64) * 8 * kCoopBanks;       // tile slots of a block x banks = the lanes of a walking wave
static_assert(kCoopJobs == 64, "a walking wave serves every job of its block, one per lane");
constexpr int kCoopCacheStride = 36;                                   // dwords of region cache per walking lane (32 used)

struct CoopJob {                 // posting wave -> walking lane
    uint32_t ws_off;             // the tile's pointer words: byte offset of its (wsA | wsB) from ws_all
    int R, Q;
    int k0;                      // stored step of the start cell (L::walk_start)
    int v0;                      // H[R][Q] from the pass
    int band_lim;                // see walk_chain_lin; -1: every block is there
    uint32_t where;              // bits 0-7: staged position of the ref slice's first base, 8-15: the query slice's,
                                 // 16: AlignWithBT's `reverse`, 17-31: dword index of the tile's ref segment in the stage array
    int pad_;
};
struct CoopDone { int ref_steps, query_steps, dv, redo; };
// state word of a job: (sequence number << 2) | phase; phase 1 posted, 2 claimed by a walking lane, 3 done

template <class L> constexpr size_t coop_bank_words() { return (size_t)L::G::kMaxFlush * L::kWalkQuads * kWsRow * 4; }
template <class L> __host__ constexpr size_t coop_ws_words(int blocks) { return (size_t)blocks * (kBlockThreads / 64) * kCoopBanks * coop_bank_words<L>(); }

// One batch: lane j claims job j if it is posted, all claimed jobs are walked in lock step, results written.
template <class L>
__device__ __forceinline__ void coop_walk_batch(const KParams &kp, const uint32_t *__restrict__ ws_all, const CoopJob *jobs, uint32_t *jstate,
                                                CoopDone *done, const uint32_t *stage_all, uint32_t *cache_all)
{
    constexpr int CW = L::kWalkCols, QN = L::kWalkQuads, ROW = kWsRow;
    constexpr int kSeg = StageGeom<L::kSlotsPerLane, L::kLanes>::kSeg;
    constexpr uint32_t kMagic = (65536u + CW - 1) / CW;
    constexpr uint32_t kM = 3u, kI = 2u, kD = 1u;               // align.h:23 numbering, as the pass tags them
    typedef __attribute__((address_space(3))) const uint8_t LdsByte;
    typedef __attribute__((address_space(3))) const uint32_t LdsWord;
    const int lane = threadIdx.x & 63;
    uint32_t *scratch = cache_all + lane * kCoopCacheStride;
    LdsByte *cache = (LdsByte *)scratch;
    LdsWord *stage = (LdsWord *)stage_all;
    const int early = kp.early;
    int v_gap, v_mism, v_match;
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_gap) : "s"(__builtin_amdgcn_readfirstlane(kp.ext)));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_mism) : "s"(__builtin_amdgcn_readfirstlane(kp.mismatch)));
    asm volatile("s_nop 1\n\tv_mov_b32 %0, %1" : "=v"(v_match) : "s"(__builtin_amdgcn_readfirstlane(kp.match)));

    // ---- claim
    const uint32_t st0 = __hip_atomic_load(&jstate[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    bool active = false;
    if ((st0 & 3u) == 1u) active = atomicCAS(&jstate[lane], st0, st0 + 1u) == st0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (!__any(active)) return;
    int p0 = 0, kA = 0, nlim_i = 0, nlim_j = 0, nis = 0, njs = 0, v = 0, v0 = 0, band_lim = -1;
    uint32_t ws_off = 0;
    int rpos0 = 0, qpos0 = 0, dir = 0, seg = 0;
    int l = 0, c = 0, k = 0;
    bool go = false, redo = false;
    if (active) {
        const CoopJob jb = jobs[lane];
        const int R = jb.R, Q = jb.Q;
        ws_off = jb.ws_off; v0 = jb.v0; band_lim = jb.band_lim;
        int l0, c0, k0;
        L::walk_start(R, Q, 0, l0, c0, k0);                       // (lane and column of (R, Q); the step comes with the job)
        k0 = jb.k0;
        p0 = l0 * CW + c0; kA = k0 - l0;
        nlim_i = -imin(early, R); nlim_j = -imin(early, Q);
        const bool rev = (jb.where >> 16) & 1u;
        seg = (int)(jb.where >> 17);
        dir = rev ? -1 : 1;                                        // a step up the tile (nis - 1): slice index + 1 when reversed
        rpos0 = (int)(jb.where & 0xffu) + (rev ? 0 : R - 1);
        qpos0 = (int)((jb.where >> 8) & 0xffu) + (rev ? 0 : Q - 1);
        v = v0;
        l = l0; c = c0; k = k0;
        go = (R >= 1) & (Q >= 1) & (early > 0) & (v != 0);
    }
    uint32_t cur = 0, rcode = 0, qcode = 0;
    TbRegion<CW> rg;
    rg.l0 = 0; rg.fbase[0] = rg.fbase[1] = rg.fbase[2] = 0; rg.qbase0 = 0;
    int off0 = 0, off1 = 0;
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
#ifdef GACT_STAMPS
    unsigned long long n_trips = 0, n_lane_trips = 0;
#endif
    // ---- refill (every walking lane at the same trip: one memory round trip), then eight moves: walk_chain_lin, statement
    //      for statement
    while (__any(go)) {
        if (go) {
            tb_refill_oct<CW, QN, ROW>(ws_all, ws_off, scratch, l, c, k, rg);
            off0 = 4 * (-8 * rg.fbase[0] - 4 * rg.qbase0);
            off1 = 4 * (16 - 8 * rg.fbase[1] - 4 * (QN - 2));
            cur = fetch(l, c, k);
            bases();
        }
#pragma unroll 1
        for (int m = 0; m < 8; m++) {
            if (!__any(go)) break;
#ifdef GACT_STAMPS
            n_trips++; n_lane_trips += (unsigned long long)__builtin_popcountll(__ballot(go));
#endif
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
    }
    if (active) {
        done[lane].ref_steps = -nis; done[lane].query_steps = -njs; done[lane].dv = v0 - v; done[lane].redo = redo ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&jstate[lane], st0 + 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifdef GACT_STAMPS
    {
        const unsigned long long n_jobs = (unsigned long long)__builtin_popcountll(__ballot(active));
        if (lane == 0) { atomicAdd(&g_coop_counts[0], 1ull); atomicAdd(&g_coop_counts[1], n_trips); atomicAdd(&g_coop_counts[2], n_lane_trips);
                         atomicAdd(&g_coop_counts[3], n_jobs); }
    }
#endif
}

// ---------------------------------------------------------------------------
// L: SplitLayoutLin<7, 13> (2-bit sets, linear gaps).  TWO_SETS as in extend_p16_kernel (overlapped seeding).
template <class L, bool TWO_SETS = false>
__global__ __launch_bounds__(kBlockThreads, 3) void extend_coop_kernel(
    KParams kp, P16Consts kc, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc,
    int same_file, gact_overlap *__restrict__ out, ChainQueues cq,
    uint32_t *__restrict__ ws_all)
{
    using G = typename L::G;
    constexpr int LANES = L::kLanes;
    static_assert(LANES == kGroup && L::kWalkFmt == 3 && L::kEndAligned, "the split linear-gap layout");
    constexpr int kGroupsOfWave = 64 / LANES;
    constexpr int kGroupsPerBlock = (kBlockThreads / 64) * kGroupsOfWave;
    constexpr int kStageWords = StageGeom<L::kSlotsPerLane, LANES>::kWords;
    constexpr int kSeg = StageGeom<L::kSlotsPerLane, LANES>::kSeg;
    static_assert(((kGroupsPerBlock * kCoopBanks * kStageWords) >> 15) == 0, "stage index fits CoopJob::where");
    constexpr int kRefBytes = (G::kRefBytes + 15) & ~15;          // the ref stream alone: the walker keeps no query bytes
    __shared__ __attribute__((aligned(16))) uint8_t lds[kGroupsPerBlock * kRefBytes];
    __shared__ ChainState chain_lds[kGroupsPerBlock][kCoopBanks][kSlots];
    // the loader's staging words of a group's two tiles, per bank: they stay until the bank is loaded again -- a walking
    // lane reads the bases of its cells from them
    __shared__ __attribute__((aligned(16))) uint32_t stage_lds[kGroupsPerBlock][kCoopBanks][kStageWords];
    __shared__ __attribute__((aligned(16))) CoopJob jobs[kCoopJobs];
    __shared__ CoopDone done[kCoopJobs];
    __shared__ uint32_t jstate[kCoopJobs];
    __shared__ __attribute__((aligned(16))) uint32_t cache_lds[kCoopJobs * kCoopCacheStride];
    __shared__ int bank_lds[kGroupsPerBlock][kCoopBanks][6];   // {R_A, Q_A, R_B, Q_B, state word of the jobs in flight (0: none), -}
    __shared__ uint32_t walk_lock;

    for (int n = threadIdx.x; n < kCoopJobs; n += kBlockThreads) jstate[n] = 0;
    if (threadIdx.x == 0) walk_lock = 0;
    __syncthreads();

    const int wave_in_block = threadIdx.x >> 6;
    WaveCtx w;
    {
        const int lane = threadIdx.x & 63;
        w.gl = lane & (LANES - 1);
        w.g = lane / LANES;
        w.slot = (blockIdx.x * (kBlockThreads / 64) + wave_in_block) * kGroupsOfWave + w.g;
        w.n_slots = gridDim.x * (kBlockThreads / 64) * kGroupsOfWave;
    }
    const int group_in_block = wave_in_block * kGroupsOfWave + w.g;
    uint8_t *ref8 = lds + group_in_block * kRefBytes;
    uint8_t *q8 = ref8;                                                 // (never written: load_pair_packed<..., false>)
    const uint16_t *ref16_lane = reinterpret_cast<const uint16_t *>(ref8) + (L::kRow0 - 1 - w.gl);
    // a wave's pointer words: [bank][flush block][uint4 n][tile A | tile B][the wave's 64 lanes]
    uint32_t *ws_wave = ws_all + (size_t)(blockIdx.x * (kBlockThreads / 64) + wave_in_block) * (kCoopBanks * coop_bank_words<L>());
    // this lane's job of bank b: ((group in block) * 2 + slot) * 2 + b   (lanes 0 and 1 of a group: slots A and B)
    const int job_base = (group_in_block * kSlots + (w.gl & 1)) * kCoopBanks;

    for (int b = 0; b < kCoopBanks; b++)
        if (w.gl < kSlots) { chain_lds[group_in_block][b][w.gl].phase = 2; chain_lds[group_in_block][b][w.gl].cand = -1; }
    if (w.gl == 0) {
        int *bt = &bank_lds[group_in_block][0][0];
        for (int n = 0; n < kCoopBanks * 6; n++) bt[n] = 0;
    }
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
            const int sv_R[kSlots] = {bk[0], bk[2]}, sv_Q[kSlots] = {bk[1], bk[3]};
            const bool mine = w.gl < kSlots && ((w.gl & 1) ? sv_R[1] : sv_R[0]) > 0;
            const int my_job = job_base + bank;
            // somebody has to walk: whoever needs a result first walks every posted job of the block (see the header)
            const unsigned long long wd0 = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                const uint32_t s = mine ? __hip_atomic_load(&jstate[my_job], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : want;
                if (__all(s == want)) break;
                if (__any(mine && (s & 3u) == 1u)) {
                    uint32_t got = 1;
                    if ((threadIdx.x & 63) == 0) got = atomicCAS(&walk_lock, 0u, 1u);
                    got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                    if (got == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        coop_walk_batch<L>(kp, ws_all, jobs, jstate, done, &stage_lds[0][0][0], cache_lds);
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if ((threadIdx.x & 63) == 0) __hip_atomic_store(&walk_lock, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        continue;
                    }
                }
                __builtin_amdgcn_s_sleep(4);
                if (__builtin_amdgcn_s_memrealtime() - wd0 > 200000000ull) {      // two seconds: never hang the machine
                    if ((threadIdx.x & 63) == 0) atomicOr(cq.band_redos, 1 << 30);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            int ref_steps = 0, query_steps = 0, nst = 0, redo = 0;
            ScoreWalk wk;
            wk.score = 0; wk.pend_gap = 0; wk.open_flag = 0; wk.have_left = 0; wk.left_first_gap = 0;
            if (mine) {
                const CoopDone d = done[my_job];
                wk.load(st[w.gl & 1]);
                wk.score += d.dv;                                        // (open == extend: the gap bookkeeping decides nothing)
                ref_steps = d.ref_steps; query_steps = d.query_steps; nst = d.ref_steps + d.query_steps; redo = d.redo;
            }
#pragma unroll
            for (int h = 0; h < kSlots; h++) {
                if (sv_R[h] > 0) {
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
        // ---- the queues are empty: chains move out of bank 1 into free slots of bank 0, so that what is left of the launch runs
        //      one bank per wave -- a chain then advances once per pass + walk, not once per two passes (with both banks
        //      in use to the end the launch of ecoli10x ran 37 ms where its queues were empty after 25).  Bank 1's chains are at
        //      rest here (their results have just been consumed); a free slot of bank 0 has no job in flight even while bank 0
        //      has, and its next pick finds the chain.
        if (bank == 1 && exhausted) {
            ChainState *st0 = chain_lds[group_in_block][0];
#pragma unroll
            for (int h = 0; h < kSlots; h++) {
                const int ph1 = st[h].phase, f0 = st0[0].phase == 2 ? 0 : st0[1].phase == 2 ? 1 : -1;
                wave_sync();
                if (ph1 != 2 && f0 >= 0 && w.gl == 0) {
                    st0[f0] = st[h];
                    st[h].phase = 2; st[h].cand = -1;
                }
                wave_sync();
            }
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
                    // (bank 1 takes chains to the end: closing it once the launch is down to its classes of fewer than 16 / 24 / 40
                    //  tiles, so that the last chains run one bank per wave, cost 4 / 6 / 9 % with four runs in flight --
                    //  8,223 -> 7,903 / 7,751 / 7,446 GCUPS on ecoli10x, profiles/r05/ab_coop_bank1_throttle.txt.  Nor does it make
                    //  this launch the faster one for a run that has the machine to itself: "lean" waves -- no new chains into the
                    //  other bank of a wave that holds a chain longer than 4..13 sixteenths of the longest one running -- bring
                    //  ecoli10x alone from 37.7 ms to 34.1-35.6 where extend_p16_kernel takes 32.0, and cost pacbio50mb alone
                    //  1-4 %: with twice the chains in flight the queues run dry at 21 ms and twice as many half-done chains are
                    //  left to finish in waves that are no longer full, profiles/r05/ab_coop_lean_waves.txt; closing bank 1 to the
                    //  classes of fewer than 8 ... 48 tiles for such a run: ecoli10x 38.0 -> 32.5-36.3 ms, still behind; pacbio50mb
                    //  alone has no tail to cut -- queues dry at 132.9 of 135.4 ms --, single_run_vs_in_flight_experiments.txt)
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
        load_pair_packed<L::kSlotsPerLane, LANES, typename L::Cols, false>(refs, qfwd, qrc, pt, w.gl, ref8, G::kRefBytes, G::kRow0, q8, G::kTileMax, qb,
                                                                          stage, typename L::Cols{});
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
        uint32_t *wsA = ws_wave + (size_t)bank * coop_bank_words<L>() + (w.g * LANES) * 4;
        uint32_t *wsB = wsA + 64 * 4;
        const uint32_t fin = L::template pass<false>(kc, w.gl, ref16_lane, qb, T_end, tB, wsA, wsB, pt);
        __builtin_amdgcn_s_setprio(3);
        const int v0A = (int)(int16_t)(__shfl(fin, L::fin_lane(pt.Q[0]), LANES) & 0xffffu);
        const int v0B = (int)(int16_t)(__shfl(fin, L::fin_lane(pt.Q[1]), LANES) >> 16);
        GACT_STAMP(t_d);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pointer stores have reached the L2 before a walker is told
        GACT_STAMP(t_e);

        // ---- post the walks of this bank: lane h of the group writes slot h's job, its state word last
        // (jobs of this wave are numbered 1, 2, ... over both banks; the counter sits in the first group's spare LDS word)
        const uint32_t wave_seq = (uint32_t)__builtin_amdgcn_readfirstlane(bank_lds[wave_in_block * kGroupsOfWave][0][5]) + 1u;
        if (w.gl < kSlots) {
            const int h = w.gl;
            if (h ? have[1] : have[0]) {
                const int Rh = h ? pt.R[1] : pt.R[0], Qh = h ? pt.Q[1] : pt.Q[0], sh = h ? pt.shift[1] : pt.shift[0];
                int l0, c0, k0;
                L::walk_start(Rh, Qh, L::tile_tB(tB, sh), l0, c0, k0);
                CoopJob *jb = &jobs[job_base + bank];
                jb->ws_off = (uint32_t)((const char *)(h ? wsB : wsA) - (const char *)ws_all);
                jb->R = Rh; jb->Q = Qh; jb->k0 = k0; jb->v0 = h ? v0B : v0A;
                jb->band_lim = ((kp.band & 0xffff) > 0 && !(h ? pt.full[1] : pt.full[0])) ? (kp.band & 0xffff) - kLinWalkSpan : -1;
                const uint32_t seg_index = (uint32_t)((group_in_block * kCoopBanks + bank) * kStageWords + (2 * h) * kSeg);
                constexpr uint32_t kFrontBits = 16u * StageGeom<L::kSlotsPerLane, LANES>::kFront;
                jb->where = (kFrontBits + (uint32_t)((h ? pt.rp0[1] : pt.rp0[0]) & 15)) |
                            ((kFrontBits + (uint32_t)((h ? pt.qp0[1] : pt.qp0[0]) & 15)) << 8) |
                            ((h ? pt.reverse[1] : pt.reverse[0]) ? 1u << 16 : 0u) | (seg_index << 17);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&jstate[job_base + bank], (wave_seq << 2) | 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        wave_sync();
        if (w.gl == 0) bank_lds[group_in_block][bank][4] = (int)((wave_seq << 2) | 3u);      // what the bank's jobs read when done
        if ((threadIdx.x & 63) == 0) bank_lds[wave_in_block * kGroupsOfWave][0][5] = (int)wave_seq;
        wave_sync();
        // (walking eagerly -- a wave that has just posted walks at once when 16 / 24 / 32 / 48 jobs of the block are pending, so that
        //  nobody waits for a result later -- takes the waiting out of the stamps, 24 % -> 9 % of a wave's time, and changes
        //  nothing: pacbio50mb with four runs in flight 8,433 lazy against 8,211 / 8,347 / 8,362 / 8,406,
        //  profiles/r05/ab_coop_eager_batches.txt.  The machine is bound by issue, not by waves that wait: DESIGN 3.13)
        GACT_STAMP(t_f);
        GACT_ACC(0, t_a, t_b); GACT_ACC(1, t_b, t_c); GACT_ACC(2, t_c, t_d); GACT_ACC(3, t_d, t_e);
        GACT_ACC(4, t_w0, t_a); GACT_ACC(5, t_e, t_f);
#ifdef GACT_STAMPS
        stamp_acc[6] += 1; stamp_acc[7] += (unsigned long long)(T_end - tB + 1);
#endif
        bank ^= 1;
    }
#ifdef GACT_STAMPS
    if ((threadIdx.x & 63) == 0) {
        for (int k = 0; k < 8; k++) atomicAdd(&g_stamps[k], stamp_acc[k]);
        const int wv = blockIdx.x * (kBlockThreads / 64) + wave_in_block;
        if (wv < 4096) {
            g_timeline[4 * wv] = tl_start; g_timeline[4 * wv + 1] = tl_empty;
            g_timeline[4 * wv + 2] = __builtin_amdgcn_s_memrealtime(); g_timeline[4 * wv + 3] = stamp_acc[6];
            g_wave_cycles[wv] = __builtin_amdgcn_s_memtime() - tl_cyc0;
        }
    }
#endif
}

}  // namespace gact
