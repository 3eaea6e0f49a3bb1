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
#include "gact_chain.hpp"

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
// Persistent chain kernel (int32 scores).  Per-group state mirrors the locals of
// GACT() (gact_chain.hpp); every lane of the group carries an identical copy,
// only the traceback runs on one lane and its results are broadcast.
//
// seed_mode: walk every candidate only until its first tile has been consumed
// (first_tile == false) or the chain is over, then hand the chain state to the
// main launch (ChainQueues): the seed launch of the packed main kernel where the
// packed arg-max keys do not fit (seed_p16_kernel otherwise).
template <int C>
__global__ __launch_bounds__(kBlockThreads, 3) void extend_kernel(
    KParams kp, SeqSetDev refs, SeqSetDev qfwd, SeqSetDev qrc,
    const gact_candidate *__restrict__ cands, int first_cand, int n,
    int rc_from, int same_file,
    gact_overlap *__restrict__ out, ChainQueues cq, int seed_mode,
    uint32_t *__restrict__ ws_all)
{
    using G = Geometry<C>;
    __shared__ uint8_t lds[(kBlockThreads / 64) * kGroupsPerWave * G::kGroupLds];
    __shared__ __attribute__((aligned(16))) uint32_t tb_lds[(kBlockThreads / 64) * kGroupsPerWave][kTbScratchWords];

    const WaveCtx w = wave_ctx();
    const int wave_in_block = threadIdx.x >> 6;
    uint8_t *ref_lds_g = lds + (wave_in_block * kGroupsPerWave + w.g) * G::kGroupLds;
    uint8_t *q_lds_g = ref_lds_g + G::kRefLds;
    const uint8_t *ref_lds_lane = ref_lds_g + (kGroup - 1 - w.gl);
    uint32_t *ws = ws_all + (size_t)w.slot * kp.ws_words;
    const bool raw = refs.use_raw | qfwd.use_raw | qrc.use_raw;

    ChainState s;
    s.comp = 0; s.cand = -1; s.phase = 2;
    bool exhausted = false;
    __builtin_amdgcn_s_setprio(3);

    for (;;) {
        // ---- pick the next tile of this group, finishing / fetching candidates on the way
        TilePick pk;
        pk.have = false; pk.R = 0; pk.Q = 0; pk.reverse = false; pk.rp0 = 0; pk.qp0 = 0;
        for (int guard = 0; guard < 3 && !pk.have; guard++) {
            if (s.phase == 2) {
                if (exhausted) break;
                if (!seed_pop(s, cq, w.gl == 0, [](int v) { return __shfl(v, 0, kGroup); }, cands, first_cand, n, rc_from, refs, qfwd,
                              qrc)) { exhausted = true; break; }
            }
            pk = chain_pick(s, kp, same_file, out, w.gl == 0);
            if (!pk.have && seed_mode && w.gl == 0)
                atomicAdd(cq.seed_cells, (unsigned long long)s.cells);     // finished inside the seed launch
        }
        if (!__any(pk.have)) {
            // nobody in this wave has a tile: either all exhausted, or some group
            // still has transitions pending (guard ran out) -- loop again for those
            if (__all(exhausted && s.phase == 2)) break;
            continue;
        }
        GroupTile gt{pk.R, pk.Q, pk.have ? s.first_tile : 0, 0};

        const bool active = gt.R > 0 && gt.Q > 0;
        const WavePlan wp = align_starts(last_step<C>(gt.R, gt.Q),
                                         first_pointer_step<C>(gt.R, gt.Q, kp.early, gt.first), active, gt.shift);
        uint32_t qb[C];
        load_tile<C>(refs, s.comp ? qrc : qfwd, raw, pk.rp0, pk.qp0, gt.R, gt.Q, pk.reverse, w.gl, ref_lds_g,
                     q_lds_g, qb, gt.shift);
        wave_sync();
        const bool any_first = __any(gt.first != 0);

        PassOut po;
        __builtin_amdgcn_s_setprio(0);          // throughput work; the serial sections around it run at priority 3
        if (any_first) dp_pass<C, true>(kp, w.gl, ref_lds_lane, qb, gt, wp.T_end, wp.tB, ws, po);
        else           dp_pass<C, false>(kp, w.gl, ref_lds_lane, qb, gt, wp.T_end, wp.tB, ws, po);
        __builtin_amdgcn_s_setprio(3);
        po.tB -= gt.shift;      // the traceback indexes steps in the tile's own (undelayed) time

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // see align_tiles_kernel

        // ---- consume the tile exactly as gact.cpp:95-133 / :158-194 do
        if (pk.have) {
            s.n_tiles++;
            s.cells += (int64_t)gt.R * gt.Q;
            int i0 = gt.R, j0 = gt.Q;
            bool stop = false;
            if (s.first_tile) {
                i0 = po.bi; j0 = po.bj;
                stop = chain_first_tile(s, kp, gt.R, gt.Q, po.best, po.bi, po.bj);
            }
            int ref_steps = 0, query_steps = 0, nst = 0;
            ScoreWalk wk;
            wk.load(s);
            if (!stop && w.gl == 0) {
                const int l0 = (j0 - 1) / C;
                walk_chain<C, 0>(ws, tb_lds[wave_in_block * kGroupsPerWave + w.g], i0, j0, l0, (j0 - 1) - l0 * C,
                                 i0 + l0 - po.tB, kp.early, ref_lds_g + kGroup + gt.shift, 1, q_lds_g, s.phase, kp,
                                 wk, ref_steps, query_steps, nst);
            }
            chain_advance(s, stop, wk, ref_steps, query_steps, nst, 0);
            if (seed_mode && !s.first_tile) {
                // first tile done: the rest of the chain belongs to the main launch
                if (w.gl == 0) {
                    cq.states[s.cand] = s;
                    const int b = chain_bucket(s, kp);
                    const int slot = atomicAdd(&cq.bucket_count[b], 1);
                    cq.live[(size_t)b * cq.live_stride + slot] = s.cand;
                    atomicAdd(cq.seed_cells, (unsigned long long)s.cells);
                }
                s.phase = 2;
            }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------
// 2-bit packer: 16 bases -> one word; flags[0] |= 1 when a byte is not A/C/G/T
// (or the GPU build's 0..3 recode, darwin.cpp:320-332).  Code order A0 C1 G2 T3
// as ntcoding.h:25-28.  The image follows NtToTwoBit (ntcoding.cpp:57-70) for every
// byte -- lower case decodes like upper case, anything else is A -- because the seed
// filter reads the queries from it; a set holding such bytes is flagged, and GACT then
// compares its raw bytes (align.cpp:134: case matters, N == N) and never this image.
__device__ __forceinline__ uint32_t base_code(uint32_t b, bool &bad)
{
    switch (b) {
        case 'A': case 0: return 0;
        case 'C': case 1: return 1;
        case 'G': case 3: return 2;
        case 'T': case 2: return 3;
        case 'a': bad = true; return 0;
        case 'c': bad = true; return 1;
        case 'g': bad = true; return 2;
        case 't': bad = true; return 3;
        default: bad = true; return 0;
    }
}

// seq_other[r] = 1 for every sequence r that holds such a byte (a word with one looks its sequences up by bisection:
// rare).  A candidate whose two reads are clean is aligned from the 2-bit image even when others of its set are not.
__global__ void pack_kernel(const uint8_t *__restrict__ raw, int64_t n_bases,
                            uint32_t *__restrict__ packed, int64_t n_words, int *__restrict__ flags,
                            const int64_t *__restrict__ offsets, int n_seqs, int32_t *__restrict__ seq_other)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += stride) {
        uint32_t word = 0;
        const int64_t b0 = wi * 16;
        bool bad_here = false;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t p = b0 + k;
            if (p < n_bases) {
                bool b = false;
                word |= base_code(raw[p], b) << (2 * k);
                if (b && seq_other) {
                    int lo = 0, hi = n_seqs - 1;         // last sequence with offsets[r] <= p
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (offsets[mid] <= p) lo = mid; else hi = mid - 1;
                    }
                    seq_other[lo] = 1;
                }
                bad_here |= b;
            }
        }
        bad |= bad_here;
        packed[wi] = word;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}


// ---------------------------------------------------------------------------
// reverse complement of a whole read set (darwin.cpp:110-147): out[off[r] + len-1-i] = comp(in[off[r] + i]);
// flags[0] |= 2 when a byte is none of acgtnACGTN
__global__ void revcomp_kernel(const uint8_t *__restrict__ raw, const int64_t *__restrict__ offsets, int n_seqs,
                               int64_t n_bases, uint8_t *__restrict__ out, int *__restrict__ flags)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_bases; p += stride) {
        int lo = 0, hi = n_seqs - 1;                     // last sequence with offsets[r] <= p
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (offsets[mid] <= p) lo = mid; else hi = mid - 1;
        }
        uint8_t c;
        switch (raw[p]) {
            case 'a': c = 't'; break; case 'A': c = 'T'; break;
            case 'c': c = 'g'; break; case 'C': c = 'G'; break;
            case 'g': c = 'c'; break; case 'G': c = 'C'; break;
            case 't': c = 'a'; break; case 'T': c = 'A'; break;
            case 'n': c = 'n'; break; case 'N': c = 'N'; break;
            default: c = 'N'; bad = true;
        }
        out[offsets[lo] + (offsets[lo + 1] - 1 - p)] = c;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flags, 2);
}

// ---------------------------------------------------------------------------
// Routing by read content: candidates [first, first + n) sorted into two index lists, lists[0 .. n) = both reads plain
// A/C/G/T (aligned from the 2-bit image), lists[n .. 2n) = one of them holds another byte (aligned from the raw bytes,
// align.cpp:134); counts[0], counts[1] their lengths.  The order inside a list is whatever the atomics make it: a
// candidate's record does not depend on where it sits in a list.
__global__ void route_kernel(const gact_candidate *__restrict__ cands, int first, int n, int rc_from,
                             const int32_t *__restrict__ ref_other, const int32_t *__restrict__ qf_other,
                             const int32_t *__restrict__ qr_other, int *__restrict__ lists, int *__restrict__ counts)
{
    const int stride = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63;
    for (int k0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63); k0 < n; k0 += stride) {      // (wave-uniform trip count)
        const int k = k0 + lane;
        const bool in = k < n;
        const int cand = first + (in ? k : 0);
        const gact_candidate c = cands[cand];
        int query_id;
        const int32_t *qo = cand_strand(c, cand, rc_from, query_id) ? qr_other : qf_other;
        const bool dirty = (ref_other && ref_other[c.ref_id]) || (qo && qo[query_id]);
        // one atomic per wave and list
        const unsigned long long m1 = __ballot(in && dirty), m0 = __ballot(in && !dirty);
        int b0 = 0, b1 = 0;
        if (lane == 0) {
            if (m0) b0 = atomicAdd(&counts[0], __popcll(m0));
            if (m1) b1 = atomicAdd(&counts[1], __popcll(m1));
        }
        b0 = __shfl(b0, 0);
        b1 = __shfl(b1, 0);
        const unsigned long long below = (1ull << lane) - 1;
        if (in) lists[dirty ? n + b1 + __popcll(m1 & below) : b0 + __popcll(m0 & below)] = cand;
    }
}

// ---------------------------------------------------------------------------
// Merged runs (gact_engine.hip, the call combiner): the candidate ranges of up to kMaxMerge callers copied into one array,
// each candidate's strand (index >= its caller's rc_from) moved into bit 30 of query_id (kCompInCand).
constexpr int kMaxMerge = 32;
struct MergeSeg { const gact_candidate *src; int first, n, rc_from, base; };
struct MergeSegs { MergeSeg s[kMaxMerge]; int n_segs; };
__global__ void gather_kernel(MergeSegs segs, gact_candidate *__restrict__ out, int total)
{
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        int g = 0;
        while (g + 1 < segs.n_segs && k >= segs.s[g + 1].base) g++;
        const MergeSeg &sg = segs.s[g];
        const int idx = sg.first + (k - sg.base);
        gact_candidate c = sg.src[idx];
        if (idx >= sg.rc_from) c.query_id |= kCompBit;
        out[k] = c;
    }
}

// ---------------------------------------------------------------------------
// Ordered seeding: the candidates of a run sorted by the length class of the chain they can make at most (what
// chain_bucket files a chain under once its first tile is done: bases left of the seed hit + bases right of it, in
// tiles), longest class first -- a counting sort in two launches.  The seed launch then takes them in that order, so
// that the chains which decide how long the run lasts are handed to the main launch first.
__device__ __forceinline__ int order_bucket(const gact_candidate &c, int cand, int rc_from, const int64_t *__restrict__ ref_off,
                                            const int64_t *__restrict__ qf_off, const int64_t *__restrict__ qr_off, int early)
{
    int query_id;
    const int64_t *qo = cand_strand(c, cand, rc_from, query_id) ? qr_off : qf_off;
    const int ref_len = (int)(ref_off[c.ref_id + 1] - ref_off[c.ref_id]);
    const int query_len = (int)(qo[query_id + 1] - qo[query_id]);
    const int rem = imax(0, imin(c.ref_pos, c.query_pos)) + imax(0, imin(ref_len - c.ref_pos, query_len - c.query_pos));
    return kBuckets - 1 - length_class(rem / imax(early, 1));
}
__global__ void order_hist_kernel(const gact_candidate *__restrict__ cands, int first, int n, int rc_from,
                                  const int64_t *__restrict__ ref_off, const int64_t *__restrict__ qf_off,
                                  const int64_t *__restrict__ qr_off, int early, int *__restrict__ hist)
{
    __shared__ int l_cnt[kBuckets];
    if (threadIdx.x < kBuckets) l_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        atomicAdd(&l_cnt[order_bucket(cands[first + k], first + k, rc_from, ref_off, qf_off, qr_off, early)], 1);
    __syncthreads();
    if (threadIdx.x < kBuckets && l_cnt[threadIdx.x]) atomicAdd(&hist[threadIdx.x], l_cnt[threadIdx.x]);
}
// cursor[kBuckets] zeroed by the caller; order[0 .. n) receives candidate indices (first + k), bucket 0 (longest) first
__global__ void order_scatter_kernel(const gact_candidate *__restrict__ cands, int first, int n, int rc_from,
                                     const int64_t *__restrict__ ref_off, const int64_t *__restrict__ qf_off,
                                     const int64_t *__restrict__ qr_off, int early, const int *__restrict__ hist,
                                     int *__restrict__ cursor, int *__restrict__ order)
{
    __shared__ int l_scan[kBuckets], l_cnt[kBuckets], l_base[kBuckets];
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int b = 0; b < kBuckets; b++) { l_scan[b] = acc; acc += hist[b]; }
    }
    const int stride = gridDim.x * blockDim.x;
    // (block-uniform trip count: the barriers inside are reached by every thread)
    for (int k0 = blockIdx.x * blockDim.x; k0 < n; k0 += stride) {
        if (threadIdx.x < kBuckets) l_cnt[threadIdx.x] = 0;
        __syncthreads();
        const int k = k0 + threadIdx.x;
        int b = 0, rank = 0;
        if (k < n) {
            b = order_bucket(cands[first + k], first + k, rc_from, ref_off, qf_off, qr_off, early);
            rank = atomicAdd(&l_cnt[b], 1);
        }
        __syncthreads();
        if (threadIdx.x < kBuckets && l_cnt[threadIdx.x]) l_base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], l_cnt[threadIdx.x]);
        __syncthreads();
        if (k < n) order[l_scan[b] + l_base[b] + rank] = first + k;
    }
}

// ---------------------------------------------------------------------------
// Diagnostic (GACT_HIP_POISON_WS=<seed>): fills the traceback workspace with a seeded pseudo-random pattern in front
// of every launch, so that a walker reading a pointer word its own pass did not store gets garbage that changes
// with the seed instead of whatever an earlier tile left there.  Results must not depend on the seed.
__global__ void poison_kernel(uint32_t *__restrict__ ws, size_t n_words, uint32_t seed)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
        uint32_t x = (uint32_t)i * 0x9E3779B1u + seed;           // one round of a multiplicative hash per word
        x ^= x >> 15; x *= 0x85EBCA77u; x ^= x >> 13; x *= 0xC2B2AE3Du; x ^= x >> 16;
        ws[i] = x;
    }
}

// ---------------------------------------------------------------------------
// VALU issue-rate probe for the instruction class the packed kernels are made of: 16 independent accumulators,
// v_pk_max_i16 and v_pk_add_i16 in rotation, nothing else.  gact_hip_measure_valu_rate launches it with eight waves
// per SIMD: the sustained packed-int16 lane-op rate of this device (tools/issue_probe.hip has the whole table).
__global__ __launch_bounds__(kBlockThreads) void valu_probe_kernel(int iters, int seed, int *__restrict__ sink,
                                                                  unsigned long long *__restrict__ wave_clocks)
{
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = (uint32_t)(seed + k) + threadIdx.x;
    uint32_t inc, lim;
    asm volatile("v_mov_b32 %0, %1" : "=v"(inc) : "s"(seed | 1));
    asm volatile("v_mov_b32 %0, %1" : "=v"(lim) : "s"(seed));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[k]) : "v"(inc));
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[k]) : "v"(lim));
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) r ^= a[k];
    if (r == 0x7fffffffu) sink[0] = (int)r;
    // shader clocks this wave took (the launch as a whole also pays for uneven block placement and its own tail)
    if ((threadIdx.x & 63) == 0) wave_clocks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

}  // namespace gact
