// dsoft_device.hpp -- the D-SOFT seed filter on the device.
//
// Same semantics as the host restatement (host/dsoft.cpp), which is pinned
// candidate by candidate against the reference's SeedPosTable::DSOFT:
//   2-bit coding, non-ACGT -> A                 ntcoding.cpp:59-71,87-103
//   hash32 + (k,w) window minimizers            ntcoding.cpp:77-88,126-182
//   index build, occurrence cap                 seed_pos_table.cpp:46-98
//   DSOFT band counting, candidate emission     seed_pos_table.cpp:100-167
//   candidate -> (chr, ref_pos, query_pos)      darwin.cpp:213-224,532-543
//
// HBM-bound integer work, laid out for a 288 GB device:
//   * index = direct-address table of 4^k end offsets (1 GiB at k = 14, like the
//     reference's index_table_) + the minimizer positions grouped by seed value.
//     Built by counting sort: histogram (atomics) -> exclusive scan -> scatter
//     (the table doubles as the cursor and ends up holding END offsets, which is
//     exactly the reference's convention) -> per-seed insertion sort of the
//     short segments a query can ever read (occurrences <= the cap).
//   * the minimizer de-duplication rule "emit when the window minimum changes,
//     or w positions after the last emission" is sequential in the reference;
//     here: a position starts a run when its minimum differs from the previous
//     position's, and inside a run every w-th position is emitted, so one
//     max-scan of run starts replaces the loop-carried state.
//   * a query strand is one wave: it scans its minimizers in position order,
//     keeps the first num_seeds+1 whose occurrence count passes the cap, and
//     applies their hits seed by seed (lanes = hits of one seed) to a per-wave
//     open-addressing table of diagonal bins in HBM/L2; bins are cleared through
//     a touched-slot list as the reference does with nz_bins.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gact_hip.h"

namespace dsoft {

struct IndexDev {
    const uint32_t *ref2;        // 2-bit padded concatenation of the reference set, 16 bases per word
    const uint32_t *table;       // [4^k] END offset of every seed value in pos
    const uint32_t *pos;         // minimizer positions, ascending inside a seed value
    const int32_t *bin_chr;      // [n_bins_used] sequence a bin belongs to
    const uint32_t *start_bin;   // [n_seqs]
    const int64_t *ref_offsets;  // [n_seqs + 1] of the GACT reference set (for lengths)
    uint32_t ref_len;            // padded concatenation length
    uint32_t n_bins_used;        // bins that belong to a sequence
    uint32_t max_occ;
    int32_t k, w;
    uint32_t bin_size;
    int32_t threshold, num_seeds;
};

__host__ __device__ inline uint32_t two_bit(uint32_t c)
{
    switch (c) {                       // ntcoding.cpp:59-71
        case 'c': case 'C': return 1;
        case 'g': case 'G': return 2;
        case 't': case 'T': return 3;
        default: return 0;
    }
}

// Thomas Wang's integer hash masked to 2k bits, ntcoding.cpp:77-88
__host__ __device__ inline uint32_t hash32(uint32_t key, int k)
{
    const uint32_t m = (1u << (2 * k)) - 1;
    key = (~key + (key << 21)) & m;
    key = key ^ (key >> 24);
    key = ((key + (key << 3)) + (key << 8)) & m;
    key = key ^ (key >> 14);
    key = ((key + (key << 2)) + (key << 4)) & m;
    key = key ^ (key >> 28);
    key = (key + (key << 31)) & m;
    return key;
}

// k bases starting at `pos` of a 2-bit stream whose base 0 sits at absolute base index base0
// (ntcoding.cpp:115-124); bases at or beyond `len` read as A (the zero padding of the last word)
__device__ __forceinline__ uint32_t seed_at(const uint32_t *__restrict__ s, int64_t base0, uint32_t pos, uint32_t len,
                                            int k)
{
    const int64_t a = base0 + pos;
    const int64_t idx = a >> 4;
    const uint32_t shift = (uint32_t)(a & 15);
    const uint64_t concat = ((uint64_t)s[idx + 1] << 32) | s[idx];
    uint32_t v = (uint32_t)(concat >> (2 * shift)) & ((1u << (2 * k)) - 1);
    if (pos + (uint32_t)k > len) {
        const int keep = (int)len - (int)pos;                 // bases that exist
        v = keep > 0 ? v & ((1u << (2 * keep)) - 1) : 0u;
    }
    return v;
}

// window minimum at position p (p >= w-1): min over hash(seed_at(p-i)), i < w
__device__ __forceinline__ uint32_t window_min(const uint32_t *__restrict__ s, int64_t base0, uint32_t p, uint32_t len,
                                               int k, int w)
{
    uint32_t mn = 0xffffffffu;
    for (int i = 0; i < w; i++) mn = min(mn, hash32(seed_at(s, base0, p - (uint32_t)i, len, k), k));
    return mn;
}

constexpr int kMinPerThread = 4;
constexpr int kMinThreads = 256;
constexpr int kMinPerBlock = kMinPerThread * kMinThreads;

// ---------------------------------------------------------------------------
// reference side

// which sequence owns each bin (darwin.cpp:532-543)
__global__ void bin_chr_kernel(const uint32_t *__restrict__ start_bin, int n_seqs, uint32_t n_bins_used,
                               int32_t *__restrict__ bin_chr)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins_used) return;
    int lo = 0, hi = n_seqs - 1;                 // last sequence with start_bin <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start_bin[mid] <= b) lo = mid; else hi = mid - 1;
    }
    bin_chr[b] = lo;
}

// the padded concatenation in 2-bit form: every sequence is followed by 'N' (-> A) up to a whole bin
__global__ void pack_ref_kernel(const uint8_t *__restrict__ raw, const int64_t *__restrict__ offsets,
                                const uint32_t *__restrict__ start_bin, const int32_t *__restrict__ bin_chr,
                                uint32_t bin_size, uint32_t ref_len, uint32_t *__restrict__ out, uint32_t n_words)
{
    const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= n_words) return;
    uint32_t word = 0;
    for (int j = 0; j < 16; j++) {
        const uint32_t p = wi * 16 + (uint32_t)j;
        if (p >= ref_len) break;
        const int32_t chr = bin_chr[p / bin_size];
        const int64_t local = (int64_t)p - (int64_t)start_bin[chr] * bin_size;
        const int64_t len = offsets[chr + 1] - offsets[chr];
        if (local < len) word |= two_bit(raw[offsets[chr] + local]) << (2 * j);
    }
    out[wi] = word;
}

// m(p) for the thread's positions p0..p0+3 and m(p0-1); positions outside [w-1, end) are not evaluated
struct MinQuad {
    uint32_t m[kMinPerThread], m_prev;
};

__device__ __forceinline__ MinQuad min_quad(const uint32_t *__restrict__ s, int64_t base0, uint32_t p0, uint32_t len,
                                            uint32_t end, int k, int w)
{
    MinQuad q;
    q.m_prev = (p0 >= (uint32_t)w && p0 - 1 < end) ? window_min(s, base0, p0 - 1, len, k, w) : 0u;   // last_m starts at 0
#pragma unroll
    for (int i = 0; i < kMinPerThread; i++) {
        const uint32_t p = p0 + (uint32_t)i;
        q.m[i] = (p + 1 >= (uint32_t)w && p < end) ? window_min(s, base0, p, len, k, w) : 0u;
    }
    return q;
}

// last position of each block whose window minimum differs from its predecessor's (-1: none)
__global__ __launch_bounds__(kMinThreads) void ref_lastflag_kernel(const uint32_t *__restrict__ ref2, uint32_t ref_len,
                                                                   uint32_t end, int k, int w,
                                                                   int32_t *__restrict__ lastflag)
{
    __shared__ int32_t red[kMinThreads / 64];
    const uint32_t p0 = blockIdx.x * kMinPerBlock + threadIdx.x * kMinPerThread;
    const MinQuad q = min_quad(ref2, 0, p0, 0xffffffffu, end, k, w);
    int32_t last = -1;
    uint32_t prev = q.m_prev;
#pragma unroll
    for (int i = 0; i < kMinPerThread; i++) {
        const uint32_t p = p0 + (uint32_t)i;
        if (p + 1 >= (uint32_t)w && p < end && q.m[i] != prev) last = (int32_t)p;
        prev = q.m[i];
    }
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = last;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t v = red[0];
        for (int i = 1; i < kMinThreads / 64; i++) v = max(v, red[i]);
        lastflag[blockIdx.x] = v;
    }
}

// carry[b] = last run start in blocks < b (-1: none); one block
__global__ void carry_kernel(const int32_t *__restrict__ lastflag, int n_blocks, int32_t *__restrict__ carry)
{
    __shared__ int32_t part[1024];
    const int per = (n_blocks + blockDim.x - 1) / blockDim.x;
    const int lo = min(n_blocks, (int)threadIdx.x * per), hi = min(n_blocks, lo + per);
    int32_t v = -1;
    for (int i = lo; i < hi; i++) v = max(v, lastflag[i]);
    part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t run = -1;
        for (unsigned i = 0; i < blockDim.x; i++) { const int32_t t = part[i]; part[i] = run; run = max(run, t); }
    }
    __syncthreads();
    v = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { carry[i] = v; v = max(v, lastflag[i]); }
}

// MODE 0: count[seed]++ ; MODE 1: pos[cursor[seed]++] = p   (cursor = the scanned table)
template <int MODE>
__global__ __launch_bounds__(kMinThreads) void ref_emit_kernel(const uint32_t *__restrict__ ref2, uint32_t end, int k,
                                                               int w, const int32_t *__restrict__ carry,
                                                               uint32_t *__restrict__ table,
                                                               uint32_t *__restrict__ pos,
                                                               unsigned long long *__restrict__ n_emitted)
{
    __shared__ int32_t wave_last[kMinThreads / 64];
    const uint32_t p0 = blockIdx.x * kMinPerBlock + threadIdx.x * kMinPerThread;
    const MinQuad q = min_quad(ref2, 0, p0, 0xffffffffu, end, k, w);
    bool flag[kMinPerThread];
    int32_t last = -1;
    uint32_t prev = q.m_prev;
#pragma unroll
    for (int i = 0; i < kMinPerThread; i++) {
        const uint32_t p = p0 + (uint32_t)i;
        flag[i] = (p + 1 >= (uint32_t)w && p < end && q.m[i] != prev);
        if (flag[i]) last = (int32_t)p;
        prev = q.m[i];
    }
    // exclusive max-scan of `last` over the block's threads
    int32_t incl = last;
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl = max(incl, t);
    }
    if ((threadIdx.x & 63) == 63) wave_last[threadIdx.x >> 6] = incl;
    __syncthreads();
    int32_t before = carry[blockIdx.x];
    for (int i = 0; i < (int)(threadIdx.x >> 6); i++) before = max(before, wave_last[i]);
    int32_t excl = __shfl_up(incl, 1);
    if ((threadIdx.x & 63) == 0) excl = -1;
    int32_t run = max(before, excl);           // last run start before this thread's first position
    int emitted = 0;
#pragma unroll
    for (int i = 0; i < kMinPerThread; i++) {
        const uint32_t p = p0 + (uint32_t)i;
        if (p + 1 < (uint32_t)w || p >= end) continue;
        bool emit = flag[i];
        if (flag[i]) run = (int32_t)p;
        else {
            const uint32_t rs = run < 0 ? 0u : (uint32_t)run;      // no run yet: last_p = 0 (ntcoding.cpp:131)
            emit = p > rs && (p - rs) % (uint32_t)w == 0;
        }
        if (emit) {
            emitted++;
            if (MODE == 0) atomicAdd(&table[q.m[i]], 1u);
            else pos[atomicAdd(&table[q.m[i]], 1u)] = p;
        }
    }
    if (MODE == 0) {
        for (int o = 32; o > 0; o >>= 1) emitted += __shfl_xor(emitted, o);
        if ((threadIdx.x & 63) == 0 && emitted) atomicAdd(n_emitted, (unsigned long long)emitted);
    }
}

// ---------------------------------------------------------------------------
// exclusive scan of the 4^k counters, three phases, kScanPerBlock elements per block
constexpr int kScanThreads = 256;
constexpr int kScanPerThread = 16;
constexpr int kScanPerBlock = kScanThreads * kScanPerThread;

__global__ __launch_bounds__(kScanThreads) void scan_sums_kernel(const uint32_t *__restrict__ data, uint64_t n,
                                                                 uint32_t *__restrict__ sums)
{
    __shared__ uint32_t red[kScanThreads / 64];
    const uint64_t base = (uint64_t)blockIdx.x * kScanPerBlock;
    uint32_t v = 0;
    for (int i = 0; i < kScanPerThread; i++) {
        const uint64_t idx = base + (uint64_t)i * kScanThreads + threadIdx.x;
        if (idx < n) v += data[idx];
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int i = 0; i < kScanThreads / 64; i++) t += red[i];
        sums[blockIdx.x] = t;
    }
}

// in-place exclusive scan of `sums` (one block)
__global__ void scan_top_kernel(uint32_t *__restrict__ sums, int n)
{
    __shared__ uint32_t part[1024];
    const int per = (n + blockDim.x - 1) / blockDim.x;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    uint32_t v = 0;
    for (int i = lo; i < hi; i++) v += sums[i];
    part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (unsigned i = 0; i < blockDim.x; i++) { const uint32_t t = part[i]; part[i] = run; run += t; }
    }
    __syncthreads();
    v = part[threadIdx.x];
    for (int i = lo; i < hi; i++) { const uint32_t t = sums[i]; sums[i] = v; v += t; }
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(uint32_t *__restrict__ data, uint64_t n,
                                                                  const uint32_t *__restrict__ sums)
{
    __shared__ uint32_t wave_tot[kScanThreads / 64];
    // thread t owns kScanPerThread consecutive elements
    const uint64_t base = (uint64_t)blockIdx.x * kScanPerBlock + (uint64_t)threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread], tot = 0;
#pragma unroll
    for (int i = 0; i < kScanPerThread; i++) {
        v[i] = (base + i < n) ? data[base + i] : 0u;
        tot += v[i];
    }
    uint32_t incl = tot;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = sums[blockIdx.x] + incl - tot;
    for (int i = 0; i < (int)(threadIdx.x >> 6); i++) run += wave_tot[i];
#pragma unroll
    for (int i = 0; i < kScanPerThread; i++) {
        if (base + i < n) data[base + i] = run;
        run += v[i];
    }
}

// positions of one seed value arrive in atomic order: sort the segments a query may read
__global__ void sort_segments_kernel(const uint32_t *__restrict__ table, uint64_t n_seeds, uint32_t max_occ,
                                     uint32_t *__restrict__ pos)
{
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seeds) return;
    const uint32_t lo = s ? table[s - 1] : 0u, hi = table[s];
    const uint32_t n = hi - lo;
    if (n < 2 || n > max_occ) return;
    for (uint32_t i = 1; i < n; i++) {
        const uint32_t v = pos[lo + i];
        uint32_t j = i;
        while (j > 0 && pos[lo + j - 1] > v) { pos[lo + j] = pos[lo + j - 1]; j--; }
        pos[lo + j] = v;
    }
}

// ---------------------------------------------------------------------------
// query side: one wave per (query, strand)

struct QuerySetDev {
    const uint32_t *packed;      // the engine's 2-bit concatenation of the set (same A0 C1 G2 T3 code)
    const int64_t *offsets;
};

// band counter of one diagonal bin: low word = bin + 1 (0: empty), high word = count << 24 | last query offset
typedef unsigned long long BinSlot;

constexpr int kQueryChunk = 256;         // positions per pass of the wave (4 per lane)

// the table is written and re-read by different lanes of one wave: keep the accesses out of the L1
__device__ __forceinline__ BinSlot ld_slot(const BinSlot *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_slot(BinSlot *p, BinSlot v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_u32(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u32(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct QueryScratch {
    BinSlot *tables;             // [blocks][table_mask + 1]
    uint32_t table_mask;
    uint32_t *touched;           // [blocks][table_mask + 1] slots to clear after the task
    uint2 *staged;               // [blocks][staged_cap] (hit, offset) of the task's candidates, emission order
    uint32_t staged_cap;
};

struct QueryOut {
    int32_t *counts;             // [tasks]
    int64_t *task_base;          // [tasks] where the task's candidates sit in temp
    unsigned long long *temp_used;
    gact_candidate *temp;
    int64_t temp_cap;
    int *overflow;               // temp was too small: counts are valid, candidates are not
};

// Tasks [0, n_queries) are the forward strands of queries first_query.., [n_queries, 2 n_queries) their reverse
// complements.  Candidates of a task land contiguously in `temp` (task order there is arbitrary);
// gather_kernel puts them in task order.
__global__ __launch_bounds__(64) void query_kernel(IndexDev ix, QuerySetDev qfwd, QuerySetDev qrc, int first_query,
                                                   int n_queries, int *__restrict__ next_task, QueryScratch sc,
                                                   QueryOut qo)
{
    __shared__ uint32_t seed_off[kQueryChunk], seed_lo[kQueryChunk], seed_n[kQueryChunk];
    const int lane = threadIdx.x;
    BinSlot *table = sc.tables + (size_t)blockIdx.x * (sc.table_mask + 1);
    uint32_t *touched = sc.touched + (size_t)blockIdx.x * (sc.table_mask + 1);
    uint2 *staged = sc.staged + (size_t)blockIdx.x * sc.staged_cap;
    const uint32_t table_mask = sc.table_mask;
    const int k = ix.k, w = ix.w;
    const uint64_t below = (1ull << lane) - 1;

    for (;;) {
        int task = 0;
        if (lane == 0) task = atomicAdd(next_task, 1);
        task = __shfl(task, 0);
        if (task >= 2 * n_queries) break;
        const bool rc = task >= n_queries;
        const int qi = first_query + (rc ? task - n_queries : task);
        const QuerySetDev &qs = rc ? qrc : qfwd;
        const int64_t base0 = qs.offsets[qi];
        const uint32_t len = (uint32_t)(qs.offsets[qi + 1] - base0);
        const uint32_t s_len = (len + 15) / 16;                                  // seed_pos_table.cpp:108
        const uint32_t end = (16 * s_len >= (uint32_t)(k + w)) ? 16 * s_len - (uint32_t)k - (uint32_t)w : 0u;

        int n_used = 0;              // seeds applied so far (the reference's num_seeds counter)
        int n_cand = 0, n_touched = 0;
        int32_t run = -1;            // last run start before the chunk
        bool done = false;

        for (uint32_t c0 = 0; c0 < end && !done; c0 += kQueryChunk) {
            // ---- minimizers of positions c0 .. c0+255, in position order
            const uint32_t p0 = c0 + (uint32_t)lane * kMinPerThread;
            const MinQuad q = min_quad(qs.packed, base0, p0, len, end, k, w);
            bool flag[kMinPerThread];
            int32_t last = -1;
            uint32_t prev = q.m_prev;
#pragma unroll
            for (int i = 0; i < kMinPerThread; i++) {
                const uint32_t p = p0 + (uint32_t)i;
                flag[i] = (p + 1 >= (uint32_t)w && p < end && q.m[i] != prev);
                if (flag[i]) last = (int32_t)p;
                prev = q.m[i];
            }
            int32_t incl = last;
            for (int o = 1; o < 64; o <<= 1) {
                const int32_t t = __shfl_up(incl, o);
                if (lane >= o) incl = max(incl, t);
            }
            int32_t excl = __shfl_up(incl, 1);
            if (lane == 0) excl = -1;
            int32_t r = max(run, excl);
            run = max(run, __shfl(incl, 63));

            // emitted minimizers of this lane that pass the occurrence cap (seed_pos_table.cpp:118-124)
            uint32_t e_off[kMinPerThread], e_lo[kMinPerThread], e_n[kMinPerThread];
            int n_mine = 0;
#pragma unroll
            for (int i = 0; i < kMinPerThread; i++) {
                const uint32_t p = p0 + (uint32_t)i;
                if (p + 1 < (uint32_t)w || p >= end) continue;
                bool emit = flag[i];
                if (flag[i]) r = (int32_t)p;
                else {
                    const uint32_t rs = r < 0 ? 0u : (uint32_t)r;
                    emit = p > rs && (p - rs) % (uint32_t)w == 0;
                }
                if (emit) {
                    const uint32_t idx = q.m[i];
                    const uint32_t lo = idx ? ix.table[idx - 1] : 0u, hi = ix.table[idx];
                    if (hi - lo <= ix.max_occ) { e_off[n_mine] = p; e_lo[n_mine] = lo; e_n[n_mine] = hi - lo; n_mine++; }
                }
            }
            int incl_n = n_mine;
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl_n, o);
                if (lane >= o) incl_n += t;
            }
            const int chunk_total = __shfl(incl_n, 63);
            int slot = incl_n - n_mine;
            for (int i = 0; i < n_mine; i++, slot++) { seed_off[slot] = e_off[i]; seed_lo[slot] = e_lo[i]; seed_n[slot] = e_n[i]; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();

            // ---- apply the seeds of the chunk in order (seed_pos_table.cpp:125-152); the hits of the next seed
            //      are fetched while the current one walks the table
            uint32_t hit_next = (chunk_total > 0 && (uint32_t)lane < seed_n[0]) ? ix.pos[seed_lo[0] + lane] : 0u;
            for (int si = 0; si < chunk_total; si++) {
                if (n_used > ix.num_seeds) { done = true; break; }                // :125-127 (N+1 seeds are used)
                n_used++;
                const uint32_t offset = seed_off[si], lo = seed_lo[si], n = seed_n[si];
                uint32_t hit_first = hit_next;
                hit_next = (si + 1 < chunk_total && (uint32_t)lane < seed_n[si + 1]) ? ix.pos[seed_lo[si + 1] + lane] : 0u;
                uint32_t prev_bin = 0xffffffffu;
                for (uint32_t h0 = 0; h0 < n; h0 += 64) {
                    const bool in = h0 + (uint32_t)lane < n;
                    const uint32_t hit = h0 == 0 ? hit_first : (in ? ix.pos[lo + h0 + lane] : 0u);
                    const bool ok = in && hit >= offset;                          // :132
                    const uint32_t bin = ok ? (hit - offset) / ix.bin_size : 0xfffffffeu - (uint32_t)lane;
                    // hits ascend, so equal bins sit in adjacent lanes: only the first hit of a bin acts, the
                    // others would add offset - last_offset = 0 (:137)
                    uint32_t left = __shfl_up(bin, 1);
                    if (lane == 0) left = prev_bin;
                    const bool lead = ok && bin != left;
                    prev_bin = __shfl(bin, 63);
                    bool crossed = false, fresh = false;
                    uint32_t where = 0;
                    if (lead) {
                        uint32_t hslot = (bin * 2654435761u) & table_mask;
                        for (;;) {
                            const BinSlot cur = ld_slot(&table[hslot]);
                            const uint32_t key = (uint32_t)cur;
                            if (key == bin + 1) {
                                const uint32_t v = (uint32_t)(cur >> 32);
                                const uint32_t cnt = v >> 24, last_off = v & 0xffffffu;
                                if (cnt < (uint32_t)ix.threshold) {
                                    const uint32_t nc = (offset - last_off > (uint32_t)k) ? cnt + (uint32_t)k
                                                                                          : cnt + (offset - last_off);   // :137
                                    st_slot(&table[hslot], ((BinSlot)((nc << 24) | offset) << 32) | key);
                                    crossed = nc >= (uint32_t)ix.threshold;
                                }
                                break;
                            }
                            if (key == 0) {
                                // a new bin: count = k (:137 with curr_count == 0)
                                BinSlot expect = 0;
                                const BinSlot want = ((BinSlot)(((uint32_t)k << 24) | offset) << 32) | (bin + 1);
                                if (__hip_atomic_compare_exchange_strong(&table[hslot], &expect, want, __ATOMIC_RELAXED,
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                                    fresh = true;
                                    crossed = k >= ix.threshold;
                                    break;
                                }
                                continue;            // another lane's bin took the slot meanwhile: look again
                            }
                            hslot = (hslot + 1) & table_mask;
                        }
                        where = hslot;
                    }
                    // ordered append of the touched slots and of the candidates
                    const uint64_t fm = __ballot(fresh), cm = __ballot(crossed);
                    if (fresh) st_u32(&touched[n_touched + __popcll(fm & below)], where);
                    n_touched += __popcll(fm);
                    if (crossed) {
                        const uint32_t at = (uint32_t)n_cand + (uint32_t)__popcll(cm & below);
                        if (at < sc.staged_cap) {
                            st_u32(&staged[at].x, hit);
                            st_u32(&staged[at].y, offset);
                        }
                    }
                    n_cand += __popcll(cm);
                    // the next seed may read what this one wrote
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- hand the task's candidates over (darwin.cpp:215-224: concatenated coordinate ->
        //      (sequence, position in it), clamped) and clear the touched bins (seed_pos_table.cpp:155-158)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long tbase = 0;
        if (lane == 0) {
            tbase = (long long)atomicAdd(qo.temp_used, (unsigned long long)n_cand);
            qo.counts[task] = n_cand;
            qo.task_base[task] = tbase;
            if (tbase + n_cand > qo.temp_cap || (uint32_t)n_cand > sc.staged_cap) *qo.overflow = 1;
        }
        tbase = __shfl(tbase, 0);
        if (tbase + n_cand <= qo.temp_cap && (uint32_t)n_cand <= sc.staged_cap) {
            for (int i = lane; i < n_cand; i += 64) {
                const uint32_t hit = ld_u32(&staged[i].x), offset = ld_u32(&staged[i].y);
                const uint32_t rbin = hit / ix.bin_size;
                const int32_t chr = rbin < ix.n_bins_used ? ix.bin_chr[rbin] : 0;
                int64_t ref_pos = (int64_t)hit - (int64_t)ix.start_bin[chr] * ix.bin_size;
                const int64_t rl = ix.ref_offsets[chr + 1] - ix.ref_offsets[chr];
                if (ref_pos > rl) ref_pos = rl;
                gact_candidate cd;
                cd.ref_id = chr; cd.query_id = qi; cd.ref_pos = (int32_t)ref_pos; cd.query_pos = (int32_t)offset;
                qo.temp[tbase + i] = cd;
            }
        }
        for (int i = lane; i < n_touched; i += 64) st_slot(&table[ld_u32(&touched[i])], 0ull);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// temp (task order arbitrary) -> out (task order): one wave per task
__global__ __launch_bounds__(64) void gather_kernel(const gact_candidate *__restrict__ temp,
                                                    const int64_t *__restrict__ task_base,
                                                    const int32_t *__restrict__ counts,
                                                    const int64_t *__restrict__ out_base, int n_tasks,
                                                    gact_candidate *__restrict__ out)
{
    for (int t = blockIdx.x; t < n_tasks; t += gridDim.x) {
        const int64_t from = task_base[t], to = out_base[t];
        const int n = counts[t];
        for (int i = threadIdx.x; i < n; i += 64) out[to + i] = temp[from + i];
    }
}

}  // namespace dsoft
