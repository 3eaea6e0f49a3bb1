// gact_engine.hip -- host side of the C-ABI declared in include/gact_hip.h.
//
// Owns all device memory: the resident read sets (2-bit packed + raw bytes),
// per-slot streams, descriptor/result buffers and the per-group traceback
// workspace.  Stands where cuda_host.cu stands in the reference (GPU_init,
// Align_Batch_GPU, GPU_close) and additionally runs the whole GACT_Batch
// state machine (gact.cpp:231-560) on the device.
//
// There is deliberately no CPU path in this file.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>           // types of the RCCL entry points gact_gather.hpp looks up at run time (nothing is linked)

#include <cerrno>
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "gact_hip.h"
#include "gact_kernels.hpp"
#include "gact_p16.hpp"
#include "gact_p16s.hpp"
#include "gact_lin.hpp"
#include "gact_aff.hpp"
#include "gact_roles.hpp"
#include "gact_coop.hpp"
#include "gact_policy.hpp"
#include "gact_big.hpp"
#include "dsoft_device.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t err__ = (expr);                                                        \
        if (err__ != hipSuccess)                                                          \
            return fail(GACT_HIP_EDEVICE, "%s failed at %s:%d: %s", #expr, __FILE__,      \
                        __LINE__, hipGetErrorString(err__));                              \
    } while (0)

struct SeqSet {
    uint32_t *d_packed = nullptr;
    uint8_t *d_raw = nullptr;
    int64_t *d_offsets = nullptr;
    int32_t *d_other = nullptr;   // per sequence: holds a byte other than A/C/G/T
    std::vector<int64_t> h_offsets;
    std::vector<int32_t> h_other;     // host copy of d_other (empty: none of the set's sequences does)
    int32_t n = 0;
    int64_t total = 0;
    int64_t max_len = 0;      // longest sequence of the set
    bool has_other = false;   // holds bytes other than A/C/G/T
    size_t cap_bases = 0, cap_seqs = 0;

    gact::SeqSetDev dev(bool use_raw) const
    {
        gact::SeqSetDev d;
        d.packed = d_packed; d.raw = d_raw; d.offsets = d_offsets; d.n = n;
        d.use_raw = use_raw ? 1 : 0;
        d.other = has_other ? d_other : nullptr;
        return d;
    }
    // The tile loaders issue their loads unpredicated (idle slots read position 0 of
    // whichever set their stale state names), so a set that was never uploaded must
    // still point at readable memory: it borrows another set's buffers.
    gact::SeqSetDev dev_or(bool use_raw, const SeqSet &fallback) const
    {
        return d_raw ? dev(use_raw) : fallback.dev(use_raw);
    }
    void release()
    {
        if (d_packed) (void)hipFree(d_packed);
        if (d_raw) (void)hipFree(d_raw);
        if (d_offsets) (void)hipFree(d_offsets);
        if (d_other) (void)hipFree(d_other);
        d_packed = nullptr; d_raw = nullptr; d_offsets = nullptr; d_other = nullptr;
        cap_bases = cap_seqs = 0; n = 0; total = 0;
    }
};

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = std::max(n, (size_t)1024);
        if (hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// The figures of ONE merged launch: events and counters of its own, recorded / copied behind that launch and
// never touched again, so that a caller that asks for its run's statistics after its fetch reads them whatever the merge
// slot is doing by then (the slot's own events and counters are re-recorded as soon as its arrays are free).  Records are
// pooled per engine; a caller's slot keeps the one of its last merged run alive.
struct LaunchStats {
    hipEvent_t ev0 = nullptr, ev_mid = nullptr, ev1 = nullptr, ev_done = nullptr;
    int *h_counter = nullptr;            // pinned: 2 x kCounterInts ints (the lane's own, the side lane's)
    bool two_phase = false, wide = false, lin = false, aff = false, overlapped = false, lane = false, side_used = false;
    int routed_raw = 0, roles = 0;
};

struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;
    bool timed = false, two_phase = false, wide = false, lin = false, aff = false;
    DevBuf<gact_tile> tiles;
    DevBuf<gact_tile_result> results;
    DevBuf<uint8_t> states;
    DevBuf<gact_candidate> cands;
    size_t n_cands = 0;                  // candidates the slot really holds (uploaded, or produced by the device filter)
    std::vector<gact_candidate> h_cands; // host copy of an uploaded list, for the strand-aware position check at run time;
                                         // empty for lists the device filter made (valid by construction)
    int64_t checked_key[4] = {-1, -1, -1, -1};   // {first, n, rc_from, sets_epoch} of the last range that passed the check
    DevBuf<gact_overlap> overlaps;
    int *d_counter = nullptr;            // see queues()
    DevBuf<int> live;
    DevBuf<int> deferred;                // route_kernel's two candidate lists (launch_extend), n entries each
    DevBuf<int> order;                   // ordered seeding: the run's candidates by length class, longest first (run_overlapped)
    hipStream_t aux_stream = nullptr;    // overlapped seeding: the second seed launch and the main launch behind it
    hipEvent_t aux_ev_a = nullptr, aux_ev_b = nullptr;
    long merge_gen = -1;                 // which merged launch carried this slot's last run (Combiner::merged_launches then): its peers have the same
    long merge_prev_gen = -1;            // ... and the one before that (two halves of a group that came apart have it in common)
    bool overlapped = false;             // the last run seeded while its main launch was running
    bool lane = false;                   // ... had a critical lane (a wide main launch beside the split one)
    int roles = 0;                       // ... ran its main launch(es) with DP waves and walker waves (1, gact_roles.hpp) / cooperative walks (2, gact_coop.hpp)
    // call combiner (Combiner below; all under its mutex)
    std::thread::id last_thread;         // who called last for this slot, and when: is a run from it likely soon?
    std::chrono::steady_clock::time_point last_call{};
    bool in_flight = false;              // a run has been launched and not been fetched / waited for
    int merged_into = -1;                // index of the merge slot the last run of this slot was part of (-1: its own launches)
    int merged_callers = 1;              // how many callers' runs that launch carried
    std::shared_ptr<LaunchStats> merged_stats;   // ... and that launch's own figures (written by the leader before the run counts as launched)
    hipEvent_t ev_ready = nullptr;       // candidates of this slot are in place (recorded before a merged gather)
    int routed_raw = 0;                  // how many candidates the last run aligned from raw bytes beside the 2-bit launches
    DevBuf<gact::ChainState> chain_states;
    uint32_t *d_ws = nullptr;
    int *d_flags = nullptr;
    // side lane: queues, workspace and stream of the raw-byte launches that run BESIDE the 2-bit launches of a routed
    // run (launch_extend); made on first use
    hipStream_t side_stream = nullptr;
    hipEvent_t side_go = nullptr;          // route_kernel has run: the side lane's launches may read its lists
    hipEvent_t side_done = nullptr;
    int *side_counter = nullptr;
    DevBuf<int> side_live;
    uint32_t *side_ws = nullptr;
    int side_blocks = 0;                 // grid the side workspace is sized for
    bool side_used = false;              // the last run had launches on it
    int stream_index = -1;              // of `stream` in the process's pool (StreamPool)
    int64_t routed_key[4] = {-1, -1, -1, -1};   // (first, n, rc_from, sets_epoch) routed_host was counted for
    int routed_host[2] = {0, 0};        // candidates of that range whose reads are plain A/C/G/T / the rest (launch_extend)
    gact_candidate *h_stage = nullptr;  // pinned staging for candidates_upload: hipMemcpyAsync from the caller's pageable array
    size_t h_stage_cap = 0;             // has the runtime pin those pages first, and with eight feeder threads at it at once
                                        // that call took 8 ms for some of them (profiles/r04/upload_trace_*.txt)
    gact_overlap *h_records = nullptr;  // pinned staging for candidates_fetch (pageable D2H is staged by the runtime
    size_t h_records_cap = 0;           // in small chunks: 0.2-0.9 ms for 3.7 MB; pinned + memcpy: 0.25 ms)
    // a caller-owned output buffer page-locked on request (gact_hip_register_output): fetches whose destination lies
    // inside it go straight there, no staging copy.  Never registered behind the caller's back.
    void *reg_out = nullptr;
    size_t reg_bytes = 0;
    int64_t cands_epoch = -1;           // sets_epoch at which the device filter made this slot's list (-1: uploaded list)
    SeqSet inline_ref, inline_query;   // Align_Batch_GPU-style inline tiles
};

// device-side D-SOFT filter (dsoft_device.hpp)
struct DsoftState {
    bool built = false;
    gact_dsoft_params p{};
    uint32_t ref_len = 0, n_bins_used = 0, max_occ = 0;
    uint64_t n_table = 0;                 // 4^k
    int64_t n_min = 0;
    uint32_t *d_ref2 = nullptr, *d_table = nullptr, *d_pos = nullptr, *d_start_bin = nullptr;
    int32_t *d_bin_chr = nullptr;
    // query scratch
    int q_blocks = 0;
    uint32_t table_mask = 0, staged_cap = 0;
    dsoft::BinSlot *d_tables = nullptr;
    uint32_t *d_touched = nullptr;
    uint2 *d_staged = nullptr;
    int *d_next = nullptr;              // [0] task counter, [1] overflow flag, [2..3] temp entries used (u64)
    DevBuf<int32_t> counts;
    DevBuf<int64_t> task_base, out_base;
    DevBuf<gact_candidate> temp;

    void release_index()
    {
        for (void *q : {(void *)d_ref2, (void *)d_table, (void *)d_pos, (void *)d_start_bin, (void *)d_bin_chr})
            if (q) (void)hipFree(q);
        d_ref2 = d_table = d_pos = d_start_bin = nullptr; d_bin_chr = nullptr;
        built = false;
    }
    void release_scratch()
    {
        if (d_tables) (void)hipFree(d_tables);
        if (d_touched) (void)hipFree(d_touched);
        if (d_staged) (void)hipFree(d_staged);
        if (d_next) (void)hipFree(d_next);
        d_tables = nullptr; d_touched = nullptr; d_staged = nullptr; d_next = nullptr; q_blocks = 0;
        counts.release(); task_base.release(); out_base.release(); temp.release();
    }
};

}  // namespace

// ---------------------------------------------------------------------------
// Every switch this library reads, in ONE table: INTEGRATION.md 7 prints it, gact_hip_options_describe
// returns it, tests/test_cabi.py keeps the three in step.  A switch is read from its environment variable once, in
// gact_hip_create (`when` c), and -- where it is safe on a live engine -- also taken by gact_hip_set_option under its name
// (`when` l).  Classes: k = which kernels run (the tests reach every kernel family through these; no record depends on any
// of them), s = scheduling, d = diagnostic / tuning: read only by builds with -DGACT_EXPERIMENTS (gact_hip_create says so on
// stderr), ignored by the default build.
struct OptionDef { const char *name, *env; char when, klass; const char *doc; };
static const OptionDef kOptions[] = {
    {"force_int32", "GACT_HIP_FORCE_INT32", 'c', 'k', "the int32 chain kernel alone (one launch), whatever the scoring allows"},
    {"force_int32_seed", "GACT_HIP_FORCE_INT32_SEED", 'c', 'k', "int32 seed launch in front of the packed main launch"},
    {"force_uniform", "GACT_HIP_FORCE_UNIFORM", 'c', 'k', "uniform instead of split (two-region) column layout"},
    {"force_wide", "GACT_HIP_FORCE_WIDE", 'c', 'k', "the wide layout (32 lanes per tile pair) for every main launch"},
    {"no_wide", "GACT_HIP_NO_WIDE", 'c', 'k', "never the wide layout"},
    {"no_tagged", "GACT_HIP_NO_TAGGED", 'c', 'k', "explicit pointer comparisons instead of tagged scores"},
    {"no_lin", "GACT_HIP_NO_LIN", 'c', 'k', "the affine passes also for linear scorings (gact_lin.hpp off)"},
    {"no_aff", "GACT_HIP_NO_AFF", 'c', 'k', "round 1's tagged affine pass instead of the drifted one (gact_aff.hpp off)"},
    {"coop", "GACT_HIP_COOP", 'l', 'k', "the split linear-gap main launch with two banks of tiles per wave and cooperative, batched traceback walks (gact_coop.hpp): 1 always, 0 never, 2 / unset where throughput bounds the launch"},
    {"roles", "GACT_HIP_ROLES", 'l', 'k', "1: the split linear-gap main launch as DP waves + walker waves (gact_roles.hpp; default 0: measured no faster, DESIGN 3.13)"},
    {"no_routing", "GACT_HIP_NO_ROUTING", 'c', 'k', "a set with a non-ACGT byte moves the whole launch onto the raw-byte kernels"},
    {"no_side_lane", "GACT_HIP_NO_SIDE_LANE", 'c', 's', "routed raw-byte launches after the 2-bit ones instead of beside them"},
    {"band", "GACT_HIP_BAND", 'c', 'k', "width of the stored pointer band in columns (default 48; 0: the whole window)"},
    {"poison_ws", "GACT_HIP_POISON_WS", 'c', 'k', "<seed>: seeded garbage over the traceback workspace before every launch (tests)"},
    {"no_overlap", "GACT_HIP_NO_OVERLAP", 'c', 's', "seed launch, then one main launch, always; live: overlap_seed"},
    {"overlap_seed", nullptr, 'l', 's', "1 / 0: ordered, overlapped seeding of large runs on an idle engine"},
    {"shared_twelfths", "GACT_HIP_SHARED_TWELFTHS", 'l', 's', "<n>: a linear-gap main launch that shares the machine and has more chains than two thirds of the resident tile slots hold takes n twelfths of the resident blocks (default 6)"},
    {"lone_lane", "GACT_HIP_LONE_LANE", 'l', 's', "<n>: a run of half as many to as many chains as there are resident tile slots, alone on the machine, runs as one block per CU of two kinds: n wide blocks for its longest chains, split blocks with the look-ahead walker on the other CUs (-n: without it); 0 (default): all wide, two blocks per CU"},
    {"overlap_big", "GACT_HIP_OVERLAP_BIG", 'l', 's', "1 / 0 (default): ... also of runs of more than four chains per resident tile slot (seed launch A takes the longest eighth, B the rest beside main launch 1; +1.3 % on pacbio50mb alone)"},
    {"no_crit_lane", "GACT_HIP_NO_CRIT_LANE", 'c', 's', "no wide launch beside the split one for runs of 1-1.5 chains per tile slot"},
    {"crit_lane_always", "GACT_HIP_CRIT_LANE_ALWAYS", 'c', 's', "... also for runs of up to four chains per tile slot"},
    {"no_shared_hint", "GACT_HIP_NO_SHARED_HINT", 'c', 's', "a launch made while another slot runs may take the wide layout"},
    {"runs_in_flight", nullptr, 'l', 's', "1: the caller keeps several runs in flight (every launch takes the throughput layout)"},
    {"no_combine", "GACT_HIP_NO_COMBINE", 'c', 's', "runs of several threads are never merged into one launch; live: combine"},
    {"combine", nullptr, 'l', 's', "1 / 0: the call combiner"},
    {"combine_window_us", "GACT_HIP_COMBINE_US", 'l', 's', "how long a leader waits for the other feeder threads (default 1000)"},
    {"sdma_copies", "GACT_HIP_SDMA_COPIES", 'c', 's', "candidate lists and records through hipMemcpyAsync instead of a copy kernel"},
    {"dsoft_temp_cap", "GACT_HIP_DSOFT_TEMP_CAP", 'c', 'k', "initial size of the device filter's candidate staging area (tests: forces the regrow path)"},
    {"rccl_lib", "GACT_HIP_RCCL_LIB", 'c', 's', "<file>: the RCCL library gact_hip_comm_create loads"},
    // diagnostic / tuning: -DGACT_EXPERIMENTS builds only
    {"no_aff_seed", "GACT_HIP_NO_AFF_SEED", 'c', 'd', "round 1's packed seed pass for affine scorings"},
    {"lane_blocks", "GACT_HIP_LANE_BLOCKS", 'c', 'd', "<n>: blocks of the critical lane"},
    {"lane_small", "GACT_HIP_LANE_SMALL", 'c', 'd', "<f>: the lane also instead of the all-wide launch, for runs of at least f/2 chains per lane tile"},
    {"no_chain_prio", "GACT_HIP_NO_CHAIN_PRIO", 'c', 'd', "all DP passes at issue priority 0"},
    {"static_prio", "GACT_HIP_STATIC_PRIO", 'c', 'd', "fixed priority thresholds instead of the ranking against the longest running chain"},
    {"rank16", "GACT_HIP_RANK16", 'c', 'd', "the ranking's two thresholds in sixteenths, hi << 8 | mid (default 12, 8)"},
    {"team_when_shared", "GACT_HIP_TEAM_WHEN_SHARED", 'c', 'd', "look-ahead walker in split launches that share the machine"},
    {"wide_blocks_per_cu", "GACT_HIP_WIDE_BLOCKS_PER_CU", 'c', 'd', "<n>: resident blocks per CU of the wide launch"},
    {"band_quantum", "GACT_HIP_BAND_QUANTUM", 'c', 'd', "<1|4|8>: lanes store their band in aligned groups"},
    {"lin_blocks", "GACT_HIP_LIN_BLOCKS", 'c', 'd', "<n>: a smaller persistent grid for the one-wave-does-all split linear-gap launch"},
    {"role_blocks", "GACT_HIP_ROLE_BLOCKS", 'c', 'd', "<n>: a smaller persistent grid for the role launch"},
    {"trace", "GACT_HIP_TRACE", 'c', 'd', "every launch of a run named on stderr and waited for"},
    {"trace_upload", "GACT_HIP_TRACE_UPLOAD", 'c', 'd', "stamps inside candidates_upload / _fetch on stderr"},
};
// the value of a switch's environment variable (null: unset -- or a diagnostic switch in a default build)
static const char *opt_env(const char *name)
{
    for (const OptionDef &o : kOptions)
        if (!strcmp(o.name, name)) {
#ifndef GACT_EXPERIMENTS
            if (o.klass == 'd') return nullptr;
#endif
            return o.env ? getenv(o.env) : nullptr;
        }
    fprintf(stderr, "[gact_hip] internal: switch '%s' is not in the option table\n", name);
    abort();
}

// The call combiner (round 4).  The reference's callers are N feeder threads, each with its own GPU_storage, all
// calling GACT_Batch at about the same time (darwin.cpp:408-433,619-629).  One persistent launch per call meant N grids
// queueing for the CUs, each with an N-th of the candidates and no load balancing inside it: 3,957 GCUPS at 8 feeders
// against 6,300 for one caller with the whole list.  Now a run that arrives while runs of OTHER threads are arriving is
// merged with them into one seed + main launch on a merge slot: the callers' candidate ranges gathered into one array
// (strand in bit 30 of query_id, kCompInCand), one set of queues, longest chains of ALL callers first; the records
// are copied back into each caller's own array behind the launch and each caller's stream waits for that.  The first
// arrival leads: it waits a short window for the slots whose threads called within the last few milliseconds, launches
// for everybody, and the others return.  A caller alone (one thread, however many slots: steps in flight) never waits
// and launches on its own slot exactly as before.
struct RunReq {
    int slot, first, n, rc_from, same_file;
    int status = 0;
    bool launched = false;
    std::string err;
};
struct Combiner {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<RunReq *> pending;
    bool leader = false;
    bool enabled = true;              // GACT_HIP_NO_COMBINE unset
    int window_us = 1000;             // GACT_HIP_COMBINE_US: how long the leader waits for the others at most
    int next_merge = 0;
    int n_merge = 0;
    long merged_launches = 0, merged_runs = 0, rejoins = 0;
    std::vector<std::shared_ptr<LaunchStats>> stats_pool;      // (the leader's: one leader at a time)
};

struct gact_hip_engine {
    gact_hip_params params;
    gact::KParams kp{};
    int C = 20;                 // columns per lane
    int big_cb = 0;             // tile_size > 512: the one-wave-per-tile kernels of gact_big.hpp, 16 or 32 columns per lane
    int big_blocks = 0;         //   and their grid (the workspace is one pointer matrix per wave)
    bool p16 = false;           // scoring fits the packed-int16 main kernel
    bool split = false;         // ... in its split (two-region) layout: tile <= 320 and early <= 208
    bool tagged = false;        // the packed main launch runs its pointer phase on tagged scores (any layout)
    bool lin = false;           // linear gaps (open == extend == mismatch): the drifted pass of gact_lin.hpp on 2-bit sets
    bool aff = false;           // any other scoring that fits: the drifted affine pass of gact_aff.hpp (split main launch, 2-bit sets)
    bool aff_seed = true;       // ... and its first-tile form in the seed launch (GACT_HIP_NO_AFF_SEED: round 1's packed seed pass)
    int wide = 0;               // wide (32 lanes per tile pair) main launch: 0 auto (few chains), 1 always, -1 never
    int wide_blocks_per_cu = 0; // GACT_HIP_WIDE_BLOCKS_PER_CU: resident blocks per CU of the wide launch (default 2)
    bool crit_lane = true;      // GACT_HIP_NO_CRIT_LANE unset: a run of 1-1.5 chains per tile slot on an idle engine has a wide launch beside its split one
    int lane_blocks = 0;                // GACT_HIP_LANE_BLOCKS=<n>: blocks of the lane (default: a third of the resident blocks)
    bool lane_small = false;            // GACT_HIP_LANE_SMALL=<f>: runs of fewer chains than tile slots, but at least f/2 per tile of the lane, take the
    int lane_small_factor = 3;          //   lane + split launches instead of the all-wide launch (experiment; f defaults to 3)
    bool crit_lane_always = false;      // GACT_HIP_CRIT_LANE_ALWAYS=1: ... and larger runs (up to 4 chains per slot) too
    bool kernel_copies = true;  // GACT_HIP_SDMA_COPIES unset: candidate lists and records cross the bus in a kernel (bus_copy_kernel)
    bool shared_hint = true;    // GACT_HIP_NO_SHARED_HINT unset: a launch made while another slot is running does not pick the wide layout
    std::atomic<bool> caller_keeps_runs_in_flight{false};      // set_option("runs_in_flight", 1): the caller says so itself -- every launch takes the throughput layout
    bool team_when_shared = false;      // GACT_HIP_TEAM_WHEN_SHARED=1: a split linear-gap launch that shares the machine walks by teams
    bool overlap_seed = true;   // GACT_HIP_NO_OVERLAP unset: a large run on an idle engine seeds in length order, most of it beside its main launch
    bool side_lane = true;      // GACT_HIP_NO_SIDE_LANE unset: few raw-byte candidates run beside the 2-bit launches (launch_extend)
    bool route_other = true;    // GACT_HIP_NO_ROUTING unset: raw-byte kernels only for candidates with a non-ACGT read
    bool chain_prio = true;     // main launch: longest chains first in the DP issue order too
    bool static_prio = false;   // GACT_HIP_STATIC_PRIO: fixed thresholds instead of the ranking (read once, at create)
    uint32_t poison = 0;        // GACT_HIP_POISON_WS=<seed>: the workspace is filled with a seeded pattern before every launch
    int rank16 = (12 << 8) | 8; // GACT_HIP_RANK16: the ranking's thresholds in sixteenths of the longest running chain, hi << 8 | mid
    bool seed16 = false;        // first tiles on the packed seed kernel too (arg-max keys fit)
    int seed_grid_blocks = 0;   // persistent grid of the packed seed kernel (2 waves per SIMD)
    int seed_lin_grid_blocks = 0;       // ... of its linear-gap form (3)
    int lin_grid_blocks = 0;    // persistent grid of the linear-gap split launch (its own occupancy)
    bool roles = false;         // GACT_HIP_ROLES=1 / set_option "roles": the split linear-gap main launch runs with DP waves and walker waves (gact_roles.hpp)
    int role_grid_blocks = 0;   // ... and its persistent grid (blocks of kRoleThreads)
    int shared_twelfths = 6;    // GACT_HIP_SHARED_TWELFTHS / set_option "shared_twelfths" (gact_policy.hpp Caps::shared_twelfths)
    int lone_lane = 0;          // GACT_HIP_LONE_LANE / set_option "lone_lane" (gact_policy.hpp Caps::lone_lane)
    bool overlap_big = false;   // GACT_HIP_OVERLAP_BIG / set_option "overlap_big" (gact_policy.hpp Caps::overlap_big)
    int coop = 0;               // GACT_HIP_COOP / set_option "coop": two banks of tiles per wave and cooperative, batched walks (gact_coop.hpp):
                                // 0 where throughput bounds the launch (gact_policy.hpp), 1 always, -1 never
    int aff_grid_blocks = 0;    // ... of the drifted affine split launch (two blocks per CU)
    int wide_lin_grid_blocks = 0;       // ... of the linear-gap wide launch
    gact::P16Consts kc;
    hipDeviceProp_t prop;
    int grid_blocks = 0;        // persistent grid
    int blocks_per_cu = 0;
    size_t ws_words_total = 0;
    SeqSet sets[GACT_NUM_SETS];
    std::vector<Slot> slots;           // [0, n_user) the callers' slots, behind them the merge slots of the combiner
    int n_user = 0;
    Combiner cb;
    std::mutex upload_mu;
    int64_t sets_epoch = 0;     // bumped by every upload / derive_revcomp: range checks of older sets are void
    DsoftState dsoft;
    std::mutex dsoft_mu;        // the filter's scratch is shared by all slots
};

namespace {

int set_device(gact_hip_engine *e)
{
    HIP_TRY(hipSetDevice(e->params.device_id));
    return 0;
}

int check_slot(gact_hip_engine *e, int slot)
{
    if (!e) return fail(GACT_HIP_EINVAL, "engine is NULL");
    if (slot < 0 || slot >= e->n_user)
        return fail(GACT_HIP_EINVAL, "slot %d out of range [0,%d)", slot, e->n_user);
    return 0;
}

// a call for `slot` from this thread, now (the combiner's guess at who is about to submit a run)
void note_call(gact_hip_engine *e, int slot)
{
    if (!e->cb.enabled || e->n_user < 2) return;
    std::lock_guard<std::mutex> lk(e->cb.mu);
    Slot &sl = e->slots[slot];
    sl.last_thread = std::this_thread::get_id();
    sl.last_call = std::chrono::steady_clock::now();
}

// d_counter layout (ints): [0] pop_seed, [2..3] seed_cells (u64), [4] / [5] candidates routed to the 2-bit / the raw-byte launches,
// [6] tiles run twice because their walk left the stored band (gact_lin.hpp LinBand), [8..8+kBuckets) bucket_count,
// [8+kBuckets..8+2*kBuckets) bucket_pop
// [1] pop counter of the second seed launch, [7] "second set of queues is complete" (overlapped seeding, run_overlapped);
// behind the ring of longest_running: the second set's bucket_count / bucket_pop, the histogram and the cursors of the
// ordering sort
constexpr int kMoreCount = 8 + 2 * gact::kBuckets + gact::kEpochs, kMorePop = kMoreCount + gact::kBuckets;
constexpr int kOrderHist = kMorePop + gact::kBuckets, kOrderCursor = kOrderHist + gact::kBuckets;
constexpr int kCounterInts = kOrderCursor + gact::kBuckets;

// GACT_HIP_POISON_WS: seeded garbage over the slot's whole traceback workspace (poison_kernel)
int poison_ws(gact_hip_engine *e, Slot &sl, uint32_t salt);

// The slots' streams belong to the process, not to an engine: slot k of every engine on a device runs on the k-th stream the
// process made there.  The runtime deals a process's streams onto few hardware queues (four unless GPU_MAX_HW_QUEUES says
// otherwise) when they are first used, and launches of two streams that share a queue run one after the other.  The slots of
// a process's FIRST engine get a queue each; the streams a later engine created did not -- the same four steps in flight took
// 52-54 ms per step on an engine made after another one had been closed, 41-42 in a fresh process or with eight queues
// (ONT shape, profiles/r04/hw_queue_sharing_side_configs.txt).  Streams are never destroyed; an engine hands them back idle.
struct StreamPool {
    std::mutex mu;
    std::vector<std::vector<std::pair<int, hipStream_t>>> idle;      // [device]: (index, stream)
    std::vector<int> made;                                            // [device]: streams made so far
    int take(int dev, hipStream_t *out)
    {
        std::lock_guard<std::mutex> lk(mu);
        if ((int)idle.size() <= dev) { idle.resize((size_t)dev + 1); made.resize((size_t)dev + 1, 0); }
        auto &v = idle[(size_t)dev];
        if (!v.empty()) {
            size_t best = 0;
            for (size_t k = 1; k < v.size(); k++) if (v[k].first < v[best].first) best = k;
            *out = v[best].second;
            const int idx = v[best].first;
            v.erase(v.begin() + (long)best);
            return idx;
        }
        if (hipStreamCreateWithFlags(out, hipStreamNonBlocking) != hipSuccess) return -1;
        return made[(size_t)dev]++;
    }
    void give(int dev, int idx, hipStream_t s)
    {
        std::lock_guard<std::mutex> lk(mu);
        if ((int)idle.size() <= dev) { idle.resize((size_t)dev + 1); made.resize((size_t)dev + 1, 0); }
        idle[(size_t)dev].emplace_back(idx, s);
    }
};
StreamPool g_streams;

// stream, events, counters and traceback workspace of a slot (merge slots: on first use)
int init_slot(gact_hip_engine *e, Slot &sl)
{
    if (sl.stream) return 0;
    if ((sl.stream_index = g_streams.take(e->params.device_id, &sl.stream)) < 0 ||
        hipEventCreate(&sl.ev0) != hipSuccess || hipEventCreate(&sl.ev1) != hipSuccess ||
        hipEventCreate(&sl.ev_mid) != hipSuccess ||
        hipEventCreateWithFlags(&sl.ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void **)&sl.d_counter, kCounterInts * sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&sl.d_flags, sizeof(int)) != hipSuccess ||
        hipMalloc((void **)&sl.d_ws, (e->ws_words_total + 64) * sizeof(uint32_t)) != hipSuccess)      // (+ slack)
        return fail(GACT_HIP_ENOMEM, "slot allocation failed (workspace %zu MiB per slot)", e->ws_words_total * 4 >> 20);
    // a fresh workspace never shows what an earlier engine of this process left in the same memory
    if (hipMemsetAsync(sl.d_ws, 0xA5, (e->ws_words_total + 64) * sizeof(uint32_t), sl.stream) != hipSuccess)
        return fail(GACT_HIP_EDEVICE, "workspace initialisation failed");
    return 0;
}

// device buffers of a set for `total` bases in n_seqs sequences
int reserve_set(SeqSet &s, int64_t total, int32_t n_seqs)
{
    const size_t need_bases = (size_t)total + 64;
    if (need_bases > s.cap_bases) {
        if (s.d_raw) (void)hipFree(s.d_raw);
        if (s.d_packed) (void)hipFree(s.d_packed);
        s.d_raw = nullptr; s.d_packed = nullptr; s.cap_bases = 0;
        HIP_TRY(hipMalloc((void **)&s.d_raw, need_bases));
        HIP_TRY(hipMalloc((void **)&s.d_packed, (need_bases / 16 + 4) * sizeof(uint32_t)));
        s.cap_bases = need_bases;
    }
    const size_t need_seqs = (size_t)n_seqs + 2;
    if (need_seqs > s.cap_seqs) {
        if (s.d_offsets) (void)hipFree(s.d_offsets);
        if (s.d_other) (void)hipFree(s.d_other);
        s.d_offsets = nullptr; s.d_other = nullptr; s.cap_seqs = 0;
        HIP_TRY(hipMalloc((void **)&s.d_offsets, need_seqs * sizeof(int64_t)));
        HIP_TRY(hipMalloc((void **)&s.d_other, need_seqs * sizeof(int32_t)));
        s.cap_seqs = need_seqs;
    }
    return 0;
}

// upload + pack one set on `stream`
int upload_set(gact_hip_engine *e, SeqSet &s, Slot &sl, const uint8_t *concat, const int64_t *offsets,
               int32_t n_seqs)
{
    if (n_seqs < 0 || (n_seqs > 0 && (!offsets || !concat && offsets[n_seqs] > 0)))
        return fail(GACT_HIP_EINVAL, "upload: bad arguments");
    const int64_t total = n_seqs ? offsets[n_seqs] : 0;
    if (n_seqs && offsets[0] != 0) return fail(GACT_HIP_EINVAL, "upload: offsets[0] must be 0");
    for (int32_t k = 0; k < n_seqs; k++)
        if (offsets[k + 1] < offsets[k]) return fail(GACT_HIP_EINVAL, "upload: offsets not monotone");

    int rc_alloc = reserve_set(s, total, n_seqs);
    if (rc_alloc) return rc_alloc;
    s.h_offsets.assign(offsets, offsets + n_seqs + 1);
    if (n_seqs == 0) s.h_offsets.assign(1, 0);
    s.n = n_seqs; s.total = total;
    s.max_len = 0;
    for (int32_t k = 0; k < n_seqs; k++) s.max_len = std::max(s.max_len, offsets[k + 1] - offsets[k]);

    HIP_TRY(hipMemcpyAsync(s.d_offsets, s.h_offsets.data(), s.h_offsets.size() * sizeof(int64_t),
                           hipMemcpyHostToDevice, sl.stream));
    if (total) HIP_TRY(hipMemcpyAsync(s.d_raw, concat, (size_t)total, hipMemcpyHostToDevice, sl.stream));
    HIP_TRY(hipMemsetAsync(sl.d_flags, 0, sizeof(int), sl.stream));
    const int64_t n_words = (total + 15) / 16 + 2;
    const int threads = 256;
    const int blocks = (int)std::min<int64_t>((n_words + threads - 1) / threads, 4096);
    HIP_TRY(hipMemsetAsync(s.d_other, 0, (size_t)(n_seqs + 1) * sizeof(int32_t), sl.stream));
    hipLaunchKernelGGL(gact::pack_kernel, dim3(std::max(blocks, 1)), dim3(threads), 0, sl.stream,
                       s.d_raw, total, s.d_packed, n_words, sl.d_flags, s.d_offsets, n_seqs, s.d_other);
    HIP_TRY(hipGetLastError());
    int flags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, sl.d_flags, sizeof(int), hipMemcpyDeviceToHost, sl.stream));
    HIP_TRY(hipStreamSynchronize(sl.stream));
    s.has_other = (flags & 1) != 0;
    s.h_other.clear();
    if (s.has_other) {
        // which sequences: on the host too, so that a run over a list the host holds can count its two routes itself
        s.h_other.resize((size_t)n_seqs);
        HIP_TRY(hipMemcpy(s.h_other.data(), s.d_other, (size_t)n_seqs * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    (void)e;
    return 0;
}

// tile_size 513 .. 2048 (gact_big.hpp): one wave per tile / per candidate, raw bytes
int launch_big_tiles(gact_hip_engine *e, Slot &sl, const SeqSet &rs, const SeqSet &qf, const SeqSet &qr, int n, int states_stride)
{
    const int blocks = std::max(1, std::min((n + 3) / 4, e->big_blocks));
    auto k = e->big_cb == 16 ? gact::big_tiles_kernel<16> : gact::big_tiles_kernel<32>;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(gact::kBlockThreads), 0, sl.stream, e->kp, rs.dev(true), qf.dev_or(true, rs),
                       qr.dev_or(true, rs), sl.tiles.p, n, sl.results.p, sl.states.p, states_stride, reinterpret_cast<uint8_t *>(sl.d_ws));
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int C>
int launch_tiles(gact_hip_engine *e, Slot &sl, const SeqSet &rs, const SeqSet &qf, const SeqSet &qr, int n,
                 int states_stride)
{
    const bool raw = rs.has_other || qf.has_other || qr.has_other;
    const int waves_needed = (n + gact::kGroupsPerWave - 1) / gact::kGroupsPerWave;
    const int blocks_needed = (waves_needed + 3) / 4;
    const int blocks = std::max(1, std::min(blocks_needed, e->grid_blocks));
    hipLaunchKernelGGL((gact::align_tiles_kernel<C>), dim3(blocks), dim3(gact::kBlockThreads), 0, sl.stream,
                       e->kp, rs.dev(raw), qf.dev_or(raw, rs), qr.dev_or(raw, rs), sl.tiles.p, n, sl.results.p,
                       sl.states.p, states_stride, sl.d_ws);
    HIP_TRY(hipGetLastError());
    return 0;
}

// what a seed launch + main launch pair runs on: the slot's own stream, counters, hand-off lists and workspace, or
// the slot's side lane.  Chain states and records are indexed by candidate and shared.
struct Lane {
    hipStream_t stream;
    int *d_counter;
    int *live;
    size_t live_cap;
    uint32_t *d_ws;
    size_t ws_words;
    int max_blocks;                      // 0: whatever the kernel's occupancy allows (the workspace is sized for it)
};

Lane main_lane(gact_hip_engine *e, Slot &sl) { return Lane{sl.stream, sl.d_counter, sl.live.p, sl.live.cap, sl.d_ws, e->ws_words_total, 0}; }

gact::ChainQueues queues(const Lane &ln, Slot &sl)
{
    gact::ChainQueues q;
    q.pop_seed = ln.d_counter;
    q.seed_cells = reinterpret_cast<unsigned long long *>(ln.d_counter + 2);
    q.bucket_count = ln.d_counter + 8;
    q.bucket_pop = ln.d_counter + 8 + gact::kBuckets;
    q.live = ln.live;
    q.live_stride = (int)(ln.live_cap / (2 * gact::kBuckets));       // (the second half: the second set of overlapped seeding)
    q.states = sl.chain_states.p;
    q.longest_now = ln.d_counter + 8 + 2 * gact::kBuckets;
    q.band_redos = ln.d_counter + 6;
    q.list_count = nullptr;
    q.list = nullptr;
    q.list_n = -1;
    q.more_flag = nullptr;
    q.more_count = q.more_pop = q.more_live = nullptr;
    q.leave_longest = 0;
    return q;
}

// the second stream of overlapped seeding (run_pass)
int ensure_aux_stream(Slot &sl)
{
    if (sl.aux_stream) return 0;
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&sl.aux_stream, hipStreamNonBlocking, hi));
    HIP_TRY(hipEventCreateWithFlags(&sl.aux_ev_a, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&sl.aux_ev_b, hipEventDisableTiming));
    return 0;
}

// device arrays of a slot for a list of n candidates
int reserve_candidates(Slot &sl, size_t n)
{
    return (sl.cands.reserve(n) || sl.overlaps.reserve(n) || sl.live.reserve(2 * n * gact::kBuckets) || sl.chain_states.reserve(n) ||
            sl.deferred.reserve(2 * n) || sl.order.reserve(n)) ? -1 : 0;
}

// the second set of queues of a lane as a launch's own set
gact::ChainQueues second_queues(const Lane &ln, Slot &sl)
{
    gact::ChainQueues q = queues(ln, sl);
    q.pop_seed = ln.d_counter + 1;
    q.bucket_count = ln.d_counter + kMoreCount;
    q.bucket_pop = ln.d_counter + kMorePop;
    q.live = ln.live + (size_t)gact::kBuckets * q.live_stride;
    return q;
}

int poison_lane(gact_hip_engine *e, const Lane &ln, uint32_t salt)
{
    if (!e->poison) return 0;
    hipLaunchKernelGGL(gact::poison_kernel, dim3(e->prop.multiProcessorCount * 8), dim3(256), 0, ln.stream, ln.d_ws,
                       ln.ws_words + 64, e->poison * 0x01000193u + salt);
    HIP_TRY(hipGetLastError());
    return 0;
}

int poison_ws(gact_hip_engine *e, Slot &sl, uint32_t salt) { return poison_lane(e, main_lane(e, sl), salt); }

size_t ws_words_for(const gact_hip_engine *e, int blocks)
{
    const size_t groups = (size_t)blocks * (gact::kBlockThreads / 64) * gact::kGroupsPerWave;
    return groups * gact::kSlots * (size_t)e->kp.ws_words;             // two tiles per group in the p16 kernel
}

// the side lane of a slot, for `count` candidates on at most `blocks` blocks
int side_lane(gact_hip_engine *e, Slot &sl, int count, int blocks, Lane *out)
{
    if (!sl.side_stream) {
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // ahead of the slot's own stream: its main launch must find room while the 2-bit seed launch drains
        HIP_TRY(hipStreamCreateWithPriority(&sl.side_stream, hipStreamNonBlocking, hi));
        HIP_TRY(hipEventCreateWithFlags(&sl.side_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&sl.side_go, hipEventDisableTiming));
        HIP_TRY(hipMalloc((void **)&sl.side_counter, kCounterInts * sizeof(int)));
    }
    if (blocks > sl.side_blocks) {
        if (sl.side_ws) (void)hipFree(sl.side_ws);
        sl.side_ws = nullptr; sl.side_blocks = 0;
        if (hipMalloc((void **)&sl.side_ws, (ws_words_for(e, blocks) + 64) * sizeof(uint32_t)) != hipSuccess)
            return fail(GACT_HIP_ENOMEM, "side workspace allocation failed (%zu MiB)", ws_words_for(e, blocks) * 4 >> 20);
        sl.side_blocks = blocks;
        HIP_TRY(hipMemsetAsync(sl.side_ws, 0xA5, (ws_words_for(e, blocks) + 64) * sizeof(uint32_t), sl.side_stream));
    }
    if (sl.side_live.reserve(2 * (size_t)count * gact::kBuckets)) return fail(GACT_HIP_ENOMEM, "device allocation failed");
    *out = Lane{sl.side_stream, sl.side_counter, sl.side_live.p, sl.side_live.cap, sl.side_ws, ws_words_for(e, sl.side_blocks), sl.side_blocks};
    return 0;
}

// int32 kernel alone, or (scoring permitting) int32 seed launch for the first
// tiles followed by the packed-int16 main launch for the rest of every chain
int launch_big_extend(gact_hip_engine *e, Slot &sl, int first, int n, int rc_from, int same_file)
{
    const SeqSet &rs = e->sets[GACT_SET_REF];
    const SeqSet &qf = e->sets[GACT_SET_QUERY], &qr = e->sets[GACT_SET_QUERY_RC];
    sl.two_phase = false; sl.routed_raw = 0; sl.side_used = false;
    const Lane own = main_lane(e, sl);
    const int blocks = std::max(1, std::min((n + 3) / 4, e->big_blocks));
    auto k = e->big_cb == 16 ? gact::big_extend_kernel<16> : gact::big_extend_kernel<32>;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(gact::kBlockThreads), 0, sl.stream, e->kp, rs.dev(true), qf.dev_or(true, rs),
                       qr.dev_or(true, rs), sl.cands.p, first, n, rc_from, same_file, sl.overlaps.p, queues(own, sl),
                       reinterpret_cast<uint8_t *>(sl.d_ws));
    HIP_TRY(hipGetLastError());
    return 0;
}

// the engine's capabilities as the launch policy sees them (gact_policy.hpp)
gact_policy::Caps policy_caps(const gact_hip_engine *e)
{
    gact_policy::Caps c;
    c.C = e->C; c.p16 = e->p16; c.seed16 = e->seed16; c.lin = e->lin; c.aff = e->aff; c.aff_seed = e->aff_seed; c.split = e->split; c.tagged = e->tagged;
    c.mismatch_below_extend = e->params.mismatch < e->params.gap_extend;
    c.shared_twelfths = e->shared_twelfths; c.lone_lane = e->lone_lane; c.overlap_big = e->overlap_big; c.roles = e->roles; c.coop = (e->lin && e->split && e->C == 20) ? e->coop : -1; c.overlap_seed = e->overlap_seed; c.crit_lane = e->crit_lane; c.crit_lane_always = e->crit_lane_always;
    c.lane_small = e->lane_small; c.lane_small_factor = e->lane_small_factor; c.lane_blocks = e->lane_blocks; c.team_when_shared = e->team_when_shared;
    c.wide = e->wide; c.wide_blocks_per_cu = e->wide_blocks_per_cu; c.cus = e->prop.multiProcessorCount;
    c.grid_blocks = e->grid_blocks; c.seed_grid_blocks = e->seed_grid_blocks; c.seed_lin_grid_blocks = e->seed_lin_grid_blocks;
    c.lin_grid_blocks = e->lin_grid_blocks; c.aff_grid_blocks = e->aff_grid_blocks; c.wide_lin_grid_blocks = e->wide_lin_grid_blocks;
    c.role_grid_blocks = e->role_grid_blocks; c.role_dp_waves = gact::kRoleDp;
    c.ws_words_per_tile = (size_t)e->kp.ws_words;
    c.role_ws_words_per_block = gact::role_ws_words<gact::SplitLayoutLin<7, 13>>(1);
    c.coop_ws_words_per_block = gact::coop_ws_words<gact::SplitLayoutLin<7, 13>>(1);
    return c;
}

// strands: -1 what [first, first + n) and rc_from say; else bit 0 = forward-strand candidates present, bit 1 =
// reverse-complement ones (merged runs: rc_from == kCompInCand, the strand travels with the candidate)
template <int C>
int launch_extend(gact_hip_engine *e, Slot &sl, int first, int n, int rc_from, int same_file, int strands = -1)
{
    const SeqSet &rs = e->sets[GACT_SET_REF];
    const SeqSet &qf = e->sets[GACT_SET_QUERY], &qr = e->sets[GACT_SET_QUERY_RC];
    const bool need_f = strands >= 0 ? (strands & 1) != 0 : first < rc_from, need_r = strands >= 0 ? (strands & 2) != 0 : first + n > rc_from;
    // Sets that hold bytes other than A/C/G/T (N, lower case) are compared as raw bytes like align.cpp:134.  With the
    // packed kernels that is decided per CANDIDATE: the launches on the 2-bit image put off every candidate one of whose
    // two reads holds such a byte (SeqSetDev::other, from pack_kernel), and a second pair of launches on the raw bytes
    // takes those -- one soft-masked read no longer moves a whole launch onto the slower kernels.
    const bool any_other = rs.has_other || (need_f && qf.has_other) || (need_r && qr.has_other);
    const bool mixed = any_other && e->p16 && e->route_other;
    gact::KParams kp = e->kp;
    const int64_t longest = std::min(rs.max_len, std::max(need_f ? qf.max_len : 0, need_r ? qr.max_len : 0));
    // GACT_HIP_STATIC_PRIO: thirds of the longest possible chain instead of the ranking against what is running
    const bool static_prio = e->static_prio;
    kp.prio_bases[0] = !e->chain_prio ? 0x7fffffff : static_prio ? (int32_t)std::min<int64_t>(longest / 3, 0x7fffffff) : 0;
    const int rank16 = e->rank16;                    // above 3/4 of the longest running chain: priority 2, above 1/2: 1
    kp.prio_bases[1] = !e->chain_prio ? 0x7fffffff : static_prio ? (int32_t)std::min<int64_t>(2 * longest / 3, 0x7fffffff) : rank16;
    sl.two_phase = e->p16;
    sl.routed_raw = 0;
    sl.overlapped = false;
    sl.lane = false;
    sl.roles = 0;
    // Is another slot of this engine still running?  Then this launch shares the machine (feeder threads, steps in
    // flight) and what counts is throughput: the wide layout -- faster per chain, slower per cell, made for a launch
    // that has the CUs to itself and lasts as long as its longest chain -- is not taken on its own account.
    bool shared_machine = e->caller_keeps_runs_in_flight.load(std::memory_order_relaxed);
    if (e->shared_hint && !shared_machine)
        for (const Slot &other : e->slots)
            if (&other != &sl && other.timed && other.ev1 && hipEventQuery(other.ev1) == hipErrorNotReady) { shared_machine = true; break; }
    (void)hipGetLastError();                 // (hipErrorNotReady is an answer, not a failure)

    // one seed launch + (packed kernels) one main launch over `count` candidates at most
    // GACT_HIP_TRACE: every launch named on stderr and waited for (a faulting kernel is the last one named)
    static const bool trace = opt_env("trace") != nullptr;
    auto traced = [&](const Lane &ln, const char *what, int blocks, int count) -> int {
        if (!trace) return 0;
        fprintf(stderr, "[gact_hip] %s: %d blocks, %d candidates ... ", what, blocks, count);
        fflush(stderr);
        HIP_TRY(hipStreamSynchronize(ln.stream));
        int dbg[8];
        HIP_TRY(hipMemcpy(dbg, ln.d_counter, sizeof dbg, hipMemcpyDeviceToHost));
        fprintf(stderr, "done (popped %d)\n", dbg[0]);
        return 0;
    };
    // second_set: the pass files and pops its chains in the lane's second set of queues (the raw-byte pass behind the 2-bit
    // one on the same lane: the first set keeps its counts for the statistics, nothing is cleared between the passes).
    // WHAT the pass runs as -- kernels, grids, sequence -- is gact_policy::plan_pass's answer (gact_policy.hpp: a pure function
    // of the count and the engine's capabilities); what follows launches it.
    auto run_pass = [&](const Lane &ln, bool raw, const int *list, const int *list_count, int count, bool first_pass, bool second_set = false) -> int {
        namespace pol = gact_policy;
        pol::Inputs in;
        in.count = count; in.raw = raw; in.listed = list != nullptr; in.second_set = second_set; in.shared_machine = shared_machine;
        in.own_lane = ln.stream == sl.stream; in.trace = trace; in.poison = e->poison != 0; in.lane_max_blocks = ln.max_blocks;
        const pol::Plan plan = pol::plan_pass(policy_caps(e), in);
        const gact::SeqSetDev d_rs = rs.dev(raw), d_qf = qf.dev_or(raw, rs), d_qr = qr.dev_or(raw, rs);
        gact::ChainQueues cq = second_set ? second_queues(ln, sl) : queues(ln, sl);
        cq.list = list; cq.list_count = list_count;
        if (trace) {
            fprintf(stderr, "[gact_hip] pass raw=%d listed=%d side=%d count=%d first=%d n=%d rc_from=%d plan=%s\n", (int)raw, list != nullptr,
                    ln.stream != sl.stream, count, first, n, rc_from, pol::describe(plan).c_str());
            fprintf(stderr, "[gact_hip]   ws %p + %zu MiB, counter %p, cands %p (%zu), overlaps %p, live %p (%zu), states %p (%zu x %zu B)\n", (void *)ln.d_ws,
                    ln.ws_words * 4 >> 20, (void *)ln.d_counter, (void *)sl.cands.p, sl.cands.cap, (void *)sl.overlaps.p, (void *)ln.live,
                    ln.live_cap, (void *)sl.chain_states.p, sl.chain_states.cap, sizeof(gact::ChainState));
            fprintf(stderr, "[gact_hip]   ref raw %p packed %p offsets %p (%lld bases), query %p %p, rc %p %p\n", (void *)rs.d_raw, (void *)rs.d_packed,
                    (void *)rs.d_offsets, (long long)rs.total, (void *)qf.d_raw, (void *)qf.d_packed, (void *)qr.d_raw, (void *)qr.d_packed);
        }
        using gact::extend_p16_kernel;
        using RolesL = gact::SplitLayoutLin<7, 13>;
        // a main launch of the plan's kernel: the same arguments whatever the kernel
        auto launch_main = [&](pol::MainK k, bool two_sets, int blocks, hipStream_t stream, const gact::ChainQueues &q, uint32_t *ws) -> int {
            using M = pol::MainK;
            const dim3 g((unsigned)blocks), b256(gact::kBlockThreads), brole(gact::kRoleThreads);
#define GACT_LAUNCH_MAIN(KERNEL, BLOCK) hipLaunchKernelGGL(KERNEL, g, BLOCK, 0, stream, kp, e->kc, d_rs, d_qf, d_qr, same_file, sl.overlaps.p, q, ws)
            if constexpr (C == 20) {
                switch (k) {
                case M::RolesLin: if (two_sets) GACT_LAUNCH_MAIN((gact::extend_roles_kernel<RolesL, true>), brole); else GACT_LAUNCH_MAIN((gact::extend_roles_kernel<RolesL, false>), brole); break;
                case M::CoopLin: if (two_sets) GACT_LAUNCH_MAIN((gact::extend_coop_kernel<RolesL, true>), b256); else GACT_LAUNCH_MAIN((gact::extend_coop_kernel<RolesL, false>), b256); break;
                case M::SplitLin: if (two_sets) GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayoutLin<7, 13>, false, true>), b256); else GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayoutLin<7, 13>, false>), b256); break;
                case M::SplitLinTeam: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayoutLinTeam<7, 13>, false>), b256); break;
                case M::WideLin: if (two_sets) GACT_LAUNCH_MAIN((extend_p16_kernel<gact::WideLayoutLin, false, true>), b256); else GACT_LAUNCH_MAIN((extend_p16_kernel<gact::WideLayoutLin, false>), b256); break;
                case M::SplitAffNeg: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayoutAff<7, 13, true>, false>), b256); break;
                case M::SplitAff: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayoutAff<7, 13, false>, false>), b256); break;
                default: break;
                }
            }
            switch (k) {
            case M::WideTaggedRaw: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::WideLayoutTagged, true>), b256); break;
            case M::WideTagged: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::WideLayoutTagged, false>), b256); break;
            case M::WideRaw: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::WideLayout, true>), b256); break;
            case M::Wide: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::WideLayout, false>), b256); break;
            case M::SplitTaggedRaw: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayout<7, 13, true>, true>), b256); break;
            case M::SplitTagged: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayout<7, 13, true>, false>), b256); break;
            case M::SplitRaw: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayout<7, 13>, true>), b256); break;
            case M::Split: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::SplitLayout<7, 13>, false>), b256); break;
            case M::UniformTaggedRaw: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::UniformLayout<C, gact::kGroup, true>, true>), b256); break;
            case M::UniformTagged: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::UniformLayout<C, gact::kGroup, true>, false>), b256); break;
            case M::UniformRaw: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::UniformLayout<C>, true>), b256); break;
            case M::Uniform: GACT_LAUNCH_MAIN((extend_p16_kernel<gact::UniformLayout<C>, false>), b256); break;
            default: break;
            }
#undef GACT_LAUNCH_MAIN
            HIP_TRY(hipGetLastError());
            return 0;
        };
        auto launch_seed = [&](pol::SeedK k, int blocks, hipStream_t stream, const gact::ChainQueues &q, uint32_t *ws) -> int {
            using S = pol::SeedK;
            const dim3 g((unsigned)blocks), b256(gact::kBlockThreads);
#define GACT_LAUNCH_SEED(KERNEL) hipLaunchKernelGGL(KERNEL, g, b256, 0, stream, kp, e->kc, d_rs, d_qf, d_qr, sl.cands.p, first, n, rc_from, same_file, sl.overlaps.p, q, ws)
            if constexpr (C == 20) {
                switch (k) {
                case S::P16Lin: GACT_LAUNCH_SEED((gact::seed_p16_kernel<C, false, 1>)); break;
                case S::P16Aff: GACT_LAUNCH_SEED((gact::seed_p16_kernel<C, false, 2>)); break;
                case S::P16AffNeg: GACT_LAUNCH_SEED((gact::seed_p16_kernel<C, false, 3>)); break;
                default: break;
                }
            }
            switch (k) {
            case S::P16Raw: GACT_LAUNCH_SEED((gact::seed_p16_kernel<C, true>)); break;
            case S::P16: GACT_LAUNCH_SEED((gact::seed_p16_kernel<C, false>)); break;
            case S::Int32:
                hipLaunchKernelGGL((gact::extend_kernel<C>), g, b256, 0, stream, kp, d_rs, d_qf, d_qr, sl.cands.p, first, n, rc_from, same_file,
                                   sl.overlaps.p, q, e->p16 ? 1 : 0, ws);
                break;
            default: break;
            }
#undef GACT_LAUNCH_SEED
            HIP_TRY(hipGetLastError());
            return 0;
        };
        int rc = 0;
        // Overlapped, ordered seeding (round 4).  One at a time a run was: seed launch (every first tile, ~2 ms on ecoli10x) ->
        // main launch, whose longest chains -- the ones that end last -- started only then.  Here the candidates are first
        // sorted by the length class of the chain they can make (two tiny launches), and
        //   slot stream : seed launch A (the longest nA: as many as main launch 1 has tile slots)  ->  main launch 1, on two
        //                 thirds of the resident blocks, popping set 1 (A's hand-offs), later set 2
        //   aux stream  : [after A] seed launch B (the rest, on the last third of the blocks, into a SECOND set of queues)
        //                 -> flag -> main launch 2 (that third of the blocks, set 2)
        // Every producer / consumer pair of a set of queues is still separated by a launch boundary (or by the flag written
        // in stream order behind seed launch B plus an acquire, extend_p16_kernel): nothing is handed over between running
        // launches.  The two thirds / one third of the workspace go with the blocks.
        if (plan.seq == pol::Seq::Overlapped) {
            if ((rc = ensure_aux_stream(sl))) return rc;
            const int ob = std::max(1, std::min((count + 255) / 256, 1024));
            hipLaunchKernelGGL(gact::order_hist_kernel, dim3(ob), dim3(256), 0, ln.stream, sl.cands.p, first, count, rc_from, d_rs.offsets,
                               d_qf.offsets, d_qr.offsets, kp.early, ln.d_counter + kOrderHist);
            hipLaunchKernelGGL(gact::order_scatter_kernel, dim3(ob), dim3(256), 0, ln.stream, sl.cands.p, first, count, rc_from,
                               d_rs.offsets, d_qf.offsets, d_qr.offsets, kp.early, ln.d_counter + kOrderHist,
                               ln.d_counter + kOrderCursor, sl.order.p);
            HIP_TRY(hipGetLastError());
            const int nA = plan.nA;
            // seed launch A
            gact::ChainQueues qa = queues(ln, sl);
            qa.list = sl.order.p; qa.list_n = nA;
            if ((rc = launch_seed(plan.seed, plan.seed_blocks, ln.stream, qa, ln.d_ws))) return rc;
            if (first_pass) HIP_TRY(hipEventRecord(sl.ev_mid, ln.stream));
            HIP_TRY(hipEventRecord(sl.aux_ev_a, ln.stream));
            // aux stream: seed launch B into the second set, the flag, main launch 2
            HIP_TRY(hipStreamWaitEvent(sl.aux_stream, sl.aux_ev_a, 0));
            gact::ChainQueues qb = second_queues(ln, sl);
            qb.list = sl.order.p + nA; qb.list_n = count - nA;
            if (count > nA && (rc = launch_seed(plan.seed, plan.seedB_blocks, sl.aux_stream, qb, ln.d_ws + plan.ws_split))) return rc;
            HIP_TRY(hipMemsetAsync(ln.d_counter + 7, 0xff, sizeof(int), sl.aux_stream));
            // (both main launches take set 1 -- the longer chains, complete since seed launch A ended -- and then set 2,
            //  open by the time main launch 2 starts)
            gact::ChainQueues q2 = queues(ln, sl);
            {
                const gact::ChainQueues s2 = second_queues(ln, sl);
                q2.more_flag = ln.d_counter + 7;
                q2.more_count = s2.bucket_count; q2.more_pop = s2.bucket_pop; q2.more_live = s2.live;
            }
            // (the critical lane, ChainQueues::leave_longest: main launch 2 in the wide layout, main launch 1 leaves it that many
            //  of the longest chains -- GACT_HIP_CRIT_LANE_ALWAYS; a run this size is bound by throughput)
            if (count > nA && (rc = launch_main(plan.lane ? pol::MainK::WideLin : plan.main, true, plan.main2_blocks, sl.aux_stream, q2, ln.d_ws + plan.ws_split))) return rc;
            HIP_TRY(hipEventRecord(sl.aux_ev_b, sl.aux_stream));
            // main launch 1: set 1, then set 2
            gact::ChainQueues q1 = q2;
            q1.leave_longest = plan.leave_longest;
            if ((rc = launch_main(plan.main, true, plan.main_blocks, ln.stream, q1, ln.d_ws))) return rc;
            HIP_TRY(hipStreamWaitEvent(ln.stream, sl.aux_ev_b, 0));
            if (first_pass) { sl.wide = false; sl.lin = true; sl.lane = plan.lane; sl.roles = plan.roles ? 1 : plan.coop ? 2 : 0; }
            sl.overlapped = true;
            return 0;
        }
        // ---- seed launch, then the main launch(es)
        if ((rc = launch_seed(plan.seed, plan.seed_blocks, ln.stream, cq, ln.d_ws))) return rc;
        if ((rc = traced(ln, raw ? "seed launch (raw bytes)" : "seed launch (2-bit)", plan.seed_blocks, count))) return rc;
        if (plan.seq == pol::Seq::SingleInt32) return 0;
        if (first_pass) HIP_TRY(hipEventRecord(sl.ev_mid, ln.stream));
        if ((rc = poison_lane(e, ln, 0x5bd1e995u))) return rc;      // the main launch reads nothing the seed launch stored
        if (first_pass) { sl.wide = plan.wide; sl.lin = plan.lin; sl.aff = plan.aff; sl.roles = plan.roles ? 1 : plan.coop ? 2 : 0; }
        if (plan.seq == pol::Seq::CritLane) {
            // The critical lane beside a run's ONE split main launch (a run too small for overlapped seeding, e.g. the merged
            // forward-strand calls of eight feeder threads: 33 k chains on 24.6 k tile slots last as long as their longest
            // chain): the wide launch on a third of the blocks, on the second stream, behind the seed launch like the split one
            if ((rc = ensure_aux_stream(sl))) return rc;
            HIP_TRY(hipEventRecord(sl.aux_ev_a, ln.stream));
            HIP_TRY(hipStreamWaitEvent(sl.aux_stream, sl.aux_ev_a, 0));
            if ((rc = launch_main(pol::MainK::WideLin, false, plan.main2_blocks, sl.aux_stream, cq, ln.d_ws + plan.ws_split))) return rc;
            HIP_TRY(hipEventRecord(sl.aux_ev_b, sl.aux_stream));
            gact::ChainQueues cq1 = cq;
            cq1.leave_longest = plan.leave_longest;
            if ((rc = launch_main(plan.main, false, plan.main_blocks, ln.stream, cq1, ln.d_ws))) return rc;
            HIP_TRY(hipStreamWaitEvent(ln.stream, sl.aux_ev_b, 0));
            if (first_pass) sl.lane = true;
            return 0;
        }
        if ((rc = launch_main(plan.main, false, plan.main_blocks, ln.stream, cq, ln.d_ws))) return rc;
        return traced(ln, raw ? "main launch (raw bytes)" : "main launch (2-bit)", plan.main_blocks, count);
    };

    const Lane own = main_lane(e, sl);
    sl.side_used = false;
    if (!mixed) return run_pass(own, any_other, nullptr, nullptr, n, true);
    // the candidates sorted by what their two reads hold: two lists, their lengths back on the host (they decide grids
    // and layouts: the only host wait of a run, and only of a run over sets with such reads)
    const gact::SeqSetDev o_rs = rs.dev(false), o_qf = qf.dev_or(false, rs), o_qr = qr.dev_or(false, rs);
    hipLaunchKernelGGL(gact::route_kernel, dim3(std::max(1, std::min((n + 255) / 256, 1024))), dim3(256), 0, sl.stream, sl.cands.p, first, n,
                       rc_from, o_rs.other, need_f ? o_qf.other : nullptr, need_r ? o_qr.other : nullptr, sl.deferred.p, sl.d_counter + 4);
    HIP_TRY(hipGetLastError());
    int routed[2] = {0, 0};
    // The two lists' lengths decide grids and layouts.  A list the host holds (uploaded, not merged, not made by the device
    // filter) is counted on the host from the sets' per-sequence flags -- no wait: with other slots' persistent launches on
    // the machine, route_kernel (14 registers: it does not fit beside three 168-register waves) starts only when a block of
    // theirs has left, and the host sat in this wait for milliseconds (1 % dirty reads, four steps in flight: -10 %).
    const bool host_counts = strands < 0 && !sl.h_cands.empty() && (size_t)first + (size_t)n <= sl.h_cands.size() &&
                             (!rs.has_other || (int32_t)rs.h_other.size() == rs.n) && (!need_f || !qf.has_other || (int32_t)qf.h_other.size() == qf.n) &&
                             (!need_r || !qr.has_other || (int32_t)qr.h_other.size() == qr.n);
    if (host_counts) {
        const int64_t key[4] = {first, n, rc_from, e->sets_epoch};
        if (memcmp(key, sl.routed_key, sizeof key)) {
            int dirty = 0;
            for (int k = first; k < first + n; k++) {
                const gact_candidate &c = sl.h_cands[(size_t)k];
                const SeqSet &qs = k >= rc_from ? qr : qf;
                dirty += ((rs.has_other && rs.h_other[(size_t)c.ref_id]) || (qs.has_other && qs.h_other[(size_t)c.query_id])) ? 1 : 0;
            }
            memcpy(sl.routed_key, key, sizeof key);
            sl.routed_host[0] = n - dirty; sl.routed_host[1] = dirty;
        }
        routed[0] = sl.routed_host[0]; routed[1] = sl.routed_host[1];
    } else {
        HIP_TRY(hipMemcpyAsync(routed, sl.d_counter + 4, sizeof routed, hipMemcpyDeviceToHost, sl.stream));
        HIP_TRY(hipStreamSynchronize(sl.stream));
    }
    sl.routed_raw = routed[1];
    int rc = 0;
    // Few raw-byte candidates beside many others: their launches last as long as their longest chain (a wave alone on
    // its SIMD issues every ~8 cycles: a lone chain advances at a third of the machine's per-SIMD rate), so they run
    // BESIDE the 2-bit launches, on the slot's side lane, started first -- the 2-bit main launch's last blocks become
    // resident as the side lane's blocks leave.  Many of them (more than the side lane holds at one tile pair per
    // group): one pass after the other on the whole machine.
    const int side_cap = std::max(1, e->prop.multiProcessorCount / 2);
    const int raw_blocks = ((routed[1] + 2 * gact::kGroupsPerWave - 1) / (2 * gact::kGroupsPerWave) + 3) / 4;
    if (routed[0] > 0 && routed[1] > 0 && e->side_lane && raw_blocks <= side_cap) {
        Lane side;
        if ((rc = side_lane(e, sl, routed[1], std::max(raw_blocks, std::min(side_cap, 16)), &side))) return rc;
        // (the side lane's launches read route_kernel's lists: behind it on the device, whether or not the host waited for it)
        HIP_TRY(hipEventRecord(sl.side_go, sl.stream));
        HIP_TRY(hipStreamWaitEvent(side.stream, sl.side_go, 0));
        HIP_TRY(hipMemsetAsync(side.d_counter, 0, kCounterInts * sizeof(int), side.stream));
        // ... and the slot's own launches behind the side lane's reset: "started first" has to hold on the DEVICE.  With the
        // host no longer waiting for route_kernel the own lane's seed launch followed it at once, filled the machine, the own
        // main launch after it, and the side lane's blocks found room when that had ended: 31 + 14 ms one after the other.
        // Now both lanes' seed launches become eligible together and the side lane's stream has the higher priority.
        HIP_TRY(hipEventRecord(sl.side_go, side.stream));
        HIP_TRY(hipStreamWaitEvent(sl.stream, sl.side_go, 0));
        if ((rc = poison_lane(e, side, 0x1b873593u))) return rc;
        if ((rc = run_pass(side, true, sl.deferred.p + n, sl.d_counter + 5, routed[1], false))) return rc;
        HIP_TRY(hipEventRecord(sl.side_done, side.stream));
        sl.side_used = true;
        rc = run_pass(own, false, sl.deferred.p, sl.d_counter + 4, routed[0], true);
        // (also when the own lane's pass could not be launched: the side lane's launches are under way and write this run's records)
        if (hipStreamWaitEvent(sl.stream, sl.side_done, 0) != hipSuccess && !rc) return fail(GACT_HIP_EDEVICE, "hipStreamWaitEvent failed");
        return rc;
    }
    if (routed[0] > 0 && (rc = run_pass(own, false, sl.deferred.p, sl.d_counter + 4, routed[0], true))) return rc;
    if (routed[1] == 0) return 0;
    if (routed[0] > 0) { int prc = poison_ws(e, sl, 0x1b873593u); if (prc) return prc; }
    // (behind a 2-bit pass on the same lane: the second set of queues, still empty)
    return run_pass(own, true, sl.deferred.p + n, sl.d_counter + 5, routed[1], routed[0] == 0, routed[0] > 0);
}

template <int C> int occupancy_blocks(int *out)
{
    int a = 0, b = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, gact::extend_kernel<C>, gact::kBlockThreads, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, gact::align_tiles_kernel<C>, gact::kBlockThreads, 0));
    int m = std::min(a, b);
    for (int v = 0; v < 14; v++) {
        int c = m;
        if (v == 12 || v == 13) continue;   // the linear-gap launches have their own grids (lin_occupancy_blocks, wide_lin_grid_blocks)
        auto k = v == 13 ? gact::extend_p16_kernel<gact::WideLayoutLin, false>
               : v == 0 ? gact::extend_p16_kernel<gact::UniformLayout<C>, true>
               : v == 1 ? gact::extend_p16_kernel<gact::UniformLayout<C>, false>
               : v == 2 ? gact::extend_p16_kernel<gact::SplitLayout<7, 13>, true>
               : v == 3 ? gact::extend_p16_kernel<gact::SplitLayout<7, 13>, false>
               : v == 4 ? gact::extend_p16_kernel<gact::WideLayout, true>
               : v == 5 ? gact::extend_p16_kernel<gact::WideLayout, false>
               : v == 6 ? gact::extend_p16_kernel<gact::SplitLayout<7, 13, true>, true>
               : v == 7 ? gact::extend_p16_kernel<gact::SplitLayout<7, 13, true>, false>
               : v == 8 ? gact::extend_p16_kernel<gact::WideLayoutTagged, true>
               : v == 9 ? gact::extend_p16_kernel<gact::WideLayoutTagged, false>
               : v == 10 ? gact::extend_p16_kernel<gact::UniformLayout<C, gact::kGroup, true>, true>
                         : gact::extend_p16_kernel<gact::UniformLayout<C, gact::kGroup, true>, false>;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, k, gact::kBlockThreads, 0));
        m = std::min(m, c);
    }
    *out = std::max(1, m);
    return 0;
}

int lin_occupancy_blocks(int *out)
{
    int a = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, gact::extend_p16_kernel<gact::SplitLayoutLin<7, 13>, false>,
                                                         gact::kBlockThreads, 0));
    int b = a;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, gact::extend_p16_kernel<gact::SplitLayoutLinTeam<7, 13>, false>,
                                                         gact::kBlockThreads, 0));
    *out = std::max(1, std::min(a, b));
    return 0;
}

template <int C> int seed_occupancy_blocks(int *out, int *out_lin)
{
    int a = 0, b = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, gact::seed_p16_kernel<C, true>, gact::kBlockThreads, 0));
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, gact::seed_p16_kernel<C, false>, gact::kBlockThreads, 0));
    int c = b;
    if constexpr (C == 20) {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, gact::seed_p16_kernel<C, false, 1>, gact::kBlockThreads, 0));
        int d = b;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&d, gact::seed_p16_kernel<C, false, 3>, gact::kBlockThreads, 0));
        b = std::min(b, d);
    }
    *out = std::max(1, std::min(a, b));
    *out_lin = std::max(1, c);          // the linear-gap seed kernel fits three waves per SIMD
    return 0;
}

int validate_tiles(gact_hip_engine *e, const SeqSet &rs, const SeqSet &qf, const SeqSet &qr, int n,
                   const gact_tile *tiles)
{
    const int T = e->params.tile_size;
    for (int t = 0; t < n; t++) {
        const gact_tile &d = tiles[t];
        if (d.ref_len < 0) continue;
        const SeqSet &qs = (d.query_set == GACT_SET_QUERY_RC) ? qr : qf;
        if (d.query_set != GACT_SET_QUERY && d.query_set != GACT_SET_QUERY_RC)
            return fail(GACT_HIP_EINVAL, "tile %d: query_set %d", t, d.query_set);
        if (d.ref_len > T || d.query_len > T || d.query_len < 0)
            return fail(GACT_HIP_ERANGE, "tile %d: length %d x %d exceeds tile_size %d", t, d.ref_len,
                        d.query_len, T);
        if (d.ref_id < 0 || d.ref_id >= rs.n || d.query_id < 0 || d.query_id >= qs.n)
            return fail(GACT_HIP_ERANGE, "tile %d: sequence id out of range", t);
        const int64_t rl = rs.h_offsets[d.ref_id + 1] - rs.h_offsets[d.ref_id];
        const int64_t ql = qs.h_offsets[d.query_id + 1] - qs.h_offsets[d.query_id];
        if (d.ref_off < 0 || d.ref_off + (int64_t)d.ref_len > rl || d.query_off < 0 ||
            d.query_off + (int64_t)d.query_len > ql)
            return fail(GACT_HIP_ERANGE, "tile %d: slice outside its sequence", t);
    }
    return 0;
}

int run_tiles(gact_hip_engine *e, Slot &sl, const SeqSet &rs, const SeqSet &qf, const SeqSet &qr, int32_t n,
              const gact_tile *tiles, gact_tile_result *results, uint8_t *states, int32_t states_stride)
{
    if (n == 0) return 0;
    if (!rs.d_raw) return fail(GACT_HIP_EINVAL, "align_tiles: the reference read set has not been uploaded");
    if (sl.tiles.reserve(n) || sl.results.reserve(n) || sl.states.reserve((size_t)n * states_stride))
        return fail(GACT_HIP_ENOMEM, "device allocation failed");
    HIP_TRY(hipMemcpyAsync(sl.tiles.p, tiles, (size_t)n * sizeof(gact_tile), hipMemcpyHostToDevice, sl.stream));
    HIP_TRY(hipMemsetAsync(sl.results.p, 0, (size_t)n * sizeof(gact_tile_result), sl.stream));
    { int prc = poison_ws(e, sl, 0x2545f491u); if (prc) return prc; }
    HIP_TRY(hipEventRecord(sl.ev0, sl.stream));
    int rc = e->big_cb ? launch_big_tiles(e, sl, rs, qf, qr, n, states_stride)
           : (e->C == 20) ? launch_tiles<20>(e, sl, rs, qf, qr, n, states_stride)
                          : launch_tiles<32>(e, sl, rs, qf, qr, n, states_stride);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(sl.ev1, sl.stream));
    sl.timed = true;
    HIP_TRY(hipMemcpyAsync(results, sl.results.p, (size_t)n * sizeof(gact_tile_result), hipMemcpyDeviceToHost,
                           sl.stream));
    HIP_TRY(hipMemcpyAsync(states, sl.states.p, (size_t)n * states_stride, hipMemcpyDeviceToHost, sl.stream));
    HIP_TRY(hipStreamSynchronize(sl.stream));
    return 0;
}

}  // namespace

extern "C" {

const char *gact_hip_last_error(void) { return g_err.c_str(); }

// what the parameters and the switches allow: geometry, kernel families, scheduling flags (no HIP call: gact_hip_create and
// gact_hip_plan_describe share it).  Returns whether the tile size belongs to gact_big.hpp.
static bool derive_kernel_flags(gact_hip_engine *e)
{
    const gact_hip_params *p = &e->params;
    e->C = (p->tile_size <= 20 * gact::kGroup) ? 20 : 32;
    const bool big = p->tile_size > 32 * gact::kGroup;               // gact_big.hpp
    if (big) e->big_cb = p->tile_size <= gact::BigGeom<16>::kTileMax ? 16 : 32;
    e->kp.tile_size = p->tile_size;
    e->kp.early = p->tile_size - p->tile_overlap;
    e->kp.match = p->match; e->kp.mismatch = p->mismatch;
    e->kp.open = p->gap_open; e->kp.ext = p->gap_extend;
    e->kp.thr = p->first_tile_score_threshold;
    e->kp.ws_words = (e->C == 20) ? gact::Geometry<20>::kWsWords : gact::Geometry<32>::kWsWords;
    e->p16 = !big && gact::p16_scoring_ok(p->tile_size, p->match, p->mismatch, p->gap_open, p->gap_extend) &&
             opt_env("force_int32") == nullptr;
    e->split = e->p16 && e->C == 20 && e->kp.early <= gact::GeometrySplit<7, 13>::W2 &&
               opt_env("force_uniform") == nullptr;
    e->tagged = e->p16 && gact::p16_tagged_ok(p->tile_size, p->match, p->mismatch, p->gap_open, p->gap_extend) &&
                opt_env("no_tagged") == nullptr;
    e->lin = e->tagged && gact::p16_lin_ok(p->tile_size, p->match, p->mismatch, p->gap_open, p->gap_extend) &&
             opt_env("no_lin") == nullptr;
    e->aff = e->tagged && e->split && !e->lin && gact::p16_aff_ok(p->tile_size, p->match, p->mismatch, p->gap_open, p->gap_extend) &&
             opt_env("no_aff") == nullptr;
    e->aff_seed = opt_env("no_aff_seed") == nullptr;
    e->kernel_copies = opt_env("sdma_copies") == nullptr;
    e->crit_lane = opt_env("no_crit_lane") == nullptr;
    e->crit_lane_always = opt_env("crit_lane_always") != nullptr;
    if (const char *v = opt_env("lane_blocks")) e->lane_blocks = std::max(0, atoi(v));
    if (const char *v = opt_env("lane_small")) { e->lane_small = atoi(v) > 0; e->lane_small_factor = std::max(1, atoi(v)); }
    e->seed16 = e->p16 && gact::p16_argmax_ok(p->tile_size, p->match) && opt_env("force_int32_seed") == nullptr;
    e->chain_prio = opt_env("no_chain_prio") == nullptr;
    e->route_other = opt_env("no_routing") == nullptr;
    e->side_lane = opt_env("no_side_lane") == nullptr;
    e->shared_hint = opt_env("no_shared_hint") == nullptr;
    e->overlap_seed = opt_env("no_overlap") == nullptr;
    e->roles = opt_env("roles") != nullptr && atoi(opt_env("roles")) != 0;
    if (const char *v = opt_env("overlap_big")) e->overlap_big = atoi(v) != 0;
    if (const char *v = opt_env("shared_twelfths")) e->shared_twelfths = std::max(1, std::min(atoi(v), 12));
    if (const char *v = opt_env("lone_lane")) e->lone_lane = std::max(-255, std::min(atoi(v), 255));
    if (const char *v = opt_env("coop")) e->coop = atoi(v) == 1 ? 1 : atoi(v) == 0 ? -1 : 0;
    e->team_when_shared = opt_env("team_when_shared") != nullptr;
    e->static_prio = opt_env("static_prio") != nullptr;
    if (const char *v = opt_env("rank16")) e->rank16 = atoi(v);
    if (const char *v = opt_env("poison_ws")) e->poison = (uint32_t)strtoul(v, nullptr, 0) | 0x80000000u;
    e->wide = opt_env("force_wide") ? 1 : opt_env("no_wide") ? -1 : 0;
    if (const char *v = opt_env("wide_blocks_per_cu")) e->wide_blocks_per_cu = atoi(v);
    return big;
}

int gact_hip_create(const gact_hip_params *p, gact_hip_engine **out)
{
    if (!p || !out) return fail(GACT_HIP_EINVAL, "create: NULL argument");
    *out = nullptr;
    if (p->tile_size < 1 || p->tile_size > GACT_HIP_MAX_TILE)
        return fail(GACT_HIP_EINVAL, "tile_size %d not in [1,%d]", p->tile_size, GACT_HIP_MAX_TILE);
    if (p->tile_overlap < 0 || p->tile_overlap >= p->tile_size)
        return fail(GACT_HIP_EINVAL, "tile_overlap %d must be in [0, tile_size)", p->tile_overlap);
    if (p->mismatch > 0 || p->gap_open > 0 || p->gap_extend > 0)
        return fail(GACT_HIP_EINVAL, "scoring: mismatch, gap_open and gap_extend must be <= 0");
    if (p->match < 0 || (int64_t)p->match * p->tile_size > (1 << 28) ||
        p->mismatch < -(1 << 20) || p->gap_open < -(1 << 20) || p->gap_extend < -(1 << 20))
        return fail(GACT_HIP_EINVAL, "scoring: values out of the int32-safe range");
    if (p->first_tile_score_threshold < 1)
        return fail(GACT_HIP_EINVAL, "first_tile_score_threshold must be >= 1 (the reference loops forever otherwise, gact.cpp:82)");
    if (p->n_slots < 1 || p->n_slots > 256) return fail(GACT_HIP_EINVAL, "n_slots %d not in [1,256]", p->n_slots);

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GACT_HIP_EDEVICE, "no HIP device visible: this engine has no CPU fallback");
    if (p->device_id < 0 || p->device_id >= ndev)
        return fail(GACT_HIP_EINVAL, "device_id %d not in [0,%d)", p->device_id, ndev);

#ifdef GACT_EXPERIMENTS
    fprintf(stderr, "[gact_hip] built with -DGACT_EXPERIMENTS: timing experiments may be compiled in, results are NOT to be trusted\n");
#endif
    gact_hip_engine *e = new gact_hip_engine();
    e->params = *p;
    int rc = set_device(e);
    if (rc) { delete e; return rc; }
    if (hipGetDeviceProperties(&e->prop, p->device_id) != hipSuccess) {
        delete e;
        return fail(GACT_HIP_EDEVICE, "hipGetDeviceProperties failed");
    }
    const bool big = derive_kernel_flags(e);
    e->kp.prio_bases[0] = e->kp.prio_bases[1] = 0x7fffffff;
    // pointer words of the linear-gap main launch are stored within `band` columns of the diagonal through a tile's (R, Q);
    // a walk that comes within a refill of its edge has its tile run again with the whole window (exact either way).
    // GACT_HIP_BAND=<n>: another width (tests: a narrow one exercises the second runs); 0: the whole window, as before round 4
    e->kp.band = gact::kLinBandDefault;
    if (const char *v = opt_env("band")) {
        const int b = atoi(v);
        e->kp.band = b <= 0 ? 0 : std::max(b, gact::kLinBandMin);
    }
    if (e->kp.band >= e->kp.early) e->kp.band = 0;             // (as wide as the window: nothing to leave out)
    // lanes store in aligned groups of 1, 4 or 8 (64 / 128 bytes of a workspace row): bits 16.. of kp.band
    {
        int q = gact::kLinBandQuantum;
        if (const char *v = opt_env("band_quantum")) q = atoi(v);
        q = q >= 8 ? 8 : q >= 4 ? 4 : q >= 2 ? 2 : 1;
        if (e->kp.band) e->kp.band |= q << 16;
    }
    static_assert(gact::GeometrySplit<7, 13>::kWsWords <= gact::Geometry<20>::kWsWords, "workspace too small");
    e->kc.match = gact::pk2(p->match); e->kc.nd = gact::pk2(p->mismatch - p->match);
    e->kc.open = gact::pk2(p->gap_open); e->kc.ext = gact::pk2(p->gap_extend);
    e->kc.ninf = gact::pk2(gact::kNegInf16); e->kc.one = gact::pk2(1);
    e->kc.mism = gact::pk2(p->mismatch); e->kc.dsub = (uint32_t)(p->match - p->mismatch) << 24;
    e->kc.c3 = gact::pk2(3); e->kc.nmask = ~e->kc.c3; e->kc.tag1 = gact::pk2(1); e->kc.tag2 = gact::pk2(2);
    e->kc.match4 = gact::pk2(4 * p->match); e->kc.mism4 = gact::pk2(4 * p->mismatch);
    e->kc.nd4 = gact::pk2(4 * (p->mismatch - p->match));
    e->kc.open4m2 = gact::pk2(4 * p->gap_open - 2); e->kc.ext4m2 = gact::pk2(4 * p->gap_extend - 2);
    e->kc.ext4m1 = gact::pk2(4 * p->gap_extend - 1);
    e->kc.dsub4 = (uint32_t)(4 * (p->match - p->mismatch)) << 24; e->kc.floor4 = gact::pk2(-6000);
    e->kc.next = gact::pk2(-p->gap_extend); e->kc.next4 = gact::pk2(-4 * p->gap_extend);
    e->kc.ext4 = gact::pk2(4 * p->gap_extend);
    e->kc.s_mismatch = p->mismatch; e->kc.s_open = p->gap_open; e->kc.s_ext = p->gap_extend;

    rc = (e->C == 20) ? occupancy_blocks<20>(&e->blocks_per_cu) : occupancy_blocks<32>(&e->blocks_per_cu);
    if (rc) { delete e; return rc; }
    e->grid_blocks = e->blocks_per_cu * e->prop.multiProcessorCount;
    {
        int sb = 0, sbl = 0;
        rc = (e->C == 20) ? seed_occupancy_blocks<20>(&sb, &sbl) : seed_occupancy_blocks<32>(&sb, &sbl);
        if (rc) { delete e; return rc; }
        // the workspace is sized for grid_blocks groups
        e->seed_grid_blocks = std::min(sb * e->prop.multiProcessorCount, e->grid_blocks);
        e->seed_lin_grid_blocks = std::min(sbl * e->prop.multiProcessorCount, e->grid_blocks);
    }
    e->lin_grid_blocks = e->grid_blocks;
    auto ws_words_for = [&](int blocks) { return ::ws_words_for(e, blocks); };
    if (e->lin) {
        // the linear-gap split launch has its own occupancy, and its walker addresses the workspace with 32-bit byte
        // offsets (gact_device.hpp tb_refill_oct): engines that never run it are sized without it, and one whose
        // workspace would pass 4 GiB runs the affine passes instead
        int lb = 0;
        if ((rc = lin_occupancy_blocks(&lb))) { delete e; return rc; }
        e->lin_grid_blocks = std::max(e->grid_blocks, lb * e->prop.multiProcessorCount);
        // GACT_HIP_LIN_BLOCKS=<n>: a smaller persistent grid for the split linear-gap launch (measurements: fewer resident
        // tiles = a smaller live pointer footprint)
        if (const char *v = opt_env("lin_blocks")) e->lin_grid_blocks = std::max(1, std::min(atoi(v), e->lin_grid_blocks));
        int wb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wb, gact::extend_p16_kernel<gact::WideLayoutLin, false>, gact::kBlockThreads, 0) !=
            hipSuccess) {
            delete e;
            return fail(GACT_HIP_EDEVICE, "hipOccupancyMaxActiveBlocksPerMultiprocessor failed");
        }
        e->wide_lin_grid_blocks = std::max(e->grid_blocks, wb * e->prop.multiProcessorCount);
        if ((ws_words_for(std::max(e->lin_grid_blocks, e->wide_lin_grid_blocks)) + 64) * sizeof(uint32_t) >= (1ull << 32)) {
            e->lin = false;
            e->lin_grid_blocks = e->grid_blocks;
        }
    }
    if (!e->lin) e->wide_lin_grid_blocks = e->grid_blocks;
    e->role_grid_blocks = 0;
    if (e->lin && e->split && e->C == 20) {
        using RL = gact::SplitLayoutLin<7, 13>;
        int rb = 0, rb2 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&rb, gact::extend_roles_kernel<RL, false>, gact::kRoleThreads, 0) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&rb2, gact::extend_roles_kernel<RL, true>, gact::kRoleThreads, 0) != hipSuccess) {
            delete e;
            return fail(GACT_HIP_EDEVICE, "hipOccupancyMaxActiveBlocksPerMultiprocessor failed");
        }
        e->role_grid_blocks = std::max(1, std::min(rb, rb2)) * e->prop.multiProcessorCount;
        if (const char *v = opt_env("role_blocks")) e->role_grid_blocks = std::max(1, std::min(atoi(v), e->role_grid_blocks));
    }
    e->roles = e->roles && e->role_grid_blocks > 0;
    e->aff_grid_blocks = e->grid_blocks;
    if (e->aff) {
        int ab = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&ab, gact::extend_p16_kernel<gact::SplitLayoutAff<7, 13, true>, false>,
                                                         gact::kBlockThreads, 0) != hipSuccess) {
            delete e;
            return fail(GACT_HIP_EDEVICE, "hipOccupancyMaxActiveBlocksPerMultiprocessor failed");
        }
        e->aff_grid_blocks = std::max(1, ab) * e->prop.multiProcessorCount;
    }
    if (p->max_blocks > 0) {                // a small engine (gact_hip_params.max_blocks): every persistent grid capped, the workspace with them
        for (int *g : {&e->grid_blocks, &e->seed_grid_blocks, &e->seed_lin_grid_blocks, &e->lin_grid_blocks, &e->wide_lin_grid_blocks, &e->aff_grid_blocks})
            *g = std::max(1, std::min(*g, (int)p->max_blocks));
        // (a role block is twelve waves where the others are four)
        if (e->role_grid_blocks > 0) e->role_grid_blocks = std::max(1, std::min(e->role_grid_blocks, (int)p->max_blocks * 4 / (gact::kRoleDp + gact::kRoleWalk)));
    }
    e->ws_words_total = ws_words_for(std::max(std::max(e->grid_blocks, e->aff_grid_blocks), std::max(e->lin_grid_blocks, e->wide_lin_grid_blocks)));
    if (e->role_grid_blocks > 0) {
        // the role launch lays its pointer words out by wave and bank (gact_roles.hpp); overlapped seeding puts two thirds of
        // its blocks in front of a seed launch's and a main launch's share of the workspace
        using RL = gact::SplitLayoutLin<7, 13>;
        const int r1 = e->role_grid_blocks * 2 / 3, r2 = e->role_grid_blocks - r1;
        const size_t need = std::max(gact::role_ws_words<RL>(e->role_grid_blocks),
                                     gact::role_ws_words<RL>(r1) + std::max(gact::role_ws_words<RL>(r2), ws_words_for(e->lin_grid_blocks - e->lin_grid_blocks * 2 / 3)));
        if ((need + 64) * sizeof(uint32_t) >= (1ull << 32)) { e->roles = false; e->role_grid_blocks = 0; }
        else e->ws_words_total = std::max(e->ws_words_total, need);
        // (the launch itself is taken with GACT_HIP_ROLES=1 / set_option "roles"; the grid and the room are there either way)
    }
    if (big) {
        // one pointer matrix per wave (1 MB at 16 columns per lane, 4 MB at 32): at most 2 GiB per slot, two blocks per CU
        const size_t per_block = (gact::kBlockThreads / 64) * (e->big_cb == 16 ? gact::BigGeom<16>::kWsBytes : gact::BigGeom<32>::kWsBytes);
        e->big_blocks = (int)std::max<size_t>(1, std::min<size_t>(2 * (size_t)e->prop.multiProcessorCount, (2ull << 30) / per_block));
        if (p->max_blocks > 0) e->big_blocks = std::max(1, std::min(e->big_blocks, (int)p->max_blocks));
        e->ws_words_total = (size_t)e->big_blocks * per_block / sizeof(uint32_t);
    }
    e->n_user = p->n_slots;
    e->cb.enabled = opt_env("no_combine") == nullptr && !big;
    if (const char *v = opt_env("combine_window_us")) e->cb.window_us = std::max(0, atoi(v));
    e->cb.n_merge = (e->cb.enabled && p->n_slots > 1) ? 2 : 0;
    e->slots.resize(p->n_slots + e->cb.n_merge);            // (never resized again: references into it stay good)
    // (the first merge slot with the callers' slots: its 1.3 GB would otherwise be allocated inside the first merged launch)
    for (int k = 0; k < p->n_slots + (e->cb.n_merge ? 1 : 0); k++)
        if ((rc = init_slot(e, e->slots[k]))) { gact_hip_destroy(e); return rc; }
    *out = e;
    return 0;
}

void gact_hip_destroy(gact_hip_engine *e)
{
    if (!e) return;
    if (opt_env("trace") && e->cb.merged_launches)
        fprintf(stderr, "[gact_hip] combiner: %ld merged launches carried %ld runs, %ld times a group that had come apart was joined again\n", e->cb.merged_launches, e->cb.merged_runs, e->cb.rejoins);
    (void)hipSetDevice(e->params.device_id);
    for (auto &sl : e->slots) {
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        sl.tiles.release(); sl.results.release(); sl.states.release();
        sl.cands.release(); sl.overlaps.release(); sl.live.release(); sl.chain_states.release(); sl.deferred.release();
        sl.inline_ref.release(); sl.inline_query.release();
        if (sl.d_counter) (void)hipFree(sl.d_counter);
        if (sl.d_flags) (void)hipFree(sl.d_flags);
        if (sl.d_ws) (void)hipFree(sl.d_ws);
        if (sl.side_stream) (void)hipStreamSynchronize(sl.side_stream);
        if (sl.aux_stream) {
            (void)hipStreamSynchronize(sl.aux_stream);
            (void)hipEventDestroy(sl.aux_ev_a); (void)hipEventDestroy(sl.aux_ev_b);
            (void)hipStreamDestroy(sl.aux_stream);
        }
        sl.order.release();
        sl.side_live.release();
        if (sl.side_counter) (void)hipFree(sl.side_counter);
        if (sl.side_ws) (void)hipFree(sl.side_ws);
        if (sl.side_done) (void)hipEventDestroy(sl.side_done);
        if (sl.side_go) (void)hipEventDestroy(sl.side_go);
        if (sl.side_stream) (void)hipStreamDestroy(sl.side_stream);
        if (sl.h_stage) (void)hipHostFree(sl.h_stage);
        if (sl.h_records) (void)hipHostFree(sl.h_records);
        if (sl.reg_out) { (void)hipHostUnregister(sl.reg_out); (void)hipGetLastError(); }
        if (sl.ev_ready) (void)hipEventDestroy(sl.ev_ready);
        if (sl.ev0) (void)hipEventDestroy(sl.ev0);
        if (sl.ev1) (void)hipEventDestroy(sl.ev1);
        if (sl.ev_mid) (void)hipEventDestroy(sl.ev_mid);
        if (sl.stream) {
            if (sl.stream_index >= 0) { (void)hipStreamSynchronize(sl.stream); g_streams.give(e->params.device_id, sl.stream_index, sl.stream); }
            else (void)hipStreamDestroy(sl.stream);
        }
    }
    for (Slot &sl : e->slots) sl.merged_stats.reset();
    for (auto &r : e->cb.stats_pool) {
        if (r->ev0) (void)hipEventDestroy(r->ev0);
        if (r->ev_mid) (void)hipEventDestroy(r->ev_mid);
        if (r->ev1) (void)hipEventDestroy(r->ev1);
        if (r->ev_done) (void)hipEventDestroy(r->ev_done);
        if (r->h_counter) (void)hipHostFree(r->h_counter);
    }
    e->cb.stats_pool.clear();
    for (auto &s : e->sets) s.release();
    e->dsoft.release_index();
    e->dsoft.release_scratch();
    delete e;
}

int gact_hip_get_device_info(gact_hip_engine *e, gact_hip_device_info *info)
{
    if (!e || !info) return fail(GACT_HIP_EINVAL, "NULL argument");
    memset(info, 0, sizeof *info);
    info->compute_units = e->prop.multiProcessorCount;
    info->clock_mhz = e->prop.clockRate / 1000;
    info->waves_per_cu = e->blocks_per_cu * (gact::kBlockThreads / 64);
    info->wave_size = e->prop.warpSize;
    info->hbm_bytes = (int64_t)e->prop.totalGlobalMem;
    snprintf(info->arch, sizeof info->arch, "%s", e->prop.gcnArchName);
    return 0;
}

int gact_hip_upload_seqs(gact_hip_engine *e, int which, const uint8_t *concat, const int64_t *offsets,
                         int32_t n_seqs)
{
    if (!e) return fail(GACT_HIP_EINVAL, "engine is NULL");
    if (which < 0 || which >= GACT_NUM_SETS) return fail(GACT_HIP_EINVAL, "which_set %d", which);
    std::lock_guard<std::mutex> lk(e->upload_mu);
    int rc = set_device(e);
    if (rc) return rc;
    e->sets_epoch++;
    if (which == GACT_SET_REF) {
        // the filter's index (start bins, bin -> sequence map, positions) describes the set it was built from
        std::lock_guard<std::mutex> lk2(e->dsoft_mu);
        e->dsoft.built = false;
    }
    return upload_set(e, e->sets[which], e->slots[0], concat, offsets, n_seqs);
}

int gact_hip_derive_revcomp(gact_hip_engine *e)
{
    if (!e) return fail(GACT_HIP_EINVAL, "engine is NULL");
    std::lock_guard<std::mutex> lk(e->upload_mu);
    int rc = set_device(e);
    if (rc) return rc;
    const SeqSet &qf = e->sets[GACT_SET_QUERY];
    SeqSet &qr = e->sets[GACT_SET_QUERY_RC];
    if (qf.n == 0 || !qf.d_raw) return fail(GACT_HIP_EINVAL, "derive_revcomp: GACT_SET_QUERY has not been uploaded");
    Slot &sl = e->slots[0];
    e->sets_epoch++;
    if ((rc = reserve_set(qr, qf.total, qf.n))) return rc;
    qr.h_offsets = qf.h_offsets; qr.n = qf.n; qr.total = qf.total; qr.max_len = qf.max_len;
    HIP_TRY(hipMemcpyAsync(qr.d_offsets, qf.d_offsets, qf.h_offsets.size() * sizeof(int64_t), hipMemcpyDeviceToDevice,
                           sl.stream));
    HIP_TRY(hipMemsetAsync(sl.d_flags, 0, sizeof(int), sl.stream));
    const int threads = 256;
    const int blocks = (int)std::min<int64_t>((qf.total + threads - 1) / threads, 1 << 16);
    hipLaunchKernelGGL(gact::revcomp_kernel, dim3(std::max(blocks, 1)), dim3(threads), 0, sl.stream, qf.d_raw,
                       qf.d_offsets, qf.n, qf.total, qr.d_raw, sl.d_flags);
    const int64_t n_words = (qf.total + 15) / 16 + 2;
    const int pblocks = (int)std::min<int64_t>((n_words + threads - 1) / threads, 4096);
    HIP_TRY(hipMemsetAsync(qr.d_other, 0, (size_t)(qf.n + 1) * sizeof(int32_t), sl.stream));
    hipLaunchKernelGGL(gact::pack_kernel, dim3(std::max(pblocks, 1)), dim3(threads), 0, sl.stream, qr.d_raw, qf.total,
                       qr.d_packed, n_words, sl.d_flags, qr.d_offsets, qf.n, qr.d_other);
    HIP_TRY(hipGetLastError());
    int flags = 0;
    HIP_TRY(hipMemcpyAsync(&flags, sl.d_flags, sizeof(int), hipMemcpyDeviceToHost, sl.stream));
    HIP_TRY(hipStreamSynchronize(sl.stream));
    if (flags & 2) {
        qr.n = 0; qr.total = 0;
        return fail(GACT_HIP_EINVAL, "derive_revcomp: Bad Nt char in GACT_SET_QUERY (darwin.cpp:139-141)");
    }
    qr.has_other = (flags & 1) != 0;
    qr.h_other.clear();
    if (qr.has_other) {
        qr.h_other.resize((size_t)qf.n);
        HIP_TRY(hipMemcpy(qr.h_other.data(), qr.d_other, (size_t)qf.n * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

int gact_hip_align_tiles(gact_hip_engine *e, int slot, int32_t n, const gact_tile *tiles,
                         gact_tile_result *results, uint8_t *states, int32_t states_stride)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!tiles || !results || !states)))
        return fail(GACT_HIP_EINVAL, "align_tiles: bad arguments");
    if (states_stride < 2 * e->params.tile_size)
        return fail(GACT_HIP_EINVAL, "states_stride %d < 2*tile_size", states_stride);
    if ((rc = set_device(e))) return rc;
    const SeqSet &rs = e->sets[GACT_SET_REF], &qf = e->sets[GACT_SET_QUERY], &qr = e->sets[GACT_SET_QUERY_RC];
    if ((rc = validate_tiles(e, rs, qf, qr, n, tiles))) return rc;
    return run_tiles(e, e->slots[slot], rs, qf, qr, n, tiles, results, states, states_stride);
}

int gact_hip_align_tiles_inline(gact_hip_engine *e, int slot, int32_t n, const uint8_t *ref_bases,
                                const uint8_t *query_bases, int32_t seq_stride, const int32_t *ref_lens,
                                const int32_t *query_lens, const uint8_t *reverses, const uint8_t *firsts,
                                gact_tile_result *results, uint8_t *states, int32_t states_stride)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!ref_bases || !query_bases || !ref_lens || !query_lens || !reverses || !firsts ||
                            !results || !states)))
        return fail(GACT_HIP_EINVAL, "align_tiles_inline: bad arguments");
    if (states_stride < 2 * e->params.tile_size)
        return fail(GACT_HIP_EINVAL, "states_stride %d < 2*tile_size", states_stride);
    if (n == 0) return 0;
    if ((rc = set_device(e))) return rc;
    Slot &sl = e->slots[slot];
    // gather the slices into two temporary resident sets
    std::vector<int64_t> roff(n + 1, 0), qoff(n + 1, 0);
    std::vector<gact_tile> tiles(n);
    for (int t = 0; t < n; t++) {
        const int rl = ref_lens[t] < 0 ? 0 : ref_lens[t];
        const int ql = (ref_lens[t] < 0 || query_lens[t] < 0) ? 0 : query_lens[t];
        if (rl > seq_stride || ql > seq_stride) return fail(GACT_HIP_ERANGE, "tile %d longer than seq_stride", t);
        roff[t + 1] = roff[t] + rl;
        qoff[t + 1] = qoff[t] + ql;
        gact_tile &d = tiles[t];
        d.ref_id = t; d.query_id = t; d.ref_off = 0; d.query_off = 0;
        d.ref_len = ref_lens[t] < 0 ? -1 : rl; d.query_len = ql;
        d.reverse = reverses[t]; d.first = firsts[t]; d.query_set = GACT_SET_QUERY; d.pad = 0;
    }
    std::vector<uint8_t> rcat((size_t)roff[n]), qcat((size_t)qoff[n]);
    for (int t = 0; t < n; t++) {
        memcpy(rcat.data() + roff[t], ref_bases + (size_t)t * seq_stride, (size_t)(roff[t + 1] - roff[t]));
        memcpy(qcat.data() + qoff[t], query_bases + (size_t)t * seq_stride, (size_t)(qoff[t + 1] - qoff[t]));
    }
    if ((rc = upload_set(e, sl.inline_ref, sl, rcat.data(), roff.data(), n))) return rc;
    if ((rc = upload_set(e, sl.inline_query, sl, qcat.data(), qoff.data(), n))) return rc;
    if ((rc = validate_tiles(e, sl.inline_ref, sl.inline_query, sl.inline_query, n, tiles.data()))) return rc;
    return run_tiles(e, sl, sl.inline_ref, sl.inline_query, sl.inline_query, n, tiles.data(), results, states,
                     states_stride);
}

// Small host <-> device transfers of a run -- the candidate list up, the records down -- made by a kernel that reads /
// writes the pinned host array over the bus instead of hipMemcpyAsync.  The runtime hands such copies to the SDMA engines,
// and an SDMA copy submitted while another stream's persistent launch is running came back only when a kernel of that
// launch had ended: 8-14 ms inside hipMemcpyAsync for the feeder threads that met it (the reference's caller, 8 threads:
// profiles/r04/upload_stall_runtime_settings.txt; HSA_ENABLE_SDMA=0 made them go away, and so does this).  Eight-byte
// units: both record types are multiples of that.  GACT_HIP_SDMA_COPIES=1: hipMemcpyAsync as before.
__global__ __launch_bounds__(256) void bus_copy_kernel(const uint2 *__restrict__ src, uint2 *__restrict__ dst, size_t n8)
{
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n8; k += (size_t)gridDim.x * 256) dst[k] = src[k];
}
static int bus_copy(hipStream_t stream, const void *src, void *dst, size_t bytes)
{
    const size_t n8 = bytes / 8;
    if (!n8) return 0;
    hipLaunchKernelGGL(bus_copy_kernel, dim3((unsigned)std::max<size_t>(1, std::min<size_t>((n8 + 255) / 256, 512))), dim3(256), 0, stream,
                       static_cast<const uint2 *>(src), static_cast<uint2 *>(dst), n8);
    HIP_TRY(hipGetLastError());
    return 0;
}
// the device's address of a pinned host array (hipHostMalloc / hipHostRegister); null: not mapped
static void *device_view(void *host)
{
    void *d = nullptr;
    if (!host || hipHostGetDevicePointer(&d, host, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return d;
}

// pinned host staging of a slot for jobs of n candidates: the upload's source, the fetch's destination (no pinned memory
// to be had: the copies go through the caller's own arrays, as before)
static void reserve_host_staging(Slot &sl, size_t n)
{
    if (n > sl.h_stage_cap) {
        if (sl.h_stage) (void)hipHostFree(sl.h_stage);
        sl.h_stage = nullptr; sl.h_stage_cap = 0;
        const size_t want = std::max<size_t>(n + n / 4, 4096);
        if (hipHostMalloc((void **)&sl.h_stage, want * sizeof(gact_candidate), hipHostMallocDefault) == hipSuccess) sl.h_stage_cap = want;
        else (void)hipGetLastError();
    }
    if (n > sl.h_records_cap) {
        if (sl.h_records) (void)hipHostFree(sl.h_records);
        sl.h_records = nullptr; sl.h_records_cap = 0;
        const size_t want = std::max<size_t>(n + n / 4, 4096);
        if (hipHostMalloc((void **)&sl.h_records, want * sizeof(gact_overlap), hipHostMallocDefault) == hipSuccess) sl.h_records_cap = want;
        else (void)hipGetLastError();
    }
}

int gact_hip_candidates_upload(gact_hip_engine *e, int slot, int32_t n, const gact_candidate *cands)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !cands)) return fail(GACT_HIP_EINVAL, "candidates_upload: bad arguments");
    // GACT_HIP_TRACE_UPLOAD: where an upload's time goes, per call, on stderr (microseconds since the call began)
    static const bool trace_up = opt_env("trace_upload") != nullptr;
    const auto tu0 = std::chrono::steady_clock::now();
    long tu[6] = {0, 0, 0, 0, 0, 0};
    auto mark = [&](int k) { if (trace_up) tu[k] = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tu0).count(); };
    if ((rc = set_device(e))) return rc;
    mark(0);
    note_call(e, slot);
    mark(1);
    Slot &sl = e->slots[slot];
    const SeqSet &rs = e->sets[GACT_SET_REF];
    const SeqSet &qf = e->sets[GACT_SET_QUERY], &qr = e->sets[GACT_SET_QUERY_RC];
    const int32_t qn = std::min(std::max(qf.n, qr.n), gact::kCompBit);      // (bit 30 of query_id is the strand of a merged run)
    for (int32_t k = 0; k < n; k++) {
        const gact_candidate &c = cands[k];
        if (c.ref_id < 0 || c.ref_id >= rs.n || c.query_id < 0 || c.query_id >= qn)
            return fail(GACT_HIP_ERANGE, "candidate %d: sequence id out of range", k);
        const int64_t rl = rs.h_offsets[c.ref_id + 1] - rs.h_offsets[c.ref_id];
        // darwin.cpp:222-224 clamps ref_pos to the read length; positions beyond
        // the reads would make GACT slice outside them.  query_pos is checked by
        // candidates_run*, which knows the strand and so the set the candidate reads
        if (c.ref_pos < 0 || c.ref_pos > rl || c.query_pos < 0)
            return fail(GACT_HIP_ERANGE, "candidate %d: position outside its read", k);
    }
    mark(2);
    if (reserve_candidates(sl, (size_t)n))
        return fail(GACT_HIP_ENOMEM, "device allocation failed");
    sl.n_cands = 0;
    sl.h_cands.assign(cands, cands + n);
    sl.checked_key[0] = -1;
    sl.routed_key[0] = -1;
    sl.cands_epoch = -1;
    reserve_host_staging(sl, (size_t)n);
    const gact_candidate *src = cands;
    if (n && sl.h_stage_cap >= (size_t)n) {
        // (nothing reads the staging array now: the one copy that does is waited for before its upload returns)
        memcpy(sl.h_stage, cands, (size_t)n * sizeof(gact_candidate));
        src = sl.h_stage;
    }
    mark(3);
    void *src_dev = (e->kernel_copies && src == sl.h_stage) ? device_view(sl.h_stage) : nullptr;
    if (n && src_dev) { if ((rc = bus_copy(sl.stream, src_dev, sl.cands.p, (size_t)n * sizeof(gact_candidate)))) return rc; }
    else if (n) HIP_TRY(hipMemcpyAsync(sl.cands.p, src, (size_t)n * sizeof(gact_candidate), hipMemcpyHostToDevice,
                                       sl.stream));
    mark(4);
    HIP_TRY(hipStreamSynchronize(sl.stream));
    mark(5);
    if (trace_up)
        fprintf(stderr, "[gact_hip] upload slot %d, %d candidates: set_device %ld, note_call %ld, checks %ld, reserve + host copy %ld, memcpyAsync %ld, sync %ld us\n",
                slot, n, tu[0], tu[1], tu[2], tu[3], tu[4], tu[5]);
    sl.n_cands = (size_t)n;
    return 0;
}

// one run on one slot's own stream, queues and workspace: seed + main launch(es) behind a reset of the counters
static int launch_run(gact_hip_engine *e, Slot &sl, int first, int n, int rc_from, int same_file, int strands)
{
    int rc = 0;
    HIP_TRY(hipMemsetAsync(sl.d_counter, 0, kCounterInts * sizeof(int), sl.stream));
    if ((rc = poison_ws(e, sl, 0))) return rc;
    HIP_TRY(hipEventRecord(sl.ev0, sl.stream));
    sl.two_phase = false;
    sl.merged_into = -1; sl.merged_callers = 1;
    if (n > 0) {
        rc = e->big_cb ? launch_big_extend(e, sl, first, n, rc_from, same_file)
           : (e->C == 20) ? launch_extend<20>(e, sl, first, n, rc_from, same_file, strands)
                          : launch_extend<32>(e, sl, first, n, rc_from, same_file, strands);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(sl.ev1, sl.stream));
    sl.timed = true;
    return 0;
}

// the runs of `batch` (two or more, same same_file) as ONE run on a merge slot
static int launch_merged(gact_hip_engine *e, const std::vector<RunReq *> &batch)
{
    Combiner &cb = e->cb;
    // a merge slot whose last launch has ended (its arrays are the right size already, most of the time); all busy: take turns
    int m = -1;
    for (int k = 0; k < cb.n_merge && m < 0; k++) {
        const Slot &c = e->slots[e->n_user + k];
        if (!c.stream || !c.timed || hipEventQuery(c.ev_ready) == hipSuccess) m = e->n_user + k;
    }
    (void)hipGetLastError();
    if (m < 0) m = e->n_user + (cb.next_merge++ % cb.n_merge);
    Slot &ms = e->slots[m];
    int rc = init_slot(e, ms);
    if (rc) return rc;
    gact::MergeSegs segs;
    int total = 0, strands = 0;
    segs.n_segs = (int)batch.size();
    for (size_t k = 0; k < batch.size(); k++) {
        const RunReq &r = *batch[k];
        Slot &src = e->slots[r.slot];
        segs.s[k] = gact::MergeSeg{src.cands.p, r.first, r.n, r.rc_from, total};
        total += r.n;
        if (r.first < r.rc_from) strands |= 1;
        if (r.first + r.n > r.rc_from) strands |= 2;
        // the caller's candidates are in place (an upload has waited for its copy; the device filter has not)
        HIP_TRY(hipEventRecord(src.ev_ready, src.stream));
        HIP_TRY(hipStreamWaitEvent(ms.stream, src.ev_ready, 0));
    }
    if (reserve_candidates(ms, (size_t)total)) return fail(GACT_HIP_ENOMEM, "device allocation failed (merged run of %d candidates)", total);
    ms.n_cands = (size_t)total;
    // this launch's own statistics record: a pooled one nobody refers to any more, else a new one
    std::shared_ptr<LaunchStats> rec;
    for (auto &r : cb.stats_pool) if (r.use_count() == 1) { rec = r; break; }
    if (!rec) {
        rec = std::make_shared<LaunchStats>();
        if (hipEventCreate(&rec->ev0) != hipSuccess || hipEventCreate(&rec->ev_mid) != hipSuccess || hipEventCreate(&rec->ev1) != hipSuccess ||
            hipEventCreateWithFlags(&rec->ev_done, hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc((void **)&rec->h_counter, 2 * kCounterInts * sizeof(int), hipHostMallocDefault) != hipSuccess)
            return fail(GACT_HIP_ENOMEM, "merged run: statistics record");
        cb.stats_pool.push_back(rec);
    }
    // (launch_run records the slot's ev0 / ev_mid / ev1: for this launch they are the record's)
    std::swap(ms.ev0, rec->ev0); std::swap(ms.ev_mid, rec->ev_mid); std::swap(ms.ev1, rec->ev1);
    struct SwapBack { Slot &ms; LaunchStats &r; ~SwapBack() { std::swap(ms.ev0, r.ev0); std::swap(ms.ev_mid, r.ev_mid); std::swap(ms.ev1, r.ev1); } };
    hipLaunchKernelGGL(gact::gather_kernel, dim3(std::max(1, std::min((total + 255) / 256, 2048))), dim3(256), 0, ms.stream, segs, ms.cands.p, total);
    HIP_TRY(hipGetLastError());
    {
        SwapBack back{ms, *rec};
        if ((rc = launch_run(e, ms, 0, total, gact::kCompInCand, batch[0]->same_file, strands))) return rc;
    }
    rec->two_phase = ms.two_phase; rec->wide = ms.wide; rec->lin = ms.lin; rec->aff = ms.aff; rec->overlapped = ms.overlapped;
    rec->lane = ms.lane; rec->side_used = ms.side_used; rec->routed_raw = ms.routed_raw; rec->roles = ms.roles;
    if (ms.two_phase) {
        // the counters as this launch leaves them, into the record's pinned array (a kernel copy: see bus_copy)
        void *hc = device_view(rec->h_counter);
        if (hc) {
            if ((rc = bus_copy(ms.stream, ms.d_counter, hc, kCounterInts * sizeof(int)))) return rc;
            if (ms.side_used && (rc = bus_copy(ms.stream, ms.side_counter, (char *)hc + kCounterInts * sizeof(int), kCounterInts * sizeof(int)))) return rc;
        } else {
            HIP_TRY(hipMemcpyAsync(rec->h_counter, ms.d_counter, kCounterInts * sizeof(int), hipMemcpyDeviceToHost, ms.stream));
            if (ms.side_used) HIP_TRY(hipMemcpyAsync(rec->h_counter + kCounterInts, ms.side_counter, kCounterInts * sizeof(int), hipMemcpyDeviceToHost, ms.stream));
        }
    }
    HIP_TRY(hipEventRecord(rec->ev_done, ms.stream));
    HIP_TRY(hipEventRecord(ms.ev1, ms.stream));          // ("is this slot still running?" is asked of the slot's own event)
    // every caller's records back into its own array, and its stream behind that
    for (size_t k = 0; k < batch.size(); k++) {
        const RunReq &r = *batch[k];
        Slot &dst = e->slots[r.slot];
        HIP_TRY(hipMemcpyAsync(dst.overlaps.p + r.first, ms.overlaps.p + segs.s[k].base, (size_t)r.n * sizeof(gact_overlap),
                               hipMemcpyDeviceToDevice, ms.stream));
    }
    HIP_TRY(hipEventRecord(ms.ev_ready, ms.stream));
    for (size_t k = 0; k < batch.size(); k++) {
        Slot &dst = e->slots[batch[k]->slot];
        HIP_TRY(hipStreamWaitEvent(dst.stream, ms.ev_ready, 0));
        dst.merged_into = m; dst.merged_callers = (int)batch.size();
        dst.merged_stats = rec;
        dst.merge_prev_gen = dst.merge_gen;
        dst.merge_gen = cb.merged_launches;
        dst.timed = true;
    }
    cb.merged_launches++; cb.merged_runs += (long)batch.size();
    return 0;
}

// A run enters here.  Whoever finds no leader becomes it: waits for the callers that are likely to follow, takes
// everything that is pending, launches, reports.
static int combined_submit(gact_hip_engine *e, int slot, int first, int n, int rc_from, int same_file)
{
    using clock = std::chrono::steady_clock;
    Combiner &cb = e->cb;
    RunReq req;
    req.slot = slot; req.first = first; req.n = n; req.rc_from = rc_from; req.same_file = same_file;
    std::unique_lock<std::mutex> lk(cb.mu);
    {
        Slot &sl = e->slots[slot];
        sl.last_thread = std::this_thread::get_id();
        sl.last_call = clock::now();
    }
    cb.pending.push_back(&req);
    cb.cv.notify_all();
    while (!req.launched) {
        if (cb.leader) { cb.cv.wait(lk); continue; }
        cb.leader = true;
        // ---- collect: slots another thread called for within the last few milliseconds, with nothing in flight and
        //      nothing pending, are about to submit (feeder threads behind their barrier, darwin.cpp:408-422)
        const auto me = std::this_thread::get_id();
        // (a thread's first run on this engine: it may be one of a group that has just been started -- every idle slot
        //  counts, whoever called for it last and however long ago, also the ones nobody has called for yet, and the
        //  window is three times as long: once per thread)
        thread_local const gact_hip_engine *submitted_to = nullptr;
        const bool first_run = submitted_to != e;
        submitted_to = e;
        // (... and three times as long for a thread whose last run was merged with others': they are known to be there, between
        //  their fetch and their next call -- line formatting and file writes in the reference's caller, 1-3 ms apart --, and a
        //  run that misses the launch of its peers costs both launches far more than the wait: the reference's caller with a
        //  group split 2 + 4 + 8 over its two calls printed 59 ms where 48 were due.  A peer that does not come back costs the
        //  long wait once: the next launch's members are the ones that did.)
        const bool had_peers = e->slots[slot].merged_into >= 0 && e->slots[slot].merged_callers > 1;
        // (the peers of that launch are expected however long ago they called last: after a pause of the whole group -- the
        //  reference's threads between their phases, a barrier -- the first arrivals found every other slot "stale", launched
        //  at once with whoever had made it, and the two halves of the group then took turns for good: 4 + 4 runs per launch,
        //  40 ms per step where 32 were due.  A peer that has left keeps its old number and is waited for once.)
        const long my_gen = had_peers ? e->slots[slot].merge_gen : -1;
        // (... and a group that did come apart joins again: the other half -- in flight on a launch of its own, the launch
        //  before that shared with this thread -- is waited for, once, for as long as a launch may last: two halves taking
        //  turns share the machine and need 40 ms a step each, for good; waiting for the other launch to end costs half of
        //  that, once)
        const long my_prev_gen = had_peers ? e->slots[slot].merge_prev_gen : -1;
        auto deadline = clock::now() + std::chrono::microseconds((first_run || had_peers) ? 3 * cb.window_us : cb.window_us);
        const auto rejoin_deadline = clock::now() + std::chrono::milliseconds(60);
        bool rejoined = false;
        for (;;) {
            const auto now = clock::now();
            int expected = 0;
            bool other_half_running = false;
            for (int k = 0; k < e->n_user; k++) {
                const Slot &o = e->slots[k];
                bool is_pending = false;
                for (const RunReq *r : cb.pending) is_pending |= r->slot == k;
                if (is_pending || o.last_thread == me || (o.last_thread == std::thread::id() && !first_run)) continue;
                if (o.in_flight) {
                    // a run in flight whose launch has ended: its thread is inside its fetch, or about to call it, and will
                    // be back in a moment (the other members of the launch this thread has just fetched from)
                    const Slot &src = o.merged_into >= 0 ? e->slots[o.merged_into] : o;
                    const hipEvent_t ev = o.merged_into >= 0 ? src.ev_ready : src.ev1;
                    if (!ev || hipEventQuery(ev) != hipSuccess) {
                        (void)hipGetLastError();
                        if (my_prev_gen >= 0 && o.merge_gen != my_gen && o.merge_prev_gen == my_prev_gen && o.merged_callers > 1) other_half_running = true;
                        continue;
                    }
                } else if (!first_run && now - o.last_call >= std::chrono::milliseconds(3) && !(my_gen >= 0 && o.merge_gen == my_gen)) {
                    continue;
                }
                expected++;
            }
            if (other_half_running && now < rejoin_deadline && (int)cb.pending.size() < gact::kMaxMerge) {
                // its launch ends, its threads fetch and call again: from then on they are "expected" like any peer
                rejoined = true;
                deadline = std::max(deadline, now + std::chrono::microseconds(3 * cb.window_us));
                cb.cv.wait_until(lk, now + std::chrono::microseconds(100));
                continue;
            }
            if (expected == 0 || now >= deadline || (int)cb.pending.size() >= gact::kMaxMerge) break;
            cb.cv.wait_until(lk, std::min(deadline, now + std::chrono::microseconds(50)));
        }
        if (rejoined) cb.rejoins++;
        std::vector<RunReq *> batch;
        batch.swap(cb.pending);
        for (RunReq *r : batch) e->slots[r->slot].in_flight = true;
        lk.unlock();
        // ---- launch: runs with the same same_file together, a run that is alone on its own slot
        (void)hipSetDevice(e->params.device_id);
        std::vector<bool> done(batch.size(), false);
        for (size_t a = 0; a < batch.size(); a++) {
            if (done[a]) continue;
            std::vector<RunReq *> group;
            for (size_t b = a; b < batch.size() && (int)group.size() < gact::kMaxMerge; b++)
                if (!done[b] && batch[b]->same_file == batch[a]->same_file) { group.push_back(batch[b]); done[b] = true; }
            int rc;
            if (group.size() == 1) {
                RunReq &r = *group[0];
                rc = launch_run(e, e->slots[r.slot], r.first, r.n, r.rc_from, r.same_file, -1);
            } else {
                rc = launch_merged(e, group);
            }
            for (RunReq *r : group) { r->status = rc; if (rc) r->err = g_err; }
        }
        lk.lock();
        for (RunReq *r : batch) r->launched = true;
        cb.leader = false;
        cb.cv.notify_all();
    }
    if (req.status) g_err = req.err;          // (the message was the leader's thread's)
    return req.status;
}

// candidates [first, first+n) of an uploaded list against the sets this run will read them from
// (index >= rc_from: GACT_SET_QUERY_RC); remembered, so that repeated runs of one range check once
static int check_candidate_range(gact_hip_engine *e, Slot &sl, int32_t first, int32_t n, int32_t rc_from)
{
    if (sl.h_cands.empty()) {
        // made by the device filter: valid for the sets it was made from, and for those only
        if (n > 0 && sl.cands_epoch != e->sets_epoch)
            return fail(GACT_HIP_EINVAL, "candidates_run: a read set was uploaded after the device filter made this slot's "
                                         "candidates; run gact_hip_dsoft_query again");
        return 0;
    }
    const int64_t key[4] = {first, n, rc_from, e->sets_epoch};
    if (!memcmp(key, sl.checked_key, sizeof key)) return 0;
    const SeqSet &rs = e->sets[GACT_SET_REF];
    for (int32_t k = first; k < first + n; k++) {
        const gact_candidate &c = sl.h_cands[(size_t)k];
        const SeqSet &qs = e->sets[k >= rc_from ? GACT_SET_QUERY_RC : GACT_SET_QUERY];
        if (c.ref_id >= rs.n || c.query_id >= qs.n)
            return fail(GACT_HIP_ERANGE, "candidate %d: sequence id out of range", k);
        const int64_t rl = rs.h_offsets[c.ref_id + 1] - rs.h_offsets[c.ref_id];
        const int64_t ql = qs.h_offsets[c.query_id + 1] - qs.h_offsets[c.query_id];
        if (c.ref_pos > rl || c.query_pos > ql)
            return fail(GACT_HIP_ERANGE, "candidate %d: position outside its read", k);
    }
    memcpy(sl.checked_key, key, sizeof key);
    return 0;
}

int gact_hip_candidates_run_mixed(gact_hip_engine *e, int slot, int32_t first, int32_t n, int32_t rc_from,
                                  int same_file)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    Slot &sl = e->slots[slot];
    if (first < 0 || n < 0 || (size_t)first + (size_t)n > sl.n_cands)
        return fail(GACT_HIP_EINVAL, "candidates_run: range [%d,%d) outside the %zu candidates of slot %d", first,
                    first + n, sl.n_cands, slot);
    if ((rc = set_device(e))) return rc;
    const bool need_f = first < rc_from, need_r = first + n > rc_from;
    if (n > 0 && (e->sets[GACT_SET_REF].n == 0 || (need_f && e->sets[GACT_SET_QUERY].n == 0) ||
                  (need_r && e->sets[GACT_SET_QUERY_RC].n == 0)))
        return fail(GACT_HIP_EINVAL, "candidates_run: read sets not uploaded");
    if ((rc = check_candidate_range(e, sl, first, n, rc_from))) return rc;
    if (e->cb.enabled && e->n_user > 1 && n > 0) return combined_submit(e, slot, first, n, rc_from, same_file);
    return launch_run(e, sl, first, n, rc_from, same_file, -1);
}

int gact_hip_candidates_run_range(gact_hip_engine *e, int slot, int32_t first, int32_t n, int complement,
                                  int same_file)
{
    return gact_hip_candidates_run_mixed(e, slot, first, n, complement ? 0 : 0x7fffffff, same_file);
}

int gact_hip_candidates_run(gact_hip_engine *e, int slot, int32_t n, int complement, int same_file)
{
    return gact_hip_candidates_run_range(e, slot, 0, n, complement, same_file);
}

int gact_hip_candidates_fetch(gact_hip_engine *e, int slot, int32_t n, gact_overlap *out)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    Slot &sl = e->slots[slot];
    if (n < 0 || (size_t)n > sl.n_cands || (n > 0 && !out))
        return fail(GACT_HIP_EINVAL, "candidates_fetch: bad arguments (slot %d holds %zu candidates)", slot, sl.n_cands);
    if ((rc = set_device(e))) return rc;
    note_call(e, slot);
    // (the slot's stream is behind whatever launch carried its last run, merged or not: the copy below waits for it)
    struct Landed {            // the run is over when this call returns, whichever way
        gact_hip_engine *e; int slot;
        ~Landed()
        {
            if (!e->cb.enabled || e->n_user < 2) return;
            std::lock_guard<std::mutex> lk(e->cb.mu);
            e->slots[slot].in_flight = false;
            e->slots[slot].last_thread = std::this_thread::get_id();
            e->slots[slot].last_call = std::chrono::steady_clock::now();
        }
    } landed{e, slot};
    const size_t bytes = (size_t)n * sizeof(gact_overlap);
    if (n > 0 && sl.reg_out && (char *)out >= (char *)sl.reg_out &&
        (char *)out + bytes <= (char *)sl.reg_out + sl.reg_bytes) {
        void *out_dev = e->kernel_copies ? device_view(out) : nullptr;
        if (out_dev) { if ((rc = bus_copy(sl.stream, sl.overlaps.p, out_dev, bytes))) return rc; }
        else HIP_TRY(hipMemcpyAsync(out, sl.overlaps.p, bytes, hipMemcpyDeviceToHost, sl.stream));
        HIP_TRY(hipStreamSynchronize(sl.stream));
        return 0;
    }
    static const bool trace_fetch = opt_env("trace_upload") != nullptr;       // (the same switch as the upload's trace)
    const auto tf0 = std::chrono::steady_clock::now();
    long tf[3] = {0, 0, 0};
    auto fmark = [&](int k) { if (trace_fetch) tf[k] = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - tf0).count(); };
    if ((size_t)n > sl.h_records_cap) reserve_host_staging(sl, (size_t)n);     // (none to be had: straight into the caller's buffer)
    gact_overlap *dst = sl.h_records_cap >= (size_t)n ? sl.h_records : out;
    void *dst_dev = (e->kernel_copies && dst == sl.h_records) ? device_view(sl.h_records) : nullptr;
    if (n && dst_dev) { if ((rc = bus_copy(sl.stream, sl.overlaps.p, dst_dev, (size_t)n * sizeof(gact_overlap)))) return rc; }
    else if (n) HIP_TRY(hipMemcpyAsync(dst, sl.overlaps.p, (size_t)n * sizeof(gact_overlap), hipMemcpyDeviceToHost,
                                       sl.stream));
    fmark(0);
    HIP_TRY(hipStreamSynchronize(sl.stream));
    fmark(1);
    if (n && dst != out) memcpy(out, dst, (size_t)n * sizeof(gact_overlap));
    fmark(2);
    if (trace_fetch)
        fprintf(stderr, "[gact_hip] fetch slot %d, %d records: copy queued %ld, stream idle %ld, host copy %ld us (since the call began)\n", slot, n, tf[0], tf[1], tf[2]);
    return 0;
}

int gact_hip_register_output(gact_hip_engine *e, int slot, void *buf, int64_t bytes)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if (!buf || bytes <= 0) return fail(GACT_HIP_EINVAL, "register_output: bad arguments");
    if ((rc = set_device(e))) return rc;
    Slot &sl = e->slots[slot];
    HIP_TRY(hipStreamSynchronize(sl.stream));
    if (sl.reg_out) { (void)hipHostUnregister(sl.reg_out); (void)hipGetLastError(); sl.reg_out = nullptr; sl.reg_bytes = 0; }
    HIP_TRY(hipHostRegister(buf, (size_t)bytes, hipHostRegisterDefault));
    sl.reg_out = buf; sl.reg_bytes = (size_t)bytes;
    return 0;
}

int gact_hip_unregister_output(gact_hip_engine *e, int slot)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if ((rc = set_device(e))) return rc;
    Slot &sl = e->slots[slot];
    if (!sl.reg_out) return 0;
    HIP_TRY(hipStreamSynchronize(sl.stream));
    void *p = sl.reg_out;
    sl.reg_out = nullptr; sl.reg_bytes = 0;
    HIP_TRY(hipHostUnregister(p));
    return 0;
}

int gact_hip_extend_candidates(gact_hip_engine *e, int slot, int32_t n, const gact_candidate *cands,
                               int complement, int same_file, gact_overlap *out)
{
    int rc = gact_hip_candidates_upload(e, slot, n, cands);
    if (rc) return rc;
    if ((rc = gact_hip_candidates_run(e, slot, n, complement, same_file))) return rc;
    return gact_hip_candidates_fetch(e, slot, n, out);
}

int gact_hip_sync(gact_hip_engine *e, int slot)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if ((rc = set_device(e))) return rc;
    HIP_TRY(hipStreamSynchronize(e->slots[slot].stream));
    if (e->cb.enabled && e->n_user > 1) {
        std::lock_guard<std::mutex> lk(e->cb.mu);
        e->slots[slot].in_flight = false;
    }
    return 0;
}

int gact_hip_last_kernel_ms(gact_hip_engine *e, int slot, float *ms)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if (!ms) return fail(GACT_HIP_EINVAL, "ms is NULL");
    if (!e->slots[slot].timed) return fail(GACT_HIP_EINVAL, "no kernel has been launched on slot %d", slot);
    if ((rc = set_device(e))) return rc;
    if (e->slots[slot].merged_into >= 0) {
        // a merged run: the launch's own record (the merge slot's events may be another launch's by now)
        const std::shared_ptr<LaunchStats> rec = e->slots[slot].merged_stats;
        if (!rec) return fail(GACT_HIP_EINVAL, "slot %d: no statistics of its merged run", slot);
        HIP_TRY(hipEventSynchronize(rec->ev1));
        HIP_TRY(hipEventElapsedTime(ms, rec->ev0, rec->ev1));
        return 0;
    }
    Slot &sl = e->slots[slot];
    HIP_TRY(hipEventSynchronize(sl.ev1));
    HIP_TRY(hipEventElapsedTime(ms, sl.ev0, sl.ev1));
    return 0;
}

int gact_hip_last_run_stats(gact_hip_engine *e, int slot, gact_hip_run_stats *st)
{
    int rc = check_slot(e, slot);
    if (rc) return rc;
    if (!st) return fail(GACT_HIP_EINVAL, "stats is NULL");
    if (!e->slots[slot].timed) return fail(GACT_HIP_EINVAL, "no kernel has been launched on slot %d", slot);
    if ((rc = set_device(e))) return rc;
    memset(st, 0, sizeof *st);
    if (e->slots[slot].merged_into >= 0) {
        // a run that was merged with other callers' runs: the figures of THAT launch, from its own immutable record -- the
        // merge slot's events and counters are re-recorded as soon as another leader takes the slot
        const std::shared_ptr<LaunchStats> rec = e->slots[slot].merged_stats;
        if (!rec) return fail(GACT_HIP_EINVAL, "slot %d: no statistics of its merged run", slot);
        st->merged_callers = e->slots[slot].merged_callers;
        st->overlapped_seeding = rec->overlapped ? 1 : 0;
        st->critical_lane = rec->lane ? 1 : 0;
        st->role_waves = rec->roles;
        HIP_TRY(hipEventSynchronize(rec->ev_done));
        HIP_TRY(hipEventElapsedTime(&st->total_ms, rec->ev0, rec->ev1));
        st->main_ms = st->total_ms;
        st->packed16 = rec->two_phase ? (rec->wide ? 3 : e->split ? 2 : 1) : 0;
        st->seed_packed16 = (rec->two_phase && e->seed16) ? 1 : 0;
        st->tagged_pointers = (rec->two_phase && e->tagged) ? 1 : 0;
        st->linear_gap = (rec->two_phase && rec->lin) ? 1 : (rec->two_phase && rec->aff) ? 2 : 0;
        st->raw_candidates = rec->routed_raw;
        if (rec->two_phase) {
            HIP_TRY(hipEventElapsedTime(&st->seed_ms, rec->ev0, rec->ev_mid));
            HIP_TRY(hipEventElapsedTime(&st->main_ms, rec->ev_mid, rec->ev1));
            const int *c = rec->h_counter;
            for (int b = 0; b < gact::kBuckets; b++) st->handed_off += c[8 + b] + c[kMoreCount + b];
            memcpy(&st->seed_cells, &c[2], sizeof(int64_t));
            st->band_redos = c[6];
            if (rec->side_used) {
                const int *cs = rec->h_counter + kCounterInts;
                for (int b = 0; b < gact::kBuckets; b++) st->handed_off += cs[8 + b];
                int64_t sc = 0;
                memcpy(&sc, &cs[2], sizeof(int64_t));
                st->seed_cells += sc;
            }
        }
        return 0;
    }
    Slot &sl = e->slots[slot];
    st->merged_callers = 1;
    st->overlapped_seeding = sl.overlapped ? 1 : 0;
    st->critical_lane = sl.lane ? 1 : 0;
    st->role_waves = sl.roles;
    HIP_TRY(hipEventSynchronize(sl.ev1));
    HIP_TRY(hipEventElapsedTime(&st->total_ms, sl.ev0, sl.ev1));
    st->main_ms = st->total_ms;
    st->packed16 = sl.two_phase ? (sl.wide ? 3 : e->split ? 2 : 1) : 0;
    st->seed_packed16 = (sl.two_phase && e->seed16) ? 1 : 0;
    st->tagged_pointers = (sl.two_phase && e->tagged) ? 1 : 0;
    st->linear_gap = (sl.two_phase && sl.lin) ? 1 : (sl.two_phase && sl.aff) ? 2 : 0;
    st->raw_candidates = sl.routed_raw;
    if (sl.two_phase) {
        HIP_TRY(hipEventElapsedTime(&st->seed_ms, sl.ev0, sl.ev_mid));
        HIP_TRY(hipEventElapsedTime(&st->main_ms, sl.ev_mid, sl.ev1));
        int c[kCounterInts];
        HIP_TRY(hipMemcpy(c, sl.d_counter, sizeof c, hipMemcpyDeviceToHost));
        st->handed_off = 0;
        for (int b = 0; b < gact::kBuckets; b++) st->handed_off += c[8 + b] + c[kMoreCount + b];
        memcpy(&st->seed_cells, &c[2], sizeof(int64_t));
        st->band_redos = c[6];
        if (sl.side_used) {                 // launches beside: their share of both figures
            HIP_TRY(hipMemcpy(c, sl.side_counter, sizeof c, hipMemcpyDeviceToHost));
            for (int b = 0; b < gact::kBuckets; b++) st->handed_off += c[8 + b];
            int64_t sc = 0;
            memcpy(&sc, &c[2], sizeof(int64_t));
            st->seed_cells += sc;
        }
    }
    return 0;
}

int gact_hip_prepare(gact_hip_engine *e, int32_t expected_candidates)
{
    if (!e) return fail(GACT_HIP_EINVAL, "prepare: engine is NULL");
    if (expected_candidates < 0) return fail(GACT_HIP_EINVAL, "prepare: expected_candidates %d", expected_candidates);
    int rc = set_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(e->cb.mu);
    const int n_slots = e->n_user + (e->cb.n_merge ? 1 : 0);
    for (int k = 0; k < n_slots; k++) {
        Slot &sl = e->slots[k];
        if ((rc = init_slot(e, sl))) return rc;
        // a merge slot holds a whole job, a caller's slot its share of it (twice over: shares are uneven)
        const size_t n = k >= e->n_user ? (size_t)expected_candidates
                                        : std::min<size_t>((size_t)expected_candidates, 2 * (size_t)expected_candidates / (size_t)e->n_user + 1024);
        if (n && reserve_candidates(sl, n)) return fail(GACT_HIP_ENOMEM, "prepare: device allocation failed (%zu candidates)", n);
        if (n && k < e->n_user) reserve_host_staging(sl, n);
        if (e->p16 && e->lin && e->split && e->C == 20) {
            // every stream this slot launches on sees the chain kernels once, with nothing to do: code and scratch are in place
            if ((rc = ensure_aux_stream(sl))) return rc;
            if (!n && reserve_candidates(sl, 1024)) return fail(GACT_HIP_ENOMEM, "prepare: device allocation failed");
            const SeqSet &rs = e->sets[GACT_SET_REF];
            if (!rs.d_raw) continue;                                   // (no read set yet: nothing valid to pass)
            const gact::SeqSetDev d = rs.dev(false);
            HIP_TRY(hipMemsetAsync(sl.d_counter, 0, kCounterInts * sizeof(int), sl.stream));
            HIP_TRY(hipMemsetAsync(sl.d_counter + 7, 0xff, sizeof(int), sl.stream));       // "the second set is open" (and empty)
            HIP_TRY(hipStreamSynchronize(sl.stream));
            const Lane own = main_lane(e, sl);
            gact::ChainQueues q = queues(own, sl);
            q.list = sl.order.p; q.list_n = 0;
            gact::ChainQueues q2 = q;                                  // the two-set variant: both sets empty, the flag set
            {
                const gact::ChainQueues s2 = second_queues(own, sl);
                q2.more_flag = sl.d_counter + 7;
                q2.more_count = s2.bucket_count; q2.more_pop = s2.bucket_pop; q2.more_live = s2.live;
            }
            for (hipStream_t st : {sl.stream, sl.aux_stream}) {
                hipLaunchKernelGGL((gact::seed_p16_kernel<20, false, 1>), dim3(1), dim3(gact::kBlockThreads), 0, st, e->kp, e->kc, d, d, d,
                                   sl.cands.p, 0, 0, 0, 0, sl.overlaps.p, q, sl.d_ws);
                hipLaunchKernelGGL((gact::extend_p16_kernel<gact::SplitLayoutLin<7, 13>, false>), dim3(1), dim3(gact::kBlockThreads), 0, st, e->kp,
                                   e->kc, d, d, d, 0, sl.overlaps.p, q, sl.d_ws);
                hipLaunchKernelGGL((gact::extend_p16_kernel<gact::SplitLayoutLin<7, 13>, false, true>), dim3(1), dim3(gact::kBlockThreads), 0, st,
                                   e->kp, e->kc, d, d, d, 0, sl.overlaps.p, q2, sl.d_ws);
                hipLaunchKernelGGL((gact::extend_coop_kernel<gact::SplitLayoutLin<7, 13>, false>), dim3(1), dim3(gact::kBlockThreads), 0, st,
                                   e->kp, e->kc, d, d, d, 0, sl.overlaps.p, q, sl.d_ws);
                hipLaunchKernelGGL((gact::extend_coop_kernel<gact::SplitLayoutLin<7, 13>, true>), dim3(1), dim3(gact::kBlockThreads), 0, st,
                                   e->kp, e->kc, d, d, d, 0, sl.overlaps.p, q2, sl.d_ws);
                if (e->role_grid_blocks > 0) {
                    hipLaunchKernelGGL((gact::extend_roles_kernel<gact::SplitLayoutLin<7, 13>, false>), dim3(1), dim3(gact::kRoleThreads), 0, st,
                                       e->kp, e->kc, d, d, d, 0, sl.overlaps.p, q, sl.d_ws);
                    hipLaunchKernelGGL((gact::extend_roles_kernel<gact::SplitLayoutLin<7, 13>, true>), dim3(1), dim3(gact::kRoleThreads), 0, st,
                                       e->kp, e->kc, d, d, d, 0, sl.overlaps.p, q2, sl.d_ws);
                }
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipStreamSynchronize(sl.stream));
            HIP_TRY(hipStreamSynchronize(sl.aux_stream));
        }
    }
    return 0;
}

int gact_hip_set_option(gact_hip_engine *e, const char *name, int32_t value)
{
    if (!e || !name) return fail(GACT_HIP_EINVAL, "set_option: NULL argument");
    const std::string n(name);
    if (n == "overlap_seed") e->overlap_seed = value != 0;
    // the caller keeps several runs in flight on this engine (steps of a pipeline, one slot each): what the engine otherwise
    // guesses from the other slots' events at launch time -- and guesses differently from run to run for the first launches
    else if (n == "runs_in_flight") e->caller_keeps_runs_in_flight.store(value != 0);
    else if (n == "combine") {
        std::lock_guard<std::mutex> lk(e->cb.mu);
        e->cb.enabled = value != 0 && e->cb.n_merge > 0;
    } else if (n == "combine_window_us") {
        std::lock_guard<std::mutex> lk(e->cb.mu);
        e->cb.window_us = std::max(0, (int)value);
    } else if (n == "overlap_big") {
        e->overlap_big = value != 0;
    } else if (n == "shared_twelfths") {
        e->shared_twelfths = std::max(1, std::min((int)value, 12));
    } else if (n == "lone_lane") {
        e->lone_lane = std::max(-255, std::min((int)value, 255));
    } else if (n == "coop") {
        if (value != 0 && !(e->lin && e->split && e->C == 20)) return fail(GACT_HIP_EINVAL, "set_option: this engine has no split linear-gap launch");
        e->coop = value == 1 ? 1 : value == 0 ? -1 : 0;
    } else if (n == "roles") {
        if (value != 0 && e->role_grid_blocks <= 0) return fail(GACT_HIP_EINVAL, "set_option: this engine was created without the role launch");
        e->roles = value != 0;
    } else {
        for (const OptionDef &o : kOptions)
            if (n == o.name) return fail(GACT_HIP_EINVAL, "set_option: '%s' is read once, in gact_hip_create%s%s", name, o.env ? ", from " : "", o.env ? o.env : "");
        return fail(GACT_HIP_EINVAL, "set_option: unknown option '%s'", name);
    }
    return 0;
}

// The launch plan (gact_policy.hpp) an engine of these parameters makes for a pass of `count` candidates, on a machine of
// `compute_units` CUs at the kernels' nominal occupancies (three blocks of four waves per CU for the main and the linear-gap
// seed launches, two for the other seed launches, one role block): no engine, no device.  flags: bit 0 the sets hold bytes
// other than A/C/G/T (raw-byte kernels), bit 1 the launch shares the machine, bit 2 the role launch is switched on, bit 3 / 4
// cooperative walks always / never, bit 5 overlapped seeding also beyond four chains per tile slot ("overlap_big" 1), bit 6 the lone
// mix ("lone_lane" 48).
int64_t gact_hip_plan_describe(const gact_hip_params *p, int32_t compute_units, int32_t count, int32_t flags, char *buf, int64_t cap)
{
    if (!p || compute_units < 1 || count < 0) return fail(GACT_HIP_EINVAL, "plan_describe: bad arguments");
    gact_hip_engine e;
    e.params = *p;
    const bool big = derive_kernel_flags(&e);
    std::string t;
    if (big) t = "{\"sequence\": \"one wave per tile (gact_big.hpp)\"}";
    else {
        e.prop.multiProcessorCount = compute_units;
        e.kp.ws_words = (e.C == 20) ? gact::Geometry<20>::kWsWords : gact::Geometry<32>::kWsWords;
        e.grid_blocks = e.lin_grid_blocks = e.wide_lin_grid_blocks = e.aff_grid_blocks = e.seed_lin_grid_blocks = 3 * compute_units;
        e.seed_grid_blocks = 2 * compute_units;
        e.role_grid_blocks = (e.lin && e.split && e.C == 20) ? compute_units : 0;
        if (flags & 4) e.roles = e.role_grid_blocks > 0;
        e.coop = (flags & 8) ? 1 : (flags & 16) ? -1 : 0;
        if (flags & 32) e.overlap_big = true;
        if (flags & 64) e.lone_lane = 48;
        gact_policy::Inputs in;
        in.count = count; in.raw = (flags & 1) != 0; in.shared_machine = (flags & 2) != 0;
        t = gact_policy::describe(gact_policy::plan_pass(policy_caps(&e), in));
    }
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>((size_t)cap - 1, t.size());
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return (int64_t)t.size() + 1;
}

int64_t gact_hip_options_describe(char *buf, int64_t cap)
{
    std::string t;
    for (const OptionDef &o : kOptions) {
        t += o.name; t += " | "; t += o.env ? o.env : "-"; t += " | ";
        t += o.when == 'l' ? "live (gact_hip_set_option)" : "gact_hip_create"; t += " | ";
        t += o.klass == 'k' ? "kernels" : o.klass == 's' ? "scheduling" : "diagnostic (-DGACT_EXPERIMENTS)";
        t += " | "; t += o.doc; t += "\n";
    }
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>((size_t)cap - 1, t.size());
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return (int64_t)t.size() + 1;
}

void *gact_hip_device_overlaps(gact_hip_engine *e, int slot)
{
    if (check_slot(e, slot)) return nullptr;
    return e->slots[slot].overlaps.p;
}

void *gact_hip_stream(gact_hip_engine *e, int slot)
{
    if (check_slot(e, slot)) return nullptr;
    return (void *)e->slots[slot].stream;
}

int gact_hip_measure_valu_rate(gact_hip_engine *e, double *lane_ops_per_s)
{
    if (!e || !lane_ops_per_s) return fail(GACT_HIP_EINVAL, "NULL argument");
    int rc = set_device(e);
    if (rc) return rc;
    Slot &sl = e->slots[0];
    const int iters = 2048, waves_per_simd = 8;
    const int blocks = e->prop.multiProcessorCount * waves_per_simd;       // 256 threads = one wave per SIMD of a CU
    const size_t n_waves = (size_t)blocks * (gact::kBlockThreads / 64);
    unsigned long long *d_clocks = nullptr;
    HIP_TRY(hipMalloc((void **)&d_clocks, n_waves * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(n_waves);
    double best = 1e300;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(gact::valu_probe_kernel, dim3(blocks), dim3(gact::kBlockThreads), 0, sl.stream, iters,
                           12345 + rep, sl.d_flags, d_clocks);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(sl.stream) != hipSuccess ||
            hipMemcpy(h.data(), d_clocks, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipFree(d_clocks);
            return fail(GACT_HIP_EDEVICE, "valu probe failed");
        }
        std::nth_element(h.begin(), h.begin() + h.size() / 2, h.end());
        if (rep) best = std::min(best, (double)h[h.size() / 2]);
    }
    (void)hipFree(d_clocks);
    // median wave: clocks between two of its issues; eight such waves share a SIMD
    const double wave_interval = best / ((double)iters * 32.0);
    const double simd_interval = wave_interval / waves_per_simd;
    *lane_ops_per_s = (double)e->prop.multiProcessorCount * 4.0 * 64.0 * ((double)e->prop.clockRate * 1e3) / simd_interval;
    return 0;
}

#include "dsoft_engine.hpp"      // gact_hip_dsoft_build / _query / candidates_download
#include "gact_gather.hpp"       // gact_hip_comm_*: the RCCL gather of a sharded job

#ifdef GACT_STAMPS
// diagnostic build: per-wave (start, queues empty, end, iterations) of the last main launch
int gact_hip_debug_timeline(gact_hip_engine *e, unsigned long long *out, int n_waves)
{
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(gact::g_timeline), (size_t)std::min(n_waves, 4096) * 4 * sizeof(unsigned long long)));
    return 0;
}

// diagnostic build: shader clocks each of those waves lived (with the timeline's 100 MHz stamps: the clock the chip held)
int gact_hip_debug_wave_cycles(gact_hip_engine *e, unsigned long long *out, int n_waves)
{
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(gact::g_wave_cycles), (size_t)std::min(n_waves, 4096) * sizeof(unsigned long long)));
    return 0;
}

// diagnostic build: read and clear the per-phase clock totals of extend_p16_kernel
int gact_hip_debug_stamps(gact_hip_engine *e, unsigned long long *out8)
{
    int rc = set_device(e);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(gact::g_stamps), 8 * sizeof(unsigned long long)));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gact::g_stamps), z, sizeof z));
    unsigned long long y[8];
    HIP_TRY(hipMemcpyFromSymbol(y, HIP_SYMBOL(gact::g_stamps2), sizeof y));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gact::g_stamps2), z, sizeof z));
    if (y[3]) printf("  load_pair: fill %llu, loads %llu, lds-writes %llu clocks/iter\n", y[0] / y[3], y[1] / y[3], y[2] / y[3]);
    if (y[6]) printf("  load_pair_packed: fill + staging %llu, cutting %llu clocks/iter\n", y[4] / y[6], y[5] / y[6]);
    unsigned long long rf = 0, zero = 0;
    HIP_TRY(hipMemcpyFromSymbol(&rf, HIP_SYMBOL(gact::g_refill_clocks), sizeof rf));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gact::g_refill_clocks), &zero, sizeof zero));
    if (y[6] && rf) printf("  walker region refills (upper bound, -DGACT_STAMPS_REFILL): %llu clocks/iter\n", rf / y[6]);
#ifdef GACT_STAMPS_FLUSH
    {
        unsigned long long fc[3], fz[3] = {0, 0, 0};
        HIP_TRY(hipMemcpyFromSymbol(fc, HIP_SYMBOL(gact::g_flush_clocks), sizeof fc));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gact::g_flush_clocks), fz, sizeof fz));
        if (fc[1]) printf("  pointer flushes of the split pass: %llu, %.0f clocks each (clock readings included), of which from the first store on: %.0f\n",
                          fc[1], (double)fc[0] / fc[1], (double)fc[2] / fc[1]);
    }
#endif
    {
        unsigned long long cc[4], cz[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpyFromSymbol(cc, HIP_SYMBOL(gact::g_coop_counts), sizeof cc));
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gact::g_coop_counts), cz, sizeof cz));
        if (cc[0]) printf("  cooperative walks: %llu batches, %.1f jobs and %.1f loop trips per batch, %.1f lanes walking per trip\n", cc[0],
                          (double)cc[3] / cc[0], (double)cc[1] / cc[0], cc[1] ? (double)cc[2] / cc[1] : 0.0);
    }
    unsigned long long wc[5], wz[5] = {0, 0, 0, 0, 0};
    HIP_TRY(hipMemcpyFromSymbol(wc, HIP_SYMBOL(gact::g_walk_counts), sizeof wc));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(gact::g_walk_counts), wz, sizeof wz));
    if (wc[0]) printf("  look-ahead walker: %llu walks, %.1f columns, %.1f team iterations and %.1f refills per walk; %.1f loop trips per wave pass\n",
                      wc[0], (double)wc[4] / wc[0], (double)wc[1] / wc[0], (double)wc[2] / wc[0], y[6] ? (double)wc[3] / y[6] : 0.0);
    return 0;
}
#endif

int gact_hip_format_overlap(const gact_overlap *o, const char *ref_name, const char *query_name, char *buf,
                            int32_t cap)
{
    if (!o || !ref_name || !query_name || !buf || cap <= 0) return fail(GACT_HIP_EINVAL, "format: bad arguments");
    // exact bytes of gact.cpp:214-224
    return snprintf(buf, (size_t)cap, "ref_id: %s, query_id: %s, ab: %d, ae: %d, bb: %d, be: %d, score: %d, comp: %d\n",
                    ref_name, query_name, o->ab, o->ae, o->bb, o->be, o->score, o->comp);
}

}  // extern "C"
