/*
 * gact_hip.h -- C-ABI of the MI355X-native GACT tiled-alignment engine.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch
 * types.  Each entry point names the reference interface it stands under
 * (file:line in Tongdongq/darwin-gpu); the C++ shim that keeps the
 * reference's own gact.h / align.h signatures on top of it is
 * darwin-gpu_amd/host/, and INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative GACT_HIP_E* code on
 *     failure; gact_hip_last_error() returns the calling thread's message.
 *     (The reference has no return codes: cudaSafeCall prints and exit(-1)s,
 *     cuda_header.h:309-319; the shim reproduces that.)
 *   - `slot` is the feeder-thread index the reference passes around as a
 *     GPU_storage (gact.h:51-67, darwin.cpp:625): each slot owns a HIP stream
 *     and its buffers, so N host threads may call concurrently with N
 *     different slots.
 *   - caller owns every host buffer; the engine owns all device memory.
 *   - there is no CPU fallback: without a usable gfx950 device
 *     gact_hip_create fails.
 */
#ifndef GACT_HIP_H
#define GACT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GACT_HIP_OK          0
#define GACT_HIP_EINVAL     -1   /* bad argument / unsupported parameter   */
#define GACT_HIP_EDEVICE    -2   /* HIP runtime error                      */
#define GACT_HIP_ENOMEM     -3
#define GACT_HIP_ERANGE     -4   /* descriptor points outside a sequence   */

/* Largest tile_size: the reference's CPU AlignWithBT asserts ref_len, query_len < 2049 (align.h:19, align.cpp:66-67); its
 * CUDA kernel stops at 324 (cuda_header.h:45), and every caller in the reference passes tile_size = 320 (params.cfg:22).
 * Up to GACT_HIP_FAST_TILE the register-tiled kernels run (packed int16 where the scoring allows); beyond it, one wave
 * per tile with the pointer matrix in HBM (csrc/gact_big.hpp): the whole interface, at a fraction of the speed. */
#define GACT_HIP_FAST_TILE  512
#define GACT_HIP_MAX_TILE   2048

/* traceback states, align.h:23 */
#define GACT_STATE_Z 0
#define GACT_STATE_D 1   /* '-' in ref, consumes a query base (gact.cpp:126-130) */
#define GACT_STATE_I 2   /* '-' in query, consumes a ref base (gact.cpp:121-125) */
#define GACT_STATE_M 3

/* which resident sequence set a descriptor addresses */
#define GACT_SET_REF       0   /* reference_seqs   (gact.cpp:40) */
#define GACT_SET_QUERY     1   /* reads_seqs       (gact.cpp:42) */
#define GACT_SET_QUERY_RC  2   /* rev_reads_seqs   (gact.cpp:43) */
#define GACT_NUM_SETS      3

typedef struct gact_hip_engine gact_hip_engine;

/*
 * Replaces the argument list of GPU_init (gact.h:85-87, cuda_host.cu:193-237)
 * plus the globals gact.cpp reads (gact.h:25-32): params.cfg values.
 * Scoring domain: mismatch <= 0, gap_open <= 0, gap_extend <= 0 (the
 * reference's traceback reads out of bounds otherwise, align.cpp:211-226).
 */
typedef struct {
    int32_t tile_size;                  /* params.cfg GACT_extend.tile_size   (<= GACT_HIP_MAX_TILE) */
    int32_t tile_overlap;               /* early_terminate = tile_size - tile_overlap, gact.cpp:94 */
    int32_t match, mismatch, gap_open, gap_extend;
    int32_t first_tile_score_threshold; /* gact.cpp:107 */
    int32_t device_id;                  /* reference hard-wires 0, cuda_host.cu:195 */
    int32_t n_slots;                    /* = num_threads of GPU_init */
    int32_t max_blocks;                 /* 0: every launch may fill the device (each slot then owns a traceback workspace for that
                                           many resident tiles, 1.3 GB at the reference's parameters); > 0: no launch of this
                                           engine uses more blocks, and the workspaces are sized for that -- an engine that only
                                           ever sees a handful of tiles or candidates per call (the shim's AlignWithBT / GACT) */
} gact_hip_params;

/* GPU_init: cuda_host.cu:193-237 */
int gact_hip_create(const gact_hip_params *params, gact_hip_engine **out);
/* GPU_close: cuda_host.cu:239-258 */
void gact_hip_destroy(gact_hip_engine *e);
const char *gact_hip_last_error(void);

/* device facts used by the measurement harness (rocminfo values) */
typedef struct {
    int32_t compute_units;
    int32_t clock_mhz;
    int32_t waves_per_cu;     /* resident waves per CU of the DP kernel */
    int32_t wave_size;
    int64_t hbm_bytes;
    char    arch[32];
} gact_hip_device_info;
int gact_hip_get_device_info(gact_hip_engine *e, gact_hip_device_info *info);

/*
 * Makes a read set resident in HBM (replaces the per-batch substr + interleave
 * + 2 H2D copies + gasal_pack_kernel of gact.cpp:400-407, cuda_host.cu:85-169
 * and cuda_header.h:47-90).  `concat` holds n_seqs sequences back to back,
 * sequence s = concat[offsets[s] .. offsets[s+1]).  Bytes may be ASCII
 * (CPU build) or the GPU build's 0..3 recode (A0 C1 T2 G3, darwin.cpp:320-332).
 * Bases are kept 2 bits each, 16 per uint32, when every byte is one of
 * A/C/G/T (or 0..3); a set holding anything else (N, lower case) is also kept
 * as raw bytes and aligned by raw byte equality like align.cpp:134.
 */
int gact_hip_upload_seqs(gact_hip_engine *e, int which_set,
                         const uint8_t *concat, const int64_t *offsets, int32_t n_seqs);

/*
 * GACT_SET_QUERY_RC := reverse complement of every sequence of GACT_SET_QUERY, made on the device
 * (stands where darwin.cpp:110-147 builds rev_reads_seqs on the host; saves the third upload).
 * Complement as darwin.cpp:122-142: a<->t, c<->g in either case, n and N stay; any other byte is an
 * error (the reference prints "Bad Nt char" and exits).
 */
int gact_hip_derive_revcomp(gact_hip_engine *e);

/* ---- per-tile batch: what Align_Batch_GPU does (cuda_host.cu:23-190) ---- */

/*
 * One tile = one AlignWithBT call (align.cpp:60-63).  The tile slices are
 * ref[ref_off, ref_off+ref_len) of sequence ref_id in GACT_SET_REF and
 * query[query_off, ...) of sequence query_id in `query_set`.
 * `reverse` and `first` have AlignWithBT's meaning (align.cpp:130: reverse
 * reads the slice back to front).  NOTE Align_Batch_GPU's reverses[] has the
 * opposite sense (cuda_host.cu:92-142 byte-reverses when reverses[t]==0);
 * the shim flips it.
 */
typedef struct {
    int32_t ref_id, query_id;
    int32_t ref_off, query_off;
    int32_t ref_len, query_len;      /* 0..tile_size; ref_len < 0 marks an idle slot (cuda_host.cu:70) */
    uint8_t reverse, first, query_set, pad;
} gact_tile;

/* out[0..4] of the reference's per-tile int block (cuda_header.h:254-302) */
typedef struct {
    int32_t score;        /* first ? max_score : pos_score  (align.cpp:190-199) */
    int32_t max_i, max_j; /* 1-based arg-max, first tiles only, else 0 */
    int32_t ref_steps;    /* i_steps of align.cpp:187 (ref bases consumed)   */
    int32_t query_steps;  /* j_steps (query bases consumed)                  */
    int32_t n_states;
} gact_tile_result;

/*
 * Aligns n tiles.  states receives, for tile t, n_states bytes (GACT_STATE_*)
 * at states + t*states_stride, in traceback order (the order AlignWithBT
 * pushes them).  states_stride must be >= 2*tile_size.
 */
int gact_hip_align_tiles(gact_hip_engine *e, int slot, int32_t n, const gact_tile *tiles,
                         gact_tile_result *results, uint8_t *states, int32_t states_stride);

/*
 * Same, with the tile slices passed inline like Align_Batch_GPU's
 * std::vector<std::string> arguments: tile t's ref bytes at
 * ref_bases + t*seq_stride (ref_lens[t] of them), likewise query.
 */
int gact_hip_align_tiles_inline(gact_hip_engine *e, int slot, int32_t n,
                                const uint8_t *ref_bases, const uint8_t *query_bases,
                                int32_t seq_stride,
                                const int32_t *ref_lens, const int32_t *query_lens,
                                const uint8_t *reverses, const uint8_t *firsts,
                                gact_tile_result *results, uint8_t *states, int32_t states_stride);

/* ---- per-candidate batch: what GACT / GACT_Batch do (gact.cpp:48-560) ---- */

/* the seed hit darwin.cpp:216-238 turns into a GACT_call */
typedef struct {
    int32_t ref_id, query_id, ref_pos, query_pos;
} gact_candidate;

/* one record per candidate, same order as the input */
typedef struct {
    int32_t ref_id, query_id;
    int32_t ab, ae, bb, be;     /* gact.cpp:219-222 */
    int32_t score;              /* total_score, gact.cpp:197-210 */
    int32_t comp;
    int32_t emitted;            /* 1 iff gact.cpp:213 would print the line */
    int32_t first_tile_score;
    int32_t n_tiles;            /* AlignWithBT-equivalents executed */
    int32_t reserved;
    int64_t cells;              /* sum of ref_len*query_len over those tiles */
} gact_overlap;

/*
 * Extends n candidates to overlaps on the device: the whole tile chain of
 * GACT (gact.cpp:82-195), the rescoring (gact.cpp:197-210) and the emit test
 * (gact.cpp:213) run inside one persistent kernel; no host round trip per
 * tile.  complement selects GACT_SET_QUERY_RC as darwin.cpp:279 does.
 */
int gact_hip_extend_candidates(gact_hip_engine *e, int slot, int32_t n,
                               const gact_candidate *cands, int complement, int same_file,
                               gact_overlap *out);

/* the same in three steps so a harness can time the device part alone with
 * the inputs already resident in HBM */
int gact_hip_candidates_upload(gact_hip_engine *e, int slot, int32_t n, const gact_candidate *cands);
int gact_hip_candidates_run(gact_hip_engine *e, int slot, int32_t n, int complement, int same_file);
/* fetch: records [0, n) into the caller's buffer (through the engine's own pinned staging area unless `out` lies
 * inside a buffer registered with gact_hip_register_output). */
int gact_hip_candidates_fetch(gact_hip_engine *e, int slot, int32_t n, gact_overlap *out);
/* Opt-in for a caller that fetches into one long-lived buffer again and again (a feeder thread's result array; the
 * reference's counterpart is the malloc'd outs_b of cuda_host.cu:67,183-189): page-locks [buf, buf + bytes) so that
 * fetches into it are one DMA with no staging copy.  One registered buffer per slot; it must stay allocated until
 * gact_hip_unregister_output (or gact_hip_destroy).  The engine never page-locks caller memory on its own. */
int gact_hip_register_output(gact_hip_engine *e, int slot, void *buf, int64_t bytes);
int gact_hip_unregister_output(gact_hip_engine *e, int slot);
/* runs on candidates [first, first+n) of the uploaded array (multi-GPU shards) */
int gact_hip_candidates_run_range(gact_hip_engine *e, int slot, int32_t first, int32_t n,
                                  int complement, int same_file);

/* one launch over both strands: uploaded candidates with index >= rc_from are
 * the reverse-complement ones (GACT_calls_rev of darwin.cpp:266-277), the rest
 * forward (GACT_calls_for, :227-238) */
int gact_hip_candidates_run_mixed(gact_hip_engine *e, int slot, int32_t first, int32_t n,
                                  int32_t rc_from, int same_file);

/* ------------------------------------------------------------------------
 * D-SOFT seed filter on the device (the stage in front of the path; optional:
 * the reference's host filter keeps working against the calls above).
 *
 * gact_hip_dsoft_build stands where darwin.cpp:532-560 builds the padded
 * reference string and `new SeedPosTable(...)` (seed_pos_table.cpp:46-98): the
 * index is built over GACT_SET_REF as uploaded (ASCII bases; anything but
 * acgtACGT counts as A, ntcoding.cpp:59-71).
 * gact_hip_dsoft_query stands where AlignReads calls sa->DSOFT for every read
 * and its reverse complement and decodes the hits (darwin.cpp:209-224,252-263):
 * queries [first_query, first_query + n_queries) of GACT_SET_QUERY / _RC are
 * filtered and the slot's candidate array is filled ON THE DEVICE, all forward
 * candidates first (query order, then emission order, as the reference's
 * GACT_calls_for), then the reverse-complement ones; run them with
 * gact_hip_candidates_run_mixed(e, slot, 0, *n_forward + *n_reverse, *n_forward, same_file).
 */
typedef struct {
    int32_t seed_size;                 /* params.cfg DSOFT_params.seed_size, 4..15 */
    int32_t bin_size;
    int32_t window_size;               /* < seed_size */
    int32_t threshold;
    int32_t num_seeds;
    int32_t seed_occurence_multiple;
    int32_t max_candidates;            /* per query strand: the first max_candidates threshold crossings are kept */
} gact_dsoft_params;

typedef struct {
    int64_t ref_length;                /* padded concatenation, darwin.cpp:544 */
    int64_t n_minimizers;
    int64_t table_bytes, pos_bytes;
    int32_t max_occurrence;            /* kmer_max_occurence_, seed_pos_table.cpp:59 */
    int32_t n_bins;
    float build_ms;                    /* HIP events, all build kernels */
} gact_dsoft_info;

int gact_hip_dsoft_build(gact_hip_engine *e, const gact_dsoft_params *p, gact_dsoft_info *info);
int gact_hip_dsoft_query(gact_hip_engine *e, int slot, int32_t first_query, int32_t n_queries,
                         int32_t *n_forward, int32_t *n_reverse, float *query_ms);
/* copies candidates [0, n) of the slot's device array to the host */
int gact_hip_candidates_download(gact_hip_engine *e, int slot, int32_t n, gact_candidate *out);

int gact_hip_sync(gact_hip_engine *e, int slot);
/* HIP-event time of the last kernel launched on this slot's stream, in ms */
int gact_hip_last_kernel_ms(gact_hip_engine *e, int slot, float *ms);
/* How the last candidates_run* on this slot was executed.  With the default
 * scoring the chain runs as two launches: a seed launch (first tile of every
 * candidate, arg-max + full pointer matrix) and the main launch (packed-int16
 * kernel, every later tile). */
typedef struct {
    float total_ms, seed_ms, main_ms;   /* HIP events on the slot's stream */
    int32_t packed16;                   /* 0: one int32 launch; 1: seed + packed-int16 main launch, uniform
                                           column layout; 2: the same, split (two-region) layout; 3: wide layout
                                           (32 lanes per tile pair, chosen when there are few chains) */
    int32_t handed_off;                 /* candidates the main launch continued */
    int32_t seed_packed16;              /* 1: the seed launch ran the packed-int16 arg-max kernel, 0: the int32 one */
    int32_t tagged_pointers;            /* 1: the split layout ran its pointer phase on tagged scores */
    int32_t linear_gap;                 /* which drifted pass the main launch ran: 1 the linear-gap pass (open == extend ==
                                           mismatch, gact_lin.hpp), 2 the drifted affine pass (gact_aff.hpp), 0 neither */
    int64_t seed_cells;                 /* DP cells executed by the seed launch */
    int32_t raw_candidates;             /* candidates the run aligned from raw bytes because one of their two reads holds a byte
                                           other than A/C/G/T (align.cpp:134), while the rest ran on the 2-bit image; 0 when no
                                           set holds such a byte, or when the whole run compared raw bytes */
    int32_t band_redos;                 /* linear-gap main launch: tiles run a second time with their whole pointer window stored,
                                           because the traceback left the band around the diagonal the first run had stored
                                           (exact either way; some tenths of a percent of the tiles at 15 % read error) */
    int32_t merged_callers;             /* how many callers' runs the launch carried: > 1 when the engine merged this run with runs
                                           other threads submitted on other slots at about the same time (feeder threads,
                                           darwin.cpp:619-629) into one seed + main launch; the times above are that launch's */
    int32_t overlapped_seeding;         /* 1: the candidates were seeded in order of chain length, most of them beside the main
                                           launch (large runs on an otherwise idle engine); seed_ms is then the first seed launch
                                           alone and main_ms holds the rest */
    int32_t critical_lane;              /* 1: beside the split main launch a wide one ran on a third of the blocks and took the
                                           longest chains (runs of 1-4 chains per tile slot on an otherwise idle engine) */
    int32_t role_waves;                 /* how the split linear-gap main launch walked its tracebacks: 0 every wave its own eight tiles
                                           behind their pass; 1 DP waves + walker waves (gact_roles.hpp); 2 two banks of tiles per wave,
                                           walks of the whole block batched on whichever wave needs a result first (gact_coop.hpp) */
} gact_hip_run_stats;
int gact_hip_last_run_stats(gact_hip_engine *e, int slot, gact_hip_run_stats *stats);

/* Optional, once, after the read sets are resident and before the first job: device arrays for jobs of up to
 * expected_candidates candidates (0: none), the second stream of every slot, and one empty launch of the chain kernels on
 * every stream -- what the first run would otherwise do inside the time its caller measures (allocations, stream
 * creation, code and scratch set-up: ~10 ms on a 65,766-candidate job).  The shim's GPU_init calls it; the reference's
 * caller makes exactly two GACT_Batch calls per feeder thread and per process (darwin.cpp:429-433). */
int gact_hip_prepare(gact_hip_engine *e, int32_t expected_candidates);

/* Scheduling switches of a live engine (none of them changes a record); unknown names are refused.
 *   "overlap_seed"       1 (default): a large run on an idle engine seeds its candidates in order of chain length, most of them
 *                        beside the main launch (gact_hip_run_stats.overlapped_seeding); 0: seed launch, then one main launch
 *   "combine"            1 (default when n_slots > 1): runs that different threads submit on different slots at about the same
 *                        time are merged into one launch (gact_hip_run_stats.merged_callers); 0: every run its own launches
 *   "combine_window_us"  how long the first of such runs waits for the others at most (default 1000)
 *   "runs_in_flight"     1: the caller keeps several runs in flight on this engine (a pipeline of steps, one slot each): every
 *                        launch takes the layout with the better throughput.  0 (default): the engine looks at the other slots'
 *                        events when a run is launched, which the first launches of a pipeline answer differently from run to run
 *   "coop"               the split linear-gap main launch with two banks of tiles per wave and cooperative, batched traceback
 *                        walks (gact_hip_run_stats.role_waves == 2): 1 always, 0 never, 2 (default) where throughput bounds the
 *                        launch -- it shares the machine and has 1.5 chains and more per resident tile slot, or has six and more
 *   "lone_lane"          a run of half as many to as many chains as there are resident tile slots, alone on the machine, runs as ONE
 *                        block per CU of two kinds: value wide blocks (e.g. 48) for its longest chains, split blocks with the
 *                        look-ahead walker on the other CUs (value < 0: without it); 0 (default): all wide, two blocks per CU, which
 *                        is faster.  (gact_hip_run_stats with the mix: layout split, critical_lane 1)
 *   "shared_twelfths"    a linear-gap main launch that shares the machine and has more chains than two thirds of the resident tile slots
 *                        hold takes value / 12 of the resident blocks (default 6)
 *   "overlap_big"        1: ordered, overlapped seeding also for runs of more than four chains per resident tile slot (seed launch A
 *                        takes the longest eighth of the list, B the rest beside main launch 1); 0 (default): seed launch, then one
 *                        main launch
 *   "roles"              1: the split linear-gap main launch runs as DP waves + walker waves (gact_hip_run_stats.role_waves);
 *                        0 (default): one wave does everything for its tiles.  Same records; measured no faster
 * Every other switch of the library is read once, in gact_hip_create, from an environment variable; set_option names the
 * variable when asked for one of those.  The whole table: gact_hip_options_describe, INTEGRATION.md 7. */
int gact_hip_set_option(gact_hip_engine *e, const char *name, int32_t value);
/* The table of every switch the library reads, one line per switch: `name | environment variable | when it is read |
 * class | what it does`.  Writes at most cap bytes (NUL-terminated) into buf, returns the size the whole table needs.
 * No engine, no device. */
int64_t gact_hip_options_describe(char *buf, int64_t cap);
/* The launch plan -- sequence, kernels, grids -- that an engine of parameters p makes for one pass over `count` candidates on a
 * device of compute_units CUs (kernels at their nominal occupancy), as one JSON object.  flags: bit 0 = the read sets hold
 * bytes other than A/C/G/T, bit 1 = the launch shares the machine (other runs in flight), bit 2 = role launch on, bit 3 =
 * cooperative launch always, bit 4 = never (neither: where the policy takes it), bit 5 = "overlap_big" 1, bit 6 = "lone_lane" 48
 * (the two optional sequences of gact_hip_set_option).
 * The policy is a pure function (csrc/gact_policy.hpp); this entry exists so that it can be swept and tested without a
 * device.  Same buffer convention as gact_hip_options_describe. */
int64_t gact_hip_plan_describe(const gact_hip_params *p, int32_t compute_units, int32_t count, int32_t flags, char *buf, int64_t cap);

/* device address of the slot's gact_overlap array (for an RCCL gather) */
void *gact_hip_device_overlaps(gact_hip_engine *e, int slot);
/* the slot's hipStream_t as an opaque pointer */
void *gact_hip_stream(gact_hip_engine *e, int slot);

/* ---- multi-GPU: the one collective of a sharded job, for C / C++ callers ----
 * One process per GPU, every process with its own engine and its share of the candidates (they shard with no exchange on the
 * data path); what is left to do is bring the records together.  The reference is single-device (cuda_host.cu:195) and joins
 * the output files of separate processes with `cat darwin.*.out | sort | uniq` (README:25); here rank 0 receives every
 * rank's records with RCCL, out of the engines' device-resident record arrays, narrowed on the device to the 32 bytes a
 * line is printed from (gact.cpp:214-224).  bench.py makes the same gather through torch.distributed (gact_amd/dist.py).
 * RCCL is looked up when the first communicator is made (librccl.so.1; GACT_HIP_RCCL_LIB names another file): a
 * single-GPU caller needs none. */
typedef struct {
    int32_t ref_id, query_id;
    int32_t ab, ae, bb, be, score;
    int32_t comp_emitted;       /* bit 0: comp, bit 1: emitted (gact_overlap) */
} gact_line;
typedef struct gact_hip_comm gact_hip_comm;
/* Rank `rank` of `world` (collective: returns when every rank has called it).  The ranks find each other through id_path, a
 * file name all of them can reach and that does not exist yet: rank 0 puts RCCL's unique id there, the others wait for it
 * (timeout_s seconds, 0: 120), rank 0 removes it again.  The communicator works on the engine's device. */
int gact_hip_comm_create(gact_hip_engine *e, int32_t rank, int32_t world, const char *id_path, int32_t timeout_s,
                         gact_hip_comm **out);
/* Collective.  The first n records of `slot` (its last run's; the call waits for that run on the device) travel to rank 0.
 * counts[world] (every rank, may be NULL): records per rank.  lines (rank 0): all of them, rank after rank, each rank's in
 * candidate order; lines_cap = room in `lines`, in records.  Other ranks pass NULL, 0.
 * Too little room on rank 0 (or lines == NULL) does not break the collective: the records are received all the same, the other
 * ranks return 0, rank 0 returns GACT_HIP_EINVAL with counts[] filled in -- call again with room for their sum. */
int gact_hip_comm_gather_lines(gact_hip_comm *c, int slot, int32_t n, int64_t *counts, gact_line *lines, int64_t lines_cap);
int gact_hip_comm_destroy(gact_hip_comm *c);

/* measurement aid: sustained lane-ops/s of this device on the packed-int16 instructions the kernels are made of
 * (independent v_pk_add_i16 / v_pk_max_i16 streams, eight waves per SIMD, no memory) */
int gact_hip_measure_valu_rate(gact_hip_engine *e, double *lane_ops_per_s);

/* formats the exact bytes of gact.cpp:214-224 */
int gact_hip_format_overlap(const gact_overlap *o, const char *ref_name, const char *query_name,
                            char *buf, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* GACT_HIP_H */
