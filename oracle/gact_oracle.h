/*
 * gact_oracle.h -- CPU restatement of the reference's GACT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it, and only as the checker / reported CPU baseline.  The HIP
 * engine in darwin-gpu_amd/csrc never includes this header.
 *
 * Parity pin: this restatement is checked (tests/test_oracle_vs_ref.py and
 * tests/golden/) against the reference's own align.cpp / gact.cpp compiled
 * unchanged from /root/reference into oracle/_ref/libdarwin_ref.so, and
 * against the known-answer tiles of SURVEY.md Appendix B.
 *
 * Reference semantics restated here:
 *   AlignWithBT   /root/reference/align.cpp:60-233
 *   Align_Batch   /root/reference/align.cpp:17-54
 *   GACT          /root/reference/gact.cpp:48-228
 */
#ifndef GACT_ORACLE_H
#define GACT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* traceback states, align.h:23 (enum states {Z, D, I, M}) */
enum { ORACLE_Z = 0, ORACLE_D = 1, ORACLE_I = 2, ORACLE_M = 3 };

typedef struct {
    int match, mismatch, gap_open, gap_extend;
} oracle_scoring;

/*
 * AlignWithBT restated (align.cpp:60-233).  The reference returns a
 * std::queue<int>; here the queue is written front-first into out[]:
 *   first  -> [max_score, max_i, max_j, states...]
 *   !first -> [pos_score, states...]
 * Returns the number of ints written, or -1 if out_cap is too small,
 * or -2 if a length is >= MAX_TILE_SIZE (the reference asserts, align.cpp:66).
 */
int oracle_align_with_bt(const char *ref_seq, long long ref_len,
                         const char *query_seq, long long query_len,
                         int match_score, int mismatch_score,
                         int gap_open, int gap_extend,
                         int query_pos, int ref_pos,
                         int reverse, int first, int early_terminate,
                         int *out, int out_cap);

/* one executed tile of a chain, for tile-level parity traces */
typedef struct {
    int32_t ref_off;      /* start of the tile slice in the ref read   */
    int32_t query_off;    /* start of the tile slice in the query read */
    int32_t ref_len;
    int32_t query_len;
    int32_t reverse;      /* AlignWithBT's `reverse` argument (0 = left phase) */
    int32_t first;        /* AlignWithBT's `first` argument */
    int32_t tile_score;
    int32_t max_i, max_j; /* first tiles only, else 0 */
    int32_t n_states;
    int32_t i_steps;      /* query bases consumed (gact.cpp's `i`) */
    int32_t j_steps;      /* ref bases consumed   (gact.cpp's `j`) */
} oracle_tile_trace;

/* result of one GACT call (gact.cpp:48-228) */
typedef struct {
    int32_t ref_id, query_id;
    int32_t ab, ae, bb, be;   /* abpos, ref_pos, bbpos, query_pos at gact.cpp:219-222 */
    int32_t score;            /* total_score, gact.cpp:197-210 */
    int32_t comp;             /* complement */
    int32_t emitted;          /* 1 iff the reference would print the line (gact.cpp:213) */
    int32_t first_tile_score;
    int32_t n_tiles;          /* AlignWithBT calls made */
    int64_t cells;            /* sum ref_len*query_len over those calls */
} oracle_overlap;

/*
 * GACT restated (gact.cpp:48-228).  `trace` may be NULL; if not, up to
 * trace_cap executed tiles are recorded (n_tiles still counts all).
 */
void oracle_gact(const char *ref_str, const char *query_str,
                 int ref_length, int query_length,
                 int tile_size, int tile_overlap,
                 int ref_pos, int query_pos, int first_tile_score_threshold,
                 int ref_id, int query_id, int complement,
                 int match_score, int mismatch_score,
                 int gap_open, int gap_extend,
                 int same_file,
                 oracle_overlap *out,
                 oracle_tile_trace *trace, int trace_cap);

/* candidate as darwin.cpp:227-238 builds it (CPU build passes the same five
 * numbers straight to GACT, darwin.cpp:240-246) */
typedef struct {
    int32_t ref_id, query_id, ref_pos, query_pos;
} oracle_candidate;

/*
 * Runs oracle_gact over candidates [0,n) with n_threads host threads over
 * contiguous candidate ranges (darwin.cpp:619-629 splits reads the same way).
 * seqs are given as one concatenated byte buffer + offsets (n_seqs+1 entries).
 * `complement` picks query_rc_* instead of query_* exactly like
 * darwin.cpp:279 passes rev_reads_char.  Returns total cells.
 */
int64_t oracle_gact_many(const char *ref_concat, const int64_t *ref_offsets,
                         const char *query_concat, const int64_t *query_offsets,
                         const oracle_candidate *cands, int n,
                         int complement, int same_file,
                         int tile_size, int tile_overlap, int first_tile_score_threshold,
                         int match_score, int mismatch_score, int gap_open, int gap_extend,
                         int n_threads, oracle_overlap *out);

/* formats the exact bytes of gact.cpp:214-224; returns strlen */
int oracle_format_line(const oracle_overlap *o, const char *ref_name,
                       const char *query_name, char *buf, int cap);

#ifdef __cplusplus
}
#endif
#endif
