/*
 * gact_oracle.c -- CPU restatement of the reference's GACT hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY (see gact_oracle.h).  Pinned against the compiled
 * reference (oracle/_ref) and SURVEY.md Appendix B known-answer tiles.
 *
 * The arithmetic follows the reference statement by statement; what differs
 * is storage: the reference allocates a 2050x2050 vector<vector<int>> and
 * copies 4x2049 ints per column (align.cpp:85,115-120); here one byte per
 * cell in a (ref_len+1)x(query_len+1) scratch and two rolling columns.
 */
#include "gact_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_INF (1 << 30)          /* align.h:18 */
#define ORACLE_MAX_TILE_SIZE 2049     /* align.h:19 */

/* pointer byte: low 2 bits = AlnOperands {ZERO,DELETE,INSERT,MATCH} (align.h:22),
 * +4 when del_open >= del_extend, +8 when ins_open >= ins_extend (align.cpp:170-171) */
#define PTR_DEL_OPEN 4
#define PTR_INS_OPEN 8

typedef struct {
    unsigned char *dir;   /* (R+1)*(Q+1) pointer bytes */
    size_t dir_cap;
    int *col;             /* 8 * (MAX_TILE_SIZE+1) ints: h,m,i,d for rd and wr */
} oracle_scratch;

static __thread oracle_scratch tls_scratch;

static oracle_scratch *get_scratch(size_t need)
{
    oracle_scratch *s = &tls_scratch;
    if (!s->col) s->col = (int *)malloc(sizeof(int) * 8 * (ORACLE_MAX_TILE_SIZE + 1));
    if (s->dir_cap < need) {
        free(s->dir);
        s->dir = (unsigned char *)malloc(need);
        s->dir_cap = need;
    }
    return s;
}

int oracle_align_with_bt(const char *ref_seq, long long ref_len,
                         const char *query_seq, long long query_len,
                         int match_score, int mismatch_score,
                         int gap_open, int gap_extend,
                         int query_pos, int ref_pos,
                         int reverse, int first, int early_terminate,
                         int *out, int out_cap)
{
    /* align.cpp:66-67 asserts */
    if (ref_len >= ORACLE_MAX_TILE_SIZE || query_len >= ORACLE_MAX_TILE_SIZE) return -2;
    if (ref_len < 0 || query_len < 0) return -2;
    if (ref_pos < 0 || query_pos < 0 || ref_pos >= ORACLE_MAX_TILE_SIZE || query_pos >= ORACLE_MAX_TILE_SIZE) return -2;

    const int R = (int)ref_len, Q = (int)query_len;
    const size_t stride = (size_t)Q + 1;
    oracle_scratch *sc = get_scratch(((size_t)R + 1) * stride);
    unsigned char *dir = sc->dir;
    const int W = ORACLE_MAX_TILE_SIZE + 1;
    int *h_rd = sc->col, *m_rd = h_rd + W, *i_rd = m_rd + W, *d_rd = i_rd + W;
    int *h_wr = d_rd + W, *m_wr = h_wr + W, *i_wr = m_wr + W, *d_wr = i_wr + W;

    /* borders, align.cpp:87-107: H=M=0, I=D=-INF, pointer ZERO */
    for (int j = 0; j <= Q; j++) {
        h_rd[j] = 0; m_rd[j] = 0; i_rd[j] = -ORACLE_INF; d_rd[j] = -ORACLE_INF;
        h_wr[j] = 0; m_wr[j] = 0; i_wr[j] = -ORACLE_INF; d_wr[j] = -ORACLE_INF;
    }
    for (int i = 0; i <= R; i++) dir[(size_t)i * stride] = ORACLE_Z;
    for (int j = 0; j <= Q; j++) dir[j] = ORACLE_Z;

    int max_score = 0, pos_score = 0, max_i = 0, max_j = 0;

    for (int i = 1; i <= R; i++) {
        /* align.cpp:115-120 copies wr->rd for k>=1 (index 0 keeps the border);
         * swapping the column pointers and restoring index 0 is equivalent */
        int *t;
        t = h_rd; h_rd = h_wr; h_wr = t;
        t = m_rd; m_rd = m_wr; m_wr = t;
        t = i_rd; i_rd = i_wr; i_wr = t;
        t = d_rd; d_rd = d_wr; d_wr = t;
        h_rd[0] = 0; m_rd[0] = 0; i_rd[0] = -ORACLE_INF; d_rd[0] = -ORACLE_INF;
        h_wr[0] = 0; m_wr[0] = 0; i_wr[0] = -ORACLE_INF; d_wr[0] = -ORACLE_INF;

        /* align.cpp:130 (the #else branch is the live one) */
        const char ref_nt = reverse ? ref_seq[R - i] : ref_seq[i - 1];
        unsigned char *drow = dir + (size_t)i * stride;

        for (int j = 1; j <= Q; j++) {
            const char query_nt = reverse ? query_seq[Q - j] : query_seq[j - 1];  /* :131 */
            const int sub = (query_nt == ref_nt) ? match_score : mismatch_score;  /* :134 */

            /* :138-147  M = max(M,I,D)[i-1][j-1] + sub, floored at 0 */
            int best_prev;
            if (m_rd[j - 1] > i_rd[j - 1] && m_rd[j - 1] > d_rd[j - 1]) best_prev = m_rd[j - 1];
            else if (i_rd[j - 1] > d_rd[j - 1]) best_prev = i_rd[j - 1];
            else best_prev = d_rd[j - 1];
            int m = best_prev + sub;
            if (m < 0) m = 0;
            m_wr[j] = m;

            /* :149-156 */
            const int ins_open = m_rd[j] + gap_open;
            const int ins_extend = i_rd[j] + gap_extend;
            const int del_open = m_wr[j - 1] + gap_open;
            const int del_extend = d_wr[j - 1] + gap_extend;
            const int ins = (ins_open > ins_extend) ? ins_open : ins_extend;
            const int del = (del_open > del_extend) ? del_open : del_extend;
            i_wr[j] = ins;
            d_wr[j] = del;

            /* :158-160 */
            const int max1 = m > ins ? m : ins;
            const int max2 = del > 0 ? del : 0;
            const int h = max1 > max2 ? max1 : max2;
            h_wr[j] = h;

            /* :162-171 */
            int p = (m >= ins) ? ((m >= del) ? ORACLE_M : ORACLE_D)
                               : ((ins >= del) ? ORACLE_I : ORACLE_D);
            if (m <= 0 && ins <= 0 && del <= 0) p = ORACLE_Z;
            if (ins_open >= ins_extend) p += PTR_INS_OPEN;
            if (del_open >= del_extend) p += PTR_DEL_OPEN;
            drow[j] = (unsigned char)p;

            /* :173-181 */
            if (h >= max_score) { max_score = h; max_i = i; max_j = j; }
            if (i == ref_pos && j == query_pos) pos_score = h;
        }
    }

    /* traceback, align.cpp:185-230 */
    int n = 0;
    int i_curr = ref_pos, j_curr = query_pos;
    int i_steps = 0, j_steps = 0;
#define PUSH(v) do { if (n >= out_cap) return -1; out[n++] = (v); } while (0)
    if (first) {
        i_curr = max_i; j_curr = max_j;
        PUSH(max_score); PUSH(i_curr); PUSH(j_curr);
    } else {
        PUSH(pos_score);
    }
    /* align.cpp:85 zero-initialises the whole 2050 x 2050 matrix, and cells outside the tile are never
     * written: a start position beyond the tile reads ZERO_OP (the stride here is the tile's own) */
    int state = (i_curr > R || j_curr > Q) ? ORACLE_Z : dir[(size_t)i_curr * stride + j_curr] % 4;
    while (state != ORACLE_Z) {
        if (i_steps >= early_terminate || j_steps >= early_terminate) break;   /* :205 */
        PUSH(state);
        if (state == ORACLE_M) {
            state = dir[(size_t)(i_curr - 1) * stride + (j_curr - 1)] % 4;
            i_curr--; j_curr--; i_steps++; j_steps++;
        } else if (state == ORACLE_I) {
            state = (dir[(size_t)i_curr * stride + j_curr] & PTR_INS_OPEN) ? ORACLE_M : ORACLE_I;
            i_curr--; i_steps++;
        } else { /* ORACLE_D */
            state = (dir[(size_t)i_curr * stride + j_curr] & PTR_DEL_OPEN) ? ORACLE_M : ORACLE_D;
            j_curr--; j_steps++;
        }
    }
#undef PUSH
    return n;
}

/* ------------------------------------------------------------------ GACT */

typedef struct {
    char *ref, *query;   /* aligned strings, grown from the middle */
    int cap, lo, hi;     /* valid range [lo,hi) */
} aln_buf;

static __thread aln_buf tls_aln;
static __thread int *tls_bt;
static __thread int tls_bt_cap;

static void aln_reset(aln_buf *a, int left_room, int right_room)
{
    const int need = left_room + right_room + 2;
    if (a->cap < need) {
        free(a->ref); free(a->query);
        a->ref = (char *)malloc((size_t)need);
        a->query = (char *)malloc((size_t)need);
        a->cap = need;
    }
    a->lo = a->hi = left_room + 1;
}

void oracle_gact(const char *ref_str, const char *query_str,
                 int ref_length, int query_length,
                 int tile_size, int tile_overlap,
                 int ref_pos, int query_pos, int first_tile_score_threshold,
                 int ref_id, int query_id, int complement,
                 int match_score, int mismatch_score,
                 int gap_open, int gap_extend,
                 int same_file,
                 oracle_overlap *out,
                 oracle_tile_trace *trace, int trace_cap)
{
    const int bt_need = 2 * ORACLE_MAX_TILE_SIZE + 8;
    if (tls_bt_cap < bt_need) {
        free(tls_bt);
        tls_bt = (int *)malloc(sizeof(int) * (size_t)bt_need);
        tls_bt_cap = bt_need;
    }
    int *bt = tls_bt;

    /* the left phase can prepend at most ref_pos+query_pos columns and the
     * right phase append at most the remaining bases of both reads */
    aln_buf *al = &tls_aln;
    aln_reset(al, ref_pos + query_pos + 2,
              (ref_length > 0 ? ref_length : 0) + (query_length > 0 ? query_length : 0) + 2);

    const int early = tile_size - tile_overlap;
    int abpos, bbpos;
    int rev_ref_pos = ref_pos, rev_query_pos = query_pos;   /* gact.cpp:72-73 */
    int i = 0, j = 0;
    int first_tile_score = 0;
    int first_tile = 1;
    int n_tiles = 0;
    int64_t cells = 0;

    /* left extension, gact.cpp:82-134 */
    while (ref_pos > 0 && query_pos > 0 && ((i > 0 && j > 0) || first_tile)) {
        const int rlen = (ref_pos > tile_size) ? tile_size : ref_pos;
        const int qlen = (query_pos > tile_size) ? tile_size : query_pos;
        const int n = oracle_align_with_bt(ref_str + ref_pos - rlen, rlen,
                                           query_str + query_pos - qlen, qlen,
                                           match_score, mismatch_score, gap_open, gap_extend,
                                           qlen, rlen, 0, first_tile, early, bt, tls_bt_cap);
        oracle_tile_trace tr;
        tr.ref_off = ref_pos - rlen; tr.query_off = query_pos - qlen;
        tr.ref_len = rlen; tr.query_len = qlen; tr.reverse = 0; tr.first = first_tile;
        tr.max_i = tr.max_j = 0;
        cells += (int64_t)rlen * qlen;
        i = 0; j = 0;
        int k = 0;
        const int tile_score = bt[k++];
        tr.tile_score = tile_score;
        int stop = 0;
        if (first_tile) {
            tr.max_i = bt[k]; tr.max_j = bt[k + 1];
            ref_pos = ref_pos - rlen + bt[k++];
            query_pos = query_pos - qlen + bt[k++];
            rev_ref_pos = ref_pos;
            rev_query_pos = query_pos;
            first_tile_score = tile_score;
            if (tile_score < first_tile_score_threshold) stop = 1;   /* :107-109 */
        }
        tr.n_states = stop ? 0 : n - k;
        if (!stop) {
            for (; k < n; k++) {
                first_tile = 0;
                const int state = bt[k];
                al->lo--;
                if (state == ORACLE_M) {
                    al->ref[al->lo] = ref_str[ref_pos - j - 1];
                    al->query[al->lo] = query_str[query_pos - i - 1];
                    i++; j++;
                } else if (state == ORACLE_I) {
                    al->ref[al->lo] = ref_str[ref_pos - j - 1];
                    al->query[al->lo] = '-';
                    j++;
                } else { /* D */
                    al->ref[al->lo] = '-';
                    al->query[al->lo] = query_str[query_pos - i - 1];
                    i++;
                }
            }
            ref_pos -= j;
            query_pos -= i;
        }
        tr.i_steps = i; tr.j_steps = j;
        if (trace && n_tiles < trace_cap) trace[n_tiles] = tr;
        n_tiles++;
        if (stop) break;
    }

    abpos = ref_pos; bbpos = query_pos;        /* :136-141 */
    ref_pos = rev_ref_pos; query_pos = rev_query_pos;
    i = tile_size; j = tile_size;

    /* right extension, gact.cpp:144-195 */
    while (ref_pos < ref_length && query_pos < query_length && ((i > 0 && j > 0) || first_tile)) {
        const int rlen = (ref_pos + tile_size < ref_length) ? tile_size : ref_length - ref_pos;
        const int qlen = (query_pos + tile_size < query_length) ? tile_size : query_length - query_pos;
        const int n = oracle_align_with_bt(ref_str + ref_pos, rlen, query_str + query_pos, qlen,
                                           match_score, mismatch_score, gap_open, gap_extend,
                                           qlen, rlen, 1, first_tile, early, bt, tls_bt_cap);
        oracle_tile_trace tr;
        tr.ref_off = ref_pos; tr.query_off = query_pos;
        tr.ref_len = rlen; tr.query_len = qlen; tr.reverse = 1; tr.first = first_tile;
        tr.max_i = tr.max_j = 0;
        cells += (int64_t)rlen * qlen;
        i = 0; j = 0;
        int k = 0;
        const int tile_score = bt[k++];
        tr.tile_score = tile_score;
        int stop = 0;
        if (first_tile) {
            tr.max_i = bt[k]; tr.max_j = bt[k + 1];
            ref_pos = ref_pos + rlen - bt[k++];
            query_pos = query_pos + qlen - bt[k++];
            first_tile_score = tile_score;
            if (tile_score < first_tile_score_threshold) stop = 1;   /* :168-170 */
        }
        tr.n_states = stop ? 0 : n - k;
        if (!stop) {
            for (; k < n; k++) {
                first_tile = 0;
                const int state = bt[k];
                if (state == ORACLE_M) {
                    al->ref[al->hi] = ref_str[ref_pos + j];
                    al->query[al->hi] = query_str[query_pos + i];
                    i++; j++;
                } else if (state == ORACLE_I) {
                    al->ref[al->hi] = ref_str[ref_pos + j];
                    al->query[al->hi] = '-';
                    j++;
                } else {
                    al->ref[al->hi] = '-';
                    al->query[al->hi] = query_str[query_pos + i];
                    i++;
                }
                al->hi++;
            }
            ref_pos += j;
            query_pos += i;
        }
        tr.i_steps = i; tr.j_steps = j;
        if (trace && n_tiles < trace_cap) trace[n_tiles] = tr;
        n_tiles++;
        if (stop) break;
    }

    /* rescore, gact.cpp:197-210: one `open` flag shared by both gap kinds */
    int total_score = 0;
    int open = 1;
    for (int c = al->lo; c < al->hi; c++) {
        const char r = al->ref[c], q = al->query[c];
        if (r == '-' || q == '-') {
            total_score += open ? gap_open : gap_extend;
            open = 0;
        } else {
            total_score += (q == r) ? match_score : mismatch_score;
            open = 1;
        }
    }

    out->ref_id = ref_id; out->query_id = query_id;
    out->ab = abpos; out->ae = ref_pos; out->bb = bbpos; out->be = query_pos;
    out->score = total_score;
    out->comp = complement ? 1 : 0;
    out->emitted = (!(same_file && ref_id == query_id) && total_score > 0) ? 1 : 0;  /* :213 */
    out->first_tile_score = first_tile_score;
    out->n_tiles = n_tiles;
    out->cells = cells;
}

/* ------------------------------------------------------- threaded driver */

typedef struct {
    const char *ref_concat; const int64_t *ref_offsets;
    const char *query_concat; const int64_t *query_offsets;
    const oracle_candidate *cands; int lo, hi;
    int complement, same_file, tile_size, tile_overlap, thr;
    int match, mismatch, gap_open, gap_extend;
    oracle_overlap *out; int64_t cells;
} many_job;

static void *many_worker(void *p)
{
    many_job *jb = (many_job *)p;
    int64_t cells = 0;
    for (int k = jb->lo; k < jb->hi; k++) {
        const oracle_candidate *c = &jb->cands[k];
        const int64_t r0 = jb->ref_offsets[c->ref_id], r1 = jb->ref_offsets[c->ref_id + 1];
        const int64_t q0 = jb->query_offsets[c->query_id], q1 = jb->query_offsets[c->query_id + 1];
        oracle_gact(jb->ref_concat + r0, jb->query_concat + q0, (int)(r1 - r0), (int)(q1 - q0),
                    jb->tile_size, jb->tile_overlap, c->ref_pos, c->query_pos, jb->thr,
                    c->ref_id, c->query_id, jb->complement,
                    jb->match, jb->mismatch, jb->gap_open, jb->gap_extend,
                    jb->same_file, &jb->out[k], NULL, 0);
        cells += jb->out[k].cells;
    }
    jb->cells = cells;
    return NULL;
}

int64_t oracle_gact_many(const char *ref_concat, const int64_t *ref_offsets,
                         const char *query_concat, const int64_t *query_offsets,
                         const oracle_candidate *cands, int n,
                         int complement, int same_file,
                         int tile_size, int tile_overlap, int first_tile_score_threshold,
                         int match_score, int mismatch_score, int gap_open, int gap_extend,
                         int n_threads, oracle_overlap *out)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    many_job jobs[256];
    pthread_t th[256];
    const int per = (n + n_threads - 1) / n_threads;
    int used = 0;
    for (int t = 0; t < n_threads; t++) {
        const int lo = t * per, hi = (lo + per > n) ? n : lo + per;
        if (lo >= hi) break;
        many_job *jb = &jobs[used];
        jb->ref_concat = ref_concat; jb->ref_offsets = ref_offsets;
        jb->query_concat = query_concat; jb->query_offsets = query_offsets;
        jb->cands = cands; jb->lo = lo; jb->hi = hi;
        jb->complement = complement; jb->same_file = same_file;
        jb->tile_size = tile_size; jb->tile_overlap = tile_overlap;
        jb->thr = first_tile_score_threshold;
        jb->match = match_score; jb->mismatch = mismatch_score;
        jb->gap_open = gap_open; jb->gap_extend = gap_extend;
        jb->out = out; jb->cells = 0;
        used++;
    }
    if (used == 1) {
        many_worker(&jobs[0]);
    } else {
        for (int t = 0; t < used; t++) pthread_create(&th[t], NULL, many_worker, &jobs[t]);
        for (int t = 0; t < used; t++) pthread_join(th[t], NULL);
    }
    int64_t cells = 0;
    for (int t = 0; t < used; t++) cells += jobs[t].cells;
    return cells;
}

int oracle_format_line(const oracle_overlap *o, const char *ref_name,
                       const char *query_name, char *buf, int cap)
{
    /* gact.cpp:214-224 */
    return snprintf(buf, (size_t)cap,
                    "ref_id: %s, query_id: %s, ab: %d, ae: %d, bb: %d, be: %d, score: %d, comp: %d\n",
                    ref_name, query_name, o->ab, o->ae, o->bb, o->be, o->score, o->comp);
}
