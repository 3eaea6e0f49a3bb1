// gact_shim.cpp -- the reference's C++ entry points implemented on the C-ABI
// (include/gact_hip.h).  Error behaviour follows the reference: a failed
// device call prints and exit(-1)s (cudaSafeCall, cuda_header.h:309-319).
#include "gact.h"
#include "align.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "gact_hip.h"

// globals of the driver that gact.cpp reads (gact.cpp:39-46)
extern bool same_file;
extern std::vector<std::string> reference_seqs;
extern std::vector<long long int> reference_lengths;
extern std::vector<std::string> reads_seqs;
extern std::vector<std::string> rev_reads_seqs;
extern std::vector<long long int> reads_lengths;
extern std::vector<std::vector<std::string> > reference_descrips;
extern std::vector<std::vector<std::string> > reads_descrips;

namespace {

void die(const char *what)
{
    printf("\n%s failed: %s\n\n", what, gact_hip_last_error());
    exit(-1);
}
#define SAFE(call) do { if ((call) != 0) die(#call); } while (0)

struct Main {
    gact_hip_engine *engine = nullptr;
    gact_hip_params params;
    std::once_flag uploaded;
    int scores[4];
} g_main;

void upload_set(gact_hip_engine *e, int which, const std::vector<std::string> &seqs)
{
    std::vector<int64_t> offs(seqs.size() + 1, 0);
    for (size_t k = 0; k < seqs.size(); k++) offs[k + 1] = offs[k] + (int64_t)seqs[k].size();
    std::vector<uint8_t> cat((size_t)offs.back());
    for (size_t k = 0; k < seqs.size(); k++) memcpy(cat.data() + offs[k], seqs[k].data(), seqs[k].size());
    SAFE(gact_hip_upload_seqs(e, which, cat.data(), offs.data(), (int32_t)seqs.size()));
}

// the driver's read sets become resident the first time a batch needs them
// (after darwin.cpp:314-398 has recoded them; ASCII and 0..3 are both accepted)
void ensure_reads_resident()
{
    std::call_once(g_main.uploaded, [] {
        upload_set(g_main.engine, GACT_SET_REF, reference_seqs);
        upload_set(g_main.engine, GACT_SET_QUERY, reads_seqs);
        upload_set(g_main.engine, GACT_SET_QUERY_RC, rev_reads_seqs);
    });
}

// the bytes of gact.cpp:214-224 appended to `buf` (what gact_hip_format_overlap's snprintf makes, without the format
// parsing: a feeder thread prints thousands of lines per batch, inside the time the reference reports as "Time GACT calling")
void append_int(std::string &buf, int v)
{
    char tmp[12];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) tmp[n++] = '-';
    while (n) buf.push_back(tmp[--n]);
}
void append_overlap(std::string &buf, const gact_overlap &o)
{
    buf += "ref_id: "; buf += reference_descrips[o.ref_id][0];
    buf += ", query_id: "; buf += reads_descrips[o.query_id][0];
    buf += ", ab: "; append_int(buf, o.ab);
    buf += ", ae: "; append_int(buf, o.ae);
    buf += ", bb: "; append_int(buf, o.bb);
    buf += ", be: "; append_int(buf, o.be);
    buf += ", score: "; append_int(buf, o.score);
    buf += ", comp: "; append_int(buf, o.comp);
    buf += '\n';
}

void print_overlap(std::ofstream &fout, const gact_overlap &o)
{
    char line[1024];
    int n = gact_hip_format_overlap(&o, reference_descrips[o.ref_id][0].c_str(),
                                    reads_descrips[o.query_id][0].c_str(), line, sizeof line);
    if (n < 0) die("gact_hip_format_overlap");
    fout.write(line, n);
}

// The reference is single-device (cudaSetDevice(0), cuda_host.cu:195).  One process per GPU is how this engine
// scales: a launcher gives each process its GPU through GACT_HIP_DEVICE (darwin_hip --device sets it).
int shim_device()
{
    const char *v = getenv("GACT_HIP_DEVICE");
    return v ? atoi(v) : 0;
}

// small engines for the align.h / GACT() surface, keyed by what the kernels bake in
typedef std::tuple<int, int, int, int, int, int, int> Key;   // tile, overlap, match, mismatch, open, ext, thr
std::mutex g_side_mu;
std::map<Key, gact_hip_engine *> g_side;

// (one tile or one candidate per call: a few blocks per launch, and a traceback workspace of that size -- 7 MB where a
// full engine's slot holds 1.3 GB)
gact_hip_engine *side_engine(int tile, int overlap, int match, int mismatch, int open, int ext, int thr)
{
    const Key k(tile, overlap, match, mismatch, open, ext, thr);
    auto it = g_side.find(k);
    if (it != g_side.end()) return it->second;
    gact_hip_params p;
    memset(&p, 0, sizeof p);
    p.tile_size = tile; p.tile_overlap = overlap;
    p.match = match; p.mismatch = mismatch; p.gap_open = open; p.gap_extend = ext;
    p.first_tile_score_threshold = thr; p.device_id = shim_device(); p.n_slots = 1;
    p.max_blocks = 4;
    gact_hip_engine *e = nullptr;
    SAFE(gact_hip_create(&p, &e));
    if (g_side.empty())
        atexit([] {                                  // the reference's CPU entry points have no close call to hang this on
            for (auto &kv : g_side) gact_hip_destroy(kv.second);
            g_side.clear();
        });
    g_side[k] = e;
    return e;
}

// GACT() for the driver's own reads (darwin.cpp:240-246,279-285 pass reference_seqs[id].c_str() and its per-read copies of
// reads_seqs / rev_reads_seqs): an engine with the three global sets resident, a slot per calling thread, no lock around
// the call.  (Until round 4 every call took a process-wide mutex and uploaded its two reads.)
constexpr int kGactSlots = 32;
struct GactEngine {
    gact_hip_engine *e = nullptr;
    std::atomic<int> next_slot{0};
};
std::map<Key, GactEngine *> g_gact;

GactEngine *gact_engine(int tile, int overlap, int match, int mismatch, int open, int ext, int thr)
{
    std::lock_guard<std::mutex> lk(g_side_mu);
    const Key k(tile, overlap, match, mismatch, open, ext, thr);
    auto it = g_gact.find(k);
    if (it != g_gact.end()) return it->second;
    gact_hip_params p;
    memset(&p, 0, sizeof p);
    p.tile_size = tile; p.tile_overlap = overlap;
    p.match = match; p.mismatch = mismatch; p.gap_open = open; p.gap_extend = ext;
    p.first_tile_score_threshold = thr; p.device_id = shim_device(); p.n_slots = kGactSlots;
    p.max_blocks = 4;
    GactEngine *g = new GactEngine();
    SAFE(gact_hip_create(&p, &g->e));
    upload_set(g->e, GACT_SET_REF, reference_seqs);
    upload_set(g->e, GACT_SET_QUERY, reads_seqs);
    upload_set(g->e, GACT_SET_QUERY_RC, rev_reads_seqs);
    if (g_gact.empty())
        atexit([] {
            for (auto &kv : g_gact) { gact_hip_destroy(kv.second->e); delete kv.second; }
            g_gact.clear();
        });
    g_gact[k] = g;
    return g;
}

// is [str, str + len) the driver's sequence `id` of `set`?
bool is_resident(const std::vector<std::string> &set, int id, const char *str, int len)
{
    return id >= 0 && (size_t)id < set.size() && set[id].size() == (size_t)len &&
           (set[id].data() == str || memcmp(set[id].data(), str, (size_t)len) == 0);
}

}  // namespace

namespace {

// GACT_HIP_PAIR_STRANDS=1 (opt-in).  darwin.cpp:429-433 calls GACT_Batch twice per feeder thread: the forward-strand
// calls, then the reverse-complement ones.  Either call is half of the thread's work, and a launch with half the
// candidates lasts almost as long as one with all of them (it lasts as long as its longest chain): one after the other
// they take nearly twice the time.  With the switch set the forward call only KEEPS its calls; the reverse-complement
// call that follows on the same GPU_storage runs both strands in one launch (gact_hip_candidates_run_mixed) and writes
// the forward lines, then its own.  Any other call in between (or GPU_close) first runs what was kept, on its own, and
// writes it to the ofstream it came with -- which therefore has to be alive then: that is why this is a switch and not
// the default.  The lines are the same either way.
struct KeptForward {
    std::vector<gact_candidate> cands;
    std::ofstream *fout = nullptr;
};
std::mutex g_kept_mu;
std::map<std::pair<void *, int>, KeptForward> g_kept;          // by (engine, slot)

bool pair_strands() { static const bool on = getenv("GACT_HIP_PAIR_STRANDS") && atoi(getenv("GACT_HIP_PAIR_STRANDS")) != 0; return on; }
bool time_prints() { static const bool on = getenv("GACT_HIP_TIME") != nullptr; return on; }

void to_candidates(const std::vector<GACT_call> &calls, int num_calls, std::vector<gact_candidate> &cands)
{
    const size_t base = cands.size();
    cands.resize(base + (size_t)num_calls);
    for (int k = 0; k < num_calls; k++) {
        gact_candidate &c = cands[base + k];
        c.ref_id = calls[k].ref_id; c.query_id = calls[k].query_id;
        c.ref_pos = calls[k].ref_pos; c.query_pos = calls[k].query_pos;
    }
}

// candidates [0, rc_from) forward, the rest reverse-complement: one run, lines to the two streams; returns ms inside the engine
long run_and_print(gact_hip_engine *e, int slot, const std::vector<gact_candidate> &cands, int rc_from, std::ofstream &fout_f,
                   std::ofstream &fout_r)
{
    const int n = (int)cands.size();
    if (n == 0) return 0;
    std::vector<gact_overlap> out((size_t)n);
    const auto t1 = std::chrono::high_resolution_clock::now();
    SAFE(gact_hip_candidates_upload(e, slot, n, cands.data()));
    const auto tu = std::chrono::high_resolution_clock::now();
    SAFE(gact_hip_candidates_run_mixed(e, slot, 0, n, rc_from, same_file ? 1 : 0));
    const auto tr = std::chrono::high_resolution_clock::now();
    SAFE(gact_hip_candidates_fetch(e, slot, n, out.data()));
    const auto t2 = std::chrono::high_resolution_clock::now();
    if (time_prints()) {
        auto us = [](std::chrono::high_resolution_clock::time_point a, std::chrono::high_resolution_clock::time_point b) {
            return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        gact_hip_run_stats st;
        memset(&st, 0, sizeof st);
        (void)gact_hip_last_run_stats(e, slot, &st);
        printf("time_gpu split, slot %d: upload %ld us, submit %ld us, wait + fetch %ld us (launch: %.2f ms on the device, %d callers merged)\n",
               slot, us(t1, tu), us(tu, tr), us(tr, t2), st.total_ms, st.merged_callers);
    }
    // (gact.cpp:214-224 ends every line with std::endl; the lines of a batch are written and flushed together here)
    std::string buf;
    buf.reserve((size_t)n * 96);
    for (int k = 0; k < rc_from && k < n; k++)
        if (out[k].emitted) append_overlap(buf, out[k]);
    if (&fout_r != &fout_f) { fout_f.write(buf.data(), (std::streamsize)buf.size()); fout_f.flush(); buf.clear(); }
    for (int k = rc_from < n ? (rc_from < 0 ? 0 : rc_from) : n; k < n; k++)
        if (out[k].emitted) append_overlap(buf, out[k]);
    fout_r.write(buf.data(), (std::streamsize)buf.size());
    fout_r.flush();
    return (long)std::chrono::duration_cast<std::chrono::milliseconds>(t2 - t1).count();
}

// what an earlier forward call left with this (engine, slot), taken out of the table
bool take_kept(void *engine, int slot, KeptForward &out)
{
    std::lock_guard<std::mutex> lk(g_kept_mu);
    auto it = g_kept.find(std::make_pair(engine, slot));
    if (it == g_kept.end()) return false;
    out = std::move(it->second);
    g_kept.erase(it);
    return true;
}

}  // namespace

void GPU_init(int tile_size_, int tile_overlap_, int gap_open, int gap_extend, int match, int mismatch,
              int early_terminate, std::vector<GPU_storage> *s, int num_threads)
{
    (void)early_terminate;   // always tile_size - tile_overlap (darwin.cpp:611)
    gact_hip_params p;
    memset(&p, 0, sizeof p);
    p.tile_size = tile_size_; p.tile_overlap = tile_overlap_;
    p.match = match; p.mismatch = mismatch; p.gap_open = gap_open; p.gap_extend = gap_extend;
    p.first_tile_score_threshold = first_tile_score_threshold;
    p.device_id = shim_device();           // 0 like cuda_host.cu:195 unless GACT_HIP_DEVICE names another GPU
    p.n_slots = num_threads;
    SAFE(gact_hip_create(&p, &g_main.engine));
    g_main.params = p;
    for (int i = 0; i < num_threads; ++i) {
        GPU_storage st;
        st.engine = g_main.engine; st.slot = i; st.reserved = 0;
        s->push_back(st);
    }
    // darwin.cpp has its three read sets in memory when it calls GPU_init (FASTA and reverse complements are read before
    // :611): they become resident now, outside the feeder threads' "Time GACT calling".  (The -DGPU build recodes the
    // strings to 0..3 afterwards, darwin.cpp:314-398; the engine takes either form and packs both to the same 2-bit image.)
    // A caller whose reads come later gets them uploaded by its first batch, as before.
    if (!reference_seqs.empty() && !reads_seqs.empty() && !rev_reads_seqs.empty()) {
        ensure_reads_resident();
        // (arrays for a job of ~16 candidates per read -- the filter's yield on 10x PacBio-shape reads is 14 --, streams, one
        //  empty launch of the chain kernels: gact_hip.h gact_hip_prepare)
        const size_t guess = std::min<size_t>(16 * reads_seqs.size(), (size_t)4 << 20);
        SAFE(gact_hip_prepare(g_main.engine, (int32_t)guess));
    }
    gact_hip_device_info info;
    SAFE(gact_hip_get_device_info(g_main.engine, &info));
    printf("%s: %d CUs, %lld MB HBM, %d feeder slots\n", info.arch, info.compute_units,
           (long long)(info.hbm_bytes >> 20), num_threads);
}

void GPU_close(std::vector<GPU_storage> *s, int num_threads)
{
    (void)num_threads;
    for (const GPU_storage &st : *s) {                       // (GACT_HIP_PAIR_STRANDS: forward calls nobody followed up)
        KeptForward kept;
        if (take_kept(st.engine, st.slot, kept)) run_and_print((gact_hip_engine *)st.engine, st.slot, kept.cands, (int)kept.cands.size(), *kept.fout, *kept.fout);
    }
    if (g_main.engine) gact_hip_destroy(g_main.engine);
    g_main.engine = nullptr;
    s->clear();
}

void GACT_Batch(std::vector<GACT_call> calls, int num_calls, bool complement, int offset, GPU_storage *s,
                int match_score, int mismatch_score, int gap_open, int gap_extend, std::ofstream &fout)
{
    (void)offset; (void)match_score; (void)mismatch_score; (void)gap_open; (void)gap_extend;
    printf("GACT_Batch, num_calls: %d, complement: %d\n", num_calls, complement);   // gact.cpp:249
    const auto t0 = std::chrono::high_resolution_clock::now();
    long time_gpu = 0;
    gact_hip_engine *e = (gact_hip_engine *)s->engine;
    KeptForward kept;
    const bool have_kept = pair_strands() && take_kept(s->engine, s->slot, kept);
    if (num_calls > 0 || have_kept) ensure_reads_resident();
    if (pair_strands() && !complement) {
        // a forward call: what an earlier one left is run now, on its own; this one is kept for the reverse-complement call
        if (have_kept) time_gpu += run_and_print(e, s->slot, kept.cands, (int)kept.cands.size(), *kept.fout, *kept.fout);
        if (num_calls > 0) {
            KeptForward k;
            to_candidates(calls, num_calls, k.cands);
            k.fout = &fout;
            std::lock_guard<std::mutex> lk(g_kept_mu);
            g_kept[std::make_pair(s->engine, s->slot)] = std::move(k);
        }
    } else {
        std::vector<gact_candidate> cands;
        int rc_from = 0x7fffffff;
        std::ofstream *fout_f = &fout;
        if (have_kept) { cands = std::move(kept.cands); fout_f = kept.fout; }
        if (complement) rc_from = (int)cands.size();
        if (num_calls > 0) to_candidates(calls, num_calls, cands);
        if (!complement) rc_from = (int)cands.size();
        time_gpu += run_and_print(e, s->slot, cands, rc_from, *fout_f, fout);
    }
    if (time_prints()) {
        // the reference's -D TIME line (gact.cpp:291-295,412-424,554-558): milliseconds in the host loop / in the device calls
        const long total = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        printf("time_loop: %ld ms, time_gpu: %ld ms\n", total - time_gpu, time_gpu);
    }
}

int *Align_Batch_GPU(std::vector<std::string> ref_seqs, std::vector<std::string> query_seqs,
                     std::vector<int> ref_lens, std::vector<int> query_lens,
                     int *sub_mat, int gap_open, int gap_extend,
                     std::vector<int> ref_poss, std::vector<int> query_poss,
                     std::vector<char> reverses, std::vector<char> firsts,
                     int early_terminate, int tile_size_, GPU_storage *s,
                     int num_blocks, int threads_per_block)
{
    (void)sub_mat; (void)gap_open; (void)gap_extend; (void)ref_poss; (void)query_poss; (void)early_terminate;
    const int batch = num_blocks * threads_per_block;
    gact_hip_engine *e = (gact_hip_engine *)s->engine;
    const int stride = tile_size_ > 0 ? tile_size_ : 1;
    std::vector<uint8_t> rb((size_t)batch * stride), qb((size_t)batch * stride), rev(batch), fst(batch);
    std::vector<int32_t> rl(batch), ql(batch);
    for (int t = 0; t < batch; t++) {
        rl[t] = ref_lens[t];
        ql[t] = (ref_lens[t] == -1) ? 0 : query_lens[t];
        if (ref_lens[t] == -1) { rev[t] = 0; fst[t] = 0; continue; }
        // the staging rows are tile_size wide (cuda_host.cu:56-61 sizes its own the same way)
        if (rl[t] < 0 || rl[t] > stride || ql[t] < 0 || ql[t] > stride || (size_t)rl[t] > ref_seqs[t].size() ||
            (size_t)ql[t] > query_seqs[t].size()) {
            printf("\nAlign_Batch_GPU failed: tile %d has lengths %d x %d, tile_size is %d\n\n", t, rl[t], ql[t], tile_size_);
            exit(-1);
        }
        memcpy(rb.data() + (size_t)t * stride, ref_seqs[t].data(), (size_t)rl[t]);
        memcpy(qb.data() + (size_t)t * stride, query_seqs[t].data(), (size_t)ql[t]);
        // reverses[t]==1 (towards 0) keeps the bytes as they are, ==0 byte-reverses them
        // (cuda_host.cu:92-142): the opposite of AlignWithBT's `reverse`
        rev[t] = reverses[t] == 1 ? 0 : 1;
        fst[t] = firsts[t] == 1 ? 1 : 0;
    }
    std::vector<gact_tile_result> res(batch);
    const int sstride = 2 * tile_size_;
    std::vector<uint8_t> st((size_t)batch * sstride);
    SAFE(gact_hip_align_tiles_inline(e, s->slot, batch, rb.data(), qb.data(), stride, rl.data(), ql.data(),
                                     rev.data(), fst.data(), res.data(), st.data(), sstride));
    int *out = (int *)malloc((size_t)batch * sizeof(int) * 2 * tile_size_);
    for (int t = 0; t < batch; t++) {
        int *o = out + (size_t)t * 2 * tile_size_;
        if (ref_lens[t] == -1) { o[0] = 0; o[1] = o[2] = o[3] = o[4] = 0; o[5] = -1; continue; }
        o[0] = res[t].score; o[1] = res[t].ref_steps; o[2] = res[t].query_steps;
        o[3] = res[t].max_i; o[4] = res[t].max_j;
        const int n = res[t].n_states < sstride - 6 ? res[t].n_states : sstride - 6;
        for (int k = 0; k < n; k++) o[5 + k] = st[(size_t)t * sstride + k];
        o[5 + n] = -1;
    }
    return out;
}

std::queue<int> AlignWithBT(char *ref_seq, long long int ref_len, char *query_seq, long long int query_len,
                            int match_score, int mismatch_score, int gap_open, int gap_extend,
                            int query_pos, int ref_pos, bool reverse, bool first, int early_terminate)
{
    std::queue<int> q;
    if (ref_len < 0 || query_len < 0 || ref_pos < 0 || query_pos < 0) {
        printf("\nAlignWithBT: negative length or position (%lld x %lld, pos %d,%d)\n\n", ref_len, query_len, ref_pos,
               query_pos);
        exit(-1);
    }
    if (!first) {
        // (ref_pos, query_pos) is where pos_score is read and where the traceback starts (align.cpp:179-181,186).
        // A cell depends on nothing below or right of it, so the answer is that of the tile cut down to ref_pos x
        // query_pos: the first ref_pos bases in DP order, i.e. the slice's prefix, or its suffix when `reverse`
        // (align.cpp:130-131).  On or beyond the border the pointer is ZERO (:101-107; cells past the tile are
        // never written) and pos_score stays 0 (:104).
        if (ref_pos == 0 || query_pos == 0 || ref_pos > ref_len || query_pos > query_len) {
            q.push(0);
            return q;
        }
        if (reverse) { ref_seq += ref_len - ref_pos; query_seq += query_len - query_pos; }
        ref_len = ref_pos; query_len = query_pos;
    }
    if (ref_len > GACT_HIP_MAX_TILE || query_len > GACT_HIP_MAX_TILE) {
        printf("\nAlignWithBT: tile %lld x %lld is larger than %d (the reference asserts the same, align.cpp:66-67)\n\n", ref_len,
               query_len, GACT_HIP_MAX_TILE);
        exit(-1);
    }
    std::lock_guard<std::mutex> lk(g_side_mu);
    // an engine per tile class: the register-tiled kernels up to GACT_HIP_FAST_TILE, the one-wave-per-tile kernels beyond
    const int tile = (ref_len <= GACT_HIP_FAST_TILE && query_len <= GACT_HIP_FAST_TILE) ? GACT_HIP_FAST_TILE : GACT_HIP_MAX_TILE;
    int early = early_terminate < 1 ? 1 : (early_terminate > tile ? tile : early_terminate);
    gact_hip_engine *e = side_engine(tile, tile - early, match_score, mismatch_score, gap_open, gap_extend, 1);
    const int32_t rl = (int32_t)ref_len, ql = (int32_t)query_len;
    const uint8_t rv = reverse ? 1 : 0, fs = first ? 1 : 0;
    gact_tile_result res;
    std::vector<uint8_t> st(2 * tile);
    std::vector<uint8_t> rbuf(tile, 0), qbuf(tile, 0);
    memcpy(rbuf.data(), ref_seq, (size_t)rl);
    memcpy(qbuf.data(), query_seq, (size_t)ql);
    SAFE(gact_hip_align_tiles_inline(e, 0, 1, rbuf.data(), qbuf.data(), tile, &rl, &ql, &rv, &fs, &res, st.data(),
                                     2 * tile));
    q.push(res.score);
    if (first) { q.push(res.max_i); q.push(res.max_j); }
    if (early_terminate >= 1)
        for (int k = 0; k < res.n_states; k++) q.push(st[k]);
    return q;
}

std::vector<std::queue<int> > Align_Batch(std::vector<std::string> ref_seqs, std::vector<std::string> query_seqs,
                                          std::vector<int> ref_lens, std::vector<int> query_lens,
                                          int match_score, int mismatch_score, int gap_open, int gap_extend,
                                          std::vector<int> ref_poss_b, std::vector<int> query_poss_b,
                                          std::vector<char> reverses, std::vector<char> firsts, int early_terminate)
{
    std::vector<std::queue<int> > result;
    for (size_t j = 0; j < ref_seqs.size(); ++j) {
        if (ref_lens[j] == -1) { result.push_back(std::queue<int>()); continue; }    // align.cpp:40-44
        result.push_back(AlignWithBT((char *)ref_seqs[j].c_str(), ref_lens[j], (char *)query_seqs[j].c_str(),
                                     query_lens[j], match_score, mismatch_score, gap_open, gap_extend,
                                     query_poss_b[j], ref_poss_b[j], reverses[j] == 1, firsts[j] == 1,
                                     early_terminate));
    }
    return result;
}

void GACT(char *ref_str, char *query_str, int ref_length, int query_length, int tile_size_, int tile_overlap_,
          int ref_pos, int query_pos, int first_tile_score_threshold_, int ref_id, int query_id, bool complement,
          int match_score, int mismatch_score, int gap_open, int gap_extend, std::ofstream &fout)
{
    // the driver's own reads: resident once, this thread's slot, no lock
    if (tile_size_ <= GACT_HIP_FAST_TILE && is_resident(reference_seqs, ref_id, ref_str, ref_length) &&
        is_resident(complement ? rev_reads_seqs : reads_seqs, query_id, query_str, query_length)) {
        GactEngine *g = gact_engine(tile_size_, tile_overlap_, match_score, mismatch_score, gap_open, gap_extend,
                                    first_tile_score_threshold_);
        thread_local std::map<GactEngine *, int> my_slot;
        auto it = my_slot.find(g);
        if (it == my_slot.end()) it = my_slot.insert(std::make_pair(g, g->next_slot.fetch_add(1))).first;
        if (it->second < kGactSlots) {
            gact_candidate c;
            c.ref_id = ref_id; c.query_id = query_id; c.ref_pos = ref_pos; c.query_pos = query_pos;
            gact_overlap o;
            SAFE(gact_hip_extend_candidates(g->e, it->second, 1, &c, complement ? 1 : 0, same_file ? 1 : 0, &o));
            if (o.emitted) { print_overlap(fout, o); fout.flush(); }                      // gact.cpp:213
            return;
        }
    }
    std::lock_guard<std::mutex> lk(g_side_mu);
    gact_hip_engine *e = side_engine(tile_size_, tile_overlap_, match_score, mismatch_score, gap_open, gap_extend,
                                     first_tile_score_threshold_);
    // the two reads of this one call become a two-sequence resident set
    const int64_t roffs[2] = {0, ref_length}, qoffs[2] = {0, query_length};
    SAFE(gact_hip_upload_seqs(e, GACT_SET_REF, (const uint8_t *)ref_str, roffs, 1));
    SAFE(gact_hip_upload_seqs(e, complement ? GACT_SET_QUERY_RC : GACT_SET_QUERY, (const uint8_t *)query_str, qoffs, 1));
    gact_candidate c;
    c.ref_id = 0; c.query_id = 0; c.ref_pos = ref_pos; c.query_pos = query_pos;
    gact_overlap o;
    SAFE(gact_hip_extend_candidates(e, 0, 1, &c, complement ? 1 : 0, 0, &o));
    o.ref_id = ref_id; o.query_id = query_id;
    if (!(same_file && ref_id == query_id) && o.score > 0) { print_overlap(fout, o); fout.flush(); }     // gact.cpp:213
}
