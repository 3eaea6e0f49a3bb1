// darwin_hip.cpp -- a darwin.cpp-shaped driver around the GACT shim.
//
//   darwin_hip <REF.fasta> <READS.fasta> CPU_THREADS [--params params.cfg]
//              [--candidates FILE | --dump-candidates FILE [--dsoft-only]] [--device-dsoft]
//              [--device D] [--shard R/W] [--recode]
//
// Plays the part of reference darwin.cpp:451-646 for the GACT stage: owns the
// globals gact.cpp reads, loads params.cfg and the two FASTA files, builds the
// reverse complements (darwin.cpp:110-147), GPU_init, fans candidates out over
// feeder threads (contiguous ranges, darwin.cpp:619-629), each thread calling
// GACT_Batch for its forward and then its reverse-complement calls
// (darwin.cpp:429-433) into darwin.<thread>.out, GPU_close.
//
// Candidates come from the D-SOFT restatement (dsoft.cpp; darwin.cpp:209-288 per read:
// forward strand, then reverse complement) or, with --candidates, from FILE: int32
// records {ref_id, query_id, ref_pos, query_pos, comp}.  --dump-candidates writes the
// same format; --dsoft-only stops after the filter (no GPU is touched).
// --device-dsoft runs the filter on the GPU as well (gact_hip_dsoft_build / _query): every feeder thread
// filters its read range straight into its slot's device candidate array and extends it from there.
//
// --device D: the GPU this process uses (the reference is single-device, cuda_host.cu:195).  --shard R/W: this
// process is rank R of W -- it extends every W-th candidate (host filter) or the R-th contiguous range of reads
// (--device-dsoft) and writes darwin.<R>.<thread>.out; the union of all ranks' files is the W = 1 output, and the
// reference's canonical form `cat darwin.*.out | sort | uniq` (README:25) is the gather.  --recode: hand the
// read sets to GACT_Batch the way the reference's -DGPU build does, recoded in place to A0 C1 T2 G3
// (darwin.cpp:314-398).  Stage timers are printed with the reference's labels (darwin.cpp:300,405,441,553-639).
//
//   darwin_hip --selftest FILE   exercises AlignWithBT / Align_Batch / Align_Batch_GPU / GACT
//                                on the cases in FILE and prints what they return.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <chrono>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "align.h"
#include "dsoft.h"
#include "gact.h"
#include "gact_hip.h"

// ---- the globals of darwin.cpp:39-93 that gact.cpp / the shim read
bool same_file = false;
int NUM_BLOCKS = 32, THREADS_PER_BLOCK = 64, BATCH_SIZE = 2048;
int match_score = 1, mismatch_score = -1, gap_open = -1, gap_extend = -1;
int first_tile_score_threshold = 35;
int tile_size = 320, tile_overlap = 120;
int num_threads = 1;
std::vector<long long int> reference_lengths, reads_lengths;
std::vector<std::string> reference_seqs, reads_seqs, rev_reads_seqs;
std::vector<std::vector<std::string> > reference_descrips, reads_descrips;

static std::string rev_comp(const std::string &seq)
{
    std::string rc;
    rc.reserve(seq.size());
    for (size_t k = seq.size(); k-- > 0;) {
        switch (seq[k]) {                       // darwin.cpp:122-142
            case 'a': rc += 't'; break; case 'A': rc += 'T'; break;
            case 'c': rc += 'g'; break; case 'C': rc += 'G'; break;
            case 'g': rc += 'c'; break; case 'G': rc += 'C'; break;
            case 't': rc += 'a'; break; case 'T': rc += 'A'; break;
            case 'n': rc += 'n'; break; case 'N': rc += 'N'; break;
            default: std::cerr << "Bad Nt char: " << seq[k] << std::endl; exit(1);
        }
    }
    return rc;
}

// header fields split on anything but [A-Za-z0-9_] (fasta.cpp:19-33); field 0 is the printed name
static std::vector<std::string> split_header(const std::string &h)
{
    std::vector<std::string> out;
    std::string cur;
    for (char c : h) {
        const bool ok = (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == '_';
        if (ok) cur += c;
        else if (!cur.empty()) { out.push_back(cur); cur.clear(); }
    }
    if (!cur.empty()) out.push_back(cur);
    if (out.empty()) out.push_back("");
    return out;
}

static void parse_fasta(const std::string &path, std::vector<std::vector<std::string> > &descrips,
                        std::vector<std::string> &seqs, std::vector<long long int> &lengths)
{
    std::ifstream in(path);
    if (!in) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(1); }
    std::string line, cur;
    bool have = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            if (have) { seqs.push_back(cur); lengths.push_back((long long)cur.size()); }
            descrips.push_back(split_header(line.substr(1)));
            cur.clear(); have = true;
        } else if (have) {
            cur += line;
        }
    }
    if (have) { seqs.push_back(cur); lengths.push_back((long long)cur.size()); }
}

// [section] key = value, '#' / ';' comments (ConfigFile.cpp:30-56); values through atof (Chameleon.cpp:89-91)
static std::map<std::string, double> parse_cfg(const std::string &path)
{
    std::map<std::string, double> kv;
    std::ifstream in(path);
    std::string line, section;
    while (std::getline(in, line)) {
        const size_t a = line.find_first_not_of(" \t\r");
        if (a == std::string::npos) continue;
        if (line[a] == '#' || line[a] == ';') continue;
        if (line[a] == '[') { section = line.substr(a + 1, line.find(']') - a - 1); continue; }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        std::string key = line.substr(a, eq - a), val = line.substr(eq + 1);
        key.erase(key.find_last_not_of(" \t") + 1);
        kv[section + "/" + key] = atof(val.c_str());
    }
    return kv;
}

struct Cand { int ref_id, query_id, ref_pos, query_pos, comp; };

// ---- stage timers with the reference's labels (darwin.cpp:300,405,441 per thread; :553,574,596,639 in main)
static std::mutex io_lock;
typedef std::chrono::steady_clock::time_point Tick;
static Tick now() { return std::chrono::steady_clock::now(); }
static void print_stage(const char *label, Tick a, Tick b)
{
    const long ms = (long)(std::chrono::duration<double, std::milli>(b - a).count() + 0.5);
    std::lock_guard<std::mutex> lk(io_lock);
    std::cout << label << ": " << ms << " msec" << std::endl;
}

static int shard_rank = 0, shard_world = 1;
static std::string out_name(int cpu_id)
{
    // darwin.cpp:174; ranks of a sharded run keep their files apart
    return shard_world > 1 ? "darwin." + std::to_string(shard_rank) + "." + std::to_string(cpu_id) + ".out"
                           : "darwin." + std::to_string(cpu_id) + ".out";
}

// darwin.cpp:314-398: the -DGPU build recodes every base in place before GACT_Batch; anything else stays
static void recode_in_place(std::vector<std::string> &seqs)
{
    for (std::string &r : seqs)
        for (char &c : r)
            switch (c) { case 'A': c = 0; break; case 'C': c = 1; break; case 'T': c = 2; break; case 'G': c = 3; break; default: break; }
}

static void feeder(int cpu_id, const std::vector<Cand> *all, size_t lo, size_t hi, GPU_storage s)
{
    std::ofstream fout(out_name(cpu_id));
    const Tick t0 = now();
    std::vector<GACT_call> calls_for, calls_rev;
    for (size_t k = lo; k < hi; k++) {
        const Cand &c = (*all)[k];
        GACT_call g;                                  // darwin.cpp:227-238
        g.ref_id = c.ref_id; g.query_id = c.query_id;
        g.ref_pos = c.ref_pos; g.query_pos = c.query_pos;
        g.ref_bpos = c.ref_pos; g.query_bpos = c.query_pos;
        g.score = 0; g.first_tile_score = 0; g.first = 1; g.reverse = 1;
        (c.comp ? calls_rev : calls_for).push_back(g);
    }
    GACT_Batch(calls_for, (int)calls_for.size(), false, 0, &s, match_score, mismatch_score, gap_open, gap_extend, fout);
    GACT_Batch(calls_rev, (int)calls_rev.size(), true, (int)calls_for.size(), &s, match_score, mismatch_score,
               gap_open, gap_extend, fout);
    fout.close();
    print_stage("Time GACT calling", t0, now());                 // darwin.cpp:441
}

// --device-dsoft: reads [lo, hi) are filtered and extended on the device, slot s.slot
static void device_feeder(int cpu_id, int lo, int hi, GPU_storage s, std::vector<Cand> *dump)
{
    gact_hip_engine *e = (gact_hip_engine *)s.engine;
    std::ofstream fout(out_name(cpu_id));
    int32_t nf = 0, nr = 0;
    float ms = 0;
    auto check = [](int rc, const char *what) {
        if (rc != 0) { printf("\n%s failed: %s\n\n", what, gact_hip_last_error()); exit(-1); }
    };
    const Tick t0 = now();
    check(gact_hip_dsoft_query(e, s.slot, lo, hi - lo, &nf, &nr, &ms), "gact_hip_dsoft_query");
    const Tick t1 = now();
    print_stage("Time finding seeds", t0, t1);                   // darwin.cpp:300
    const int32_t n = nf + nr;
    if (dump) {
        std::vector<gact_candidate> c((size_t)n);
        check(gact_hip_candidates_download(e, s.slot, n, c.data()), "gact_hip_candidates_download");
        for (int32_t k = 0; k < n; k++) dump->push_back(Cand{c[k].ref_id, c[k].query_id, c[k].ref_pos, c[k].query_pos, k >= nf});
        return;
    }
    std::vector<gact_overlap> o((size_t)n);
    check(gact_hip_candidates_run_mixed(e, s.slot, 0, n, nf, same_file), "gact_hip_candidates_run_mixed");
    check(gact_hip_candidates_fetch(e, s.slot, n, o.data()), "gact_hip_candidates_fetch");
    char line[1024];
    for (const gact_overlap &r : o) {
        if (!r.emitted) continue;
        const int len = gact_hip_format_overlap(&r, reference_descrips[r.ref_id][0].c_str(),
                                                reads_descrips[r.query_id][0].c_str(), line, sizeof line);
        fout.write(line, len);
    }
    print_stage("Time GACT calling", t1, now());                 // darwin.cpp:441
}

// --rccl-gather IDFILE (with --shard R/W): this rank's share as ONE run on the engine (no feeder threads: a run is fastest
// with all of its candidates in one launch), then the job's one collective -- every rank's records to rank 0 over RCCL,
// out of the engines' device arrays (gact_hip_comm_gather_lines) -- and rank 0 writes darwin.gathered.out: the lines of
// rank 0, then rank 1's, ..., each rank's forward-strand lines first.  `cat darwin.*.out | sort | uniq` of the file-based
// form and `sort | uniq` of this file are the same lines.
static int gathered_run(const std::vector<std::vector<Cand> > &per_thread, const std::string &id_path)
{
    auto check = [](int rc, const char *what) {
        if (rc != 0) { printf("\n%s failed: %s\n\n", what, gact_hip_last_error()); exit(-1); }
    };
    std::vector<gact_candidate> cands;
    for (int comp = 0; comp < 2; comp++)
        for (const auto &v : per_thread)
            for (const Cand &c : v)
                if ((c.comp != 0) == (comp != 0)) cands.push_back(gact_candidate{c.ref_id, c.query_id, c.ref_pos, c.query_pos});
    int32_t nf = 0;
    for (const auto &v : per_thread)
        for (const Cand &c : v) nf += c.comp ? 0 : 1;
    const int32_t n = (int32_t)cands.size();
    std::vector<GPU_storage> s;
    GPU_init(tile_size, tile_overlap, gap_open, gap_extend, match_score, mismatch_score, tile_size - tile_overlap, &s, 1);
    gact_hip_engine *e = (gact_hip_engine *)s[0].engine;
    gact_hip_comm *comm = nullptr;
    check(gact_hip_comm_create(e, shard_rank, shard_world, id_path.c_str(), 0, &comm), "gact_hip_comm_create");
    const Tick t0 = now();
    check(gact_hip_candidates_upload(e, 0, n, cands.data()), "gact_hip_candidates_upload");
    check(gact_hip_candidates_run_mixed(e, 0, 0, n, nf, same_file), "gact_hip_candidates_run_mixed");
    const Tick t1 = now();
    std::vector<int64_t> counts((size_t)shard_world, 0);
    // (rank 0 cannot know the total before the call: room for every rank's share at its largest -- the deal is round-robin)
    std::vector<gact_line> lines(shard_rank == 0 ? ((size_t)n + 1) * (size_t)shard_world : 0);
    check(gact_hip_comm_gather_lines(comm, 0, n, counts.data(), shard_rank == 0 ? lines.data() : nullptr, (int64_t)lines.size()),
          "gact_hip_comm_gather_lines");
    const Tick t2 = now();
    print_stage("Time GACT calling", t0, t1);                                   // darwin.cpp:441 (launch; the wait is in the gather)
    print_stage("Time gathering records (RCCL)", t1, t2);
    if (shard_rank == 0) {
        int64_t total = 0;
        printf("gathered records per rank:");
        for (int r = 0; r < shard_world; r++) { printf(" %lld", (long long)counts[(size_t)r]); total += counts[(size_t)r]; }
        printf("\n");
        std::ofstream fout("darwin.gathered.out");
        char line[1024];
        for (int64_t k = 0; k < total; k++) {
            const gact_line &l = lines[(size_t)k];
            if (!(l.comp_emitted & 2)) continue;
            gact_overlap o;
            memset(&o, 0, sizeof o);
            o.ref_id = l.ref_id; o.query_id = l.query_id; o.ab = l.ab; o.ae = l.ae; o.bb = l.bb; o.be = l.be;
            o.score = l.score; o.comp = l.comp_emitted & 1; o.emitted = 1;
            const int len = gact_hip_format_overlap(&o, reference_descrips[o.ref_id][0].c_str(), reads_descrips[o.query_id][0].c_str(),
                                                    line, sizeof line);
            fout.write(line, len);
        }
    }
    gact_hip_comm_destroy(comm);
    GPU_close(&s, 1);
    return 0;
}

static void upload_set(gact_hip_engine *e, int which, const std::vector<std::string> &seqs)
{
    std::vector<int64_t> offs(seqs.size() + 1, 0);
    for (size_t k = 0; k < seqs.size(); k++) offs[k + 1] = offs[k] + (int64_t)seqs[k].size();
    std::vector<uint8_t> cat((size_t)offs.back());
    for (size_t k = 0; k < seqs.size(); k++) memcpy(cat.data() + offs[k], seqs[k].data(), seqs[k].size());
    if (gact_hip_upload_seqs(e, which, cat.data(), offs.data(), (int32_t)seqs.size()) != 0) {
        printf("\nupload failed: %s\n\n", gact_hip_last_error());
        exit(-1);
    }
}

static void print_queue(const char *tag, std::queue<int> q)
{
    printf("%s", tag);
    while (!q.empty()) { printf(" %d", q.front()); q.pop(); }
    printf("\n");
}

// FILE lines:  T ref query match mismatch open ext reverse first early     -> AlignWithBT + Align_Batch_GPU
//              G ref query ref_pos query_pos tile overlap thr match mismatch open ext comp -> GACT
static int selftest(const char *path)
{
    std::ifstream in(path);
    std::string kind;
    std::vector<GPU_storage> s;
    bool inited = false;
    int n = 0;
    while (in >> kind) {
        if (kind == "T") {
            std::string r, q; int m, x, o, e, rev, first, early;
            in >> r >> q >> m >> x >> o >> e >> rev >> first >> early;
            print_queue("AlignWithBT", AlignWithBT((char *)r.c_str(), (long long)r.size(), (char *)q.c_str(),
                                                   (long long)q.size(), m, x, o, e, (int)q.size(), (int)r.size(),
                                                   rev != 0, first != 0, early));
            if (m == 1 && x == -1 && o == -1 && e == -1 && early == tile_size - tile_overlap &&
                (int)r.size() <= tile_size && (int)q.size() <= tile_size) {
                if (!inited) {
                    NUM_BLOCKS = 1; THREADS_PER_BLOCK = 4; BATCH_SIZE = 4;
                    GPU_init(tile_size, tile_overlap, gap_open, gap_extend, match_score, mismatch_score,
                             tile_size - tile_overlap, &s, 1);
                    inited = true;
                }
                // slot 1 of a 4-slot batch, the others idle (ref_len -1, gact.cpp:303-305)
                std::vector<std::string> rs(4), qs(4);
                std::vector<int> rl(4, -1), ql(4, 0);
                std::vector<char> rv(4, 0), fs(4, 0);
                rs[1] = r; qs[1] = q; rl[1] = (int)r.size(); ql[1] = (int)q.size();
                rv[1] = rev ? 0 : 1;      // Align_Batch_GPU's sense is the opposite of AlignWithBT's
                fs[1] = (char)first;
                int *out = Align_Batch_GPU(rs, qs, rl, ql, nullptr, gap_open, gap_extend, rl, ql, rv, fs,
                                           tile_size - tile_overlap, tile_size, &s[0], NUM_BLOCKS, THREADS_PER_BLOCK);
                const int *o1 = out + 2 * tile_size;
                printf("Align_Batch_GPU %d %d %d %d %d :", o1[0], o1[1], o1[2], o1[3], o1[4]);
                for (int k = 5; o1[k] != -1; k++) printf(" %d", o1[k]);
                printf("\n");
                free(out);
            }
        } else if (kind == "P") {
            // P ref query match mismatch open ext reverse first early ref_pos query_pos -> AlignWithBT at a position
            std::string r, q; int m, x, o, e, rev, first, early, rp, qp;
            in >> r >> q >> m >> x >> o >> e >> rev >> first >> early >> rp >> qp;
            print_queue("AlignWithBT", AlignWithBT((char *)r.c_str(), (long long)r.size(), (char *)q.c_str(),
                                                   (long long)q.size(), m, x, o, e, qp, rp, rev != 0, first != 0, early));
        } else if (kind == "B") {
            // B n match mismatch open ext early, then n lines "ref query reverse first" ("-" "-" = idle, ref_len -1)
            int nb, m, x, o, e, early;
            in >> nb >> m >> x >> o >> e >> early;
            std::vector<std::string> rs(nb), qs(nb);
            std::vector<int> rl(nb), ql(nb);
            std::vector<char> rv(nb), fs(nb);
            for (int k = 0; k < nb; k++) {
                int rev, first;
                in >> rs[k] >> qs[k] >> rev >> first;
                if (rs[k] == "-") { rs[k].clear(); qs[k].clear(); rl[k] = -1; ql[k] = 0; }     // align.cpp:40-44
                else { rl[k] = (int)rs[k].size(); ql[k] = (int)qs[k].size(); }
                rv[k] = (char)rev; fs[k] = (char)first;
            }
            std::vector<std::queue<int> > res = Align_Batch(rs, qs, rl, ql, m, x, o, e, rl, ql, rv, fs, early);
            for (int k = 0; k < nb; k++) print_queue("Align_Batch", res[k]);
        } else if (kind == "G") {
            std::string r, q; int rp, qp, t, ov, thr, m, x, o, e, comp;
            in >> r >> q >> rp >> qp >> t >> ov >> thr >> m >> x >> o >> e >> comp;
            reference_descrips.assign(1, std::vector<std::string>(1, "refname"));
            reads_descrips.assign(2, std::vector<std::string>(1, "queryname"));
            same_file = false;
            const char *tmp = "selftest_gact.out";
            { std::ofstream fout(tmp);
              GACT((char *)r.c_str(), (char *)q.c_str(), (int)r.size(), (int)q.size(), t, ov, rp, qp, thr, 0, 1,
                   comp != 0, m, x, o, e, fout); }
            std::ifstream back(tmp); std::stringstream ss; ss << back.rdbuf();
            printf("GACT %s", ss.str().empty() ? "\n" : ss.str().c_str());
            remove(tmp);
        }
        n++;
    }
    if (inited) GPU_close(&s, 1);
    return n > 0 ? 0 : 1;
}

int main(int argc, char *argv[])
{
    if (argc >= 3 && strcmp(argv[1], "--selftest") == 0) return selftest(argv[2]);
    if (argc < 4) {
        fprintf(stderr, "Usage: darwin_hip <REFERENCE>.fasta <READS>.fasta CPU_THREADS --candidates FILE "
                        "[--params params.cfg]\n");
        return 1;
    }
    std::string cand_path, dump_path, cfg_path = "params.cfg", gather_id;
    bool dsoft_only = false, device_dsoft = false, recode = false;
    for (int a = 4; a < argc; a++) {
        if (!strcmp(argv[a], "--candidates") && a + 1 < argc) cand_path = argv[++a];
        else if (!strcmp(argv[a], "--dump-candidates") && a + 1 < argc) dump_path = argv[++a];
        else if (!strcmp(argv[a], "--params") && a + 1 < argc) cfg_path = argv[++a];
        else if (!strcmp(argv[a], "--dsoft-only")) dsoft_only = true;
        else if (!strcmp(argv[a], "--device-dsoft")) device_dsoft = true;
        else if (!strcmp(argv[a], "--recode")) recode = true;
        else if (!strcmp(argv[a], "--rccl-gather") && a + 1 < argc) gather_id = argv[++a];
        else if (!strcmp(argv[a], "--device") && a + 1 < argc) setenv("GACT_HIP_DEVICE", argv[++a], 1);   // read by GPU_init
        else if (!strcmp(argv[a], "--shard") && a + 1 < argc) {
            if (sscanf(argv[++a], "%d/%d", &shard_rank, &shard_world) != 2 || shard_world < 1 || shard_rank < 0 ||
                shard_rank >= shard_world) { fprintf(stderr, "--shard wants R/W with 0 <= R < W\n"); return 1; }
        } else { fprintf(stderr, "unknown option %s\n", argv[a]); return 1; }
    }
    if (recode && device_dsoft) { fprintf(stderr, "--recode: the device filter reads ASCII sets\n"); return 1; }
    std::map<std::string, double> cfg = parse_cfg(cfg_path);
    auto get = [&](const char *k, int dflt) { return cfg.count(k) ? (int)cfg[k] : dflt; };
    match_score = get("GACT_scoring/match", 1); mismatch_score = get("GACT_scoring/mismatch", -1);
    gap_open = get("GACT_scoring/gap_open", -1); gap_extend = get("GACT_scoring/gap_extend", -1);
    first_tile_score_threshold = get("GACT_first_tile/first_tile_score_threshold", 35);
    tile_size = get("GACT_extend/tile_size", 320); tile_overlap = get("GACT_extend/tile_overlap", 120);
    DsoftParams dp;
    dp.seed_size = get("DSOFT_params/seed_size", 14); dp.bin_size = (uint32_t)get("DSOFT_params/bin_size", 64);
    dp.window_size = (uint32_t)get("DSOFT_params/window_size", 4); dp.threshold = get("DSOFT_params/threshold", 21);
    dp.num_seeds = get("DSOFT_params/num_seeds", 800);
    dp.seed_occurence_multiple = (uint32_t)get("DSOFT_params/seed_occurence_multiple", 32);
    dp.max_candidates = get("DSOFT_params/max_candidates", 1000000);
    num_threads = std::stoi(argv[3]);
    if (num_threads < 1) num_threads = 1;
    const std::string ref_path(argv[1]), reads_path(argv[2]);
    same_file = (ref_path == reads_path);                         // darwin.cpp:500-502
    printf("same_file: %d\n", same_file);
    printf("Scores: match = %d, mismatch = %d, gap_open = %d, gap_extend = %d\n", match_score, mismatch_score,
           gap_open, gap_extend);

    Tick t_stage = now();
    parse_fasta(ref_path, reference_descrips, reference_seqs, reference_lengths);
    print_stage("Time elapsed (loading reference genome)", t_stage, now());      // darwin.cpp:553
    t_stage = now();
    parse_fasta(reads_path, reads_descrips, reads_seqs, reads_lengths);
    if (!device_dsoft)
        for (const std::string &r : reads_seqs) rev_reads_seqs.push_back(rev_comp(r));
    std::cout << "Number of reads: " << reads_seqs.size() << std::endl;
    print_stage("Time elapsed (loading reads)", t_stage, now());                 // darwin.cpp:574

    if (device_dsoft) {
        std::vector<GPU_storage> s;
        GPU_init(tile_size, tile_overlap, gap_open, gap_extend, match_score, mismatch_score, tile_size - tile_overlap,
                 &s, num_threads);
        gact_hip_engine *e = (gact_hip_engine *)s[0].engine;
        upload_set(e, GACT_SET_REF, reference_seqs);
        upload_set(e, GACT_SET_QUERY, reads_seqs);
        if (gact_hip_derive_revcomp(e) != 0) { std::cerr << gact_hip_last_error() << std::endl; return 1; }   // darwin.cpp:110-147
        gact_dsoft_params gp = {dp.seed_size, (int32_t)dp.bin_size, (int32_t)dp.window_size, dp.threshold, dp.num_seeds,
                                (int32_t)dp.seed_occurence_multiple, dp.max_candidates};
        gact_dsoft_info info;
        t_stage = now();
        if (gact_hip_dsoft_build(e, &gp, &info) != 0) { printf("\ndsoft_build failed: %s\n\n", gact_hip_last_error()); return 1; }
        printf("Reference length: %lld, %zu pieces; device index: %lld minimizers, %.1f ms\n", (long long)info.ref_length,
               reference_seqs.size(), (long long)info.n_minimizers, info.build_ms);
        print_stage("Time elapsed (seed position table construction)", t_stage, now());      // darwin.cpp:596
        // this rank's contiguous share of the reads, then contiguous ranges per feeder thread (darwin.cpp:619-629)
        const int all_reads = (int)reads_seqs.size();
        const int per_rank = (int)std::ceil(1.0 * all_reads / shard_world);
        const int r_lo = std::min(all_reads, shard_rank * per_rank), r_hi = std::min(all_reads, r_lo + per_rank);
        const int num_reads = r_hi - r_lo;
        const int reads_per_thread = (int)std::ceil(1.0 * num_reads / num_threads);
        std::vector<std::vector<Cand> > dumps(num_threads);
        const bool dumping = !dump_path.empty() && dsoft_only;
        std::vector<std::thread> threads;
        t_stage = now();
        for (int i = 0; i < num_threads; i++) {
            const int lo = r_lo + std::min(num_reads, i * reads_per_thread), hi = std::min(r_hi, lo + reads_per_thread);
            threads.push_back(std::thread(device_feeder, i, lo, hi, s[i], dumping ? &dumps[i] : nullptr));
        }
        for (auto &t : threads) t.join();
        print_stage("Time elapsed (seed table querying + aligning)", t_stage, now());         // darwin.cpp:639
        if (dumping) {
            std::ofstream out(dump_path, std::ios::binary);
            size_t total = 0;
            for (auto &v : dumps) { out.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(Cand))); total += v.size(); }
            printf("num_candidates: %zu\n", total);
        }
        GPU_close(&s, num_threads);
        return 0;
    }

    // per-thread candidate lists, contiguous read ranges like darwin.cpp:619-629
    std::vector<std::vector<Cand> > per_thread(num_threads);
    Tick t_query = now();
    if (!cand_path.empty()) {
        std::vector<Cand> cands;
        std::ifstream in(cand_path, std::ios::binary);
        if (!in) { fprintf(stderr, "cannot open candidates file '%s'\n", cand_path.c_str()); return 1; }
        Cand c;
        while (in.read((char *)&c, sizeof c)) cands.push_back(c);
        const size_t per = (cands.size() + num_threads - 1) / num_threads;
        for (int i = 0; i < num_threads; i++) {
            const size_t lo = std::min(cands.size(), i * per), hi = std::min(cands.size(), lo + per);
            per_thread[i].assign(cands.begin() + lo, cands.begin() + hi);
        }
    } else {
        DsoftIndex index;
        t_stage = now();
        index.build(reference_seqs, dp);
        printf("Reference length: %u, %zu pieces\n", index.reference_length(), reference_seqs.size());
        print_stage("Time elapsed (seed position table construction)", t_stage, now());      // darwin.cpp:596
        t_query = now();
        const int num_reads = (int)reads_seqs.size();
        const int reads_per_thread = (int)std::ceil(1.0 * num_reads / num_threads);
        std::vector<std::thread> filt;
        for (int i = 0; i < num_threads; i++) {
            filt.push_back(std::thread([&, i] {
                const Tick t0 = now();
                const int lo = std::min(num_reads, i * reads_per_thread), hi = std::min(num_reads, lo + reads_per_thread);
                DsoftScratch sc;
                std::vector<DsoftCandidate> f, r;
                for (int k = lo; k < hi; k++) {
                    f.clear(); r.clear();
                    index.query(reads_seqs[k].data(), (uint32_t)reads_seqs[k].size(), k, sc, f);        // darwin.cpp:213
                    index.query(rev_reads_seqs[k].data(), (uint32_t)rev_reads_seqs[k].size(), k, sc, r);  // :252
                    for (const DsoftCandidate &c : f) per_thread[i].push_back(Cand{c.ref_id, c.query_id, c.ref_pos, c.query_pos, 0});
                    for (const DsoftCandidate &c : r) per_thread[i].push_back(Cand{c.ref_id, c.query_id, c.ref_pos, c.query_pos, 1});
                }
                print_stage("Time finding seeds", t0, now());            // darwin.cpp:300
            }));
        }
        for (auto &t : filt) t.join();
    }
    size_t total = 0;
    for (auto &v : per_thread) total += v.size();
    printf("num_candidates: %zu\n", total);
    if (!dump_path.empty()) {
        std::ofstream out(dump_path, std::ios::binary);
        for (auto &v : per_thread) out.write((const char *)v.data(), (std::streamsize)(v.size() * sizeof(Cand)));
    }
    if (dsoft_only) return 0;
    if (shard_world > 1) {
        // this rank's share: every W-th candidate of the whole list (chain lengths vary widely),
        // dealt out again in contiguous ranges to its feeder threads
        std::vector<Cand> mine;
        size_t g = 0;
        for (auto &v : per_thread)
            for (const Cand &c : v)
                if ((int)(g++ % (size_t)shard_world) == shard_rank) mine.push_back(c);
        const size_t per = (mine.size() + num_threads - 1) / num_threads;
        for (int i = 0; i < num_threads; i++) {
            const size_t lo = std::min(mine.size(), i * per), hi = std::min(mine.size(), lo + per);
            per_thread[i].assign(mine.begin() + lo, mine.begin() + hi);
        }
    }
    if (recode) {
        const Tick t0 = now();
        recode_in_place(reference_seqs); recode_in_place(reads_seqs); recode_in_place(rev_reads_seqs);
        print_stage("Time converting bases", t0, now());                          // darwin.cpp:405
    }

    if (!gather_id.empty()) return gathered_run(per_thread, gather_id);

    std::vector<GPU_storage> s;
    GPU_init(tile_size, tile_overlap, gap_open, gap_extend, match_score, mismatch_score, tile_size - tile_overlap,
             &s, num_threads);
    std::vector<std::thread> threads;
    for (int i = 0; i < num_threads; i++)
        threads.push_back(std::thread(feeder, i, &per_thread[i], (size_t)0, per_thread[i].size(), s[i]));
    for (auto &t : threads) t.join();
    print_stage("Time elapsed (seed table querying + aligning)", t_query, now());             // darwin.cpp:639
    GPU_close(&s, num_threads);
    return 0;
}
