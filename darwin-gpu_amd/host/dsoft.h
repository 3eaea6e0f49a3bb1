// dsoft.h -- D-SOFT seed filter: minimizer index over the
// concatenated reference and diagonal-band seed counting per query, restated
// from the reference so that the driver runs from FASTA alone.  Outside the
// GACT hot path; plain host C++.
//
// Reference semantics restated (quirks kept, they decide which candidates exist):
//   2-bit coding, non-ACGT -> A                 ntcoding.cpp:59-71,87-103
//   hash32 + (k,w) window minimizers            ntcoding.cpp:77-88,126-182
//   index build, occurrence cap                 seed_pos_table.cpp:46-98
//   DSOFT band counting, candidate emission     seed_pos_table.cpp:100-167
//   candidate -> (chr, ref_pos, query_pos)      darwin.cpp:213-224,532-543
#ifndef DARWIN_HIP_DSOFT_H
#define DARWIN_HIP_DSOFT_H

#include <cstdint>
#include <string>
#include <vector>

struct DsoftParams {
    int seed_size = 14;                 // params.cfg DSOFT_params.seed_size
    uint32_t bin_size = 64;
    uint32_t window_size = 4;
    int threshold = 21;
    int num_seeds = 800;
    uint32_t seed_occurence_multiple = 32;
    int max_candidates = 1000000;
};

struct DsoftCandidate {
    int ref_id, query_id, ref_pos, query_pos;
};

// per-thread scratch (darwin.cpp:193-199 allocates the same per AlignReads thread)
struct DsoftScratch {
    std::vector<uint64_t> bin_count_offset;   // [num_bins], (count << 32) | last query offset
    std::vector<uint32_t> nz_bins;
    std::vector<uint64_t> hits;               // (ref hit << 32) | query offset, seed_pos_table.cpp:143
};

class DsoftIndex {
public:
    // reference_seqs as darwin.cpp:526 loads them; builds the padded concatenation
    // (darwin.cpp:532-543) and the minimizer index over it (seed_pos_table.cpp:46-98)
    void build(const std::vector<std::string> &reference_seqs, const DsoftParams &p);

    // one query strand (darwin.cpp:213-224 / 252-263): appends the decoded candidates
    int query(const char *q, uint32_t len, int query_id, DsoftScratch &sc, std::vector<DsoftCandidate> &out) const;

    uint32_t reference_length() const { return ref_len_; }
    uint32_t num_bins() const { return num_bins_; }

private:
    DsoftParams p_;
    uint32_t ref_len_ = 0, num_bins_ = 0, max_occ_ = 0;
    std::vector<uint64_t> mins_;              // sorted (seed << 32) | position
    std::vector<uint32_t> start_bin_;         // per reference sequence
    std::vector<int> bin_to_chr_;             // per bin
    std::vector<long long> ref_lengths_;
};

#endif
