// dsoft.cpp -- see dsoft.h.  Own restatement; the reference's candidate lists are the
// test pin (tests/test_dsoft.py compares against seed_pos_table.cpp compiled unchanged).
#include "dsoft.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>

namespace {

inline uint32_t two_bit(char c)
{
    switch (c) {                       // ntcoding.cpp:59-71: everything but acgt/ACGT is 0
        case 'c': case 'C': return 1;
        case 'g': case 'G': return 2;
        case 't': case 'T': return 3;
        default: return 0;
    }
}

// ntcoding.cpp:87-103: 16 bases per word, base j of a word at bits 2j, 1+len/16 words
std::vector<uint32_t> to_two_bit(const char *seq, uint32_t len)
{
    std::vector<uint32_t> w(1 + len / 16 + 1, 0);   // one spare word: seed_at reads idx+1
    for (uint32_t i = 0; i < len; i++) w[i / 16] += two_bit(seq[i]) << (2 * (i % 16));
    return w;
}

// Thomas Wang's integer hash masked to 2k bits, ntcoding.cpp:77-88
inline uint32_t hash32(uint32_t key, int k)
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

inline uint32_t seed_at(const std::vector<uint32_t> &s, uint32_t pos, int k)   // ntcoding.cpp:115-124
{
    const uint32_t m = (1u << (2 * k)) - 1;
    const uint32_t idx = pos / 16, shift = pos % 16;
    const uint64_t concat = ((uint64_t)s[idx + 1] << 32) + s[idx];
    return (uint32_t)(concat >> (2 * shift)) & m;
}

// ntcoding.cpp:126-182.  s_len = number of 2-bit words the caller declares (the
// loop bound 16*s_len - k - w is taken in uint32 like the reference); emits
// (hash, position) pairs in position order.
template <class F> void minimizers(const std::vector<uint32_t> &s, uint32_t s_len, int k, int w, F emit)
{
    std::vector<uint32_t> window((size_t)w, 0);
    uint64_t last_m = 0;
    uint32_t last_p = 0;
    if (16 * s_len < (uint32_t)(k + w)) return;   // the reference's unsigned bound wraps here and it reads out of bounds
    for (int p = 0; p < w - 1; p++) window[p] = hash32(seed_at(s, p, k), k);
    const uint32_t end = 16 * s_len - (uint32_t)k - (uint32_t)w;
    for (uint32_t p = (uint32_t)(w - 1); p < end; p++) {
        window[p % w] = hash32(seed_at(s, p, k), k);
        uint32_t mn = 0xffffffffu;
        for (int i = 0; i < w; i++) mn = std::min(mn, window[i]);
        const uint64_t m = mn;
        if (m != last_m || p - last_p >= (uint32_t)w) {
            emit(mn, p);
            last_m = m;
            last_p = p;
        }
    }
}

}  // namespace

void DsoftIndex::build(const std::vector<std::string> &reference_seqs, const DsoftParams &p)
{
    p_ = p;
    assert(p.seed_size <= 15 && p.seed_size > 3 && (uint32_t)p.seed_size > p.window_size);   // seed_pos_table.cpp:48-50
    // darwin.cpp:532-543: every sequence padded with 'N' to a whole number of bins
    std::string concat;
    start_bin_.clear(); bin_to_chr_.clear(); ref_lengths_.clear();
    uint32_t curr_bin = 0;
    for (size_t i = 0; i < reference_seqs.size(); i++) {
        const std::string &r = reference_seqs[i];
        start_bin_.push_back(curr_bin);
        ref_lengths_.push_back((long long)r.size());
        concat += r;
        for (size_t j = 0; j < r.size() / p.bin_size; j++) { bin_to_chr_.push_back((int)i); curr_bin++; }
        if (r.size() % p.bin_size > 0) {
            concat += std::string(p.bin_size - r.size() % p.bin_size, 'N');
            bin_to_chr_.push_back((int)i); curr_bin++;
        }
    }
    ref_len_ = (uint32_t)concat.size();
    const uint32_t log_bin = (uint32_t)std::log2((double)p.bin_size);
    num_bins_ = 1 + (ref_len_ >> log_bin);                                  // darwin.cpp:184
    max_occ_ = p.seed_occurence_multiple * (1 + (ref_len_ >> (2 * p.seed_size)));   // seed_pos_table.cpp:59

    const std::vector<uint32_t> r2 = to_two_bit(concat.data(), ref_len_);
    const uint32_t rlen_2bit = 1 + ref_len_ / 16;                           // :61
    mins_.clear();
    minimizers(r2, rlen_2bit, p.seed_size, (int)p.window_size,
               [&](uint32_t h, uint32_t pos) { mins_.push_back(((uint64_t)h << 32) + pos); });
    std::sort(mins_.begin(), mins_.end());                                  // :71
}

int DsoftIndex::query(const char *q, uint32_t len, int query_id, DsoftScratch &sc,
                      std::vector<DsoftCandidate> &out) const
{
    if (sc.bin_count_offset.size() != num_bins_) sc.bin_count_offset.assign(num_bins_, 0);
    sc.nz_bins.clear();
    sc.hits.clear();
    const std::vector<uint32_t> q2 = to_two_bit(q, len);
    const uint32_t qlen_2bit = (len + 15) / 16;                             // seed_pos_table.cpp:108
    const uint32_t k = (uint32_t)p_.seed_size;
    int num_seeds = 0;
    bool stop = false;
    minimizers(q2, qlen_2bit, p_.seed_size, (int)p_.window_size, [&](uint32_t index, uint32_t offset) {
        if (stop) return;
        // the reference keeps end offsets per seed value (index_table_, :73-93); the range of seed
        // `index` in the sorted minimizer array is the same thing
        const auto lo = std::lower_bound(mins_.begin(), mins_.end(), (uint64_t)index << 32);
        const auto hi = std::lower_bound(lo, mins_.end(), ((uint64_t)index + 1) << 32);
        const uint32_t occ = (uint32_t)(hi - lo);
        if (occ > max_occ_) return;                                         // :124
        if (num_seeds > p_.num_seeds) { stop = true; return; }              // :125-127 (N+1 seeds are used)
        num_seeds++;
        for (auto it = lo; it != hi; ++it) {
            const uint32_t hit = (uint32_t)(*it & 0xffffffffu);
            if (hit < offset) continue;                                     // :132
            const uint32_t bin = (hit - offset) / p_.bin_size;
            const uint32_t curr_count = (uint32_t)(sc.bin_count_offset[bin] >> 32);
            const uint32_t last_offset = (uint32_t)(sc.bin_count_offset[bin] & 0xffffffffu);
            if (curr_count < (uint32_t)p_.threshold) {
                const uint32_t new_count = ((offset - last_offset > k) || curr_count == 0)
                                               ? curr_count + k : curr_count + (offset - last_offset);   // :137
                sc.bin_count_offset[bin] = ((uint64_t)new_count << 32) + offset;
                if (new_count >= (uint32_t)p_.threshold) {
                    if ((int)sc.hits.size() >= p_.max_candidates) break;    // :141-143 (leaves the hit loop only)
                    sc.hits.push_back(((uint64_t)hit << 32) + offset);
                }
                if (curr_count == 0) sc.nz_bins.push_back(bin);             // :146-149
            }
        }
    });
    for (uint32_t b : sc.nz_bins) sc.bin_count_offset[b] = 0;               // :155-158

    // darwin.cpp:215-224: concatenated coordinate -> (sequence, position in it), clamped
    for (uint64_t h : sc.hits) {
        int ref_pos = (int)(h >> 32);
        const uint32_t bin = (uint32_t)ref_pos / p_.bin_size;
        const int chr = bin < bin_to_chr_.size() ? bin_to_chr_[bin] : 0;
        ref_pos -= (int)(start_bin_[chr] * p_.bin_size);
        if (ref_pos > ref_lengths_[chr]) ref_pos = (int)ref_lengths_[chr];
        DsoftCandidate c;
        c.ref_id = chr; c.query_id = query_id; c.ref_pos = ref_pos; c.query_pos = (int)(h & 0xffffffffu);
        out.push_back(c);
    }
    return (int)sc.hits.size();
}
