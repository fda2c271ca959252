// refine.hpp — host-side refinement stages that follow the GPU hot path in kbo::map / kbo::call
// (reference src/variant_calling.rs, src/gap_filling.rs, src/translate.rs::add_variants).
// They are sparse, branchy post-processing (one unit of work per variant site / per gap);
// every matching-statistics pass they need is delegated to the GPU through MsFn.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "sbwt_index.hpp"

namespace kbo {

struct MsVal {
    uint32_t d, lo, hi; // (d, lo..hi) as returned by index::query_sbwt
};

// variant_calling::Variant (variant_calling.rs:8-26)
struct Variant {
    size_t query_pos;
    std::vector<uint8_t> query_chars, ref_chars;
};

// Navigation over the host copy of the index: rank/select on the subset-matrix rows.
// Replaces the sbwt calls SbwtIndex::{search, access_kmer, push_kmer_to_vec}
// (reference call sites variant_calling.rs:276, gap_filling.rs:144,217).
class HostNav {
public:
    explicit HostNav(const HostIndex &h);
    uint64_t rank(int c, uint64_t i) const;            // set bits of B_c in [0, i)
    uint64_t select(int c, uint64_t q) const;          // position of the q-th (0-based) set bit of B_c
    bool search(const uint8_t *pattern, size_t len, uint64_t &lo, uint64_t &hi) const; // sbwt search()
    void access_kmer(uint64_t colex, std::vector<uint8_t> &out) const;                // k chars, '$'-padded
    const HostIndex &index() const { return h_; }

private:
    const HostIndex &h_;
    std::vector<uint64_t> samples_[4]; // rank before every 512-bit block
};

// Batched matching statistics on the GPU: one MsVal vector per input sequence.
using MsFn = std::function<void(const std::vector<std::vector<uint8_t>> &seqs, std::vector<std::vector<MsVal>> &out)>;

// variant_calling::call_variants (variant_calling.rs:249-294).  Parameter names follow the
// callee: `ms_ref` walks the index the variants are called against (kbo's query index),
// `ms_query` walks the index built from `query` (kbo's reference sequence).
std::vector<Variant> call_variants(const HostNav &nav_ref, const MsFn &ms_ref, const MsFn &ms_query, uint32_t k,
                                   const uint8_t *query, size_t len, size_t threshold_d);

// the two halves of call_variants behind the breakpoint scan, shared with the batched entry point (call_batch.cpp):
// a site = breakpoint i, the unique match j to its right and that match's row (variant_calling.rs:268-276)
struct CallSite {
    size_t i, j;
    uint32_t lo;
};
void call_site_kmers(const HostNav &nav_ref, const uint8_t *query, uint32_t k, const std::vector<CallSite> &sites,
                     std::vector<std::vector<uint8_t>> &query_kmers, std::vector<std::vector<uint8_t>> &ref_kmers);
// arrays of sites.size() entries each
std::vector<Variant> resolve_call_sites(const std::vector<CallSite> &sites, const std::vector<uint8_t> *query_kmers,
                                        const std::vector<uint8_t> *ref_kmers, const std::vector<MsVal> *ms_vs_ref,
                                        const std::vector<MsVal> *ms_vs_query, size_t threshold_d);

// resolve_variant (variant_calling.rs:139-201) on plain arrays of k entries each (the depths of the two walks); the
// variant's characters come back as index ranges into the two k-mers.  false = Err(ResolveVariantErr).
bool resolve_variant_ranges(const uint8_t *query_kmer, const uint8_t *ref_kmer, const uint32_t *d_vs_query, const uint32_t *d_vs_ref,
                            size_t k, size_t thr, size_t &q_from, size_t &q_to, size_t &r_from, size_t &r_to);
bool resolve_variant_peaks(size_t k, size_t csl, bool hq, size_t qpeak, bool hr, size_t rpeak, size_t &q_from, size_t &q_to, size_t &r_from,
                           size_t &r_to);

// translate::add_variants (translate.rs:350-386)
void add_variants(std::vector<uint8_t> &translation, const std::vector<Variant> &variants);

// gap_filling::fill_gaps (gap_filling.rs:444-526)
std::vector<uint8_t> fill_gaps(const std::vector<uint8_t> &translation, const std::vector<MsVal> &noisy_ms,
                               const uint8_t *ref_seq, size_t len, const HostNav &nav, size_t threshold,
                               double max_err_prob);

// pieces exposed for the reference's own unit tests (gap_filling.rs:535-638)
std::pair<size_t, std::vector<uint8_t>> nearest_unique_context(const std::vector<MsVal> &ms, const HostNav &nav,
                                                               size_t range_start, size_t range_end);
std::vector<uint8_t> left_extend_kmer(const std::vector<uint8_t> &kmer_start, const HostNav &nav, size_t max_extension_len);
std::vector<uint8_t> left_extend_over_gap(const std::vector<MsVal> &ms, const uint8_t *ref_seq, size_t ref_len,
                                          const HostNav &nav, size_t left_overlap_req, size_t right_overlap_req,
                                          size_t gap_start, size_t gap_end, size_t search_radius);

double log_rm_max_cdf_host(size_t t, size_t alphabet_size, size_t n_kmers); // derandomize.rs:91-100

// thrown where the reference would panic (index out of bounds, usize underflow, assert!)
struct RefPanic : std::runtime_error {
    explicit RefPanic(const std::string &m) : std::runtime_error(m) {}
};

} // namespace kbo
