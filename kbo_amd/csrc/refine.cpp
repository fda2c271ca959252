// refine.cpp — host-side refinement after the GPU hot path: variant calling, gap filling,
// add_variants.  Statement-level restatements of reference src/variant_calling.rs,
// src/gap_filling.rs and src/translate.rs:350-386; usize arithmetic that would panic in the
// reference (underflow, out-of-bounds index, failed assert!) throws RefPanic here.
#include "refine.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace kbo {
namespace {

inline size_t usub(size_t a, size_t b, const char *what)
{
    if (b > a) throw RefPanic(std::string("attempt to subtract with overflow: ") + what);
    return a - b;
}
template <typename V> inline auto &at(V &v, size_t i, const char *what)
{
    if (i >= v.size()) throw RefPanic(std::string("index out of bounds: ") + what);
    return v[i];
}
inline int code_of(uint8_t ch)
{
    switch (ch) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}

} // namespace

// ------------------------------------------------------------------ HostNav

HostNav::HostNav(const HostIndex &h) : h_(h)
{
    const uint64_t nw = (h.n_sets + 63) / 64;
    for (int c = 0; c < 4; c++) {
        samples_[c].assign(nw / 8 + 2, 0);
        uint64_t acc = 0;
        for (uint64_t w = 0; w < nw; w++) {
            if ((w & 7) == 0) samples_[c][w >> 3] = acc;
            acc += (uint64_t)__builtin_popcountll(h.rows[c][w]);
        }
        for (uint64_t b = (nw + 7) / 8; b < samples_[c].size(); b++) samples_[c][b] = acc;
        if ((nw & 7) == 0) samples_[c][nw >> 3] = acc;
    }
}

uint64_t HostNav::rank(int c, uint64_t i) const
{
    const uint64_t blk = i >> 9, w1 = i >> 6;
    uint64_t r = samples_[c][blk];
    const auto &b = h_.rows[c];
    for (uint64_t w = blk << 3; w < w1; w++) r += (uint64_t)__builtin_popcountll(b[w]);
    if (i & 63) r += (uint64_t)__builtin_popcountll(b[w1] & ((1ull << (i & 63)) - 1));
    return r;
}

uint64_t HostNav::select(int c, uint64_t q) const
{
    // last block whose sample is <= q
    const auto &s = samples_[c];
    uint64_t lo = 0, hi = (h_.n_sets + 511) / 512;
    while (lo + 1 < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (s[mid] <= q) lo = mid; else hi = mid;
    }
    uint64_t rem = q - s[lo];
    const auto &b = h_.rows[c];
    const uint64_t nw = (h_.n_sets + 63) / 64;
    for (uint64_t w = lo << 3; w < nw; w++) {
        uint64_t word = b[w];
        uint64_t pc = (uint64_t)__builtin_popcountll(word);
        if (rem < pc) {
            for (;; rem--) {
                if (rem == 0) return w * 64 + (uint64_t)__builtin_ctzll(word);
                word &= word - 1;
            }
        }
        rem -= pc;
    }
    throw RefPanic("select past the last set bit");
}

bool HostNav::search(const uint8_t *pattern, size_t len, uint64_t &lo, uint64_t &hi) const
{
    lo = 0;
    hi = h_.n_sets;
    for (size_t i = 0; i < len; i++) {
        int c = code_of(pattern[i]);
        if (c < 0) return false;
        lo = h_.C[c] + rank(c, lo);
        hi = h_.C[c] + rank(c, hi);
        if (lo >= hi) return false;
    }
    return true;
}

void HostNav::access_kmer(uint64_t colex, std::vector<uint8_t> &out) const
{
    if (colex >= h_.n_sets) throw RefPanic("access_kmer: row out of range");
    const uint32_t k = h_.k;
    out.assign(k, '$');
    uint64_t i = colex;
    for (uint32_t t = k; t-- > 0;) {
        if (i == 0) break; // root: everything to the left is '$'
        int c = 3;
        while (c > 0 && h_.C[c] > i) c--;
        out[t] = (uint8_t)"ACGT"[c];
        i = select(c, i - h_.C[c]); // first row of the (k-1)-suffix group the edge leaves from
    }
}

// ------------------------------------------------------------------ variant calling

namespace {

// variant_calling.rs:46-58
std::vector<uint8_t> get_kmer_ending_at(const uint8_t *query, size_t end_pos, size_t k)
{
    std::vector<uint8_t> km;
    if (end_pos >= k - 1) km.assign(query + end_pos + 1 - k, query + end_pos + 1);
    else {
        km.assign(k - 1 - end_pos, '$');
        km.insert(km.end(), query, query + end_pos + 1);
    }
    return km;
}

// variant_calling.rs:60-71
size_t longest_common_suffix(const uint8_t *x, const uint8_t *y, size_t k)
{
    size_t len = 0;
    for (size_t i = 0; i < k; i++) {
        if (x[k - 1 - i] == y[k - 1 - i]) len++;
        else break;
    }
    return len;
}

// variant_calling.rs:73-83
bool rightmost_significant_peak(const uint32_t *d, size_t n, size_t thr, size_t &peak)
{
    if (n == 0) throw RefPanic("assert!(!ms.is_empty())");
    for (size_t i = n - 1; i-- > 0;) {
        if (d[i] >= thr && d[i] > d[i + 1]) { peak = i; return true; }
    }
    return false;
}

} // namespace

// variant_calling.rs:139-201 on plain arrays; the variant's characters as index ranges into the two k-mers
bool resolve_variant_ranges(const uint8_t *query_kmer, const uint8_t *ref_kmer, const uint32_t *d_vs_query, const uint32_t *d_vs_ref,
                            size_t k, size_t thr, size_t &q_from, size_t &q_to, size_t &r_from, size_t &r_to)
{
    const size_t csl = longest_common_suffix(query_kmer, ref_kmer, k);
    if (csl == 0) throw RefPanic("assert!(common_suffix_len > 0)");
    size_t qpeak = 0, rpeak = 0;
    const bool hq = rightmost_significant_peak(d_vs_ref, k, thr, qpeak);
    const bool hr = rightmost_significant_peak(d_vs_query, k, thr, rpeak);
    return resolve_variant_peaks(k, csl, hq, qpeak, hr, rpeak, q_from, q_to, r_from, r_to);
}

// the same from the three values resolve_variant reads off its inputs (the device's second pass delivers them per site:
// call_second_kernels.hip): the common suffix length and the two rightmost significant peaks
bool resolve_variant_peaks(size_t k, size_t csl, bool hq, size_t qpeak, bool hr, size_t rpeak, size_t &q_from, size_t &q_to, size_t &r_from,
                           size_t &r_to)
{
    if (csl == 0) throw RefPanic("assert!(common_suffix_len > 0)");
    if (!(hq && hr)) return false;
    const size_t sms = k - csl;
    const long query_gap = (long)sms - (long)qpeak - 1, ref_gap = (long)sms - (long)rpeak - 1;
    q_from = q_to = r_from = r_to = 0;
    if (query_gap > 0 && ref_gap > 0) {
        q_from = qpeak + 1; q_to = sms;
        r_from = rpeak + 1; r_to = sms;
        return true;
    }
    const long qo = -query_gap, ro = -ref_gap;
    if (qo == ro) return false;
    const size_t vlen = (size_t)std::labs(qo - ro);
    if (qo > ro) { // deletion in query
        if (rpeak + 1 + vlen > k) throw RefPanic("ref_kmer slice out of range");
        r_from = rpeak + 1; r_to = rpeak + 1 + vlen;
    } else { // insertion in query
        if (qpeak + 1 + vlen > k) throw RefPanic("query_kmer slice out of range");
        q_from = qpeak + 1; q_to = qpeak + 1 + vlen;
    }
    return true;
}

namespace {

// the same on the vectors the single-sequence path carries; returns false for Err(ResolveVariantErr)
bool resolve_variant(const std::vector<uint8_t> &query_kmer, const std::vector<uint8_t> &ref_kmer,
                     const std::vector<MsVal> &ms_vs_query, const std::vector<MsVal> &ms_vs_ref, size_t thr,
                     std::vector<uint8_t> &qchars, std::vector<uint8_t> &rchars)
{
    const size_t k = query_kmer.size();
    if (ref_kmer.size() != k || ms_vs_query.size() != k || ms_vs_ref.size() != k) throw RefPanic("resolve_variant: length asserts");
    std::vector<uint32_t> dq(k), dr(k);
    for (size_t i = 0; i < k; i++) { dq[i] = ms_vs_query[i].d; dr[i] = ms_vs_ref[i].d; }
    size_t qf, qt, rf, rt;
    if (!resolve_variant_ranges(query_kmer.data(), ref_kmer.data(), dq.data(), dr.data(), k, thr, qf, qt, rf, rt)) return false;
    qchars.assign(query_kmer.begin() + qf, query_kmer.begin() + qt);
    rchars.assign(ref_kmer.begin() + rf, ref_kmer.begin() + rt);
    return true;
}

} // namespace

void call_site_kmers(const HostNav &nav_ref, const uint8_t *query, uint32_t k, const std::vector<CallSite> &sites,
                     std::vector<std::vector<uint8_t>> &query_kmers, std::vector<std::vector<uint8_t>> &ref_kmers)
{
    for (const CallSite &st : sites) {
        query_kmers.push_back(get_kmer_ending_at(query, st.j, k)); // variant_calling.rs:275
        std::vector<uint8_t> rk;
        nav_ref.access_kmer(st.lo, rk);                            // :276
        ref_kmers.push_back(std::move(rk));
    }
}

std::vector<Variant> resolve_call_sites(const std::vector<CallSite> &sites, const std::vector<uint8_t> *query_kmers,
                                        const std::vector<uint8_t> *ref_kmers, const std::vector<MsVal> *ms_vs_ref,
                                        const std::vector<MsVal> *ms_vs_query, size_t d)
{
    std::vector<Variant> calls;
    for (size_t s = 0; s < sites.size(); s++) { // variant_calling.rs:282-284
        Variant v;
        if (resolve_variant(query_kmers[s], ref_kmers[s], ms_vs_query[s], ms_vs_ref[s], d, v.query_chars, v.ref_chars)) {
            v.query_pos = sites[s].i;
            calls.push_back(std::move(v));
        }
    }
    return calls;
}

std::vector<Variant> call_variants(const HostNav &nav_ref, const MsFn &ms_ref, const MsFn &ms_query, uint32_t k,
                                   const uint8_t *query, size_t len, size_t d)
{
    // first pass: whole-sequence MS on the GPU (variant_calling.rs:266)
    std::vector<std::vector<uint8_t>> one(1, std::vector<uint8_t>(query, query + len));
    std::vector<std::vector<MsVal>> msv;
    ms_ref(one, msv);
    const std::vector<MsVal> &ms = msv[0];
    // breakpoint scan (variant_calling.rs:268-273) - the batched entry point runs this scan on the device
    // (call_kernels.hip) and arrives at the same site list
    std::vector<CallSite> sites;
    for (size_t i = 1; i < len; i++) {
        if (ms[i].d < ms[i - 1].d && ms[i - 1].d >= d && ms[i].d < d) {
            for (size_t j = i + 1; j < std::min(i + k + 1, len); j++) {
                if (ms[j].d >= d && ms[j].hi - ms[j].lo == 1) {
                    sites.push_back({i, j, ms[j].lo});
                    break;
                }
            }
        }
    }
    if (sites.empty()) return {};
    std::vector<std::vector<uint8_t>> query_kmers, ref_kmers;
    call_site_kmers(nav_ref, query, k, sites, query_kmers, ref_kmers);
    // second pass: the two k-length walks of every site, batched on the GPU (:279-280)
    std::vector<std::vector<MsVal>> ms_vs_ref, ms_vs_query;
    ms_ref(query_kmers, ms_vs_ref);
    ms_query(ref_kmers, ms_vs_query);
    return resolve_call_sites(sites, query_kmers.data(), ref_kmers.data(), ms_vs_ref.data(), ms_vs_query.data(), d);
}

void add_variants(std::vector<uint8_t> &refined, const std::vector<Variant> &variants)
{ // translate.rs:357-383
    for (const Variant &var : variants) {
        const size_t ql = var.query_chars.size(), rl = var.ref_chars.size();
        if (ql == rl) {
            for (size_t i = 0; i < rl; i++) at(refined, var.query_pos + i, "add_variants") = var.ref_chars[i];
        } else if (ql == 0) {
            at(refined, usub(var.query_pos, 1, "query_pos - 1"), "add_variants") = 'I';
            at(refined, var.query_pos, "add_variants") = 'I';
        } else if (rl == 0) {
            for (size_t i = 0; i < ql; i++) at(refined, var.query_pos + i, "add_variants") = 'D';
        } else {
            bool all_equal = true;
            for (uint8_t c : var.ref_chars) all_equal = all_equal && c == var.ref_chars[0];
            const uint8_t fill = all_equal ? var.ref_chars[0] : (uint8_t)'N';
            for (size_t i = 0; i < ql; i++) at(refined, var.query_pos + i, "add_variants") = fill;
        }
    }
}

// ------------------------------------------------------------------ gap filling

namespace {

// gap_filling.rs:20-42
size_t count_right_overlaps(const std::vector<uint8_t> &kmer, const uint8_t *ref_seq, size_t ref_len, size_t ref_match_end)
{
    if (kmer.empty() || ref_len == 0 || ref_len < ref_match_end) throw RefPanic("count_right_overlaps asserts");
    size_t kmer_pos = kmer.size() - 1;
    size_t ref_pos = usub(ref_match_end, 1, "ref_match_end - 1");
    size_t matches = 0;
    while (kmer_pos > 0) {
        if (ref_pos >= ref_len) throw RefPanic("ref_seq[ref_pos]");
        if (ref_seq[ref_pos] == kmer[kmer_pos]) matches++;
        else break;
        kmer_pos -= 1;
        ref_pos = usub(ref_pos, 1, "ref_pos -= 1");
    }
    return matches;
}

// gap_filling.rs:44-67
size_t count_left_overlaps(const std::vector<uint8_t> &kmer, const uint8_t *ref_seq, size_t ref_len, size_t ref_match_start)
{
    if (kmer.empty() || ref_len == 0 || !(ref_len > ref_match_start)) throw RefPanic("count_left_overlaps asserts");
    size_t kmer_pos = 0, ref_pos = ref_match_start, matches = 0;
    while (kmer_pos < kmer.size()) {
        if (ref_pos >= ref_len) throw RefPanic("ref_seq[ref_pos]");
        if (ref_seq[ref_pos] == kmer[kmer_pos]) matches++;
        else break;
        kmer_pos++;
        ref_pos++;
    }
    return matches;
}

} // namespace

// gap_filling.rs:127-151
std::pair<size_t, std::vector<uint8_t>> nearest_unique_context(const std::vector<MsVal> &ms, const HostNav &nav,
                                                               size_t range_start, size_t range_end)
{
    if (nav.index().k == 0 || ms.empty() || !(range_end >= range_start) || !(range_end < ms.size()))
        throw RefPanic("nearest_unique_context asserts");
    std::vector<uint8_t> kmer;
    size_t kmer_idx = range_end;
    while (kmer_idx >= range_start) {
        const MsVal &iv = at(ms, kmer_idx, "ms[kmer_idx]");
        if (iv.hi - iv.lo == 1) {
            nav.access_kmer(iv.lo, kmer); // push_kmer_to_vec
            break;
        }
        kmer_idx = usub(kmer_idx, 1, "kmer_idx -= 1");
    }
    return {kmer_idx, kmer};
}

// gap_filling.rs:205-232
std::vector<uint8_t> left_extend_kmer(const std::vector<uint8_t> &kmer_start, const HostNav &nav, size_t max_extension_len)
{
    if (kmer_start.empty()) throw RefPanic("assert!(!kmer_start.is_empty())");
    size_t left_extension_len = 0;
    std::vector<uint8_t> kmer = kmer_start;
    while (left_extension_len < max_extension_len) {
        const size_t keep = usub(kmer.size(), left_extension_len + 1, "kmer.len() - (left_extension_len + 1)");
        int n_found = 0;
        uint8_t found_c = 0;
        uint64_t flo = 0, fhi = 0;
        for (uint8_t c : {(uint8_t)'A', (uint8_t)'C', (uint8_t)'G', (uint8_t)'T'}) {
            std::vector<uint8_t> nk;
            nk.reserve(keep + 1);
            nk.push_back(c);
            nk.insert(nk.end(), kmer.begin(), kmer.begin() + keep);
            uint64_t lo, hi;
            if (nav.search(nk.data(), nk.size(), lo, hi)) {
                if (n_found == 0) { found_c = c; flo = lo; fhi = hi; }
                n_found++;
            }
        }
        if (n_found == 1 && fhi - flo == 1) kmer.insert(kmer.begin(), found_c);
        else break;
        left_extension_len++;
    }
    return kmer;
}

// gap_filling.rs:295-361
std::vector<uint8_t> left_extend_over_gap(const std::vector<MsVal> &ms, const uint8_t *ref_seq, size_t ref_len,
                                          const HostNav &nav, size_t left_overlap_req, size_t right_overlap_req,
                                          size_t gap_start, size_t gap_end, size_t search_radius)
{
    const size_t k = nav.index().k;
    if (k == 0 || ms.size() != ref_len || !(left_overlap_req <= gap_start) || ref_len < gap_end ||
        !(right_overlap_req <= ref_len - gap_end) || !(gap_end > gap_start) || !(gap_end < ms.size()))
        throw RefPanic("left_extend_over_gap asserts");
    const size_t search_start = std::min(gap_end + search_radius, usub(ref_len, 1, "ref_seq.len() - 1"));
    const size_t search_end = gap_end + right_overlap_req;
    std::vector<uint8_t> kmer;
    size_t kmer_idx = search_start;
    while (kmer_idx >= search_end) {
        auto ctx = nearest_unique_context(ms, nav, search_end, kmer_idx);
        kmer_idx = ctx.first;
        kmer = std::move(ctx.second);
        if (!kmer.empty()) {
            const size_t right_matches_want =
                usub(usub(search_start, gap_end - 1, "search_start - (gap_end - 1)"), usub(search_start, kmer_idx, "search_start - kmer_idx"),
                     "right_matches_want");
            const size_t right_matches_got = count_right_overlaps(kmer, ref_seq, ref_len, gap_end + right_matches_want);
            const size_t ref_start_pos = gap_start > left_overlap_req ? gap_start - left_overlap_req : 0;
            const size_t left_matches_got = count_left_overlaps(kmer, ref_seq, ref_len, ref_start_pos);
            const bool should_extend = kmer.size() < left_overlap_req + (gap_end - gap_start) + right_matches_got;
            if (right_matches_got >= std::min(right_matches_want, k) && left_matches_got >= left_overlap_req) {
                const size_t s = left_matches_got - left_overlap_req;
                const size_t e = usub(kmer.size(), usub(right_matches_got, right_overlap_req, "right_matches_got - right_overlap_req"), "kmer end");
                if (s > e || e > kmer.size()) throw RefPanic("kmer[start..end]");
                kmer = std::vector<uint8_t>(kmer.begin() + s, kmer.begin() + e);
                break;
            } else if (should_extend && right_matches_got >= std::min(right_matches_want, k) && left_matches_got < left_overlap_req) {
                const size_t left_extend_length =
                    usub(left_overlap_req + (gap_end - gap_start) + right_matches_got, k, "left_extend_length");
                kmer = left_extend_kmer(kmer, nav, left_extend_length);
                const size_t lm = count_left_overlaps(kmer, ref_seq, ref_len, ref_start_pos);
                if (lm >= left_overlap_req) {
                    const size_t s = lm - left_overlap_req;
                    const size_t e = usub(kmer.size(), usub(right_matches_got, right_overlap_req, "right_matches_got - right_overlap_req"), "kmer end");
                    if (s > e || e > kmer.size()) throw RefPanic("kmer[start..end]");
                    kmer = std::vector<uint8_t>(kmer.begin() + s, kmer.begin() + e);
                    break;
                }
            }
            kmer.clear();
        }
        kmer_idx = usub(kmer_idx, 1, "kmer_idx -= 1");
    }
    return kmer;
}

double log_rm_max_cdf_host(size_t t, size_t alphabet_size, size_t n_kmers)
{ // derandomize.rs:99, same operation order as kbo_capi.cpp
    double a = std::exp(std::log(1.0) - std::log((double)alphabet_size));
    int b = (int)t + 1;
    double r = 1;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return (double)n_kmers * std::log1p(-r);
}

// gap_filling.rs:444-526
std::vector<uint8_t> fill_gaps(const std::vector<uint8_t> &translation, const std::vector<MsVal> &noisy_ms,
                               const uint8_t *ref_seq, size_t len, const HostNav &nav, size_t threshold, double max_err_prob)
{
    const size_t n_elements = translation.size();
    if (translation.empty() || translation.size() != noisy_ms.size() || len != n_elements) throw RefPanic("fill_gaps asserts");
    const size_t k = nav.index().k;
    if (k == 0) throw RefPanic("assert!(sbwt.k() > 0)");
    std::vector<uint8_t> refined = translation;
    size_t i = threshold + 1;
    while (i < usub(refined.size(), threshold, "refined.len() - threshold")) {
        if (refined[i - 1] == '-' || refined[i - 1] == 'X') {
            const size_t start_index = i - 1;
            while (i < n_elements && refined[i] == '-') i++;
            const size_t end_index = std::min(i, refined.size() - threshold);
            const bool overlap_without_extend = end_index - start_index + 2 * threshold <= k;
            const size_t search_radius = usub(k, threshold * (overlap_without_extend ? 1 : 0), "search_radius");
            const std::vector<uint8_t> kmer = left_extend_over_gap(noisy_ms, ref_seq, len, nav, threshold, threshold,
                                                                   start_index, end_index, search_radius);
            const bool kmer_found = !kmer.empty() && std::find(kmer.begin(), kmer.end(), (uint8_t)'$') == kmer.end();
            const size_t gap = end_index - start_index;
            const bool no_indels = kmer.size() == threshold + gap + threshold;
            const size_t a = std::min(threshold, kmer.size()), b = std::min(threshold + gap, kmer.size());
            std::vector<bool> matching;
            for (size_t j = a, r = start_index; j < b && r < end_index; j++, r++) matching.push_back(kmer[j] == ref_seq[r]);
            size_t total_overlaps = 0;
            for (bool m : matching) total_overlaps += m;
            double log_probs = 0.0;
            size_t consecutive = 0;
            for (size_t w = 0; w + 1 < matching.size(); w++) {
                if (matching[w] && matching[w + 1]) consecutive++;
                else {
                    if (consecutive > 0) log_probs += log_rm_max_cdf_host(consecutive + 1, 4, 1);
                    consecutive = 0;
                }
            }
            const bool fill_overlaps = log_probs > std::log1p(-max_err_prob);
            const bool fill_flanked = !matching.empty() && !matching.front() && !matching.back() && total_overlaps + 2 == gap;
            const bool pass = kmer_found && no_indels && (overlap_without_extend || fill_overlaps || fill_flanked);
            if (pass)
                for (size_t j = 0; j < gap; j++)
                    refined[start_index + j] = kmer[threshold + j] == ref_seq[start_index + j] ? (uint8_t)'M' : kmer[threshold + j];
        }
        i++;
    }
    return refined;
}

} // namespace kbo
