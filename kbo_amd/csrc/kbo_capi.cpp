// kbo_capi.cpp — the extern "C" boundary declared in include/kbo_hip.h.
//
// Host logic only: argument checks mirroring the reference's asserts, index ownership,
// device-memory plumbing and kernel launches.  All matching-statistics / derandomize /
// translate / run-length compute happens in the *_kernels.hip files; nothing here falls back to the CPU.
#include "../../include/kbo_hip.h"
#include "../../include/kbo_hip_tuning.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <condition_variable>
#include <functional>
#include <memory>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"
#include "../../include/kbo_hip_tuning.h"

using namespace kbo_host;

namespace kbo_host {

// ---- A3: derandomize.rs:91-145, f64, identical operation order -------------------------
double powi_f64(double a, int b) // Rust f64::powi == llvm.powi == compiler-rt __powidf2
{
    const bool recip = b < 0;
    double r = 1;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1 / r : r;
}

double log_rm_max_cdf(size_t t, size_t alphabet_size, size_t n_kmers)
{
    KBO_REQUIRE(n_kmers > 0, KBO_E_BAD_ARG, "n_kmers > 0 (derandomize.rs:96)");
    KBO_REQUIRE(alphabet_size > 0, KBO_E_BAD_ARG, "alphabet_size > 0 (derandomize.rs:97)");
    return (double)n_kmers *
           std::log1p(-powi_f64(std::exp(std::log(1.0) - std::log((double)alphabet_size)), (int)t + 1));
}

size_t random_match_threshold(size_t k, size_t n_kmers, size_t alphabet_size, double p)
{
    KBO_REQUIRE(k > 0, KBO_E_BAD_ARG, "k > 0 (derandomize.rs:133)");
    KBO_REQUIRE(n_kmers > 0, KBO_E_BAD_ARG, "n_kmers > 0 (derandomize.rs:134)");
    KBO_REQUIRE(alphabet_size > 0, KBO_E_BAD_ARG, "alphabet_size > 0 (derandomize.rs:135)");
    KBO_REQUIRE(p <= 1.0 && p > 0.0, KBO_E_BAD_ARG, "0 < max_error_prob <= 1 (derandomize.rs:136-137)");
    for (size_t i = 1; i < k; i++)
        if (log_rm_max_cdf(i, alphabet_size, n_kmers) > std::log1p(-p)) return i;
    return k;
}

} // namespace kbo_host

namespace {

// matching statistics with intervals of a list of sequences, batched on the GPU
kbo::MsFn make_ms_fn(kbo_index *idx)
{
    return [idx](const std::vector<std::vector<uint8_t>> &seqs, std::vector<std::vector<kbo::MsVal>> &out) {
        out.assign(seqs.size(), {});
        if (seqs.empty()) return;
        std::vector<uint64_t> off(seqs.size() + 1, 0);
        for (size_t s = 0; s < seqs.size(); s++) off[s + 1] = off[s] + seqs[s].size();
        std::vector<uint8_t> concat(off.back());
        for (size_t s = 0; s < seqs.size(); s++) std::memcpy(concat.data() + off[s], seqs[s].data(), seqs[s].size());
        hipStream_t stream = nullptr;
        BatchOnDevice B;
        run_walk_host(idx, concat.data(), off.data(), seqs.size(), true, B, stream);
        std::vector<uint8_t> d(B.total);
        std::vector<uint32_t> lo(B.total), hi(B.total);
        HIP_OK(hipMemcpy(d.data(), B.ms.p, B.total, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(lo.data(), B.lo.p, B.total * 4, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hi.data(), B.hi.p, B.total * 4, hipMemcpyDeviceToHost));
        for (size_t s = 0; s < seqs.size(); s++) {
            out[s].resize(seqs[s].size());
            for (size_t i = 0; i < seqs[s].size(); i++) out[s][i] = kbo::MsVal{d[off[s] + i], lo[off[s] + i], hi[off[s] + i]};
        }
    };
}

// lib.rs:735-738 for one sequence with an explicit threshold: MS (with intervals) + A5 + A6 on the GPU
void ms_and_translation(kbo_index *idx, const uint8_t *seq, size_t len, size_t threshold, std::vector<kbo::MsVal> &ms,
                        std::vector<uint8_t> &chars)
{
    KBO_REQUIRE(len > 0, KBO_E_EMPTY_QUERY, "assert!(!query.is_empty()) (index.rs:248)");
    const uint64_t off[2] = {0, len};
    check_len_threshold(off, 1, idx->host.k, threshold);
    hipStream_t stream = nullptr;
    BatchOnDevice B;
    run_walk_host(idx, seq, off, 1, true, B, stream);
    DevBuf dch(((len + 15) / 16) * 16 + 16);
    derand_translate_host_offsets(B.ms.as<uint8_t>(), B.off.as<uint64_t>(), off, 1, idx->host.k, (uint32_t)threshold,
                                  nullptr, dch.as<uint8_t>(), nullptr, stream);
    std::vector<uint8_t> d(len);
    std::vector<uint32_t> lo(len), hi(len);
    chars.resize(len);
    HIP_OK(hipMemcpy(d.data(), B.ms.p, len, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(lo.data(), B.lo.p, len * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hi.data(), B.hi.p, len * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(chars.data(), dch.p, len, hipMemcpyDeviceToHost));
    ms.resize(len);
    for (size_t i = 0; i < len; i++) ms[i] = kbo::MsVal{d[i], lo[i], hi[i]};
}

// kbo::call (lib.rs:547-573)
std::vector<kbo::Variant> call_impl(kbo_index *query_idx, const uint8_t *ref_seq, size_t len, const kbo_call_opts &o)
{
    kbo_index ref_idx; // lib.rs:553: an index of ref_seq is built on every call
    kbo::BuildParams p;
    p.k = o.sbwt_build_opts.k;
    p.add_revcomp = o.sbwt_build_opts.add_revcomp != 0;
    p.num_threads = std::max(1u, o.sbwt_build_opts.num_threads);
    const uint8_t *seqs[1] = {ref_seq};
    const size_t lens[1] = {len};
    kbo::build_host_index(seqs, lens, 1, p, ref_idx.host);
    KBO_REQUIRE(ref_idx.host.k == query_idx->host.k, KBO_E_K_MISMATCH, "assert!(sbwt_ref.k() == sbwt_query.k()) (lib.rs:559)");
    // variant_calling.rs:260 — callee's sbwt_ref is kbo's query index (lib.rs:561-568)
    const size_t d = random_match_threshold(query_idx->host.k, query_idx->host.n_kmers, 4, o.max_error_prob);
    kbo::HostNav nav(query_idx->host);
    return kbo::call_variants(nav, make_ms_fn(query_idx), make_ms_fn(&ref_idx), query_idx->host.k, ref_seq, len, d);
}

kbo_variant *pack_variants(const std::vector<kbo::Variant> &v)
{
    size_t chars = 0;
    for (const auto &x : v) chars += x.query_chars.size() + x.ref_chars.size();
    const size_t head = std::max<size_t>(1, v.size()) * sizeof(kbo_variant);
    uint8_t *mem = static_cast<uint8_t *>(std::malloc(head + chars + 1));
    if (!mem) throw std::bad_alloc();
    kbo_variant *out = reinterpret_cast<kbo_variant *>(mem);
    uint8_t *cp = mem + head;
    for (size_t i = 0; i < v.size(); i++) {
        out[i].query_pos = v[i].query_pos;
        out[i].query_chars = cp;
        out[i].query_len = v[i].query_chars.size();
        std::memcpy(cp, v[i].query_chars.data(), v[i].query_chars.size());
        cp += v[i].query_chars.size();
        out[i].ref_chars = cp;
        out[i].ref_len = v[i].ref_chars.size();
        std::memcpy(cp, v[i].ref_chars.data(), v[i].ref_chars.size());
        cp += v[i].ref_chars.size();
    }
    return out;
}

// format.rs:143-193, statement for statement (sequential, variable-length output: host).  Returns false where the reference
// panics: an 'R' at position 0 that starts a run makes format.rs:175 evaluate aln[i - 1] with i = 0 (usize underflow).
bool run_lengths_gapped_impl(const uint8_t *aln, size_t len, size_t max_gap_len, std::vector<kbo_rle> &out)
{
    size_t i = 0;
    bool match_start = false;
    while (i < len) {
        match_start = (aln[i] != '-' && aln[i] != ' ') && !match_start;
        if (match_start) {
            kbo_rle rle{(uint64_t)i, 0, 0, 0, 0, 0, 0};
            size_t within_gap_bases = 0;
            bool within_gap_start = false;
            while (i < len && aln[i] != ' ') {
                const bool is_true_gap = aln[i] == '-';
                if (is_true_gap && !within_gap_start) {
                    within_gap_start = true;
                    rle.gap_opens += 1;
                    within_gap_bases = 0;
                }
                if (!is_true_gap && within_gap_start) within_gap_start = false;
                const bool is_match = aln[i] == 'M' || aln[i] == 'R' || aln[i] == 'I';
                const bool is_gap = is_true_gap || aln[i] == 'D';
                rle.matches += is_match;
                rle.gap_bases += is_gap;
                rle.mismatches += (!is_match && !is_gap);
                rle.end = (is_match || !is_gap) ? i + 1 : rle.end;
                if (aln[i] == 'R' && i == 0) return false; // (format.rs:175: aln[i - 1] with i = 0)
                rle.jumps += (aln[i] == 'R' && aln[i - 1] == 'R');
                within_gap_bases += (aln[i] == '-');
                i += 1;
                if (within_gap_bases > max_gap_len || (is_gap && i == len && rle.gap_opens > 0)) {
                    rle.gap_opens -= 1;
                    rle.gap_bases -= within_gap_bases;
                    break;
                }
            }
            out.push_back(rle);
            match_start = false;
        } else {
            i += 1;
        }
    }
    return true;
}

kbo_rle *copy_rles(const std::vector<kbo_rle> &v)
{
    kbo_rle *p = static_cast<kbo_rle *>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(kbo_rle)));
    if (!p) throw std::bad_alloc();
    if (!v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(kbo_rle));
    return p;
}

} // namespace


extern "C" {

const char *kbo_last_error(void) { return last_error().c_str(); }
const char *kbo_version(void) { return "kbo-hip 0.1.0 (gfx950)"; }

void kbo_build_opts_default(kbo_build_opts *o)
{
    if (!o) return;
    o->k = 31; o->add_revcomp = 0; o->num_threads = 1; o->prefix_precalc = 8;
    o->build_select = 0; o->mem_gb = 4; o->dedup_batches = 0; o->temp_dir = nullptr;
}
void kbo_find_opts_default(kbo_find_opts *o)
{
    if (!o) return;
    o->max_error_prob = 0.0000001; o->max_gap_len = 0;
}
void kbo_call_opts_default(kbo_call_opts *o)
{
    if (!o) return;
    o->max_error_prob = 0.0000001;
    kbo_build_opts_default(&o->sbwt_build_opts);
    o->sbwt_build_opts.build_select = 1;
}
void kbo_map_opts_default(kbo_map_opts *o)
{
    if (!o) return;
    o->max_error_prob = 0.0000001; o->fill_gaps = 1; o->call_variants = 1; o->format = 1;
    kbo_build_opts_default(&o->sbwt_build_opts);
    o->sbwt_build_opts.build_select = 1;
}

namespace {
// A sharded index (capi_internal.hpp kbo_index::shards): `want` shards or more over disjoint parts of the input - groups of
// whole sequences of about equal size, and with add_revcomp the forward and the reverse-complement strand of every group
// apart.  The set of strings (of at most k characters) that are suffixes of rows of an SBWT is the set of substrings of its
// input's ACGT-runs of at least k characters - a property of the runs, hence of the sequences one by one - so the depth of a
// walk against the index of everything is the maximum of the depths against the shards.  The threshold of the derandomisation
// needs the number of distinct k-mers of the union: every shard hands back its sorted k-mers and a merge counts them once.
void build_sharded(const uint8_t *const *seqs, const size_t *lens, size_t n_seqs, const kbo::BuildParams &p, size_t want, kbo_index &out)
{
    const size_t strands = p.add_revcomp ? 2 : 1;
    const size_t groups = std::max<size_t>(1, std::min(n_seqs, (want + strands - 1) / strands));
    uint64_t bases = 0;
    for (size_t s = 0; s < n_seqs; s++) bases += lens[s];
    // contiguous groups of sequences with about bases / groups bases each
    std::vector<size_t> first(groups + 1, n_seqs);
    first[0] = 0;
    {
        uint64_t acc = 0;
        size_t g = 1;
        for (size_t s = 0; s < n_seqs && g < groups; s++) {
            acc += lens[s];
            if (acc >= bases * g / groups && s + 1 < n_seqs) first[g++] = s + 1;
        }
    }
    std::vector<std::vector<uint64_t>> keys;
    uint32_t key_words = 0;
    uint64_t n_sets = 0;
    for (size_t g = 0; g < groups; g++) {
        if (first[g] >= first[g + 1]) continue;
        for (size_t st = 0; st < strands; st++) {
            kbo::BuildParams q = p;
            q.add_revcomp = false;
            q.revcomp_only = st == 1;
            keys.emplace_back();
            q.keys_out = &keys.back();
            q.key_words_out = &key_words;
            std::unique_ptr<kbo_index> sh(new kbo_index());
            try {
                kbo::build_host_index(seqs + first[g], lens + first[g], first[g + 1] - first[g], q, sh->host);
            } catch (const std::runtime_error &e) { // (a shard that is itself too large: sequences are never cut)
                throw KboError(KBO_E_UNSUPPORTED,
                               std::string("sharded build: shard ") + std::to_string(out.shards.size()) + " (sequences " + std::to_string(first[g]) + " .. " +
                               std::to_string(first[g + 1] - 1) + ", " + (st ? "reverse-complement" : "forward") + " strand) failed: " + e.what() +
                               " - shards hold whole sequences, so one sequence with 2^32 or more distinct k-mers cannot be indexed");
            }
            n_sets += sh->host.n_sets;
            out.shards.push_back(std::move(sh));
        }
    }
    // distinct k-mers of the union: a merge over the shards' sorted key lists (equal k-mers have equal keys everywhere)
    const size_t W = key_words ? key_words : 1;
    std::vector<size_t> at(keys.size(), 0);
    uint64_t distinct = 0;
    auto less = [&](const uint64_t *a, const uint64_t *b) {
        for (size_t j = 0; j < W; j++)
            if (a[j] != b[j]) return a[j] < b[j];
        return false;
    };
    uint64_t cur[8]; // (k <= 255: at most 8 words a key)
    KBO_REQUIRE(W <= 8, KBO_E_UNSUPPORTED, "k-mer keys wider than 8 words");
    for (;;) {
        const uint64_t *best = nullptr;
        for (size_t x = 0; x < keys.size(); x++) {
            if (at[x] * W >= keys[x].size()) continue;
            const uint64_t *c = keys[x].data() + at[x] * W;
            if (!best || less(c, best)) best = c;
        }
        if (!best) break;
        distinct++;
        std::copy(best, best + W, cur); // (advance every list past this k-mer)
        for (size_t x = 0; x < keys.size(); x++)
            while (at[x] * W < keys[x].size() && std::equal(cur, cur + W, keys[x].data() + at[x] * W)) at[x]++;
    }
    out.host.k = p.k;
    out.host.n_kmers = distinct;
    out.host.n_sets = n_sets; // (rows over all shards; the index of the union would have fewer dummy rows)
}
} // namespace

int kbo_index_build(const uint8_t *const *seqs, const size_t *lens, size_t n_seqs,
                    const kbo_build_opts *opts, kbo_index_t **out)
{
    return guarded([&] {
        KBO_REQUIRE(out, KBO_E_BAD_ARG, "null out");
        *out = nullptr;
        KBO_REQUIRE(seqs && lens && n_seqs > 0, KBO_E_BAD_ARG, "assert!(!slices.is_empty()) (index.rs:60)");
        kbo_build_opts o;
        if (opts) o = *opts; else kbo_build_opts_default(&o);
        kbo::BuildParams p;
        p.k = o.k; p.add_revcomp = o.add_revcomp != 0; p.num_threads = std::max(1u, o.num_threads);
        // rows the index would have, at most: one per base and strand (+ dummies).  Beyond 32-bit row numbers (a human genome
        // with its reverse complements) - or when a test asks for it - the index is built as shards (capi_internal.hpp)
        uint64_t bases = 0;
        for (size_t s = 0; s < n_seqs; s++) bases += lens[s];
        const uint64_t est_rows = bases * (p.add_revcomp ? 2u : 1u);
        const uint64_t kShardRows = 0xE0000000ull; // (3.76 * 10^9: leaves room for the dummy rows)
        const int forced = g_index_shards.load();
        const size_t want = forced > 0 ? (size_t)forced : (size_t)((est_rows + kShardRows - 1) / kShardRows);
        std::unique_ptr<kbo_index> idx(new kbo_index());
        if (want <= 1) kbo::build_host_index(seqs, lens, n_seqs, p, idx->host);
        else build_sharded(seqs, lens, n_seqs, p, want, *idx);
        *out = idx.release();
    });
}

int kbo_index_from_parts(uint32_t k, uint64_t n_sets, uint64_t n_kmers, const uint64_t *const rows[4],
                         const uint64_t C[4], const uint8_t *lcs, kbo_index_t **out)
{
    return guarded([&] {
        KBO_REQUIRE(out, KBO_E_BAD_ARG, "null out");
        *out = nullptr;
        KBO_REQUIRE(rows && C && lcs && k > 0 && k <= 255 && n_sets > 0, KBO_E_BAD_ARG, "bad index parts");
        kbo_index *idx = new kbo_index();
        idx->host.k = k; idx->host.n_sets = n_sets; idx->host.n_kmers = n_kmers;
        const size_t nw = (n_sets + 63) / 64;
        for (int c = 0; c < 4; c++) {
            idx->host.C[c] = C[c];
            idx->host.rows[c].assign(rows[c], rows[c] + nw);
            if (n_sets & 63) idx->host.rows[c][nw - 1] &= (1ull << (n_sets & 63)) - 1;
        }
        idx->host.lcs.assign(lcs, lcs + n_sets);
        idx->host.lcs[0] = 0;
        try {
            kbo::validate_host_index(idx->host); // C[] vs edge bits, LCS < k: the kernels trust them
        } catch (...) {
            delete idx;
            throw;
        }
        *out = idx;
    });
}

int kbo_index_export_parts(const kbo_index_t *idx, uint64_t *const rows[4], uint64_t C[4], uint8_t *lcs)
{
    return guarded([&] {
        KBO_REQUIRE(idx && rows && C && lcs, KBO_E_BAD_ARG, "null argument");
        require_unsharded(idx, "kbo_index_export_parts");
        const size_t nw = (idx->host.n_sets + 63) / 64;
        for (int c = 0; c < 4; c++) {
            C[c] = idx->host.C[c];
            std::memcpy(rows[c], idx->host.rows[c].data(), nw * sizeof(uint64_t));
        }
        std::memcpy(lcs, idx->host.lcs.data(), idx->host.n_sets);
    });
}

void kbo_index_free(kbo_index_t *idx) { delete idx; }
size_t kbo_index_k(const kbo_index_t *idx) { return idx ? idx->host.k : 0; }
uint64_t kbo_index_n_kmers(const kbo_index_t *idx) { return idx ? idx->host.n_kmers : 0; }
uint64_t kbo_index_n_sets(const kbo_index_t *idx) { return idx ? idx->host.n_sets : 0; }

int kbo_index_save(const kbo_index_t *idx_c, const char *path)
{
    kbo_index_t *idx = const_cast<kbo_index_t *>(idx_c); // (the handle's cache of its path cover is filled in under its mutex)
    int rc = guarded([&] {
        KBO_REQUIRE(idx && path, KBO_E_BAD_ARG, "null argument");
        std::lock_guard<std::mutex> g(idx->mu);
        require_unsharded(idx, "kbo_index_save");
        if (plan_enabled(idx) && !idx->cover && !idx->transient && idx->host.n_sets < 0xFFFFFFF0ull) {
            idx->cover.reset(new kbo::PathCover());
            kbo::make_path_cover(idx->host, *idx->cover);
        }
        kbo::save_host_index(idx->host, path, plan_enabled(idx) ? idx->cover.get() : nullptr);
    });
    return rc == KBO_E_BAD_ARG && idx && path ? KBO_E_IO : rc;
}

int kbo_index_load(const char *path, kbo_index_t **out)
{
    int rc = guarded([&] {
        KBO_REQUIRE(path && out, KBO_E_BAD_ARG, "null argument");
        *out = nullptr;
        kbo_index *idx = new kbo_index();
        try {
            std::unique_ptr<kbo::PathCover> pc(new kbo::PathCover());
            bool have = false;
            kbo::load_host_index(path, idx->host, pc.get(), &have);
            if (have) idx->cover = std::move(pc);
        } catch (...) {
            delete idx;
            throw;
        }
        *out = idx;
    });
    return rc == KBO_E_BAD_ARG && path && out ? KBO_E_IO : rc;
}

int kbo_index_save_sbwt(const kbo_index_t *idx, const char *prefix)
{
    int rc = guarded([&] {
        KBO_REQUIRE(idx && prefix, KBO_E_BAD_ARG, "null argument");
        require_unsharded(idx, "kbo_index_save_sbwt");
        kbo::save_sbwt_pair(idx->host, prefix);
    });
    return rc == KBO_E_BAD_ARG && idx && prefix ? KBO_E_IO : rc;
}

int kbo_index_load_sbwt(const char *prefix, kbo_index_t **out)
{
    int rc = guarded([&] {
        KBO_REQUIRE(prefix && out, KBO_E_BAD_ARG, "null argument");
        *out = nullptr;
        kbo_index *idx = new kbo_index();
        bool own = false;
        try {
            own = kbo::load_sbwt_pair(prefix, idx->host);
        } catch (...) {
            delete idx;
            throw;
        }
        if (!own) {
            delete idx;
            throw KboError(KBO_E_UNSUPPORTED,
                           std::string(prefix) + ".sbwt was written by the sbwt crate: its field layout behind the SubsetMatrix tag is not "
                           "pinned by the reference (index.rs:143); hand the index over with kbo_index_from_parts");
        }
        *out = idx;
    });
    return rc == KBO_E_BAD_ARG && prefix && out ? KBO_E_IO : rc;
}

int kbo_index_to_device(kbo_index_t *idx, int device)
{
    return guarded([&] {
        KBO_REQUIRE(idx, KBO_E_BAD_ARG, "null index");
        for (kbo_index *sh : shards_of(idx)) (void)device_view(sh, device < 0 ? current_device() : device, nullptr, 0, true);
    });
}

int kbo_index_device_bytes(const kbo_index_t *idx, uint64_t *rank_bytes, uint64_t *lcs_bytes)
{
    return guarded([&] {
        KBO_REQUIRE(idx, KBO_E_BAD_ARG, "null index");
        uint64_t rb = 0, lb = 0; // (a sharded index: over all shards)
        for (kbo_index *sh : shards_of(const_cast<kbo_index *>(idx))) {
            rb += (sh->host.n_sets / kbo::kRankRowsPerBlock + 2) * 16 * 4;
            lb += (3 * (sh->host.n_sets + 1) + 4) * sizeof(uint32_t);
        }
        if (rank_bytes) *rank_bytes = rb;
        if (lcs_bytes) *lcs_bytes = lb;
    });
}

uint64_t kbo_index_device_pair_bytes(const kbo_index_t *idx)
{
    if (!idx) return 0;
    uint64_t total = 0; // what the device copies actually carry (none for the 64-bit entry layout)
    for (kbo_index *sh : shards_of(const_cast<kbo_index *>(idx))) {
        uint64_t bytes = 0;
        std::lock_guard<std::mutex> g(sh->mu);
        for (const auto &kv : sh->dev)
            if (kv.second->pair_off) bytes = std::max<uint64_t>(bytes, (kv.second->n_blocks) * 16ull * 16ull);
        total += bytes;
    }
    return total;
}

int kbo_log_rm_max_cdf(size_t t, size_t alphabet_size, size_t n_kmers, double *out)
{
    return guarded([&] {
        KBO_REQUIRE(out, KBO_E_BAD_ARG, "null out");
        *out = log_rm_max_cdf(t, alphabet_size, n_kmers);
    });
}

int kbo_random_match_threshold(size_t k, size_t n_kmers, size_t alphabet_size, double max_error_prob,
                               size_t *out)
{
    return guarded([&] {
        KBO_REQUIRE(out, KBO_E_BAD_ARG, "null out");
        *out = random_match_threshold(k, n_kmers, alphabet_size, max_error_prob);
    });
}

int kbo_ms_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                 uint8_t *d_out, uint32_t *lo_out, uint32_t *hi_out)
{
    return guarded([&] { ms_batch_impl(idx, concat, offsets, n_seqs, d_out, lo_out, hi_out); });
}

int kbo_matching_statistics(kbo_index_t *idx, const uint8_t *query, size_t len, uint64_t *d, uint64_t *lo,
                            uint64_t *hi)
{
    return guarded([&] {
        KBO_REQUIRE(idx && query && d, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(len > 0, KBO_E_EMPTY_QUERY, "assert!(!query.is_empty()) (index.rs:248)");
        KBO_REQUIRE((lo == nullptr) == (hi == nullptr), KBO_E_BAD_ARG, "lo/hi must come together");
        const uint64_t off[2] = {0, len};
        std::vector<uint8_t> d8(len);
        std::vector<uint32_t> lo32, hi32;
        if (lo) { lo32.resize(len); hi32.resize(len); }
        int rc = kbo_ms_batch(idx, query, off, 1, d8.data(), lo ? lo32.data() : nullptr,
                              lo ? hi32.data() : nullptr);
        if (rc) throw KboError(rc, last_error());
        for (size_t i = 0; i < len; i++) d[i] = d8[i]; // widen to the reference's usize
        if (lo)
            for (size_t i = 0; i < len; i++) { lo[i] = lo32[i]; hi[i] = hi32[i]; }
    });
}

int kbo_derandomize_ms_val(size_t curr, int64_t next, size_t threshold, size_t k, int64_t *out)
{
    return guarded([&] {
        KBO_REQUIRE(out, KBO_E_BAD_ARG, "null out");
        KBO_REQUIRE(k > 0, KBO_E_BAD_ARG, "k > 0 (derandomize.rs:227)");
        KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (derandomize.rs:228)");
        KBO_REQUIRE(curr <= k, KBO_E_MS_RANGE, "curr_noisy_ms <= k (derandomize.rs:229)");
        KBO_REQUIRE(next <= (int64_t)k, KBO_E_MS_RANGE, "next_derand_ms <= k (derandomize.rs:230)");
        int64_t run = next - 1;
        if (curr == k) run = (int64_t)k;
        if (curr > threshold && next < (int64_t)curr) run = (int64_t)curr;
        *out = run;
    });
}

int kbo_derandomize_ms_vec(const uint64_t *noisy, size_t len, size_t k, size_t threshold, int64_t *out)
{
    return guarded([&] {
        KBO_REQUIRE(noisy && out, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(k > 0, KBO_E_BAD_ARG, "k > 0 (derandomize.rs:274)");
        KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (derandomize.rs:275)");
        KBO_REQUIRE(len > 2, KBO_E_LEN_LE_2, "len > 2 (derandomize.rs:276)");
        KBO_REQUIRE(k <= 255, KBO_E_UNSUPPORTED, "k > 255");
        std::vector<uint8_t> n8(((len + 15) / 16) * 16 + 16, 0);
        for (size_t i = 0; i < len; i++) {
            KBO_REQUIRE(noisy[i] <= k, KBO_E_MS_RANGE, "curr_noisy_ms <= k (derandomize.rs:229)");
            n8[i] = (uint8_t)noisy[i]; // narrow usize -> u8 (values <= k <= 255)
        }
        hipStream_t stream = nullptr;
        const uint64_t off[2] = {0, len};
        DevBuf dms(n8.size()), doff(sizeof(off)), dch(n8.size()), dder(len * sizeof(int32_t));
        HIP_OK(hipMemcpyAsync(dms.p, n8.data(), n8.size(), hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(doff.p, off, sizeof(off), hipMemcpyHostToDevice, stream));
        derand_translate_host_offsets(dms.as<uint8_t>(), doff.as<uint64_t>(), off, 1, (uint32_t)k,
                                      (uint32_t)std::min<size_t>(threshold, 0x7FFFFFFF), nullptr,
                                      dch.as<uint8_t>(), dder.as<int32_t>(), stream);
        std::vector<int32_t> d32(len);
        HIP_OK(hipMemcpyAsync(d32.data(), dder.p, len * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (size_t i = 0; i < len; i++) out[i] = d32[i];
    });
}

int kbo_translate_ms_val(int64_t curr, int64_t next, int64_t prev, size_t threshold, uint32_t *aln_curr,
                         uint32_t *aln_next)
{
    return guarded([&] {
        KBO_REQUIRE(aln_curr && aln_next, KBO_E_BAD_ARG, "null out");
        KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (translate.rs:186)");
        *aln_next = ' ';
        const int64_t t = (int64_t)threshold;
        if (curr > t && next > 0 && next < t) { *aln_curr = 'R'; *aln_next = 'R'; }
        else if (curr <= 0) *aln_curr = (next == 1 && prev > 0) ? 'X' : '-';
        else *aln_curr = 'M';
    });
}

int kbo_translate_ms_vec(const int64_t *derand, size_t len, size_t k, size_t threshold, uint32_t *out)
{
    return guarded([&] {
        KBO_REQUIRE(derand && out, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(k > 0, KBO_E_BAD_ARG, "k > 0 (translate.rs:268)");
        KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (translate.rs:269)");
        KBO_REQUIRE(len > 2, KBO_E_LEN_LE_2, "len > 2 (translate.rs:270)");
        KBO_REQUIRE(len < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "vector longer than 2^32-1");
        // the stencil only compares values with 0, 1, threshold and k: clamping i64 -> i32
        // preserves every comparison as long as threshold and k fit in i32
        const int64_t lim = 0x7FFFFFF0;
        std::vector<int32_t> x(len);
        for (size_t i = 0; i < len; i++) x[i] = (int32_t)std::max<int64_t>(-lim, std::min<int64_t>(lim, derand[i]));
        const uint32_t t32 = (uint32_t)std::min<size_t>(threshold, (size_t)lim - 1);
        const uint32_t k32 = (uint32_t)std::min<size_t>(k, (size_t)lim - 1);
        hipStream_t stream = nullptr;
        DevBuf dx(len * sizeof(int32_t)), dch(len);
        HIP_OK(hipMemcpyAsync(dx.p, x.data(), len * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIP_OK(kbo::launch_translate(dx.as<int32_t>(), len, k32, t32, dch.as<uint8_t>(), stream));
        std::vector<uint8_t> ch(len);
        HIP_OK(hipMemcpyAsync(ch.data(), dch.p, len, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        for (size_t i = 0; i < len; i++) out[i] = ch[i]; // widen u8 -> Rust char
    });
}

int kbo_matches_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                      double max_error_prob, uint8_t *chars_out)
{
    return guarded([&] { matches_batch_impl(idx, concat, offsets, n_seqs, max_error_prob, false, chars_out); });
}

int kbo_map_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                  double max_error_prob, int format, uint8_t *out)
{
    return guarded([&] { matches_batch_impl(idx, concat, offsets, n_seqs, max_error_prob, format != 0, out); });
}

int kbo_matches(kbo_index_t *idx, const uint8_t *query, size_t len, double max_error_prob, uint32_t *chars_out)
{
    return guarded([&] {
        KBO_REQUIRE(idx && query && chars_out, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(len > 0, KBO_E_EMPTY_QUERY, "assert!(!query.is_empty()) (index.rs:248)");
        const uint64_t off[2] = {0, len};
        std::vector<uint8_t> ch(len);
        matches_batch_impl(idx, query, off, 1, max_error_prob, false, ch.data());
        for (size_t i = 0; i < len; i++) chars_out[i] = ch[i];
    });
}

int kbo_map(kbo_index_t *idx, const uint8_t *ref_seq, size_t len, const kbo_map_opts *opts, uint8_t *out)
{
    return guarded([&] {
        KBO_REQUIRE(idx && ref_seq && out, KBO_E_BAD_ARG, "null argument");
        kbo_map_opts o;
        if (opts) o = *opts; else kbo_map_opts_default(&o);
        if (o.call_variants)
            KBO_REQUIRE(idx->host.k == o.sbwt_build_opts.k, KBO_E_K_MISMATCH,
                        "assert!(sbwt.k() == map_opts.sbwt_build_opts.k) (lib.rs:729)");
        KBO_REQUIRE(len > 0, KBO_E_EMPTY_QUERY, "assert!(!query.is_empty()) (index.rs:248)");
        if (!o.fill_gaps && !o.call_variants) { // everything on the GPU, incl. relative_to_ref
            const uint64_t off[2] = {0, len};
            matches_batch_impl(idx, ref_seq, off, 1, o.max_error_prob, o.format != 0, out);
            return;
        }
        require_unsharded(idx, "kbo_map with fill_gaps / call_variants");
        const size_t threshold = random_match_threshold(idx->host.k, idx->host.n_kmers, 4, o.max_error_prob); // lib.rs:731
        std::vector<kbo::MsVal> ms;
        std::vector<uint8_t> refined;
        ms_and_translation(idx, ref_seq, len, threshold, ms, refined);                                       // lib.rs:735-738
        if (o.fill_gaps) {                                                                                   // lib.rs:743-747
            kbo::HostNav nav(idx->host);
            refined = kbo::fill_gaps(refined, ms, ref_seq, len, nav, threshold, o.max_error_prob);
        }
        if (o.call_variants) {                                                                               // lib.rs:749-754
            kbo_call_opts co;
            co.max_error_prob = o.max_error_prob;
            co.sbwt_build_opts = o.sbwt_build_opts;
            kbo::add_variants(refined, call_impl(idx, ref_seq, len, co));
        }
        if (o.format) {                                                                                      // lib.rs:756-760
            int rc = kbo_relative_to_ref(ref_seq, refined.data(), len, out);
            if (rc) throw KboError(rc, last_error());
        } else {
            std::memcpy(out, refined.data(), len);
        }
    });
}

int kbo_call(kbo_index_t *query_idx, const uint8_t *ref_seq, size_t len, const kbo_call_opts *opts, kbo_variant **out,
             size_t *n_out)
{
    return guarded([&] {
        KBO_REQUIRE(query_idx && ref_seq && out && n_out, KBO_E_BAD_ARG, "null argument");
        *out = nullptr;
        *n_out = 0;
        require_unsharded(query_idx, "kbo_call");
        KBO_REQUIRE(len > 0, KBO_E_EMPTY_QUERY, "assert!(!query.is_empty()) (index.rs:248)");
        kbo_call_opts o;
        if (opts) o = *opts; else kbo_call_opts_default(&o);
        std::vector<kbo::Variant> v = call_impl(query_idx, ref_seq, len, o);
        *out = pack_variants(v);
        *n_out = v.size();
    });
}

int kbo_add_variants(uint32_t *translation, size_t len, const kbo_variant *variants, size_t n_variants)
{
    return guarded([&] {
        KBO_REQUIRE(translation && (variants || n_variants == 0), KBO_E_BAD_ARG, "null argument");
        std::vector<uint8_t> t(len);
        for (size_t i = 0; i < len; i++) t[i] = (uint8_t)translation[i];
        std::vector<kbo::Variant> v(n_variants);
        for (size_t i = 0; i < n_variants; i++) {
            v[i].query_pos = variants[i].query_pos;
            v[i].query_chars.assign(variants[i].query_chars, variants[i].query_chars + variants[i].query_len);
            v[i].ref_chars.assign(variants[i].ref_chars, variants[i].ref_chars + variants[i].ref_len);
        }
        kbo::add_variants(t, v);
        for (size_t i = 0; i < len; i++) translation[i] = t[i];
    });
}

int kbo_fill_gaps(kbo_index_t *idx, const uint8_t *ref_seq, size_t len, size_t threshold, double max_err_prob,
                  uint32_t *out)
{
    return guarded([&] {
        KBO_REQUIRE(idx && ref_seq && out, KBO_E_BAD_ARG, "null argument");
        std::vector<kbo::MsVal> ms;
        require_unsharded(idx, "kbo_fill_gaps");
        std::vector<uint8_t> tr;
        ms_and_translation(idx, ref_seq, len, threshold, ms, tr);
        kbo::HostNav nav(idx->host);
        std::vector<uint8_t> refined = kbo::fill_gaps(tr, ms, ref_seq, len, nav, threshold, max_err_prob);
        for (size_t i = 0; i < len; i++) out[i] = refined[i];
    });
}

int kbo_nearest_unique_context(kbo_index_t *idx, const uint8_t *ref_seq, size_t len, size_t range_start,
                               size_t range_end, size_t *kmer_idx, uint8_t *kmer_out, size_t *kmer_len)
{
    return guarded([&] {
        KBO_REQUIRE(idx && ref_seq && kmer_idx && kmer_out && kmer_len, KBO_E_BAD_ARG, "null argument");
        require_unsharded(idx, "kbo_nearest_unique_context");
        std::vector<std::vector<uint8_t>> one(1, std::vector<uint8_t>(ref_seq, ref_seq + len));
        std::vector<std::vector<kbo::MsVal>> ms;
        make_ms_fn(idx)(one, ms);
        kbo::HostNav nav(idx->host);
        auto r = kbo::nearest_unique_context(ms[0], nav, range_start, range_end);
        *kmer_idx = r.first;
        *kmer_len = r.second.size();
        std::memcpy(kmer_out, r.second.data(), r.second.size());
    });
}

int kbo_run_lengths_gapped(const uint8_t *aln, size_t len, size_t max_gap_len, kbo_rle **out, size_t *n_out)
{
    return guarded([&] {
        KBO_REQUIRE((aln || len == 0) && out && n_out, KBO_E_BAD_ARG, "null argument");
        std::vector<kbo_rle> v;
        KBO_REQUIRE(run_lengths_gapped_impl(aln, len, max_gap_len, v), KBO_E_REF_PANIC,
                    "an 'R' at position 0: the reference indexes aln[i - 1] there (format.rs:175) and panics");
        *out = copy_rles(v);
        *n_out = v.size();
    });
}

int kbo_run_lengths_gapped_batch(const uint8_t *aln_concat, const uint64_t *offsets, size_t n_seqs, size_t max_gap_len,
                                 kbo_rle **rles, uint64_t *rle_offsets)
{
    return guarded([&] {
        KBO_REQUIRE(aln_concat && offsets && rles && rle_offsets, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(n_seqs > 0 && n_seqs < (1ull << 31), KBO_E_BAD_ARG, "1 .. 2^31-1 sequences");
        KBO_REQUIRE(offsets[0] == 0, KBO_E_BAD_ARG, "offsets[0] must be 0");
        const OffsetScan scan = scan_offsets(offsets, n_seqs);
        KBO_REQUIRE(scan.monotone, KBO_E_BAD_ARG, "offsets not monotone");
        KBO_REQUIRE(scan.longest < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "sequence longer than 2^32-1");
        const uint64_t total = offsets[n_seqs];
        hipStream_t st = nullptr;
        DevBuf chars(total + 64), off((n_seqs + 1) * sizeof(uint64_t)), count(16);
        DevBuf scratch(kbo::chunk_items_scratch_words((uint32_t)n_seqs) * sizeof(uint32_t));
        HIP_OK(hipMemsetAsync(static_cast<uint8_t *>(chars.p) + total, 0, 64, st));
        if (total) HIP_OK(hipMemcpyAsync(chars.p, aln_concat, total, hipMemcpyHostToDevice, st));
        HIP_OK(hipMemcpyAsync(off.p, offsets, (n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        const uint32_t gap = (uint32_t)std::min<size_t>(max_gap_len, 0xFFFFFFFFu);
        const uint32_t longest = (uint32_t)scan.longest;
        HIP_OK(kbo::launch_rle_count(chars.as<uint8_t>(), off.as<uint64_t>(), (uint32_t)n_seqs, gap, scratch.as<uint32_t>(),
                                     count.as<uint32_t>(), st, longest));
        uint32_t n_runs = 0;
        HIP_OK(hipMemcpy(&n_runs, count.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
        DevBuf d_runs(std::max<size_t>(16, (size_t)n_runs * kRleWords * sizeof(uint32_t)));
        HIP_OK(kbo::launch_rle_emit(chars.as<uint8_t>(), off.as<uint64_t>(), (uint32_t)n_seqs, gap, scratch.as<uint32_t>(),
                                    d_runs.as<uint32_t>(), n_runs, st, longest));
        std::vector<uint32_t> compact((size_t)n_runs * kRleWords + 1), words(kbo::chunk_items_scratch_words((uint32_t)n_seqs));
        if (n_runs) HIP_OK(hipMemcpy(compact.data(), d_runs.p, (size_t)n_runs * kRleWords * sizeof(uint32_t), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(words.data(), scratch.p, words.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        kbo_rle *all = static_cast<kbo_rle *>(std::malloc(std::max<size_t>(1, n_runs) * sizeof(kbo_rle)));
        if (!all) throw std::bad_alloc();
        widen_rles(all, compact.data(), n_runs, HostTeam::get());
        const uint32_t *local = words.data(), *sums = local + n_seqs + 1;
        for (size_t q = 0; q <= n_seqs; q++) rle_offsets[q] = (uint64_t)sums[q / 1024] + local[q];
        *rles = all;
    });
}

int kbo_relative_to_ref(const uint8_t *ref_seq, const uint8_t *aln, size_t len, uint8_t *out)
{
    return guarded([&] {
        KBO_REQUIRE(ref_seq && aln && out, KBO_E_BAD_ARG, "null argument");
        for (size_t i = 0; i < len; i++) { // format.rs:270-286
            const uint8_t a = aln[i];
            if (a == 'M' || a == 'R' || a == 'I') out[i] = ref_seq[i];
            else if (a == 'X' || a == 'D') out[i] = '-';
            else if (a != '-') out[i] = a;
            else out[i] = '-';
        }
    });
}

namespace {
// kbo::find over a batch (lib.rs:815-820: matches, then run_lengths_gapped per sequence; both on the device,
// slab by slab).  into == false: the record array is library-allocated and handed back through *rles_out;
// into == true: records go to the caller's `buf` of `capacity` records (none are written beyond it).
// Returns the number of runs of the batch.
size_t find_batch_impl(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                       const kbo_find_opts *opts, bool into, kbo_rle *buf, size_t capacity, kbo_rle **rles_out,
                       uint64_t *rle_offsets)
{
    kbo_find_opts o;
    if (opts) o = *opts; else kbo_find_opts_default(&o);
    RleSink sink;
    sink.max_gap_len = o.max_gap_len;
    sink.rle_offsets = rle_offsets;
    if (into) {
        sink.caller_owns = true;
        sink.all = buf;
        sink.all_cap = buf ? capacity : 0;
    }
    matches_batch_impl(idx, concat, offsets, n_seqs, o.max_error_prob, false, nullptr, &sink);
    if (sink.direct) { // one device: the records are already in place
        if (!into) {
            *rles_out = sink.all;
            sink.all = nullptr;
        }
        return sink.all_used;
    }
    // several devices: slabs completed out of order and were kept per slab; put them together
    const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, slab_bytes_for(idx));
    std::vector<uint64_t> base(slabs.size() + 1, 0);
    for (size_t i = 0; i < slabs.size(); i++) base[i + 1] = base[i] + sink.runs[i].size();
    kbo_rle *all = buf;
    if (!into) {
        all = static_cast<kbo_rle *>(std::malloc(std::max<uint64_t>(1, base.back()) * sizeof(kbo_rle)));
        if (!all) throw std::bad_alloc();
    }
    const bool fits = !into || (buf && base.back() <= capacity);
    rle_offsets[0] = 0;
    HostTeam::get().run(slabs.size(), [&](size_t i) {
        if (fits && !sink.runs[i].empty()) std::memcpy(all + base[i], sink.runs[i].data(), sink.runs[i].size() * sizeof(kbo_rle));
        const size_t ns = slabs[i].s1 - slabs[i].s0;
        for (size_t q = 1; q <= ns; q++) rle_offsets[slabs[i].s0 + q] = base[i] + sink.first[i][q];
    });
    if (!into) *rles_out = all;
    return base.back();
}
} // namespace

int kbo_find_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                   const kbo_find_opts *opts, kbo_rle **rles, uint64_t *rle_offsets)
{
    return guarded([&] {
        KBO_REQUIRE(idx && rles && rle_offsets, KBO_E_BAD_ARG, "null argument");
        find_batch_impl(idx, concat, offsets, n_seqs, opts, false, nullptr, 0, rles, rle_offsets);
    });
}

int kbo_find_batch_into(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                        const kbo_find_opts *opts, kbo_rle *rles, size_t capacity, uint64_t *rle_offsets, size_t *n_runs)
{
    return guarded([&] {
        KBO_REQUIRE(idx && (rles || capacity == 0) && rle_offsets && n_runs, KBO_E_BAD_ARG, "null argument");
        *n_runs = find_batch_impl(idx, concat, offsets, n_seqs, opts, true, rles, capacity, nullptr, rle_offsets);
        KBO_REQUIRE(*n_runs <= capacity, KBO_E_NOMEM, "rles holds fewer records than the batch has runs (*n_runs says how many)");
    });
}

// ---- 2-bit packed batches (pack_kernels.hip has the layout) ------------------------------------------------------------

size_t kbo_packed_words(const uint64_t *offsets, size_t n_seqs)
{
    if (!offsets) return 0;
    size_t w = 0;
    for (size_t s = 0; s < n_seqs; s++) w += (size_t)((offsets[s + 1] - offsets[s] + 15) / 16);
    return w;
}

int kbo_pack_reads(const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, uint32_t *words_out, uint64_t *exc_pos,
                   uint8_t *exc_byte, size_t exc_cap, size_t *n_exc)
{
    return guarded([&] {
        KBO_REQUIRE(concat && offsets && words_out && n_exc && (exc_cap == 0 || (exc_pos && exc_byte)), KBO_E_BAD_ARG, "null argument");
        std::vector<uint64_t> pw(n_seqs + 1, 0);
        for (size_t s = 0; s < n_seqs; s++) {
            KBO_REQUIRE(offsets[s + 1] >= offsets[s], KBO_E_BAD_ARG, "offsets not monotone");
            pw[s + 1] = pw[s] + (offsets[s + 1] - offsets[s] + 15) / 16;
        }
        const size_t piece = 4096, n_tasks = (n_seqs + piece - 1) / piece;
        std::vector<std::vector<std::pair<uint64_t, uint8_t>>> exc(n_tasks); // (in order inside a task, tasks in order)
        HostTeam::get().run(n_tasks, [&](size_t t) {
            for (size_t s = t * piece; s < std::min(n_seqs, (t + 1) * piece); s++) {
                const uint64_t b0 = offsets[s], len = offsets[s + 1] - b0;
                uint32_t *w = words_out + pw[s];
                for (uint64_t i0 = 0; i0 < len; i0 += 16) {
                    uint32_t v = 0;
                    const uint64_t nb = std::min<uint64_t>(16, len - i0);
                    for (uint64_t i = 0; i < nb; i++) {
                        const uint8_t ch = concat[b0 + i0 + i];
                        const uint32_t c = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
                        if (c == 4u) exc[t].emplace_back(b0 + i0 + i, ch);
                        v |= (c & 3u) << (2 * i);
                    }
                    w[i0 / 16] = v;
                }
            }
        });
        size_t total = 0;
        for (const auto &e : exc) total += e.size();
        *n_exc = total;
        KBO_REQUIRE(total <= exc_cap, KBO_E_NOMEM, "more non-ACGT bases than the exception list holds (*n_exc says how many)");
        size_t x = 0;
        for (const auto &e : exc)
            for (const auto &pr : e) {
                exc_pos[x] = pr.first;
                exc_byte[x++] = pr.second;
            }
    });
}

int kbo_unpack_matches(const uint32_t *words, const uint64_t *offsets, size_t n_seqs, uint8_t *chars_out)
{
    return guarded([&] {
        KBO_REQUIRE(words && offsets && chars_out, KBO_E_BAD_ARG, "null argument");
        std::vector<uint64_t> pw(n_seqs + 1, 0);
        for (size_t s = 0; s < n_seqs; s++) pw[s + 1] = pw[s] + (offsets[s + 1] - offsets[s] + 15) / 16;
        const size_t piece = 4096;
        HostTeam::get().run((n_seqs + piece - 1) / piece, [&](size_t t) {
            for (size_t s = t * piece; s < std::min(n_seqs, (t + 1) * piece); s++) {
                const uint64_t b0 = offsets[s], len = offsets[s + 1] - b0;
                const uint32_t *w = words + pw[s];
                for (uint64_t i = 0; i < len; i++) chars_out[b0 + i] = (uint8_t)"M-XR"[(w[i / 16] >> (2 * (i % 16))) & 3u];
            }
        });
    });
}

int kbo_matches_batch_packed(kbo_index_t *idx, const uint32_t *words, const uint64_t *offsets, size_t n_seqs, const uint64_t *exc_pos,
                             const uint8_t *exc_byte, size_t n_exc, double max_error_prob, uint32_t *words_out)
{
    return guarded([&] {
        KBO_REQUIRE(idx && words && words_out, KBO_E_BAD_ARG, "null argument");
        const PackedBatch in{words, exc_pos, exc_byte, n_exc};
        matches_batch_packed_impl(idx, in, offsets, n_seqs, max_error_prob, words_out, nullptr);
    });
}

int kbo_find_batch_packed(kbo_index_t *idx, const uint32_t *words, const uint64_t *offsets, size_t n_seqs, const uint64_t *exc_pos,
                          const uint8_t *exc_byte, size_t n_exc, const kbo_find_opts *opts, kbo_rle32 **rles, uint64_t *rle_offsets)
{
    return guarded([&] {
        KBO_REQUIRE(idx && words && rles && rle_offsets, KBO_E_BAD_ARG, "null argument");
        static_assert(sizeof(kbo_rle32) == kRleWords * sizeof(uint32_t), "kbo_rle32 is the device's record");
        kbo_find_opts o;
        if (opts) o = *opts; else kbo_find_opts_default(&o);
        RleSink sink;
        sink.max_gap_len = o.max_gap_len;
        sink.rle_offsets = rle_offsets;
        sink.compact = true;
        const PackedBatch in{words, exc_pos, exc_byte, n_exc};
        matches_batch_packed_impl(idx, in, offsets, n_seqs, o.max_error_prob, nullptr, &sink);
        if (sink.direct) { // one device: the records are already in place
            *rles = reinterpret_cast<kbo_rle32 *>(sink.all32);
            sink.all32 = nullptr;
            return;
        }
        // several devices: slabs completed out of order and were kept per slab; put them together
        const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, packed_slab_bytes(idx)); // (as matches_batch_packed_impl)
        std::vector<uint64_t> base(slabs.size() + 1, 0);
        for (size_t i = 0; i < slabs.size(); i++) base[i + 1] = base[i] + sink.runs32[i].size() / kRleWords;
        uint32_t *all = static_cast<uint32_t *>(std::malloc(std::max<uint64_t>(1, base.back()) * kRleWords * sizeof(uint32_t)));
        if (!all) throw std::bad_alloc();
        rle_offsets[0] = 0;
        HostTeam::get().run(slabs.size(), [&](size_t i) {
            if (!sink.runs32[i].empty()) std::memcpy(all + base[i] * kRleWords, sink.runs32[i].data(), sink.runs32[i].size() * sizeof(uint32_t));
            const size_t ns = slabs[i].s1 - slabs[i].s0;
            for (size_t q = 1; q <= ns; q++) rle_offsets[slabs[i].s0 + q] = base[i] + sink.first[i][q];
        });
        *rles = reinterpret_cast<kbo_rle32 *>(all);
    });
}

int kbo_find(kbo_index_t *idx, const uint8_t *query, size_t len, const kbo_find_opts *opts, kbo_rle **out,
             size_t *n_out)
{
    return guarded([&] {
        KBO_REQUIRE(idx && query && out && n_out, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(len > 0, KBO_E_EMPTY_QUERY, "assert!(!query.is_empty()) (index.rs:248)");
        const uint64_t off[2] = {0, len};
        uint64_t ro[2];
        int rc = kbo_find_batch(idx, query, off, 1, opts, out, ro);
        if (rc) throw KboError(rc, last_error());
        *n_out = ro[1];
    });
}

void kbo_free(void *p) { std::free(p); }

namespace {
// work buffer of kbo_ms_batch_dev: items, then the scan scratch of the chunked item list
struct DevWork {
    bool chunked;
    uint32_t chunk, n_slots;
    size_t bytes, plan_off;
    size_t ms_bytes;                 // what the walks alone need (kbo_ms_batch_dev, kbo_call_walk_dev: kbo_ms_work_bytes) - the regions below are kbo_map_batch_dev's / kbo_find_batch_dev's
    size_t long_off, long_bytes;     // batches with sequences of more than 160 bases: work of map_long_kernel (long_kernels.hip) ...
    size_t derand_off, derand_bytes; // ... and of the piece-wise derandomize + translate kernel behind the walk when it does not apply
};
DevWork dev_work(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k)
{
    DevWork w;
    w.chunk = (uint32_t)walk_chunk(total_bases, n_seqs, k);
    w.chunked = max_seq_len == 0 || max_seq_len > w.chunk;
    const uint64_t slots = w.chunked ? total_bases / w.chunk + n_seqs : n_seqs;
    w.n_slots = (uint32_t)std::min<uint64_t>(slots, 0xFFFFFFFFu);
    w.bytes = std::max<uint64_t>(1, slots) * sizeof(kbo::WalkItem);
    if (w.chunked) w.bytes += kbo::chunk_items_scratch_words((uint32_t)n_seqs) * sizeof(uint32_t) + 16;
    w.bytes = (w.bytes + 15) / 16 * 16;
    w.plan_off = w.bytes; // work of the plan-guided walk behind it
    w.bytes += kbo::plan_work_bytes(std::max<uint64_t>(1, slots), total_bases);
    w.bytes = (w.bytes + 63) / 64 * 64;
    w.long_off = w.derand_off = w.ms_bytes = w.bytes;
    w.long_bytes = w.derand_bytes = 0;
    if (max_seq_len == 0 || max_seq_len > 160) {
        w.long_bytes = (kbo::long_work_bytes(n_seqs, total_bases, k) + 63) / 64 * 64;
        w.derand_bytes = (kbo::derand_piece_work_bytes((uint32_t)std::min<size_t>(n_seqs, 0xFFFFFFFEu), total_bases) + 63) / 64 * 64;
        w.derand_off = w.long_off + w.long_bytes;
        w.bytes += w.long_bytes + w.derand_bytes;
    }
    return w;
}
} // namespace

size_t kbo_work_bytes(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k)
{
    return dev_work(n_seqs, total_bases, max_seq_len, k).bytes;
}

size_t kbo_ms_work_bytes(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k)
{
    return dev_work(n_seqs, total_bases, max_seq_len, k).ms_bytes;
}

size_t kbo_index_work_bytes(const kbo_index_t *idx, size_t n_seqs, uint64_t total_bases, size_t max_seq_len)
{
    if (!idx) return 0;
    const size_t base = dev_work(n_seqs, total_bases, max_seq_len, idx->host.k).bytes;
    return base + (idx->sharded() ? ((size_t)total_bases + 15) / 16 * 16 + 16 : 0); // + one further shard's MS values
}

namespace {
std::atomic<bool> g_ms_one_kernel{true}; // kbo_ms_batch_dev: batches of reads through map_reads_kernel's MS-emitting form (kbo_set_ms_one_kernel)
int ms_batch_dev_impl(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs,
                      uint64_t total_bases, size_t max_seq_len, uint8_t *d_ms_out, uint32_t *d_lo_out,
                      uint32_t *d_hi_out, void *d_work, size_t work_bytes, void *stream, const CallSink *call,
                      bool count_bases = true /* false: the caller's own device_view() has counted this batch for the copy's lazy plan structures */)
{
    return guarded([&] {
        KBO_REQUIRE(idx && d_concat && d_offsets && d_ms_out && d_work, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(n_seqs > 0 && total_bases > 0, KBO_E_EMPTY_QUERY, "empty batch");
        KBO_REQUIRE(n_seqs < (1ull << 28) && total_bases < 0xFFFFFF00ull, KBO_E_UNSUPPORTED,
                    "one launch covers < 2^28 sequences and < 4 GiB of query: split the batch");
        KBO_REQUIRE(((uintptr_t)d_concat & 15) == 0 && ((uintptr_t)d_ms_out & 3) == 0 &&
                        ((uintptr_t)d_work & 15) == 0,
                    KBO_E_BAD_ARG, "d_concat/d_work must be 16-byte and d_ms_out 4-byte aligned");
#ifndef KBO_WALK_DEBUG
        KBO_REQUIRE((d_lo_out == nullptr) == (d_hi_out == nullptr), KBO_E_BAD_ARG, "lo/hi must come together");
#endif
        hipStream_t s = static_cast<hipStream_t>(stream);
        const std::vector<kbo_index *> shards = shards_of(idx); // (a sharded index: every shard is walked, the maximum kept)
        KBO_REQUIRE(shards.size() == 1 || (!d_lo_out && !call), KBO_E_UNSUPPORTED,
                    "intervals and the call mode need the rows of one index; this handle is a sharded index");
        kbo::WalkItem *items = static_cast<kbo::WalkItem *>(d_work);
        // reads: one item per sequence; batches that hold (or may hold) long sequences: chunks
        const DevWork w = dev_work(n_seqs, total_bases, max_seq_len, idx->host.k);
        const size_t shard_ms = shards.size() > 1 ? ((size_t)total_bases + 15) / 16 * 16 + 16 : 0; // one further shard's MS values
        // (the walks never touch the regions of the kernels for long sequences: only kbo_map_batch_dev / kbo_find_batch_dev ask for those)
        KBO_REQUIRE(work_bytes >= w.ms_bytes + shard_ms, KBO_E_BAD_ARG,
                    "d_work is smaller than kbo_ms_work_bytes() (+ one shard's MS values for a sharded index: kbo_index_work_bytes()) for this batch");
        KBO_REQUIRE(total_bases / w.chunk + n_seqs < (1ull << 28), KBO_E_UNSUPPORTED, "more than 2^28 work items per launch");
        // a batch of reads over a copy with a depth table, nothing but the MS values asked for: map_reads_kernel in the form that puts
        // the values together in LDS (k where nothing happened, the ramps behind the mismatches, the table's values right behind them),
        // stopping there - no characters are made - and the plain walk for the reads it leaves (C2: 0.33 against 0.61 ms per
        // 1 M reads for the plan-guided walk below, which stays what larger batches' chunks, the intervals, the call mode and the
        // work counters take)
        if (shards.size() == 1 && !d_lo_out && !call && !w.chunked && max_seq_len > 0 && g_ms_one_kernel.load() != 0 && !g_plan_stats.load()) {
            DevCopy::PlanState *plan_state = nullptr;
            const kbo::DevIndexView view = device_view(idx, current_device(), &plan_state, count_bases ? total_bases : 0);
            kbo::WalkArgs a{};
            a.ix = view;
            a.q = d_concat;
            a.q_bytes = total_bases;
            a.items = items;
            a.n_items = w.n_slots;
            a.d_out = d_ms_out;
            a.max_item_len = (uint32_t)max_seq_len;
            attach_plan(a, static_cast<uint8_t *>(d_work) + w.plan_off, plan_state);
            a.chars_out = nullptr;
            a.map_thr = idx->host.k; // (no value is derandomised here)
            a.map_fmt = 0;
            a.map_want_ms = 1;
            count_bases = false; // (this batch is counted: the walk below, should the copy not have what the kernel needs, must not count it again)
            if (a.gitems && kbo::map_reads_applies(a)) {
                a.seq_off = d_offsets;
                a.host_bailed = plan_state ? plan_state->bailed : nullptr;
                HIP_OK(kbo::launch_map_reads(a, s));
                if (kbo::map_reads_finish_applies(a)) HIP_OK(kbo::launch_map_reads_finish(a, s)); // (the reads it left: their values by one kernel)
                else HIP_OK(kbo::launch_redo_pass(a, s));
                if (!a.host_bailed) plan_after_launch(a, s, plan_state);
                return;
            }
        }
        if (w.chunked) {
            uint32_t *scratch = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(d_work) +
                                                            ((size_t)w.n_slots * sizeof(kbo::WalkItem) + 15) / 16 * 16);
            HIP_OK(kbo::launch_make_chunk_items(d_offsets, (uint32_t)n_seqs, w.chunk, idx->host.k, w.n_slots, items, scratch, s, call != nullptr));
        } else {
            HIP_OK(kbo::launch_make_items(d_offsets, (uint32_t)n_seqs, items, s));
        }
        uint8_t *ms_shard = static_cast<uint8_t *>(d_work) + w.bytes; // (16-byte aligned: w.bytes is a multiple of 16)
        for (size_t sh = 0; sh < shards.size(); sh++) {
            DevCopy::PlanState *plan_state = nullptr;
            const kbo::DevIndexView view = device_view(shards[sh], current_device(), &plan_state, count_bases ? total_bases : 0);
            kbo::WalkArgs a{};
            a.ix = view;
            a.q = d_concat;
            a.q_bytes = total_bases;
            a.items = items;
            a.n_items = w.n_slots;
            a.rounds = 0;
            a.d_out = sh == 0 ? d_ms_out : ms_shard;
            a.lo_out = d_lo_out;
            a.hi_out = d_hi_out;
            a.call_sites = call ? static_cast<uint4 *>(call->d_sites) : nullptr;
            a.call_counts = call ? call->d_counts : nullptr;
            a.call_cap = call ? call->cap_per_list : 0;
            a.call_thr = call ? call->threshold : 0;
            a.max_item_len = w.chunked ? w.chunk + (call ? 2u : 1u) * idx->host.k : (uint32_t)std::min<size_t>(max_seq_len, 0xFFFFFFFFu);
            attach_plan(a, static_cast<uint8_t *>(d_work) + w.plan_off, plan_state);
            HIP_OK(kbo::launch_ms_walk(a, walk_max_waves(), s));
            plan_after_launch(a, s, plan_state);
            if (sh > 0) HIP_OK(kbo::launch_max_bytes(d_ms_out, ms_shard, total_bases, s)); // depth against the union = maximum
        }
    });
}
} // namespace

int kbo_ms_batch_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs,
                     uint64_t total_bases, size_t max_seq_len, uint8_t *d_ms_out, uint32_t *d_lo_out,
                     uint32_t *d_hi_out, void *d_work, size_t work_bytes, void *stream)
{
    return ms_batch_dev_impl(idx, d_concat, d_offsets, n_seqs, total_bases, max_seq_len, d_ms_out, d_lo_out, d_hi_out, d_work,
                             work_bytes, stream, nullptr);
}

int kbo_call_walk_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs,
                      uint64_t total_bases, size_t max_seq_len, size_t threshold, uint8_t *d_ms_out, void *d_sites,
                      size_t capacity, uint32_t *d_count, void *d_work, size_t work_bytes, void *stream)
{
    if (!d_sites || !d_count || capacity < kbo::kCallSegs || capacity > 0x7FFFFF00ull) {
        last_error() = "kbo_call_walk_dev: bad site buffer";
        return KBO_E_BAD_ARG;
    }
    if (hipMemsetAsync(d_count, 0, kbo::kCallSegs * 64 + 64, static_cast<hipStream_t>(stream)) != hipSuccess) return KBO_E_HIP;
    const CallSink sink{d_sites, d_count, (uint32_t)(capacity / kbo::kCallSegs), (uint32_t)threshold};
    return ms_batch_dev_impl(idx, d_concat, d_offsets, n_seqs, total_bases, max_seq_len, d_ms_out, nullptr, nullptr, d_work,
                             work_bytes, stream, &sink);
}

int kbo_plan_stats_dev(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k, const void *d_work,
                       uint64_t out[KBO_PLAN_STATS], void *stream)
{
    return guarded([&] {
        KBO_REQUIRE(d_work && out && n_seqs > 0 && total_bases > 0, KBO_E_BAD_ARG, "null / empty argument");
        const DevWork w = dev_work(n_seqs, total_bases, max_seq_len, k);
        const kbo::PlanLayout L = kbo::plan_layout(w.n_slots, total_bases);
        const uint8_t *plan = static_cast<const uint8_t *>(d_work) + w.plan_off;
        hipStream_t s = static_cast<hipStream_t>(stream);
        uint32_t ctl[16], st[kbo::kPlanStatSlots * kbo::kPlanStatWords], tot[2];
        const uint32_t last = 2u * w.n_slots; // the scan's extra entry: its prefix is the number of units
        HIP_OK(hipMemcpyAsync(ctl, plan + L.qctl, sizeof ctl, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(st, plan + L.pstats, sizeof st, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(&tot[0], plan + L.ucount + (size_t)last * 4, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipMemcpyAsync(&tot[1], plan + L.usums + (size_t)(last / 1024u) * 4, 4, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
        for (uint32_t i = 0; i < KBO_PLAN_STATS; i++) out[i] = 0;
        for (uint32_t sl = 0; sl < kbo::kPlanStatSlots; sl++) {
            for (uint32_t i = 0; i < 8; i++) out[i] += st[sl * kbo::kPlanStatWords + i];
            for (uint32_t i = 0; i < 3; i++) out[12 + i] += st[sl * kbo::kPlanStatWords + kbo::kPlanStatTabLookups + i];
            out[16] += st[sl * kbo::kPlanStatWords + kbo::kPlanStatTabAnchored];
        }
        out[15] = ctl[4];
        out[17] = ctl[5];
        out[8] = (uint64_t)tot[0] + tot[1];
        out[9] = ctl[1];
        out[10] = ctl[2];
        out[11] = ctl[3];
    });
}

int kbo_plan_flags_dev(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k, const void *d_work, uint8_t *flags_out, void *stream)
{
    return guarded([&] {
        KBO_REQUIRE(d_work && flags_out && n_seqs > 0 && total_bases > 0, KBO_E_BAD_ARG, "null / empty argument");
        const DevWork w = dev_work(n_seqs, total_bases, max_seq_len, k);
        KBO_REQUIRE(!w.chunked, KBO_E_UNSUPPORTED, "one item per sequence only (reads)");
        const kbo::PlanLayout L = kbo::plan_layout(w.n_slots, total_bases);
        hipStream_t s = static_cast<hipStream_t>(stream);
        HIP_OK(hipMemcpyAsync(flags_out, static_cast<const uint8_t *>(d_work) + w.plan_off + L.redo, n_seqs, hipMemcpyDeviceToHost, s));
        HIP_OK(hipStreamSynchronize(s));
    });
}

int kbo_set_map_long(int mode)
{
    kbo::set_map_long(mode);
    return KBO_OK;
}

int kbo_set_ms_one_kernel(int on)
{
    g_ms_one_kernel = on != 0;
    return KBO_OK;
}

int kbo_long_stats_dev(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k, const void *d_work, uint64_t out[KBO_LONG_STATS],
                       void *stream)
{
    return guarded([&] {
        KBO_REQUIRE(d_work && out && n_seqs > 0 && total_bases > 0, KBO_E_BAD_ARG, "null / empty argument");
        const DevWork w = dev_work(n_seqs, total_bases, max_seq_len, k);
        KBO_REQUIRE(w.long_bytes != 0, KBO_E_UNSUPPORTED, "not a batch of long sequences");
        uint32_t ctl[32], st[kbo::kPlanStatSlots * kbo::kPlanStatWords];
        HIP_OK(kbo::long_read_stats(static_cast<const uint8_t *>(d_work) + w.long_off, n_seqs, total_bases, k, ctl, st, static_cast<hipStream_t>(stream)));
        for (uint32_t i = 0; i < KBO_LONG_STATS; i++) out[i] = 0;
        out[0] = ctl[0];
        out[1] = ctl[4];
        out[2] = ctl[1];
        for (uint32_t i = 0; i < 5 && 8 + i < KBO_LONG_STATS; i++) out[8 + i] = (uint64_t)ctl[16 + i] << 4; // (KBO_LONG_X & 128: shader cycles by phase)
        for (uint32_t i = 0; i < 9 && 16 + i < KBO_LONG_STATS; i++) out[16 + i] = ctl[8 + i]; // (why flagged: list cap / ext / back / on; +4: after the band pass)
        for (uint32_t sl = 0; sl < kbo::kPlanStatSlots; sl++) {
            out[3] += st[sl * kbo::kPlanStatWords + kbo::kPlanStatSeedLookups];
            out[4] += st[sl * kbo::kPlanStatWords + kbo::kPlanStatSeedExtensions];
            out[5] += st[sl * kbo::kPlanStatWords + kbo::kPlanStatTabLookups];
            out[6] += st[sl * kbo::kPlanStatWords + kbo::kPlanStatTabAnchored];
            out[13] += st[sl * kbo::kPlanStatWords + kbo::kPlanStatUnits];
            out[14] += st[sl * kbo::kPlanStatWords + kbo::kPlanStatAccepted];
        }
    });
}

size_t kbo_derand_work_bytes(size_t n_seqs, uint64_t total_bases)
{
    return kbo::derand_piece_work_bytes((uint32_t)std::min<size_t>(n_seqs, 0xFFFFFFFEu), total_bases);
}

int kbo_derand_translate_dev(const uint8_t *d_ms, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                             size_t k, size_t threshold, const uint8_t *d_ref, uint8_t *d_chars_out,
                             size_t max_seq_len, void *d_work, size_t work_bytes, void *stream)
{
    return guarded([&] {
        KBO_REQUIRE(d_ms && d_offsets && d_chars_out, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(n_seqs > 0 && n_seqs < 0xFFFFFFFFull, KBO_E_EMPTY_QUERY, "empty batch");
        KBO_REQUIRE(k > 0 && k <= 255, KBO_E_BAD_ARG, "k in 1..255");
        KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (derandomize.rs:275)");
        KBO_REQUIRE(((uintptr_t)d_ms & 3) == 0 && ((uintptr_t)d_chars_out & 3) == 0 && ((uintptr_t)d_ref & 3) == 0 &&
                        ((uintptr_t)d_work & 15) == 0,
                    KBO_E_BAD_ARG, "device buffers must be 4-byte (d_work 16-byte) aligned");
        HIP_OK(kbo::launch_derand_translate(d_ms, d_offsets, (uint32_t)n_seqs, (uint32_t)k, (uint32_t)threshold,
                                            d_ref, d_chars_out, nullptr,
                                            (uint32_t)std::min<size_t>(max_seq_len, 0xFFFFFFFFu), 0xFFFFFFFFu,
                                            static_cast<hipStream_t>(stream), total_bases, d_work, work_bytes));
    });
}

size_t kbo_run_lengths_work_bytes(size_t n_seqs)
{
    return kbo::chunk_items_scratch_words((uint32_t)std::min<size_t>(n_seqs, 0xFFFFFFFEu)) * sizeof(uint32_t) + 16;
}

namespace {
// kbo_set_stage_timing: event triples of the one-kernel route's calls (guarded by g_timing_mu)
std::atomic<int> g_stage_timing{0};
std::mutex g_timing_mu;
struct StageEvents { hipEvent_t e0, e1, e1t, e2; };
std::vector<StageEvents> g_timing_pool, g_timing_used;
StageEvents timing_take()
{
    std::lock_guard<std::mutex> g(g_timing_mu);
    StageEvents ev{};
    if (!g_timing_pool.empty()) {
        ev = g_timing_pool.back();
        g_timing_pool.pop_back();
    } else {
        HIP_OK(hipEventCreate(&ev.e0));
        HIP_OK(hipEventCreate(&ev.e1));
        HIP_OK(hipEventCreate(&ev.e1t));
        HIP_OK(hipEventCreate(&ev.e2));
    }
    return ev;
}
} // namespace

int kbo_set_stage_timing(int on)
{
    if (on <= 1) {
        g_stage_timing = on != 0;
        return KBO_OK;
    }
    // on > 1: the events of that many calls are made here, not inside the calls that are to be timed
    return guarded([&] {
        std::vector<StageEvents> held;
        for (int i = 0; i < on; ++i)
            held.push_back(timing_take());
        std::lock_guard<std::mutex> g(g_timing_mu);
        g_timing_pool.insert(g_timing_pool.end(), held.begin(), held.end());
        g_stage_timing = 1;
    });
}

int kbo_stage_timing_read(double *kernel_ms_sum, double *redo_ms_sum, int *n_calls)
{
    return guarded([&] {
        std::lock_guard<std::mutex> g(g_timing_mu);
        double a = 0, b = 0;
        for (const StageEvents &ev : g_timing_used) {
            HIP_OK(hipEventSynchronize(ev.e2));
            float x = 0, y = 0;
            HIP_OK(hipEventElapsedTime(&x, ev.e0, ev.e1));
            HIP_OK(hipEventElapsedTime(&y, ev.e1t, ev.e2));
            a += x;
            b += y;
            g_timing_pool.push_back(ev);
        }
        if (kernel_ms_sum) *kernel_ms_sum = a;
        if (redo_ms_sum) *redo_ms_sum = b;
        if (n_calls) *n_calls = (int)g_timing_used.size();
        g_timing_used.clear();
    });
}

namespace {
// one fence event per host thread and device: hipStreamWaitEvent takes the event's state at the time of the call, so recording it
// again for the next batch does not disturb a wait that is already queued
hipEvent_t tail_fence()
{
    thread_local std::map<int, hipEvent_t> evs;
    hipEvent_t &ev = evs[current_device()];
    if (!ev) HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return ev;
}

// kbo::find behind the characters (kbo_find_batch_dev): format::run_lengths_gapped into d_records
struct FindTail {
    size_t max_gap_len;
    void *d_rle_work;
    uint32_t *d_records;
    size_t capacity;
};

int map_batch_dev_impl(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                       size_t max_seq_len, double max_error_prob, int format, int want_ms, uint8_t *d_ms, uint8_t *d_chars_out,
                       void *d_work, size_t work_bytes, void *stream, void *tail_stream, bool split, int *fused,
                       const FindTail *find = nullptr)
{
    if (fused) *fused = 0;
    bool done = false, counted = false; // (counted: device_view() has seen this batch's bases for the copy's lazy plan structures)
    size_t threshold = 0;
    int rc = guarded([&] {
        KBO_REQUIRE(idx && d_concat && d_offsets && d_ms && d_chars_out && d_work, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(n_seqs > 0 && total_bases > 0, KBO_E_EMPTY_QUERY, "empty batch");
        KBO_REQUIRE(n_seqs < (1ull << 28) && total_bases < 0xFFFFFF00ull, KBO_E_UNSUPPORTED,
                    "one launch covers < 2^28 sequences and < 4 GiB of query: split the batch");
        KBO_REQUIRE(((uintptr_t)d_concat & 15) == 0 && ((uintptr_t)d_ms & 3) == 0 && ((uintptr_t)d_chars_out & 3) == 0 &&
                        ((uintptr_t)d_work & 15) == 0,
                    KBO_E_BAD_ARG, "d_concat/d_work must be 16-byte, d_ms/d_chars_out 4-byte aligned");
        threshold = random_match_threshold(idx->host.k, idx->host.n_kmers, 4, max_error_prob);
        KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (derandomize.rs:275)");
        if (idx->sharded()) return; // (two kernels, below)
        hipStream_t s = static_cast<hipStream_t>(stream);
        const DevWork w = dev_work(n_seqs, total_bases, max_seq_len, idx->host.k);
        KBO_REQUIRE(work_bytes >= w.bytes, KBO_E_BAD_ARG, "d_work is smaller than kbo_work_bytes() for this batch");
        DevCopy::PlanState *plan_state = nullptr;
        if (max_seq_len == 0 || max_seq_len > 160) {
            // sequences of any length: one wave per piece of a sequence (long_kernels.hip), the pieces whose proof fails by the
            // plain walk + the literal recurrences behind it (on the tail stream when the caller gave one)
            if (want_ms || !w.long_bytes) return;
            const kbo::DevIndexView view = device_view(idx, current_device(), &plan_state, total_bases);
            counted = true;
            if (!kbo::map_long_applies(view, (uint32_t)threshold)) return;
            kbo::LongArgs la{};
            HIP_OK(kbo::launch_map_long(view, d_concat, d_offsets, (uint32_t)n_seqs, total_bases, (uint32_t)threshold, format != 0, d_chars_out,
                                        static_cast<uint8_t *>(d_work) + w.long_off, s, la, g_plan_stats.load()));
            hipStream_t ts = s;
            if (split && static_cast<hipStream_t>(tail_stream) != s) {
                ts = static_cast<hipStream_t>(tail_stream);
                hipEvent_t fence = tail_fence();
                HIP_OK(hipEventRecord(fence, s));
                HIP_OK(hipStreamWaitEvent(ts, fence, 0));
            }
            HIP_OK(kbo::launch_map_long_redo(la, d_ms, ts));
            if (find) {
                uint32_t *rle_scratch = static_cast<uint32_t *>(find->d_rle_work);
                uint32_t *total = rle_scratch + kbo::chunk_items_scratch_words((uint32_t)n_seqs);
                const uint32_t gap = (uint32_t)std::min<size_t>(find->max_gap_len, 0xFFFFFFFFu), cap = (uint32_t)std::min<size_t>(find->capacity, 0xFFFFFFFFu);
                const uint32_t longest = (uint32_t)std::min<size_t>(max_seq_len, 0xFFFFFFFFu);
                HIP_OK(kbo::launch_rle_count(d_chars_out, d_offsets, (uint32_t)n_seqs, gap, rle_scratch, total, ts, longest, true));
                if (cap) HIP_OK(kbo::launch_rle_emit(d_chars_out, d_offsets, (uint32_t)n_seqs, gap, rle_scratch, find->d_records, cap, ts, longest, true));
            }
            done = true;
            return;
        }
        if (w.chunked) return;
        const kbo::DevIndexView view = device_view(idx, current_device(), &plan_state, total_bases);
        counted = true;
        kbo::WalkItem *items = static_cast<kbo::WalkItem *>(d_work);
        kbo::WalkArgs a{};
        a.ix = view;
        a.q = d_concat;
        a.q_bytes = total_bases;
        a.items = items;
        a.n_items = w.n_slots;
        a.d_out = d_ms;
        a.max_item_len = (uint32_t)max_seq_len;
        attach_plan(a, static_cast<uint8_t *>(d_work) + w.plan_off, plan_state);
        a.chars_out = d_chars_out;
        a.map_thr = (uint32_t)threshold;
        a.map_fmt = format ? 1u : 0u;
        a.map_want_ms = want_ms ? 1u : 0u;
        if (!a.gitems || !kbo::map_reads_applies(a)) return; // (no plan structures, or the copy is held off: two kernels)
        a.seq_off = d_offsets; // (the kernel and redo_collect_kernel read the offsets themselves: no item list is made)
        a.host_bailed = plan_state ? plan_state->bailed : nullptr; // (set by redo_collect_kernel itself: no 8-byte copy behind the launch)
        // kbo::find with max_gap_len = 0: the kernel counts the runs of the reads it finishes (their characters are in LDS anyway), so
        // that format::run_lengths_gapped is one pass over the characters instead of two
        uint32_t *rle_scratch = find ? static_cast<uint32_t *>(find->d_rle_work) : nullptr;
        const bool count_in_kernel = find && find->max_gap_len == 0 && !format && !want_ms && kbo::map_reads_direct(a);
        if (count_in_kernel) a.run_counts = rle_scratch;
        const bool timing = g_stage_timing.load() != 0;
        StageEvents ev{};
        if (timing) {
            ev = timing_take();
            HIP_OK(hipEventRecord(ev.e0, s));
        }
        HIP_OK(kbo::launch_map_reads(a, s));
        if (timing) HIP_OK(hipEventRecord(ev.e1, s));
        hipStream_t ts = s;
        if (split && static_cast<hipStream_t>(tail_stream) != s) { // the second pass on the caller's other stream, behind the kernel
            ts = static_cast<hipStream_t>(tail_stream);
            hipEvent_t fence = tail_fence();
            HIP_OK(hipEventRecord(fence, s));
            HIP_OK(hipStreamWaitEvent(ts, fence, 0));
            // (beside another batch's kernel the pass costs by the look-ups it takes away from that kernel rather than by its longest
            // chain: longer pieces, fewer warm-up bases.  Pieces of 16 / 24 / 32 / 48 / 64 bases at C2, two batches in flight:
            // 0.341 / 0.324 / 0.316 / 0.343 / 0.337 ms per batch)
            static const bool env_piece = std::getenv("KBO_REDO_PIECE") != nullptr;
            if (!env_piece) a.redo_piece = 32u;
        }
        if (timing) HIP_OK(hipEventRecord(ev.e1t, ts)); // (when the second pass starts: behind the kernel and behind what `ts` held)
        if (kbo::map_reads_finish_applies(a)) {
            // the reads the kernel listed, finished by one kernel: walk, derandomize + translate, characters (and their runs)
            HIP_OK(kbo::launch_map_reads_finish(a, ts));
        } else {
            HIP_OK(kbo::launch_redo_pass(a, ts)); // (redo_collect_kernel reads the offsets as well: no item list at all)
            HIP_OK(kbo::launch_derand_flagged(d_ms, d_offsets, (uint32_t)n_seqs, idx->host.k, (uint32_t)threshold, format ? d_concat : nullptr,
                                              d_chars_out, a.redo, (uint32_t)max_seq_len, ts, count_in_kernel ? rle_scratch : nullptr));
        }
        if (timing) {
            HIP_OK(hipEventRecord(ev.e2, ts));
            std::lock_guard<std::mutex> g(g_timing_mu);
            g_timing_used.push_back(ev);
        }
        if (!a.host_bailed) plan_after_launch(a, ts, plan_state);
        if (find) { // the run lengths, behind the second pass
            uint32_t *total = rle_scratch + kbo::chunk_items_scratch_words((uint32_t)n_seqs); // last word of the work buffer
            const uint32_t gap = (uint32_t)std::min<size_t>(find->max_gap_len, 0xFFFFFFFFu), cap = (uint32_t)std::min<size_t>(find->capacity, 0xFFFFFFFFu);
            if (count_in_kernel) { // (the flagged reads' counts: launch_derand_flagged's)
                HIP_OK(kbo::launch_rle_scan_counts((uint32_t)n_seqs, rle_scratch, total, ts));
            } else
                HIP_OK(kbo::launch_rle_count(d_chars_out, d_offsets, (uint32_t)n_seqs, gap, rle_scratch, total, ts, (uint32_t)max_seq_len, true));
            if (cap) HIP_OK(kbo::launch_rle_emit(d_chars_out, d_offsets, (uint32_t)n_seqs, gap, rle_scratch, find->d_records, cap, ts, (uint32_t)max_seq_len, true));
        }
        done = true;
    });
    if (rc != KBO_OK || done) {
        if (fused && done) *fused = 1;
        return rc;
    }
    rc = ms_batch_dev_impl(idx, d_concat, d_offsets, n_seqs, total_bases, max_seq_len, d_ms, nullptr, nullptr, d_work, work_bytes, stream, nullptr, !counted);
    if (rc != KBO_OK) return rc;
    const DevWork w = dev_work(n_seqs, total_bases, max_seq_len, idx->host.k);
    rc = kbo_derand_translate_dev(d_ms, d_offsets, n_seqs, total_bases, idx->host.k, threshold, format ? d_concat : nullptr, d_chars_out,
                                  max_seq_len, w.derand_bytes ? static_cast<uint8_t *>(d_work) + w.derand_off : nullptr, w.derand_bytes, stream);
    if (rc != KBO_OK || !find) return rc;
    return kbo_run_lengths_dev(d_chars_out, d_offsets, n_seqs, max_seq_len, find->max_gap_len, find->d_rle_work, find->d_records, find->capacity, stream);
}
} // namespace

int kbo_find_batch_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                       size_t max_seq_len, double max_error_prob, size_t max_gap_len, uint8_t *d_ms, uint8_t *d_chars_out, void *d_work,
                       size_t work_bytes, void *d_rle_work, uint32_t *d_records, size_t capacity, void *stream, void *tail_stream, int *fused)
{
    if (!d_rle_work || (!d_records && capacity) || ((uintptr_t)d_rle_work & 3) || ((uintptr_t)d_records & 3) || n_seqs >= (1ull << 31)) {
        last_error() = "kbo_find_batch_dev: bad run-length buffers";
        return KBO_E_BAD_ARG;
    }
    const FindTail ft{max_gap_len, d_rle_work, d_records, capacity};
    return map_batch_dev_impl(idx, d_concat, d_offsets, n_seqs, total_bases, max_seq_len, max_error_prob, 0, 0, d_ms, d_chars_out, d_work,
                              work_bytes, stream, tail_stream, true, fused, &ft);
}

int kbo_map_batch_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                      size_t max_seq_len, double max_error_prob, int format, int want_ms, uint8_t *d_ms, uint8_t *d_chars_out,
                      void *d_work, size_t work_bytes, void *stream, int *fused)
{
    return map_batch_dev_impl(idx, d_concat, d_offsets, n_seqs, total_bases, max_seq_len, max_error_prob, format, want_ms, d_ms, d_chars_out,
                              d_work, work_bytes, stream, nullptr, false, fused);
}

int kbo_map_batch_dev_tail(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                           size_t max_seq_len, double max_error_prob, int format, int want_ms, uint8_t *d_ms, uint8_t *d_chars_out,
                           void *d_work, size_t work_bytes, void *stream, void *tail_stream, int *fused)
{
    return map_batch_dev_impl(idx, d_concat, d_offsets, n_seqs, total_bases, max_seq_len, max_error_prob, format, want_ms, d_ms, d_chars_out,
                              d_work, work_bytes, stream, tail_stream, true, fused);
}

// ---- kbo_map_stream_*: pipelines of (kernel stream, second-pass stream), two slots each
// A (kernel stream, second-pass stream) pair in which the KERNELS' stream is kept off some compute units (32 of the device's 256 by
// default; KBO_TAIL_CUS / tail_cus = how many, 0 = two plain streams) and the second passes' stream is a plain one: a second pass is a
// chain of dependent look-ups by a few hundred waves, and beside kernels that hold every wave slot of the device each link of the chain
// waits for a slot - 0.12 ms alone, 0.37 beside two kernels, which then wait for it in turn.  With units the kernels cannot take, the
// pass's workgroups find free slots at once - and one that is WORK (5 % substitutions, long sequences, repeats) still spreads over the
// whole device.  C2, two pipelines, same box: plain / plain 757 Gbp/s; second passes CONFINED to 16 units of their own 973, but 5 %
// substitutions 245 -> 76 (adaptive forms of that: LABNOTES round 6); kernels off 32 units, second passes plain: 967 and every variant at
// or above the plain arrangement (reserved 16 / 24 / 32 / 40 / 48 units: 878 / 958 / 967 / 904 / 896 - 32 is four per XCD).
static int tail_cus_default() // compute units the kernels' streams stay off (KBO_TAIL_CUS; 0 = plain streams, the arrangement of rounds 4 - 5)
{
    static const int v = std::getenv("KBO_TAIL_CUS") ? std::atoi(std::getenv("KBO_TAIL_CUS")) : 32;
    return v;
}
static void make_stream_pair(int device, int tail_cus, hipStream_t *ks, hipStream_t *ts)
{
    const int want = tail_cus >= 0 ? tail_cus : tail_cus_default();
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, device));
    const int n_cu = prop.multiProcessorCount;
    HIP_OK(hipStreamCreateWithFlags(ts, hipStreamNonBlocking));
    if (want <= 0 || want >= n_cu) {
        HIP_OK(hipStreamCreateWithFlags(ks, hipStreamNonBlocking));
        return;
    }
    std::vector<uint32_t> mask_k((size_t)(n_cu + 31) / 32, 0u);
    for (int cu = want; cu < n_cu; cu++) mask_k[(size_t)cu / 32] |= 1u << (cu % 32);
    // (a runtime that refuses the mask - none seen - gets a plain stream: slower, never wrong)
    if (hipExtStreamCreateWithCUMask(ks, (uint32_t)mask_k.size(), mask_k.data()) != hipSuccess) {
        (void)hipGetLastError();
        *ks = nullptr;
        HIP_OK(hipStreamCreateWithFlags(ks, hipStreamNonBlocking));
    }
}

int kbo_stream_pair_create(int tail_cus, void **stream, void **tail_stream)
{
    return guarded([&] {
        KBO_REQUIRE(stream && tail_stream, KBO_E_BAD_ARG, "null argument");
        hipStream_t ks = nullptr, ts = nullptr;
        make_stream_pair(current_device(), tail_cus, &ks, &ts);
        *stream = ks;
        *tail_stream = ts;
    });
}

void kbo_stream_pair_destroy(void *stream, void *tail_stream)
{
    if (stream) (void)hipStreamDestroy(static_cast<hipStream_t>(stream));
    if (tail_stream) (void)hipStreamDestroy(static_cast<hipStream_t>(tail_stream));
}

struct kbo_map_stream {
    kbo_index_t *idx = nullptr;
    int device = 0;
    // a pipeline: the kernels' stream `ks` (kept off 32 compute units) and the second passes' stream `ts` (plain): make_stream_pair
    struct Pipe { hipStream_t ks = nullptr, ts = nullptr; };
    struct Slot {
        DevBuf work, ms;
        hipEvent_t done = nullptr;
        uint64_t ticket = 0; // the batch that used it last (0: none yet)
    };
    std::vector<Pipe> pipes;
    std::deque<Slot> slots; // 2 per pipeline (a deque: the buffers do not move)
    size_t max_seqs = 0, max_seq_len = 0, work_bytes = 0;
    uint64_t max_bases = 0, next = 0;
    hipEvent_t ready = nullptr, kdone = nullptr;
    std::mutex mu;
    kbo_map_stream() = default;
    kbo_map_stream(const kbo_map_stream &) = delete;
    kbo_map_stream &operator=(const kbo_map_stream &) = delete;
    ~kbo_map_stream() // (also what a create that fails half-way leaves: whatever it had made so far)
    {
        for (auto &p : pipes) {
            if (p.ks) (void)hipStreamSynchronize(p.ks);
            if (p.ts) (void)hipStreamSynchronize(p.ts);
        }
        for (auto &sl : slots)
            if (sl.done) (void)hipEventDestroy(sl.done);
        if (ready) (void)hipEventDestroy(ready);
        if (kdone) (void)hipEventDestroy(kdone);
        for (auto &p : pipes) {
            if (p.ks) (void)hipStreamDestroy(p.ks);
            if (p.ts) (void)hipStreamDestroy(p.ts);
        }
    }
};

int kbo_map_stream_create(kbo_index_t *idx, int pipelines, size_t max_seqs, uint64_t max_bases, size_t max_seq_len, kbo_map_stream_t **out)
{
    return guarded([&] {
        KBO_REQUIRE(idx && out && max_seqs > 0 && max_bases > 0, KBO_E_BAD_ARG, "null / empty argument");
        KBO_REQUIRE(pipelines >= 1 && pipelines <= 8, KBO_E_BAD_ARG, "1 .. 8 pipelines");
        *out = nullptr;
        std::unique_ptr<kbo_map_stream> m(new kbo_map_stream());
        m->idx = idx;
        m->device = current_device();
        m->max_seqs = max_seqs;
        m->max_bases = max_bases;
        m->max_seq_len = max_seq_len;
        m->work_bytes = kbo_index_work_bytes(idx, max_seqs, max_bases, max_seq_len);
        m->pipes.resize((size_t)pipelines);
        m->slots.resize(2 * (size_t)pipelines);
        for (auto &p : m->pipes) {
            make_stream_pair(m->device, -1, &p.ks, &p.ts);
        }
        for (auto &sl : m->slots) {
            sl.work.alloc(m->work_bytes + 64);
            sl.ms.alloc(((size_t)max_bases + 15) / 16 * 16 + 64);
            HIP_OK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        }
        HIP_OK(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&m->kdone, hipEventDisableTiming));
        *out = m.release();
    });
}

int kbo_map_stream_submit(kbo_map_stream_t *m, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                          size_t max_seq_len, double max_error_prob, int format, uint8_t *d_ms_out, uint8_t *d_chars_out, void *ready_stream,
                          uint64_t *ticket, int *fused)
{
    if (!m) {
        last_error() = "kbo_map_stream_submit: null stream";
        return KBO_E_BAD_ARG;
    }
    std::lock_guard<std::mutex> g(m->mu);
    int rc = guarded([&] {
        KBO_REQUIRE(n_seqs <= m->max_seqs && total_bases <= m->max_bases, KBO_E_BAD_ARG, "the batch exceeds what the stream's slots were made for");
        // (the stream's queues, events and slots live on the device it was created on: a submit from a thread whose current device is another
        // one would pair that device's copy of the index with them)
        KBO_REQUIRE(current_device() == m->device, KBO_E_BAD_ARG, "kbo_map_stream_submit: the calling thread's current device is not the one the stream was created on");
        KBO_REQUIRE(kbo_index_work_bytes(m->idx, n_seqs, total_bases, max_seq_len) <= m->work_bytes, KBO_E_BAD_ARG,
                    "the batch needs more work memory than the stream's slots have (max_seq_len of kbo_map_stream_create)");
    });
    if (rc != KBO_OK) return rc;
    const uint64_t n = m->next;
    kbo_map_stream::Pipe &p = m->pipes[n % m->pipes.size()];
    kbo_map_stream::Slot &sl = m->slots[n % m->slots.size()];
    hipStream_t kern = p.ks, tail = p.ts;
    rc = guarded([&] {
        // the slot's buffers are free again behind its last batch; the inputs are there behind what ready_stream holds so far
        if (sl.ticket) HIP_OK(hipStreamWaitEvent(kern, sl.done, 0));
        if (ready_stream) {
            HIP_OK(hipEventRecord(m->ready, static_cast<hipStream_t>(ready_stream)));
            HIP_OK(hipStreamWaitEvent(kern, m->ready, 0));
        }
    });
    if (rc != KBO_OK) return rc;
    rc = map_batch_dev_impl(m->idx, d_concat, d_offsets, n_seqs, total_bases, max_seq_len, max_error_prob, format, d_ms_out ? 1 : 0,
                            d_ms_out ? d_ms_out : sl.ms.as<uint8_t>(), d_chars_out, sl.work.p, m->work_bytes, kern, tail, true, fused);
    if (rc != KBO_OK) return rc;
    rc = guarded([&] { // complete when both streams have come this far
        HIP_OK(hipEventRecord(m->kdone, kern));
        HIP_OK(hipStreamWaitEvent(tail, m->kdone, 0));
        HIP_OK(hipEventRecord(sl.done, tail));
    });
    if (rc != KBO_OK) return rc;
    m->next = n + 1;
    sl.ticket = n + 1;
    if (ticket) *ticket = n + 1;
    return KBO_OK;
}

namespace {
// the event that says the batch with this ticket is complete: its own while its slot still holds it, else that of the next batch of
// the same pipeline whose slot still holds it - a pipeline's batches complete in the order they were submitted (submit never blocks, so
// a slot taken again says nothing about the batch that had it before; the pipeline's latest batch is always held)
hipEvent_t map_stream_event(kbo_map_stream_t *m, uint64_t ticket)
{
    KBO_REQUIRE(m && ticket >= 1 && ticket <= m->next, KBO_E_BAD_ARG, "no such batch");
    for (uint64_t t = ticket; t <= m->next; t += m->pipes.size()) {
        kbo_map_stream::Slot &sl = m->slots[(t - 1) % m->slots.size()];
        if (sl.ticket == t) return sl.done;
    }
    return nullptr; // (not reached: the pipeline's latest batch holds its slot)
}
} // namespace

int kbo_map_stream_wait(kbo_map_stream_t *m, uint64_t ticket)
{
    return guarded([&] {
        hipEvent_t ev;
        {
            KBO_REQUIRE(m, KBO_E_BAD_ARG, "null stream");
            std::lock_guard<std::mutex> g(m->mu);
            ev = map_stream_event(m, ticket);
        }
        if (ev) HIP_OK(hipEventSynchronize(ev));
    });
}

int kbo_map_stream_wait_on(kbo_map_stream_t *m, uint64_t ticket, void *stream)
{
    return guarded([&] {
        KBO_REQUIRE(m, KBO_E_BAD_ARG, "null stream");
        std::lock_guard<std::mutex> g(m->mu);
        hipEvent_t ev = map_stream_event(m, ticket);
        if (ev) HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), ev, 0));
    });
}

int kbo_map_stream_sync(kbo_map_stream_t *m)
{
    return guarded([&] {
        KBO_REQUIRE(m, KBO_E_BAD_ARG, "null stream");
        std::lock_guard<std::mutex> g(m->mu);
        for (auto &p : m->pipes)
            for (hipStream_t s : {p.ks, p.ts})
                if (s) HIP_OK(hipStreamSynchronize(s));
    });
}

void kbo_map_stream_free(kbo_map_stream_t *m)
{
    delete m; // (waits for what its streams still hold)
}

namespace {
struct PackedScratch { size_t q, ms, chars, pscr, exc, bytes; };
PackedScratch packed_scratch(size_t n_seqs, uint64_t total_bases)
{
    auto up = [](size_t v) { return (v + 63) / 64 * 64; };
    PackedScratch L{};
    const size_t padded = up(total_bases + 32);
    L.q = 0;
    L.ms = L.q + padded;
    L.chars = L.ms + padded;
    L.pscr = L.chars + padded;
    L.exc = L.pscr + up(kbo::chunk_items_scratch_words((uint32_t)n_seqs) * sizeof(uint32_t));
    L.bytes = L.exc + up(n_seqs + 16);
    return L;
}
} // namespace

size_t kbo_matches_packed_dev_scratch_bytes(size_t n_seqs, uint64_t total_bases) { return packed_scratch(n_seqs, total_bases).bytes; }

int kbo_matches_packed_dev(kbo_index_t *idx, const uint32_t *d_words, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                           size_t max_seq_len, size_t uniform_len, const uint64_t *d_exc_pos, const uint8_t *d_exc_byte, size_t n_exc,
                           double max_error_prob, uint32_t *d_words_out, void *d_scratch, void *d_work, size_t work_bytes, void *stream,
                           void *tail_stream)
{
    return guarded([&] {
        KBO_REQUIRE(idx && d_words && d_offsets && d_words_out && d_scratch && d_work, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(n_exc == 0 || (d_exc_pos && d_exc_byte), KBO_E_BAD_ARG, "exception list missing");
        KBO_REQUIRE(n_seqs > 0 && total_bases > 0, KBO_E_EMPTY_QUERY, "empty batch");
        KBO_REQUIRE(n_seqs < (1ull << 28) && total_bases < 0xFFFFFF00ull && n_exc < 0xFFFFFFFFull, KBO_E_UNSUPPORTED,
                    "one launch covers < 2^28 sequences and < 4 GiB of query: split the batch");
        KBO_REQUIRE(((uintptr_t)d_scratch & 15) == 0 && ((uintptr_t)d_work & 15) == 0 && ((uintptr_t)d_words & 3) == 0 && ((uintptr_t)d_words_out & 3) == 0,
                    KBO_E_BAD_ARG, "d_scratch/d_work must be 16-byte, the words 4-byte aligned");
        KBO_REQUIRE(max_seq_len > 0 && max_seq_len <= 160 && !idx->sharded(), KBO_E_UNSUPPORTED,
                    "kbo_matches_packed_dev: reads of at most 160 bases over an unsharded index (else: kbo_matches_batch_packed)");
        const size_t threshold = random_match_threshold(idx->host.k, idx->host.n_kmers, 4, max_error_prob);
        KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (derandomize.rs:275)");
        hipStream_t s = static_cast<hipStream_t>(stream), ts = static_cast<hipStream_t>(tail_stream);
        const DevWork w = dev_work(n_seqs, total_bases, max_seq_len, idx->host.k);
        KBO_REQUIRE(!w.chunked, KBO_E_UNSUPPORTED, "reads only");
        KBO_REQUIRE(work_bytes >= w.bytes, KBO_E_BAD_ARG, "d_work is smaller than kbo_work_bytes() for this batch");
        const PackedScratch L = packed_scratch(n_seqs, total_bases);
        uint8_t *sc = static_cast<uint8_t *>(d_scratch);
        uint8_t *q = sc + L.q, *ms = sc + L.ms, *chars = sc + L.chars, *exc = sc + L.exc;
        uint32_t *pscr = uniform_len ? nullptr : reinterpret_cast<uint32_t *>(sc + L.pscr);
        const uint32_t wps = uniform_len ? (uint32_t)((uniform_len + 15) / 16) : 0u;
        DevCopy::PlanState *plan_state = nullptr;
        const kbo::DevIndexView view = device_view(idx, current_device(), &plan_state, total_bases);
        kbo::WalkItem *items = static_cast<kbo::WalkItem *>(d_work);
        kbo::WalkArgs a{};
        a.ix = view;
        a.q = q;
        a.q_bytes = total_bases;
        a.items = items;
        a.n_items = w.n_slots;
        a.d_out = ms;
        a.max_item_len = (uint32_t)max_seq_len;
        attach_plan(a, static_cast<uint8_t *>(d_work) + w.plan_off, plan_state);
        a.chars_out = chars;
        a.map_thr = (uint32_t)threshold;
        a.map_fmt = 0;
        a.map_want_ms = 0;
        KBO_REQUIRE(a.gitems && kbo::map_reads_packed_applies(a, true), KBO_E_UNSUPPORTED,
                    "this copy of the index cannot take the packed-native kernel (no depth table, a held-off copy, a threshold below the "
                    "table's order): kbo_matches_batch_packed takes any batch");
        if (pscr) HIP_OK(kbo::launch_packed_prefix(d_offsets, (uint32_t)n_seqs, pscr, s));
        a.seq_off = d_offsets; // (no item list: the kernels read the offsets)
        a.qp = d_words;
        a.qp_wps = wps;
        a.qp_data = pscr;
        a.qp_sums = pscr ? pscr + n_seqs + 1u : nullptr;
        a.packed_out = d_words_out;
        if (n_exc) {
            HIP_OK(hipMemsetAsync(exc, 0, n_seqs, s));
            HIP_OK(kbo::launch_flag_exceptions(d_exc_pos, (uint32_t)n_exc, 0, d_offsets, (uint32_t)n_seqs, exc, s));
            a.qp_exc = exc;
        }
        const bool timing = g_stage_timing.load() != 0;
        StageEvents ev{};
        if (timing) {
            ev = timing_take();
            HIP_OK(hipEventRecord(ev.e0, s));
        }
        HIP_OK(kbo::launch_map_reads(a, s));
        if (timing) HIP_OK(hipEventRecord(ev.e1, s));
        if (ts != s) {
            hipEvent_t fence = tail_fence();
            HIP_OK(hipEventRecord(fence, s));
            HIP_OK(hipStreamWaitEvent(ts, fence, 0));
            static const bool env_piece = std::getenv("KBO_REDO_PIECE") != nullptr;
            if (!env_piece) a.redo_piece = 32u;
        }
        if (timing) HIP_OK(hipEventRecord(ev.e1t, ts));
        HIP_OK(kbo::launch_unpack_flagged(d_words, d_offsets, (uint32_t)n_seqs, wps, pscr, a.redo, q, ts));
        HIP_OK(kbo::launch_exceptions(d_exc_pos, d_exc_byte, (uint32_t)n_exc, 0, q, ts));
        HIP_OK(kbo::launch_redo_pass(a, ts));
        HIP_OK(kbo::launch_derand_flagged(ms, d_offsets, (uint32_t)n_seqs, idx->host.k, (uint32_t)threshold, nullptr, chars, a.redo, (uint32_t)max_seq_len, ts));
        HIP_OK(kbo::launch_pack_flagged(chars, d_offsets, (uint32_t)n_seqs, wps, pscr, a.redo, d_words_out, ts));
        if (timing) {
            HIP_OK(hipEventRecord(ev.e2, ts));
            std::lock_guard<std::mutex> g(g_timing_mu);
            g_timing_used.push_back(ev);
        }
        plan_after_launch(a, ts, plan_state);
    });
}

int kbo_run_lengths_dev(const uint8_t *d_chars, const uint64_t *d_offsets, size_t n_seqs, size_t max_seq_len,
                        size_t max_gap_len, void *d_work, uint32_t *d_records, size_t capacity, void *stream)
{
    return guarded([&] {
        KBO_REQUIRE(d_chars && d_offsets && d_work && (d_records || capacity == 0), KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(n_seqs > 0 && n_seqs < (1ull << 31), KBO_E_BAD_ARG, "1 .. 2^31-1 sequences");
        KBO_REQUIRE(((uintptr_t)d_work & 3) == 0 && ((uintptr_t)d_records & 3) == 0, KBO_E_BAD_ARG, "4-byte alignment");
        hipStream_t s = static_cast<hipStream_t>(stream);
        const uint32_t gap = (uint32_t)std::min<size_t>(max_gap_len, 0xFFFFFFFFu);
        uint32_t *scratch = static_cast<uint32_t *>(d_work);
        uint32_t *total = scratch + kbo::chunk_items_scratch_words((uint32_t)n_seqs); // last word of the work buffer
        const uint32_t longest = (uint32_t)std::min<size_t>(max_seq_len, 0xFFFFFFFFu);
        HIP_OK(kbo::launch_rle_count(d_chars, d_offsets, (uint32_t)n_seqs, gap, scratch, total, s, longest));
        if (capacity)
            HIP_OK(kbo::launch_rle_emit(d_chars, d_offsets, (uint32_t)n_seqs, gap, scratch, d_records,
                                        (uint32_t)std::min<size_t>(capacity, 0xFFFFFFFFu), s, longest));
    });
}

int kbo_walk_geometry(int *blocks, int *threads)
{
    return guarded([&] {
        if (blocks) *blocks = walk_max_waves();
        if (threads) *threads = kbo::kWalkThreads;
    });
}

int kbo_set_devices(const int *devices, int n)
{
    return guarded([&] {
        KBO_REQUIRE(n >= 0 && (devices || n == 0), KBO_E_BAD_ARG, "bad device list");
        int count = 0;
        if (n > 0) HIP_OK(hipGetDeviceCount(&count)); // (n == 0, back to the current device, needs no device at all)
        for (int i = 0; i < n; i++) KBO_REQUIRE(devices[i] >= 0 && devices[i] < count, KBO_E_BAD_ARG, "no such device");
        std::lock_guard<std::mutex> g(g_devices_mu);
        g_devices.assign(devices, devices + n);
    });
}

// ---- per-handle options
int kbo_index_opts_default(kbo_index_opts_t *opts)
{
    return guarded([&] {
        KBO_REQUIRE(opts, KBO_E_BAD_ARG, "null argument");
        std::memset(opts, 0, sizeof *opts);
        opts->struct_size = (uint32_t)sizeof *opts;
        opts->plan = opts->depth_table = opts->depth_table_anchors = KBO_OPT_INHERIT;
        opts->n_devices = -1;
    });
}

namespace {
void apply_opts(kbo_index *idx, const kbo_index_opts_t &o)
{
    idx->opts.plan = o.plan == KBO_OPT_INHERIT ? kOptInherit : (o.plan != 0 ? 1 : 0);
    idx->opts.depth_table = o.depth_table == KBO_OPT_INHERIT ? kOptInherit : (o.depth_table < 0 ? -1 : std::min(o.depth_table, 17));
    idx->opts.depth_table_anchors =
        o.depth_table_anchors == KBO_OPT_INHERIT ? kOptInherit : (o.depth_table_anchors < 0 ? -1 : (o.depth_table_anchors != 0 ? 1 : 0));
    idx->opts.slab_bytes = o.slab_bytes ? std::max<size_t>(1u << 16, std::min<size_t>(o.slab_bytes, 0xF0000000ull)) : 0;
    {
        std::lock_guard<std::mutex> g(idx->mu);
        idx->opts.n_devices = o.n_devices;
        idx->opts.devices.assign(o.devices, o.devices + std::max(0, o.n_devices));
    }
    if (o.plan != KBO_OPT_INHERIT && o.plan != 0) plan_reset_holdoff();
    for (auto &sh : idx->shards) apply_opts(sh.get(), o); // (a sharded index: its shards are what gets copied and walked)
}
} // namespace

int kbo_index_set_opts(kbo_index_t *idx, const kbo_index_opts_t *opts)
{
    return guarded([&] {
        KBO_REQUIRE(idx && opts, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(opts->struct_size == sizeof *opts, KBO_E_BAD_ARG, "kbo_index_opts_t: struct_size is not this library's (fill it with kbo_index_opts_default)");
        KBO_REQUIRE(opts->n_devices >= -1 && opts->n_devices <= KBO_OPT_MAX_DEVICES, KBO_E_BAD_ARG, "n_devices: -1 .. KBO_OPT_MAX_DEVICES");
        if (opts->n_devices > 0) {
            int count = 0;
            HIP_OK(hipGetDeviceCount(&count));
            for (int i = 0; i < opts->n_devices; i++) KBO_REQUIRE(opts->devices[i] >= 0 && opts->devices[i] < count, KBO_E_BAD_ARG, "no such device");
        }
        apply_opts(idx, *opts);
    });
}

int kbo_index_get_opts(const kbo_index_t *idx_c, kbo_index_opts_t *opts)
{
    return guarded([&] {
        KBO_REQUIRE(idx_c && opts, KBO_E_BAD_ARG, "null argument");
        kbo_index *idx = const_cast<kbo_index *>(idx_c);
        std::memset(opts, 0, sizeof *opts);
        opts->struct_size = (uint32_t)sizeof *opts;
        auto out = [](int v) { return v == kOptInherit ? KBO_OPT_INHERIT : v; };
        opts->plan = out(idx->opts.plan.load());
        opts->depth_table = out(idx->opts.depth_table.load());
        opts->depth_table_anchors = out(idx->opts.depth_table_anchors.load());
        opts->slab_bytes = idx->opts.slab_bytes.load();
        std::lock_guard<std::mutex> g(idx->mu);
        opts->n_devices = idx->opts.n_devices;
        for (int i = 0; i < idx->opts.n_devices; i++) opts->devices[i] = idx->opts.devices[(size_t)i];
    });
}

int kbo_set_host_threads(int n)
{
    HostTeam::get().set_threads(n < 1 ? 1u : (unsigned)n);
    HostTeam::out().set_threads(n < 1 ? 1u : (unsigned)n);
    return KBO_OK;
}

int kbo_release_scratch(void)
{
    return guarded([&] {
        release_host_scratch();
    });
}

int kbo_set_pair_steps(uint64_t min_rows, int min_depth)
{
    g_pair_min_rows = min_rows; // applies to device copies made after the call
    if (min_depth >= 0) kbo::set_pair_min_depth(min_depth);
    return KBO_OK;
}

int kbo_set_plan(int enabled, int seed_depth, int seed_cap)
{
    if (enabled >= 0) g_plan_enabled = enabled != 0; // launches from now on; path covers of device copies made from now on
    if (enabled > 0) plan_reset_holdoff();
    kbo::set_plan_params(seed_depth, seed_cap); // (<= 0 keeps a value)
    if (const char *e = std::getenv("KBO_PLAN_GAP")) kbo::set_plan_params(0, 0, std::atoi(e), 0);   // experiments
    if (const char *e = std::getenv("KBO_PLAN_BAIL")) kbo::set_plan_bail(std::atoi(e));
    if (const char *e = std::getenv("KBO_PLAN_CHUNK")) kbo::set_plan_params(0, 0, 0, std::atoi(e));
    return KBO_OK;
}

int kbo_index_recovery_lines(const kbo_index_t *idx, uint8_t *lines, size_t *n_bytes)
{
    return guarded([&] {
        KBO_REQUIRE(idx && n_bytes, KBO_E_BAD_ARG, "null argument");
        require_unsharded(idx, "kbo_index_recovery_lines");
        const size_t need = (idx->host.n_sets / kbo::kFatRows + 3) * 128;
        if (lines) {
            KBO_REQUIRE(*n_bytes >= need, KBO_E_BAD_ARG, "buffer smaller than the lines");
            std::vector<uint8_t> v;
            kbo::make_recovery_lines(idx->host, v);
            std::memcpy(lines, v.data(), v.size());
        }
        *n_bytes = need;
    });
}

int kbo_index_depth_table(kbo_index_t *idx, int device, int view, uint8_t *table, size_t *n_bytes, int *order)
{
    return guarded([&] {
        KBO_REQUIRE(idx && n_bytes && order && view >= 0 && view < 3, KBO_E_BAD_ARG, "null argument / view not in 0..2");
        const int dev = device < 0 ? current_device() : device;
        const kbo::DevIndexView v = device_view(idx, dev, nullptr, 0, true); // (no table while kbo_set_depth_table(-1) is in force)
        const size_t need = v.dtab ? (size_t)1 << (2u * v.dtab_order) : 0;
        if (table && need) {
            KBO_REQUIRE(*n_bytes >= need, KBO_E_BAD_ARG, "buffer smaller than the table");
            KBO_REQUIRE(v.dtab_order <= 14u, KBO_E_UNSUPPORTED, "tables of more than 14 bases are not handed back (a test hook: GiBs through a host copy)");
            int prev = current_device();
            if (prev != dev) HIP_OK(hipSetDevice(dev));
            hipError_t e = hipSuccess;
            if (!v.dtab_grouped) e = hipMemcpy(table, v.dtab, need, hipMemcpyDeviceToHost);
            else { // the grouped table holds every entry three times: the copy `view` of them, put back in key order
                std::vector<uint8_t> g(kbo::dtab_bytes(v.dtab_order, true));
                e = hipMemcpy(g.data(), v.dtab, g.size(), hipMemcpyDeviceToHost);
                const uint32_t cb = 2u * (v.dtab_order - 2u);
                const uint64_t cm = (1ull << cb) - 1ull;
                for (uint64_t key = 0; key < need; key++) {
                    const uint64_t a = view == 0 ? ((key & cm) << 6) + (key >> cb)
                                     : view == 1 ? (((key >> 2) & cm) << 6) + 16u + ((key >> (cb + 2u)) << 2) + (key & 3u)
                                                 : ((key >> 4) << 6) + 32u + (key & 15u);
                    table[key] = g[a];
                }
            }
            if (prev != dev) (void)hipSetDevice(prev);
            HIP_OK(e);
        }
        *n_bytes = need;
        *order = (int)v.dtab_order;
    });
}

int kbo_index_path_cover(const kbo_index_t *idx, uint8_t *text, uint32_t *pos, uint32_t *node_at)
{
    return guarded([&] {
        KBO_REQUIRE(idx && text && pos && node_at, KBO_E_BAD_ARG, "null argument");
        require_unsharded(idx, "kbo_index_path_cover");
        kbo_index_t *ix = const_cast<kbo_index_t *>(idx); // (fills the handle's cache of its cover, under its mutex)
        std::lock_guard<std::mutex> g(ix->mu);
        if (!ix->cover) {
            ix->cover.reset(new kbo::PathCover());
            kbo::make_path_cover(ix->host, *ix->cover);
        }
        const kbo::PathCover &pc = *ix->cover;
        std::memcpy(text, pc.text.data() + kbo::PathCover::kPad, idx->host.n_sets);
        std::memcpy(pos, pc.pos.data(), idx->host.n_sets * 4);
        std::memcpy(node_at, pc.node_at.data(), idx->host.n_sets * 4);
    });
}

int kbo_set_walk_experiment(int lane_limit, int dummy_lds_bytes)
{
    kbo::set_walk_experiment(lane_limit, dummy_lds_bytes);
    return KBO_OK;
}

int kbo_set_plan_tuning(int gap, int chunk, int bail_x16)
{
    kbo::set_plan_params(0, 0, gap, chunk);
    if (bail_x16 >= 0) kbo::set_plan_bail(bail_x16);
    return KBO_OK;
}

int kbo_set_plan_unit_cap_divisor(int divisor)
{
    g_plan_cap_div = std::max(1, divisor);
    return KBO_OK;
}

int kbo_set_index_shards(int shards)
{
    g_index_shards = std::max(0, shards);
    return KBO_OK;
}

int kbo_index_shards(const kbo_index_t *idx) { return idx ? (idx->sharded() ? (int)idx->shards.size() : 1) : 0; }

const kbo_index_t *kbo_index_shard(const kbo_index_t *idx, int i)
{
    if (!idx || i < 0) return nullptr;
    if (!idx->sharded()) return i == 0 ? idx : nullptr;
    return (size_t)i < idx->shards.size() ? idx->shards[(size_t)i].get() : nullptr;
}

int kbo_set_plan_stats(int on)
{
    g_plan_stats = on != 0;
    return KBO_OK;
}

int kbo_set_plan_table_budget(uint64_t bytes)
{
    g_plan_table_budget = bytes;
    return KBO_OK;
}

int kbo_set_plan_lazy(int64_t bases)
{
    g_plan_lazy_bases = bases < 0 ? -1 : bases;
    return KBO_OK;
}

int kbo_set_depth_table(int order)
{
    g_depth_table = order < 0 ? -1 : std::min(order, 17);
    return KBO_OK;
}

int kbo_set_depth_table_anchors(int mode)
{
    g_depth_table_anchors = mode < 0 ? -1 : (mode != 0 ? 1 : 0);
    return KBO_OK;
}

int kbo_set_seed_table_depth(int bases)
{
    g_seed_table_depth = std::max(0, std::min(bases, 14));
    return KBO_OK;
}

int kbo_index_plan_holdoff(kbo_index_t *idx, int device, uint32_t *bails, int *holdoff)
{
    return guarded([&] {
        KBO_REQUIRE(idx, KBO_E_BAD_ARG, "null index");
        require_unsharded(idx, "kbo_index_plan_holdoff");
        const int dev = device < 0 ? current_device() : device;
        std::lock_guard<std::mutex> g(idx->mu);
        auto it = idx->dev.find(dev);
        KBO_REQUIRE(it != idx->dev.end(), KBO_E_BAD_ARG, "the index has no copy on that device");
        if (bails) *bails = it->second->plan.bails.load();
        if (holdoff) *holdoff = it->second->plan.holdoff.load();
    });
}

// test hook (kbo_hip_tuning.h): the rank blocks, contraction entries and two-base blocks of the copy on `device` (made on the device:
// layout_kernels.hip) against the host's make_device_layout -> *n_diff = bytes that differ (0: the same layout)
int kbo_index_layout_check(kbo_index_t *idx, int device, uint64_t *n_diff)
{
    return guarded([&] {
        KBO_REQUIRE(idx && n_diff, KBO_E_BAD_ARG, "null argument");
        require_unsharded(idx, "kbo_index_layout_check");
        const int dev = device < 0 ? current_device() : device;
        std::lock_guard<std::mutex> g(idx->mu);
        auto it = idx->dev.find(dev);
        KBO_REQUIRE(it != idx->dev.end(), KBO_E_BAD_ARG, "the index has no copy on that device");
        const DevCopy &dc = *it->second;
        kbo::DeviceLayout lay;
        kbo::make_device_layout(idx->host, lay, dc.pair_off != 0);
        KBO_REQUIRE(lay.n_blocks == dc.n_blocks, KBO_E_BAD_ARG, "block counts differ");
        const size_t per = lay.n_blocks * 16, ent_bytes = lay.ent.size() * 4, pair_bytes = lay.pair.size() * 4;
        uint64_t diff = 0;
        auto compare = [&](const void *d_ptr, const void *h_ptr, size_t bytes) {
            std::vector<uint8_t> got(bytes);
            HIP_OK(hipMemcpy(got.data(), d_ptr, bytes, hipMemcpyDeviceToHost));
            const uint8_t *want = static_cast<const uint8_t *>(h_ptr);
            for (size_t i = 0; i < bytes; i++) diff += got[i] != want[i];
        };
        for (int c = 0; c < 4; c++) compare(dc.arena.as<uint8_t>() + per * c, lay.rank[c].data(), per);
        compare(dc.big ? dc.ent.as<uint8_t>() : dc.arena.as<uint8_t>() + per * 4 + 16, lay.ent.data(), ent_bytes);
        if (dc.pair_off) compare(dc.arena.as<uint8_t>() + (size_t)dc.pair_off * 16, lay.pair.data(), pair_bytes);
        *n_diff = diff;
    });
}

// test hook (kbo_hip_tuning.h): the handle's path cover (laid out on the device when its first copy made it: cover_kernels.hip) against
// the host's make_path_cover -> *n_diff = positions of text / pos / node_at that differ (0: the same layout); KBO_E_BAD_ARG without a cover
int kbo_index_cover_check(kbo_index_t *idx, uint64_t *n_diff)
{
    return guarded([&] {
        KBO_REQUIRE(idx && n_diff, KBO_E_BAD_ARG, "null argument");
        require_unsharded(idx, "kbo_index_cover_check");
        std::lock_guard<std::mutex> g(idx->mu);
        KBO_REQUIRE((bool)idx->cover, KBO_E_BAD_ARG, "the handle has no path cover yet (kbo_index_to_device makes it)");
        kbo::PathCover want;
        kbo::make_path_cover(idx->host, want);
        const kbo::PathCover &got = *idx->cover;
        KBO_REQUIRE(got.text.size() == want.text.size() && got.pos.size() == want.pos.size() && got.node_at.size() == want.node_at.size(),
                    KBO_E_BAD_ARG, "sizes differ");
        uint64_t diff = 0;
        for (size_t i = 0; i < want.text.size(); i++) diff += got.text[i] != want.text[i];
        for (size_t i = 0; i < want.pos.size(); i++) diff += (got.pos[i] != want.pos[i]) + (got.node_at[i] != want.node_at[i]);
        *n_diff = diff;
    });
}

int kbo_index_device_layout(kbo_index_t *idx, int device, kbo_device_layout *out)
{
    return guarded([&] {
        KBO_REQUIRE(idx && out, KBO_E_BAD_ARG, "null argument");
        require_unsharded(idx, "kbo_index_device_layout");
        const int dev = device < 0 ? current_device() : device;
        std::lock_guard<std::mutex> g(idx->mu);
        auto it = idx->dev.find(dev);
        KBO_REQUIRE(it != idx->dev.end(), KBO_E_BAD_ARG, "the index has no copy on that device");
        const DevCopy &dc = *it->second;
        *out = kbo_device_layout{};
        out->rank_bytes = dc.setup.rank_bytes;
        out->entry_bytes = dc.setup.entry_bytes;
        out->pair_bytes = dc.setup.pair_bytes;
        out->cover_bytes = dc.setup.cover_bytes;
        out->lines_bytes = dc.setup.lines_bytes;
        out->seed_bytes = dc.setup.seed_bytes;
        out->dtab_bytes = dc.setup.dtab_bytes;
        out->anchor_bytes = dc.setup.anchor_bytes;
        out->entries_64bit = dc.big ? 1u : 0u;
        out->seed_depth = dc.seed_d;
        out->dtab_order = dc.dtab_order;
        out->dtab_grouped = dc.dtab_grouped ? 1u : 0u;
        out->layout_seconds = dc.setup.layout_s;
        out->upload_seconds = dc.setup.upload_s;
        out->cover_seconds = dc.setup.cover_s;
        out->lines_seconds = dc.setup.lines_s;
        out->seed_seconds = dc.setup.seed_s;
        out->dtab_seconds = dc.setup.dtab_s;
    });
}

uint64_t kbo_index_device_plan_bytes(const kbo_index_t *idx)
{
    if (!idx) return 0;
    uint64_t b = 0;
    for (kbo_index *sh : shards_of(const_cast<kbo_index *>(idx))) b += sh->plan_bytes;
    return b;
}

int kbo_set_force_big_layout(int on)
{
    g_force_big = on != 0; // applies to device copies made after the call
    return KBO_OK;
}

int kbo_set_host_in_place(int on)
{
    g_host_in_place = on != 0;
    return KBO_OK;
}

int kbo_set_slab_bytes(size_t bytes)
{
    g_slab_bytes = std::max<size_t>(1u << 16, std::min<size_t>(bytes, 0xF0000000ull));
    return KBO_OK;
}

int kbo_set_walk_rare(int period)
{
    kbo::set_walk_rare(period);
    return KBO_OK;
}

int kbo_set_walk_threads(int threads)
{
    kbo::set_walk_threads(threads);
    return KBO_OK;
}

int kbo_set_guided_walk(int waves_per_cu, int recovery_lines)
{
    kbo::set_guided_walk(waves_per_cu, recovery_lines);
    return KBO_OK;
}

int kbo_set_walk_waves_per_cu(int waves_per_cu)
{
    g_waves_per_cu = waves_per_cu > 0 ? waves_per_cu : 0;
    return KBO_OK;
}

} // extern "C"
