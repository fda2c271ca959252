// call_batch.cpp — kbo::call (reference lib.rs:547-573, variant_calling.rs:249-294) over a batch of sequences.
//
// The first pass of call_variants - the MS walk of every sequence against the index and the breakpoint scan over it -
// runs on the device for the whole batch (ms_walk_kernel with intervals, call_sites_kernel); only the sites
// {sequence, i, j, row} come back, about one record per mismatch instead of 9 bytes per base.  The second pass keeps
// the reference's shape: the k-mers of every site (refine.cpp), the walk of all query-side k-mers against the index
// in ONE batch, and per sequence (the reference builds an index of every sequence it is called with, lib.rs:553) the
// walk of its reference-side k-mers against that sequence's own index.
#include "capi_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <atomic>
#include <mutex>
#include <thread>

using namespace kbo_host;

namespace {

// KBO_TIMING=1 in the environment: phases of kbo_call_batch on stderr
struct CallClock {
    bool on = std::getenv("KBO_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[kbo timing] call: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

struct SiteRec {
    uint32_t seq, i, j, lo;
};

// MS values (no intervals) of a list of equally long k-mers against `idx`, one GPU batch
void ms_only(kbo_index *idx, const std::vector<std::vector<uint8_t>> &seqs, std::vector<std::vector<kbo::MsVal>> &out)
{
    out.assign(seqs.size(), {});
    if (seqs.empty()) return;
    std::vector<uint64_t> off(seqs.size() + 1, 0);
    for (size_t s = 0; s < seqs.size(); s++) off[s + 1] = off[s] + seqs[s].size();
    std::vector<uint8_t> concat(off.back());
    for (size_t s = 0; s < seqs.size(); s++) std::memcpy(concat.data() + off[s], seqs[s].data(), seqs[s].size());
    std::vector<uint8_t> d(off.back() + 16);
    ms_batch_impl(idx, concat.data(), off.data(), seqs.size(), d.data(), nullptr, nullptr);
    for (size_t s = 0; s < seqs.size(); s++) {
        out[s].resize(seqs[s].size());
        for (size_t i = 0; i < seqs[s].size(); i++) out[s][i] = kbo::MsVal{d[off[s] + i], 0u, 0u};
    }
}

// the same for the few k-mers of one sequence against that sequence's own index, from a pool thread of kbo_call_batch: the
// calling thread's own buffers and stream, no slabs, no pinned staging, no shared worker team (a millisecond of fixed
// costs per call that all threads would queue for)
// (device buffers kept per host thread; DevBuf::ensure gives a thread that has moved to another device fresh memory)
BatchOnDevice &small_batch_buffers()
{
    static thread_local BatchOnDevice B;
    return B;
}
void ms_only_small(kbo_index *idx, const std::vector<std::vector<uint8_t>> &seqs, std::vector<std::vector<kbo::MsVal>> &out)
{
    out.assign(seqs.size(), {});
    if (seqs.empty()) return;
    BatchOnDevice &B = small_batch_buffers();
    static thread_local std::vector<uint64_t> off;
    static thread_local std::vector<uint8_t> concat, d;
    off.assign(seqs.size() + 1, 0);
    for (size_t s = 0; s < seqs.size(); s++) off[s + 1] = off[s] + seqs[s].size();
    concat.resize(off.back());
    for (size_t s = 0; s < seqs.size(); s++) std::memcpy(concat.data() + off[s], seqs[s].data(), seqs[s].size());
    d.resize(off.back() + 16);
    hipStream_t stream = nullptr; // (the thread's default stream: every call below is synchronous for this thread anyway)
    run_walk_host(idx, concat.data(), off.data(), seqs.size(), false, B, stream);
    HIP_OK(hipMemcpy(d.data(), B.ms.p, off.back(), hipMemcpyDeviceToHost));
    for (size_t s = 0; s < seqs.size(); s++) {
        out[s].resize(seqs[s].size());
        for (size_t i = 0; i < seqs[s].size(); i++) out[s][i] = kbo::MsVal{d[off[s] + i], 0u, 0u};
    }
}

// first pass on the device: sites of sequences [0, n_seqs), sorted by (sequence, i).  Normally the walk itself finds them
// (call mode of ms_walk_kernel: no intervals are written at all); a slab in which a lane had more than four breakpoints
// waiting at once, or whose site lists overflowed, is done again the long way (walk with intervals + call_sites_kernel).
std::vector<SiteRec> find_sites(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, uint32_t threshold)
{
    std::vector<SiteRec> all;
    hipStream_t stream = nullptr;
    const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, g_slab_bytes);
    DevBuf d_sites, d_count;
    const size_t count_bytes = kbo::kCallSegs * 64 + 64;
    d_count.alloc(count_bytes);
    std::vector<uint32_t> counts(count_bytes / 4);
    struct Raw { uint32_t a, b, c, d; };
    std::vector<Raw> raw;
    for (const Slab &sl : slabs) {
        const size_t ns = sl.s1 - sl.s0;
        std::vector<uint64_t> off(ns + 1);
        for (size_t s = 0; s <= ns; s++) off[s] = offsets[sl.s0 + s] - sl.b0;
        uint32_t cap = (uint32_t)std::min<uint64_t>(((sl.b1 - sl.b0) / 16 + 1024) / kbo::kCallSegs * kbo::kCallSegs + kbo::kCallSegs * 16, 0x7FFFFF00u);
        bool by_walk = true; // first the call mode of the walk, then (if it gave up) the stand-alone scan
        for (;;) {
            d_sites.ensure((size_t)cap * 16);
            HIP_OK(hipMemsetAsync(d_count.p, 0, count_bytes, stream));
            BatchOnDevice B;
            std::vector<kbo::WalkItem> items;
            if (by_walk) {
                const CallSink sink{d_sites.p, d_count.as<uint32_t>(), cap / kbo::kCallSegs, threshold};
                enqueue_walk_host(idx, concat + sl.b0, off.data(), ns, false, B, items, stream, 0, nullptr, nullptr, &sink);
            } else {
                run_walk_host(idx, concat + sl.b0, off.data(), ns, true, B, stream);
                HIP_OK(kbo::launch_call_sites(B.ms.as<uint8_t>(), B.lo.as<uint32_t>(), B.hi.as<uint32_t>(), B.off.as<uint64_t>(),
                                              (uint32_t)ns, sl.b1 - sl.b0, idx->host.k, threshold, d_sites.p, cap,
                                              d_count.as<uint32_t>(), stream));
            }
            HIP_OK(hipMemcpyAsync(counts.data(), d_count.p, count_bytes, hipMemcpyDeviceToHost, stream));
            HIP_OK(hipStreamSynchronize(stream));
            const uint32_t seg_cap = cap / kbo::kCallSegs;
            uint32_t worst = 0;
            for (uint32_t g = 0; g < kbo::kCallSegs; g++) worst = std::max(worst, counts[g * 16]);
            if (by_walk && counts[kbo::kCallSegs * 16]) { // a lane ran out of room for waiting breakpoints
                by_walk = false;
                continue;
            }
            if (worst > seg_cap) { // a list overflowed (dense mismatches): once more with room for the fullest
                cap = (uint32_t)std::min<uint64_t>((uint64_t)(worst + 16) * kbo::kCallSegs, 0x7FFFFF00u);
                continue;
            }
            for (uint32_t g = 0; g < kbo::kCallSegs; g++) {
                const uint32_t n = counts[g * 16];
                if (!n) continue;
                raw.resize(n);
                HIP_OK(hipMemcpy(raw.data(), d_sites.as<uint8_t>() + (size_t)g * seg_cap * 16, (size_t)n * 16, hipMemcpyDeviceToHost));
                for (const Raw &x : raw) {
                    if (by_walk && x.a == 0xFFFFFFFFu) continue; // (a site of an item that the redo pass scanned again)
                    if (by_walk) { // {slab offset of i, of j, row}: find the sequence
                        const size_t s = (size_t)(std::upper_bound(off.begin(), off.end(), (uint64_t)x.a) - off.begin()) - 1;
                        all.push_back(SiteRec{(uint32_t)(sl.s0 + s), (uint32_t)(x.a - off[s]), (uint32_t)(x.b - off[s]), x.c});
                    } else {
                        all.push_back(SiteRec{(uint32_t)(sl.s0 + x.a), x.b, x.c, x.d});
                    }
                }
            }
            break;
        }
    }
    std::sort(all.begin(), all.end(), [](const SiteRec &a, const SiteRec &b) { return a.seq != b.seq ? a.seq < b.seq : a.i < b.i; });
    return all;
}

} // namespace

void kbo_host::release_call_thread_caches() { small_batch_buffers().release(); }

extern "C" int kbo_call_batch(kbo_index_t *query_idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                              const kbo_call_opts *opts, kbo_variant **out, uint64_t *var_offsets)
{
    return guarded([&] {
        KBO_REQUIRE(query_idx && out && var_offsets, KBO_E_BAD_ARG, "null argument");
        *out = nullptr;
        check_batch(concat, offsets, n_seqs);
        kbo_call_opts o;
        if (opts) o = *opts; else kbo_call_opts_default(&o);
        KBO_REQUIRE(o.sbwt_build_opts.k == query_idx->host.k, KBO_E_K_MISMATCH,
                    "assert!(sbwt_ref.k() == sbwt_query.k()) (lib.rs:559)");
        const uint32_t k = query_idx->host.k;
        const size_t d = random_match_threshold(k, query_idx->host.n_kmers, 4, o.max_error_prob); // variant_calling.rs:260
        CallClock clk;
        // ---- first pass, on the device
        const std::vector<SiteRec> recs = find_sites(query_idx, concat, offsets, n_seqs, (uint32_t)d);
        clk.lap("first pass (sites)");
        // ---- k-mers of every site (host: access_kmer walks the index backwards, k steps per site)
        const size_t n_sites = recs.size();
        std::vector<kbo::CallSite> sites(n_sites);
        for (size_t x = 0; x < n_sites; x++) sites[x] = kbo::CallSite{recs[x].i, recs[x].j, recs[x].lo};
        std::vector<size_t> first(n_seqs + 1, 0); // sites of sequence s: [first[s], first[s+1])
        for (const SiteRec &r : recs) first[r.seq + 1]++;
        for (size_t s = 0; s < n_seqs; s++) first[s + 1] += first[s];
        std::vector<std::vector<uint8_t>> query_kmers(n_sites), ref_kmers(n_sites);
        {
            kbo::HostNav nav(query_idx->host);
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({16, std::thread::hardware_concurrency(), n_seqs}));
            std::vector<std::thread> th;
            std::exception_ptr err;
            std::mutex mu;
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t] {
                    try {
                        for (size_t s = t; s < n_seqs; s += nt) {
                            if (first[s] == first[s + 1]) continue;
                            std::vector<kbo::CallSite> mine(sites.begin() + first[s], sites.begin() + first[s + 1]);
                            std::vector<std::vector<uint8_t>> qk, rk;
                            kbo::call_site_kmers(nav, concat + offsets[s], k, mine, qk, rk);
                            for (size_t x = 0; x < mine.size(); x++) {
                                query_kmers[first[s] + x] = std::move(qk[x]);
                                ref_kmers[first[s] + x] = std::move(rk[x]);
                            }
                        }
                    } catch (...) {
                        std::lock_guard<std::mutex> g(mu);
                        if (!err) err = std::current_exception();
                    }
                });
            for (auto &x : th) x.join();
            if (err) std::rethrow_exception(err);
        }
        clk.lap("k-mers of the sites");
        // ---- second pass: all query-side k-mers against the index, one batch (variant_calling.rs:279)
        std::vector<std::vector<kbo::MsVal>> ms_vs_ref;
        ms_only(query_idx, query_kmers, ms_vs_ref);
        clk.lap("query k-mers vs the index");
        // ---- ... and per sequence its reference-side k-mers against its own index (lib.rs:553, variant_calling.rs:280)
        // (host threads: every sequence builds its own small index, uploads it, walks a handful of k-mers and frees it
        // again - a millisecond of mostly waiting per sequence, so the sequences are spread over a pool)
        std::vector<std::vector<kbo::Variant>> calls(n_seqs);
        {
            std::atomic<size_t> next{0};
            std::exception_ptr err;
            std::mutex mu;
            const int dev = current_device();
            auto work = [&] {
                try {
                    HIP_OK(hipSetDevice(dev));
                    for (;;) {
                        const size_t s = next.fetch_add(1);
                        if (s >= n_seqs) break;
                        const size_t a = first[s], b = first[s + 1];
                        if (a == b) continue;
                        kbo_index ref_idx;
                        ref_idx.transient = true; // no path cover for an index that serves one small batch
                        kbo::BuildParams p;
                        p.k = o.sbwt_build_opts.k;
                        p.add_revcomp = o.sbwt_build_opts.add_revcomp != 0;
                        p.num_threads = 1;
                        const uint8_t *seqs1[1] = {concat + offsets[s]};
                        const size_t lens1[1] = {(size_t)(offsets[s + 1] - offsets[s])};
                        kbo::build_host_index(seqs1, lens1, 1, p, ref_idx.host);
                        std::vector<std::vector<uint8_t>> rk(ref_kmers.begin() + a, ref_kmers.begin() + b);
                        std::vector<std::vector<kbo::MsVal>> ms_vs_query;
                        ms_only_small(&ref_idx, rk, ms_vs_query);
                        std::vector<kbo::CallSite> mine(sites.begin() + a, sites.begin() + b);
                        calls[s] = kbo::resolve_call_sites(mine, query_kmers.data() + a, ref_kmers.data() + a, ms_vs_ref.data() + a,
                                                           ms_vs_query.data(), d);
                    }
                } catch (...) {
                    std::lock_guard<std::mutex> g(mu);
                    if (!err) err = std::current_exception();
                    next.store(n_seqs); // (the others stop at their next sequence)
                }
            };
            const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::thread::hardware_concurrency(), n_seqs}));
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
            work();
            for (auto &x : th) x.join();
            if (err) std::rethrow_exception(err);
        }
        clk.lap("per-sequence indexes + walks");
        size_t n_var = 0;
        for (const auto &c : calls) n_var += c.size();
        // ---- one allocation: records, then the characters
        size_t chars = 0;
        for (const auto &c : calls)
            for (const auto &v : c) chars += v.query_chars.size() + v.ref_chars.size();
        const size_t head = std::max<size_t>(1, n_var) * sizeof(kbo_variant);
        uint8_t *mem = static_cast<uint8_t *>(std::malloc(head + chars + 1));
        if (!mem) throw std::bad_alloc();
        kbo_variant *rec = reinterpret_cast<kbo_variant *>(mem);
        uint8_t *cp = mem + head;
        size_t w = 0;
        var_offsets[0] = 0;
        for (size_t s = 0; s < n_seqs; s++) {
            for (const kbo::Variant &v : calls[s]) {
                rec[w].query_pos = v.query_pos;
                rec[w].query_chars = cp;
                rec[w].query_len = v.query_chars.size();
                std::memcpy(cp, v.query_chars.data(), v.query_chars.size());
                cp += v.query_chars.size();
                rec[w].ref_chars = cp;
                rec[w].ref_len = v.ref_chars.size();
                std::memcpy(cp, v.ref_chars.data(), v.ref_chars.size());
                cp += v.ref_chars.size();
                w++;
            }
            var_offsets[s + 1] = w;
        }
        *out = rec;
    });
}

extern "C" int kbo_call_sites_dev(const uint8_t *d_ms, const uint32_t *d_lo, const uint32_t *d_hi, const uint64_t *d_offsets,
                                  size_t n_seqs, uint64_t total_bases, size_t k, size_t threshold, void *d_sites, size_t capacity,
                                  uint32_t *d_count, void *stream)
{
    return guarded([&] {
        KBO_REQUIRE(d_ms && d_lo && d_hi && d_offsets && d_sites && d_count, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE(n_seqs < 0xFFFFFFFFull && capacity <= 0x7FFFFFFFull && k > 0 && k <= 255, KBO_E_BAD_ARG, "argument out of range");
        hipStream_t s = static_cast<hipStream_t>(stream);
        KBO_REQUIRE(capacity >= kbo::kCallSegs, KBO_E_BAD_ARG, "capacity below the number of lists");
        HIP_OK(hipMemsetAsync(d_count, 0, kbo::kCallSegs * 64, s));
        HIP_OK(kbo::launch_call_sites(d_ms, d_lo, d_hi, d_offsets, (uint32_t)n_seqs, total_bases, (uint32_t)k, (uint32_t)threshold,
                                      d_sites, (uint32_t)capacity, d_count, s));
    });
}
