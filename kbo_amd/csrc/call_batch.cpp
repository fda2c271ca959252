// call_batch.cpp — kbo::call (reference lib.rs:547-573, variant_calling.rs:249-294) over a batch of sequences.
//
// The first pass of call_variants - the MS walk of every sequence against the index and the breakpoint scan over it -
// runs on the device for the whole batch (the call mode of the walk kernels); only the sites {sequence, i, j, row} come
// back, about one record per mismatch instead of 9 bytes per base - and with them, gathered by one more kernel while the
// MS values are still resident, what the second pass needs of the device: the k MS values in front of every match (the
// walk of the query-side k-mer, variant_calling.rs:279, is a function of them) and the k characters of the matched row
// (access_kmer, :276, read off the path cover).  What is left for the host is the walk of that row's k-mer against the
// index of the sequence itself (:280; the reference builds one per call, lib.rs:553) - answered by a suffix automaton of
// the sequence instead of an SBWT of it, see RunAutomaton - and resolve_variant.
#include "capi_internal.hpp"

#include <algorithm>
#include <condition_variable>
#include <deque>
#include <functional>
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

// What the second pass of a batch needs per site, produced on the device right behind the first pass (call_kernels.hip
// call_finalize_kernel): the site as {sequence, i, j, row}, the k MS values ending at the match and the k characters of the
// matched row.  One part per slab, in pinned memory (the downloads land there; nothing is copied again).
struct SiteWindows {
    struct Part {
        PinBuf recs, win; // n records of 16 / stride bytes
        PinBuf codes;     // the device's second pass: n words (call_second_kernels.hip), when has_codes
        bool has_codes = false;
        size_t n = 0;
    };
    std::vector<std::unique_ptr<Part>> parts;
    uint32_t stride = 0, kpad = 0;
    const SiteRec &rec(uint32_t part, uint32_t x) const { return parts[part]->recs.as<SiteRec>()[x]; }
    // the site's word of the device's second pass, or "left to the host"
    uint32_t code(uint32_t part, uint32_t x) const { return parts[part]->has_codes ? parts[part]->codes.as<uint32_t>()[x] : 0x01FFFFFFu; }
    const uint8_t *win(uint32_t part, uint32_t x) const { return parts[part]->win.as<uint8_t>() + (size_t)x * stride; }
};

// host threads that take tasks while the calling thread drives the device (kbo_call_batch: the sites of slab i are resolved while
// the device walks slab i + 1); wait() returns when nothing is queued or running and rethrows the first exception a task threw
class AsyncPool {
public:
    explicit AsyncPool(unsigned n)
    {
        for (unsigned t = 0; t < n; t++) threads_.emplace_back([this] { loop(); });
    }
    ~AsyncPool()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    void submit(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            queue_.push_back(std::move(f));
            pending_++;
        }
        cv_.notify_one();
    }
    void wait() // (the caller helps: a pool of 0 threads still works)
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(mu_);
                if (queue_.empty()) {
                    done_cv_.wait(g, [&] { return pending_ == 0 || !queue_.empty(); });
                    if (pending_ == 0) break;
                    continue;
                }
                f = std::move(queue_.front());
                queue_.pop_front();
            }
            run(f);
        }
        std::exception_ptr e;
        {
            std::lock_guard<std::mutex> g(mu_);
            std::swap(e, err_);
        }
        if (e) std::rethrow_exception(e);
    }
    bool failed()
    {
        std::lock_guard<std::mutex> g(mu_);
        return (bool)err_;
    }

private:
    void run(std::function<void()> &f)
    {
        try {
            f();
        } catch (...) {
            std::lock_guard<std::mutex> g(mu_);
            if (!err_) err_ = std::current_exception();
        }
        std::lock_guard<std::mutex> g(mu_);
        if (--pending_ == 0 || !queue_.empty()) done_cv_.notify_all();
    }
    void loop()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || !queue_.empty(); });
                if (queue_.empty()) return; // (stop_)
                f = std::move(queue_.front());
                queue_.pop_front();
            }
            run(f);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<std::function<void()>> queue_;
    std::vector<std::thread> threads_;
    size_t pending_ = 0;
    bool stop_ = false;
    std::exception_ptr err_;
};

// first pass on the device: sites of sequences [0, n_seqs) with their windows.  Normally the walk itself finds them
// (call mode of ms_walk_kernel: no intervals are written at all); a slab in which a lane had more than four breakpoints
// waiting at once, or whose site lists overflowed, is done again the long way (walk with intervals + call_sites_kernel).
// Sites of slab-relative sequence numbers are shifted to batch-wide ones by the caller (SiteRec::seq + Part's first).
// second_q != 0: the second pass's three values per site as well (q-mers of that many bases index the slab's sequences)
// on_part(part, its index, its first sequence, its sequences): called once the part's records are on the host
// (`all` is the caller's: the parts have to outlive whatever on_part started, also when this function throws)
void find_sites(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, uint32_t threshold,
                std::vector<size_t> &part_first_seq, uint32_t second_q, bool revcomp,
                const std::function<void(SiteWindows::Part *, uint32_t, size_t, size_t)> &on_part, SiteWindows &all)
{
    const uint32_t k = idx->host.k;
    all.stride = kbo::call_gather_stride(k);
    all.kpad = (k + 15u) / 16u * 16u;
    hipStream_t stream = nullptr;
    const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, slab_bytes_for(idx));
    DevBuf d_sites, d_count, d_prefix, d_recs, d_win, d_tab, d_tab_off, d_seq_flag, d_codes;
    std::vector<uint64_t> tab_off;
    const size_t count_bytes = kbo::kCallSegs * 64 + 64;
    d_count.alloc(count_bytes);
    d_prefix.alloc((kbo::kCallSegs + 1) * 4);
    std::vector<uint32_t> counts(count_bytes / 4), prefix(kbo::kCallSegs + 1);
    const kbo::DevIndexView view = device_view(idx, current_device(), nullptr, offsets[n_seqs]);
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
            // the sites of this slab, numbered list by list, and their windows - while its MS values are on the device
            prefix[0] = 0;
            for (uint32_t g = 0; g < kbo::kCallSegs; g++) prefix[g + 1] = prefix[g] + counts[g * 16];
            const size_t n_new = prefix[kbo::kCallSegs];
            std::unique_ptr<SiteWindows::Part> part(new SiteWindows::Part());
            part->n = n_new;
            if (n_new) {
                d_recs.ensure(n_new * 16);
                d_win.ensure(n_new * (size_t)all.stride);
                part->recs.ensure(n_new * 16);
                part->win.ensure(n_new * (size_t)all.stride);
                HIP_OK(hipMemcpyAsync(d_prefix.p, prefix.data(), prefix.size() * 4, hipMemcpyHostToDevice, stream));
                HIP_OK(kbo::launch_call_finalize(d_sites.p, d_count.as<uint32_t>(), d_prefix.as<uint32_t>(), seg_cap, worst, by_walk,
                                                 B.off.as<uint64_t>(), (uint32_t)ns, k, B.ms.as<uint8_t>(), view, d_recs.p,
                                                 d_win.as<uint8_t>(), all.stride, stream));
                if (second_q) {
                    // the second pass for these sites while the slab's bases are on the device: a table of q-mer start positions per
                    // sequence (a power of two of at least 1.5 slots per base; none for a sequence beyond 20-bit positions)
                    tab_off.assign(ns + 1, 0);
                    uint32_t max_slots = 0;
                    for (size_t s = 0; s < ns; s++) {
                        const uint64_t len = off[s + 1] - off[s];
                        uint64_t size = 0;
                        if (len > 0 && len < (1u << 20) - 1u) {
                            size = 64;
                            while (size < len + len / 2) size <<= 1;
                        }
                        tab_off[s + 1] = tab_off[s] + size;
                        max_slots = (uint32_t)std::max<uint64_t>(max_slots, size);
                    }
                    d_tab_off.ensure((ns + 1) * 8);
                    d_tab.ensure(tab_off[ns] * 4 + 16);
                    d_seq_flag.ensure(ns + 16);
                    d_codes.ensure(n_new * 4);
                    part->codes.ensure(n_new * 4);
                    HIP_OK(hipMemcpyAsync(d_tab_off.p, tab_off.data(), (ns + 1) * 8, hipMemcpyHostToDevice, stream));
                    HIP_OK(kbo::launch_call_qmer_index(B.q.as<uint8_t>(), B.off.as<uint64_t>(), (uint32_t)ns, second_q, d_tab_off.as<uint64_t>(),
                                                       d_tab.as<uint32_t>(), d_seq_flag.as<uint8_t>(), tab_off[ns], max_slots, stream));
                    HIP_OK(kbo::launch_call_depths(d_recs.p, d_win.as<uint8_t>(), all.stride, (uint32_t)n_new, B.q.as<uint8_t>(), B.off.as<uint64_t>(), k,
                                                   threshold, second_q, revcomp, d_tab_off.as<uint64_t>(), d_tab.as<uint32_t>(),
                                                   d_seq_flag.as<uint8_t>(), d_codes.as<uint32_t>(), stream));
                    HIP_OK(hipMemcpyAsync(part->codes.p, d_codes.p, n_new * 4, hipMemcpyDeviceToHost, stream));
                    part->has_codes = true;
                }
                HIP_OK(hipMemcpyAsync(part->recs.p, d_recs.p, n_new * 16, hipMemcpyDeviceToHost, stream));
                HIP_OK(hipMemcpyAsync(part->win.p, d_win.p, n_new * (size_t)all.stride, hipMemcpyDeviceToHost, stream));
                HIP_OK(hipStreamSynchronize(stream));
            }
            all.parts.push_back(std::move(part));
            part_first_seq.push_back(sl.s0);
            if (on_part) on_part(all.parts.back().get(), (uint32_t)(all.parts.size() - 1), sl.s0, ns);
            break;
        }
    }
}

// ---- the reference-side walk of a site (variant_calling.rs:280: the matched row's k-mer against the index of the sequence
// itself, which the reference builds per call, lib.rs:553) without building that index.
// What resolve_variant reads of that walk are the depths only, and the depth at position t is the length of the longest
// suffix of kmer[0 ..= t] (at most k) that is a suffix of some row of the sequence's SBWT.  The ACGT suffixes of that index's
// rows are exactly the substrings (of at most k characters) of the sequence's ACGT-runs of at least k characters: a row is
// a k-mer of such a run or a $-padded prefix of one, and every substring of a run ends some k-mer or padded prefix of it
// (shorter runs contribute no rows; index.rs:73-94 / SURVEY.md section 8(a) A0).  With add_revcomp the reverse complements
// of those runs count as well.  So the depths are plain matching statistics of the k-mer against those runs: a suffix
// automaton of the runs (linear in the sequence, a few hundred KB for a 10 kbp read) answers them in k steps per site,
// where building, uploading and walking a one-sequence SBWT cost 2.8 ms of host time per sequence.
// tests: kbo_call_batch == the oracle's literal kbo::call (which builds that index) per sequence, and the reference's goldens.
class RunAutomaton {
public:
    void build(const uint8_t *seq, size_t len, uint32_t k, bool add_revcomp)
    {
        // (state numbers are 32-bit, two states per character and strand: 96 bytes per base with reverse complements - a
        // sequence this long is a genome, and kbo_call with it as ref_seq walks a real index of it: refine.cpp)
        KBO_REQUIRE((add_revcomp ? 2 : 1) * (uint64_t)len < (1ull << 29), KBO_E_UNSUPPORTED,
                    "kbo_call_batch: a sequence of 2^29 bases or more (2^28 with add_revcomp); call kbo_call for it");
        st_.clear();
        st_.reserve(2 * (add_revcomp ? 2 * len : len) + 2); // (at most two states per character)
        new_state(0, -1);
        size_t run = 0;
        for (size_t i = 0; i <= len; i++) {
            const int c = i < len ? code(seq[i]) : -1;
            if (c >= 0) { run++; continue; }
            if (run >= k) {
                last_ = 0;
                for (size_t t = i - run; t < i; t++) extend(code(seq[t]));
                if (add_revcomp) {
                    last_ = 0;
                    for (size_t t = i; t-- > i - run;) extend(3 - code(seq[t]));
                }
            }
            run = 0;
        }
    }
    // depths of the walk of `kmer` (k characters, '$' and other non-ACGT bytes reset it) -> out[0 .. k)
    void depths(const uint8_t *kmer, uint32_t k, uint32_t *out) const
    {
        int32_t v = 0;
        uint32_t l = 0;
        for (uint32_t t = 0; t < k; t++) {
            const int c = code(kmer[t]);
            if (c < 0) { v = 0; l = 0; out[t] = 0; continue; }
            while (v != 0 && st_[v].next[c] < 0) { v = st_[v].link; l = (uint32_t)st_[v].len; }
            if (st_[v].next[c] >= 0) { v = st_[v].next[c]; l++; }
            else { v = 0; l = 0; }
            out[t] = l < k ? l : k;
        }
    }

private:
    struct State { // 24 bytes: a state's transitions, suffix link and length share a cache line
        int32_t next[4], link, len;
    };
    static int code(uint8_t ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1; }
    int32_t new_state(int32_t length, int32_t link)
    {
        st_.push_back(State{{-1, -1, -1, -1}, link, length});
        return (int32_t)st_.size() - 1;
    }
    int32_t clone_of(int32_t q, int32_t length)
    {
        State c = st_[q];
        c.len = length;
        st_.push_back(c);
        return (int32_t)st_.size() - 1;
    }
    void extend(int c) // (generalised: `last_` may be the root again at the start of every run)
    {
        int32_t p = last_;
        if (st_[p].next[c] >= 0) {
            const int32_t q = st_[p].next[c];
            if (st_[p].len + 1 == st_[q].len) { last_ = q; return; }
            const int32_t cl = clone_of(q, st_[p].len + 1);
            while (p >= 0 && st_[p].next[c] == q) { st_[p].next[c] = cl; p = st_[p].link; }
            st_[q].link = cl;
            last_ = cl;
            return;
        }
        const int32_t cur = new_state(st_[p].len + 1, 0);
        while (p >= 0 && st_[p].next[c] < 0) { st_[p].next[c] = cur; p = st_[p].link; }
        if (p >= 0) {
            const int32_t q = st_[p].next[c];
            if (st_[p].len + 1 == st_[q].len) st_[cur].link = q;
            else {
                const int32_t cl = clone_of(q, st_[p].len + 1);
                while (p >= 0 && st_[p].next[c] == q) { st_[p].next[c] = cl; p = st_[p].link; }
                st_[q].link = cl;
                st_[cur].link = cl;
            }
        }
        last_ = cur;
    }
    std::vector<State> st_;
    int32_t last_ = 0;
};

} // namespace

namespace {

// The variants of a batch as they leave the library's inside: flat arrays in the order of (sequence, query position) - what
// kbo_call_batch_flat hands out as it is and kbo_call_batch turns into the reference's records (variant_calling.rs:8-26).
template <typename T> struct GrowBuf { // a vector whose storage can be handed to the caller (malloc / realloc; kbo_call_flat_free)
    T *p = nullptr;
    size_t n = 0, cap = 0;
    GrowBuf() = default;
    GrowBuf(const GrowBuf &) = delete;
    GrowBuf &operator=(const GrowBuf &) = delete;
    ~GrowBuf() { std::free(p); }
    size_t size() const { return n; }
    T *data() { return p; }
    const T *data() const { return p; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    void reserve(size_t want)
    {
        if (want <= cap) return;
        const size_t c = std::max<size_t>({want, cap + cap / 2, 1024});
        T *q = static_cast<T *>(std::realloc(p, c * sizeof(T)));
        if (!q) throw std::bad_alloc();
        p = q;
        cap = c;
    }
    void resize(size_t m) { reserve(m); n = m; }
    void push_back(const T &v) { reserve(n + 1); p[n++] = v; }
    void append(const T *a, const T *b) { reserve(n + (size_t)(b - a)); std::memcpy(p + n, a, (size_t)(b - a) * sizeof(T)); n += (size_t)(b - a); }
    T *release() { T *q = p; p = nullptr; n = cap = 0; return q; }
};
struct FlatCalls {
    GrowBuf<uint32_t> pos;
    GrowBuf<uint16_t> qlen, rlen;
    GrowBuf<uint8_t> chars; // per variant its query characters, then its reference characters
    void push(uint32_t at, uint32_t ql, uint32_t rl)
    {
        pos.push_back(at);
        qlen.push_back((uint16_t)ql);
        rlen.push_back((uint16_t)rl);
    }
};

// ---- the route of rounds 3 - 5, now the fall-back for a slab the device cannot finish by itself (a lane of the walk's call mode with
// more than four breakpoints waiting, a site list that overflowed, more sites left to the host than its list holds) and for
// thresholds the device's second pass does not take: the first pass on the device, one 16-byte record + a window per SITE to the
// host, host threads sort, resolve and slice.  Appends the variants of sequences [0, n_seqs) to `out`, var_count[s] = those of s.
void call_slow(kbo_index *query_idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, const kbo_call_opts &o, size_t d,
               uint32_t second_q, FlatCalls &out, uint64_t *var_count)
{
    const uint32_t k = query_idx->host.k;
    std::vector<size_t> part_seq0;
    // ---- second pass, per sequence, on host threads that take a slab's sites as soon as its records are on the host - while the
    // device walks the next slabs.  (The per-sequence index of lib.rs:553 is never built: its build depends on k and add_revcomp
    // only, both checked by the caller, so it cannot fail for one sequence and not for another; a sequence without sites yields no
    // variants either way.)
    struct Ref { uint32_t part, x; };
    struct Call { uint32_t i; uint16_t q_from, q_len, r_from, r_len; Ref site; }; // characters: slices of the two k-mers
    std::vector<std::vector<Call>> calls(n_seqs);
    SiteWindows sw; // (declared in front of the pool: its parts outlive the tasks that read them)
    const kbo::HostNav nav(query_idx->host);
    const bool revcomp = o.sbwt_build_opts.add_revcomp != 0;
    const uint32_t stride = kbo::call_gather_stride(k), kpad = (k + 15u) / 16u * 16u;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)hw, (n_seqs + 15) / 16}));
    AsyncPool pool(nt - 1u); // (the calling thread joins in wait())
    struct PartOrder { // a part's sites by sequence: sequence s (of the part) has xs[first[s] .. first[s + 1])
        std::vector<uint32_t> first, xs;
    };
    // the sequences [a, b) of a part
    auto resolve_block = [&, k, d, stride, kpad, revcomp](const SiteWindows::Part *part, uint32_t part_index, size_t seq0,
                                                              std::shared_ptr<const PartOrder> po, size_t a, size_t b) {
        const SiteRec *recs = part->recs.as<SiteRec>();
        const uint32_t *codes = part->has_codes ? part->codes.as<uint32_t>() : nullptr;
        RunAutomaton sam;
        std::vector<uint32_t> dq(k), dr(k), mine;
        std::vector<uint8_t> qk(k), rk_spelled;
        for (size_t ls = a; ls < b; ls++) {
            const size_t fa = po->first[ls], fb = po->first[ls + 1];
            if (fa == fb) continue;
            const size_t s = seq0 + ls;
            const uint8_t *seq = concat + offsets[s];
            const size_t len = (size_t)(offsets[s + 1] - offsets[s]);
            mine.assign(po->xs.begin() + fa, po->xs.begin() + fb);
            std::sort(mine.begin(), mine.end(), [&](uint32_t x, uint32_t y) { return recs[x].i < recs[y].i; });
            std::vector<Call> &out_calls = calls[s];
            out_calls.reserve(mine.size());
            bool sam_built = false;
            for (const uint32_t x : mine) {
                const SiteRec &r = recs[x];
                const Ref sr{part_index, x};
                const uint32_t code = codes ? codes[x] : 0x01FFFFFFu;
                if (!(code >> 24)) { // the device did this site: the common suffix and the two peaks are all resolve_variant reads
                    const uint32_t rp = code & 0xFFu, qp = (code >> 8) & 0xFFu, csl = (code >> 16) & 0xFFu;
                    size_t qf, qt, rf, rt;
                    if (kbo::resolve_variant_peaks(k, csl, qp != 0xFFu, qp, rp != 0xFFu, rp, qf, qt, rf, rt))
                        out_calls.push_back(Call{r.i, (uint16_t)qf, (uint16_t)(qt - qf), (uint16_t)rf, (uint16_t)(rt - rf), sr});
                    continue;
                }
                if (!sam_built) { // (a site left to the host: the sequence's suffix automaton, once)
                    sam.build(seq, len, k, revcomp);
                    sam_built = true;
                }
                const uint8_t *w = part->win.as<uint8_t>() + (size_t)x * stride;
                // query-side k-mer (variant_calling.rs:46-58, 275) and its walk against the index (:279): the MS
                // values of the first pass, capped by the distance from the k-mer's (or the sequence's) first base
                for (uint32_t t = 0; t < k; t++) {
                    const int64_t pos = (int64_t)r.j - (int64_t)(k - 1u) + t;
                    if (pos < 0) { qk[t] = '$'; dr[t] = 0; continue; } // '$': the walk restarts behind it
                    qk[t] = seq[pos];
                    dr[t] = std::min<uint32_t>(w[t], (uint32_t)std::min<int64_t>(t + 1u, pos + 1));
                }
                // matched row's k-mer (:276): from the device's path cover, or spelled here when its window crosses a path start
                const uint8_t *rk = w + kpad;
                if (w[2u * kpad]) {
                    nav.access_kmer(r.lo, rk_spelled);
                    rk = rk_spelled.data();
                }
                sam.depths(rk, k, dq.data()); // its walk against the sequence's own index (:280)
                size_t qf, qt, rf, rt;
                if (kbo::resolve_variant_ranges(qk.data(), rk, dq.data(), dr.data(), k, d, qf, qt, rf, rt)) // :282-284
                    out_calls.push_back(Call{r.i, (uint16_t)qf, (uint16_t)(qt - qf), (uint16_t)rf, (uint16_t)(rt - rf), sr});
            }
        }
    };
    // a part: its sites by sequence (counting sort; void records - sites of items the redo pass scanned again - dropped), then
    // blocks of its sequences as tasks of their own
    auto on_part = [&](SiteWindows::Part *part, uint32_t part_index, size_t seq0, size_t ns) {
        if (part->n == 0 || pool.failed()) return;
        pool.submit([&, part, part_index, seq0, ns] {
            const SiteRec *recs = part->recs.as<SiteRec>();
            auto po = std::make_shared<PartOrder>();
            po->first.assign(ns + 1, 0);
            size_t valid = 0;
            for (size_t x = 0; x < part->n; x++)
                if (recs[x].seq != 0xFFFFFFFFu) { po->first[recs[x].seq + 1]++; valid++; }
            for (size_t ls = 0; ls < ns; ls++) po->first[ls + 1] += po->first[ls];
            po->xs.resize(valid);
            {
                std::vector<uint32_t> fill(po->first.begin(), po->first.end() - 1);
                for (size_t x = 0; x < part->n; x++)
                    if (recs[x].seq != 0xFFFFFFFFu) po->xs[fill[recs[x].seq]++] = (uint32_t)x;
            }
            const size_t block = 128;
            std::shared_ptr<const PartOrder> cpo = po;
            for (size_t a = 0; a < ns; a += block) {
                const size_t b = std::min(ns, a + block);
                if (po->first[a] == po->first[b]) continue;
                pool.submit([&, part, part_index, seq0, cpo, a, b] { resolve_block(part, part_index, seq0, cpo, a, b); });
            }
        });
    };
    find_sites(query_idx, concat, offsets, n_seqs, (uint32_t)d, part_seq0, second_q, revcomp, on_part, sw);
    pool.wait();
    // ---- the variants, every sequence's at its own place behind what `out` holds already
    std::vector<size_t> v0(n_seqs + 1, 0), c0(n_seqs + 1, 0);
    for (size_t s = 0; s < n_seqs; s++) {
        size_t ch = 0;
        for (const Call &c : calls[s]) ch += c.q_len + c.r_len;
        var_count[s] = calls[s].size();
        v0[s + 1] = v0[s] + calls[s].size();
        c0[s + 1] = c0[s] + ch;
    }
    const size_t vb = out.pos.size(), cb = out.chars.size();
    out.pos.resize(vb + v0[n_seqs]);
    out.qlen.resize(vb + v0[n_seqs]);
    out.rlen.resize(vb + v0[n_seqs]);
    out.chars.resize(cb + c0[n_seqs]);
    const size_t piece = 256;
    HostTeam::get().run((n_seqs + piece - 1) / piece, [&](size_t task) {
        std::vector<uint8_t> rk_spelled;
        for (size_t s = task * piece; s < std::min(n_seqs, (task + 1) * piece); s++) {
            size_t w = vb + v0[s];
            uint8_t *cp = out.chars.data() + cb + c0[s];
            const uint8_t *seq = concat + offsets[s];
            for (const Call &c : calls[s]) {
                const SiteRec &r = sw.rec(c.site.part, c.site.x);
                const uint8_t *win = sw.win(c.site.part, c.site.x);
                out.pos[w] = c.i;
                out.qlen[w] = c.q_len;
                out.rlen[w] = c.r_len;
                for (uint32_t t = 0; t < c.q_len; t++) { // (a slice of the query-side k-mer: '$' in front of the sequence)
                    const int64_t pos = (int64_t)r.j - (int64_t)(k - 1u) + c.q_from + t;
                    *cp++ = pos < 0 ? (uint8_t)'$' : seq[pos];
                }
                const uint8_t *rk = win + sw.kpad;
                if (win[2u * sw.kpad] && c.r_len) {
                    nav.access_kmer(r.lo, rk_spelled);
                    rk = rk_spelled.data();
                }
                std::memcpy(cp, rk + c.r_from, c.r_len);
                cp += c.r_len;
                w++;
            }
        }
    });
}

std::atomic<int> g_call_device_emit{1}; // kbo_set_call_device_emit: 0 = the slow route for every slab, 2 = call_depths_kernel for k <= 64 too

// ---- kbo::call over a batch, the device's way: two slots of buffers on two streams take the slabs in turn; per slab ONE small read-back
// (sixteen words: how many sites, variants, characters, sites for the host; did a list overflow) decides what is downloaded - the
// variants themselves, in their final order and form (call_emit_kernels.hip).  While the host waits for a slab's words the other
// slot's kernels run; the uploads are staged through pinned memory by the host team.
struct CallSlot {
    hipStream_t st = nullptr;
    hipEvent_t meta_ev = nullptr, done_ev = nullptr;
    PinBuf in, off, tab_off, meta, o_pos, o_lens, o_chars, o_vfirst, o_hrecs, o_hwin;
    BatchOnDevice B;
    std::vector<kbo::WalkItem> items;
    DevBuf d_sites, d_count, d_prefix, d_meta, d_recs, d_win, d_tab, d_tab_off, d_seq_flag, d_codes;
    DevBuf d_seq, d_site, d_scan, d_pos, d_lens, d_chars, d_vfirst, d_hlist, d_hrecs, d_hwin;
    const Slab *slab = nullptr;
    size_t ns = 0;
    uint32_t cap = 0, chars_cap = 0, host_cap = 0;
    int state = 0; // 0 free, 1 enqueued (meta on its way), 2 downloads on their way, 3 left to the slow route
    uint32_t n_var = 0, n_chars = 0, n_host = 0;
    void make()
    {
        if (st) return;
        HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&meta_ev, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&done_ev, hipEventDisableTiming));
    }
    ~CallSlot()
    {
        if (st) {
            (void)hipStreamSynchronize(st);
            (void)hipEventDestroy(meta_ev);
            (void)hipEventDestroy(done_ev);
            (void)hipStreamDestroy(st);
        }
    }
};

void call_fast(kbo_index *query_idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, const kbo_call_opts &o, size_t d,
               uint32_t second_q, FlatCalls &out, uint64_t *var_offsets, CallClock &clk)
{
    const uint32_t k = query_idx->host.k;
    const bool revcomp = o.sbwt_build_opts.add_revcomp != 0;
    const uint32_t stride = kbo::call_gather_stride(k), kpad = (k + 15u) / 16u * 16u;
    const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, slab_bytes_for(query_idx));
    const kbo::DevIndexView view = device_view(query_idx, current_device(), nullptr, offsets[n_seqs]);
    const size_t count_bytes = kbo::kCallSegs * 64 + 64;
    const kbo::HostNav nav(query_idx->host);
    HostTeam &team = HostTeam::get();
    CallSlot slots[2];
    double t_stage = 0, t_enq = 0, t_wait = 0, t_take = 0, t_slow = 0;
    size_t n_slow = 0, n_host_sites = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(now() - t0).count(); };

    auto enqueue = [&](CallSlot &S, const Slab &sl) {
        auto t0 = now();
        S.make();
        S.slab = &sl;
        const size_t ns = S.ns = sl.s1 - sl.s0;
        const uint64_t bytes = sl.b1 - sl.b0;
        S.off.ensure((ns + 1) * 8);
        S.tab_off.ensure((ns + 1) * 8);
        uint64_t *off = S.off.as<uint64_t>(), *tab_off = S.tab_off.as<uint64_t>();
        uint32_t max_slots = 0;
        tab_off[0] = 0;
        for (size_t s = 0; s <= ns; s++) off[s] = offsets[sl.s0 + s] - sl.b0;
        // a table of q-mer start positions per sequence (a power of two of at least 1.5 slots per base; none for a sequence beyond
        // 20-bit positions: its sites are the host's)
        for (size_t s = 0; s < ns; s++) {
            const uint64_t len = off[s + 1] - off[s];
            uint64_t size = 0;
            if (len > 0 && len < (1u << 20) - 1u) {
                size = 64;
                while (size < len + len / 2) size <<= 1;
            }
            tab_off[s + 1] = tab_off[s] + size;
            max_slots = (uint32_t)std::max<uint64_t>(max_slots, size);
        }
        S.in.ensure(bytes + 16);
        team.copy(S.in.p, concat + sl.b0, bytes);
        t_stage += since(t0);
        t0 = now();
        const uint32_t cap = S.cap = (uint32_t)std::min<uint64_t>((bytes / 16 + 1024) / kbo::kCallSegs * kbo::kCallSegs + kbo::kCallSegs * 16, 0x7FFFFF00u);
        const uint32_t seg_cap = cap / kbo::kCallSegs;
        S.chars_cap = cap * 4u + 4096u;
        S.host_cap = std::max<uint32_t>(4096u, cap / 16u);
        S.d_sites.ensure((size_t)cap * 16);
        S.d_count.ensure(count_bytes);
        S.d_prefix.ensure((kbo::kCallSegs + 1) * 4);
        S.d_meta.ensure(kbo::kCallMetaWords * 4);
        S.meta.ensure(kbo::kCallMetaWords * 4);
        S.d_recs.ensure((size_t)cap * 16);
        S.d_win.ensure((size_t)cap * stride);
        S.d_codes.ensure((size_t)cap * 4);
        S.d_tab_off.ensure((ns + 1) * 8);
        S.d_tab.ensure(tab_off[ns] * 4 + 16);
        S.d_seq_flag.ensure(ns + 16);
        const size_t seq_words = (ns + 1) + kbo::call_scan_sums_words(ns + 1) + ns + 4;
        const size_t scan_words = (size_t)cap + 1 + kbo::call_scan_sums_words((size_t)cap + 1);
        S.d_seq.ensure(seq_words * 4);
        S.d_site.ensure((size_t)cap * 16);
        S.d_scan.ensure(2 * scan_words * 4);
        S.d_pos.ensure((size_t)cap * 4);
        S.d_lens.ensure((size_t)cap * 4);
        S.d_chars.ensure((size_t)S.chars_cap + 16);
        S.d_vfirst.ensure((ns + 1) * 4);
        S.d_hlist.ensure((size_t)S.host_cap * 4);
        S.d_hrecs.ensure((size_t)S.host_cap * 16);
        S.d_hwin.ensure((size_t)S.host_cap * stride);
        hipStream_t st = S.st;
        HIP_OK(hipMemsetAsync(S.d_count.p, 0, count_bytes, st));
        HIP_OK(hipMemsetAsync(S.d_meta.p, 0, kbo::kCallMetaWords * 4, st));
        const CallSink sink{S.d_sites.p, S.d_count.as<uint32_t>(), seg_cap, (uint32_t)d};
        enqueue_walk_host(query_idx, S.in.as<uint8_t>(), off, ns, false, S.B, S.items, st, 0, nullptr, nullptr, &sink);
        HIP_OK(kbo::launch_call_prefix(S.d_count.as<uint32_t>(), seg_cap, S.d_prefix.as<uint32_t>(), S.d_meta.as<uint32_t>(), st));
        HIP_OK(kbo::launch_call_finalize(S.d_sites.p, S.d_count.as<uint32_t>(), S.d_prefix.as<uint32_t>(), seg_cap, seg_cap, true,
                                         S.B.off.as<uint64_t>(), (uint32_t)ns, k, S.B.ms.as<uint8_t>(), view, S.d_recs.p, S.d_win.as<uint8_t>(), stride, st));
        HIP_OK(hipMemcpyAsync(S.d_tab_off.p, tab_off, (ns + 1) * 8, hipMemcpyHostToDevice, st));
        HIP_OK(kbo::launch_call_qmer_index(S.B.q.as<uint8_t>(), S.B.off.as<uint64_t>(), (uint32_t)ns, second_q, S.d_tab_off.as<uint64_t>(),
                                           S.d_tab.as<uint32_t>(), S.d_seq_flag.as<uint8_t>(), tab_off[ns], max_slots, st));
        const uint32_t *d_n = S.d_prefix.as<uint32_t>() + kbo::kCallSegs;
        HIP_OK(kbo::launch_call_depths(S.d_recs.p, S.d_win.as<uint8_t>(), stride, cap, S.B.q.as<uint8_t>(), S.B.off.as<uint64_t>(), k, (uint32_t)d,
                                       second_q, revcomp, S.d_tab_off.as<uint64_t>(), S.d_tab.as<uint32_t>(), S.d_seq_flag.as<uint8_t>(),
                                       S.d_codes.as<uint32_t>(), st, d_n, g_call_device_emit.load() == 2));
        kbo::CallEmitArgs a{};
        a.recs = S.d_recs.as<uint4>();
        a.codes = S.d_codes.as<uint32_t>();
        a.win = S.d_win.as<uint8_t>();
        a.stride = stride;
        a.kpad = kpad;
        a.k = k;
        a.q = S.B.q.as<uint8_t>();
        a.off = S.B.off.as<uint64_t>();
        a.n_seqs = (uint32_t)ns;
        a.n_sites = d_n;
        a.cap = cap;
        a.seq_cnt = S.d_seq.as<uint32_t>();
        a.seq_sums = a.seq_cnt + (ns + 1);
        a.seq_fill = a.seq_sums + kbo::call_scan_sums_words(ns + 1);
        a.bucket = S.d_site.as<uint32_t>();
        a.bkey = a.bucket + cap;
        a.sorted = a.bkey + cap;
        a.vrec = a.sorted + cap;
        a.vcnt = S.d_scan.as<uint32_t>();
        a.vsums = a.vcnt + ((size_t)cap + 1);
        a.ccnt = a.vcnt + scan_words;
        a.csums = a.ccnt + ((size_t)cap + 1);
        a.out_pos = S.d_pos.as<uint32_t>();
        a.out_lens = S.d_lens.as<uint32_t>();
        a.out_chars = S.d_chars.as<uint8_t>();
        a.chars_cap = S.chars_cap;
        a.seq_vfirst = S.d_vfirst.as<uint32_t>();
        a.host_list = S.d_hlist.as<uint32_t>();
        a.host_cap = S.host_cap;
        a.meta = S.d_meta.as<uint32_t>();
        HIP_OK(kbo::launch_call_emit(a, S.d_hrecs.p, S.d_hwin.as<uint8_t>(), st));
        HIP_OK(hipMemcpyAsync(S.meta.p, S.d_meta.p, kbo::kCallMetaWords * 4, hipMemcpyDeviceToHost, st));
        HIP_OK(hipEventRecord(S.meta_ev, st));
        S.state = 1;
        t_enq += since(t0);
    };

    // the slab's words are there: what to download (or that the slab is the slow route's)
    auto downloads = [&](CallSlot &S) {
        if (S.state != 1) return;
        auto t0 = now();
        HIP_OK(hipEventSynchronize(S.meta_ev));
        t_wait += since(t0);
        const uint32_t *m = S.meta.as<uint32_t>();
        S.n_var = m[kbo::kCallMetaVariants];
        S.n_chars = m[kbo::kCallMetaChars];
        S.n_host = m[kbo::kCallMetaHost];
        if (m[kbo::kCallMetaFlags] != 0 || S.n_host > S.host_cap || S.n_chars > S.chars_cap) {
            S.state = 3;
            return;
        }
        hipStream_t st = S.st;
        S.o_vfirst.ensure((S.ns + 1) * 4);
        HIP_OK(hipMemcpyAsync(S.o_vfirst.p, S.d_vfirst.p, (S.ns + 1) * 4, hipMemcpyDeviceToHost, st));
        if (S.n_var) {
            S.o_pos.ensure((size_t)S.n_var * 4);
            S.o_lens.ensure((size_t)S.n_var * 4);
            S.o_chars.ensure((size_t)S.n_chars + 16);
            HIP_OK(hipMemcpyAsync(S.o_pos.p, S.d_pos.p, (size_t)S.n_var * 4, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(S.o_lens.p, S.d_lens.p, (size_t)S.n_var * 4, hipMemcpyDeviceToHost, st));
            if (S.n_chars) HIP_OK(hipMemcpyAsync(S.o_chars.p, S.d_chars.p, S.n_chars, hipMemcpyDeviceToHost, st));
        }
        if (S.n_host) {
            S.o_hrecs.ensure((size_t)S.n_host * 16);
            S.o_hwin.ensure((size_t)S.n_host * stride);
            HIP_OK(hipMemcpyAsync(S.o_hrecs.p, S.d_hrecs.p, (size_t)S.n_host * 16, hipMemcpyDeviceToHost, st));
            HIP_OK(hipMemcpyAsync(S.o_hwin.p, S.d_hwin.p, (size_t)S.n_host * stride, hipMemcpyDeviceToHost, st));
        }
        HIP_OK(hipEventRecord(S.done_ev, st));
        S.state = 2;
    };

    // the slab's variants behind those of the slabs in front of it
    auto take = [&](CallSlot &S) {
        if (S.state == 0) return;
        downloads(S);
        const Slab &sl = *S.slab;
        if (S.state == 3) { // the slow route, the whole slab
            auto t0 = now();
            HIP_OK(hipStreamSynchronize(S.st));
            std::vector<uint64_t> off(S.ns + 1), cnt(S.ns);
            for (size_t s = 0; s <= S.ns; s++) off[s] = offsets[sl.s0 + s] - sl.b0;
            call_slow(query_idx, concat + sl.b0, off.data(), S.ns, o, d, second_q, out, cnt.data());
            for (size_t s = 0; s < S.ns; s++) var_offsets[sl.s0 + s + 1] = var_offsets[sl.s0 + s] + cnt[s];
            S.state = 0;
            n_slow++;
            t_slow += since(t0);
            return;
        }
        auto t0 = now();
        HIP_OK(hipEventSynchronize(S.done_ev));
        t_wait += since(t0);
        t0 = now();
        const uint32_t *vfirst = S.o_vfirst.as<uint32_t>();
        const uint32_t *dpos = S.o_pos.as<uint32_t>(), *dlens = S.o_lens.as<uint32_t>();
        const uint8_t *dchars = S.o_chars.as<uint8_t>();
        const size_t vb = out.pos.size(), cb = out.chars.size();
        if (S.n_host == 0) {
            out.pos.resize(vb + S.n_var);
            out.qlen.resize(vb + S.n_var);
            out.rlen.resize(vb + S.n_var);
            out.chars.resize(cb + S.n_chars);
            if (S.n_var) {
                const size_t piece = 1u << 16, n_tasks = ((size_t)S.n_var + piece - 1) / piece;
                uint32_t *op = out.pos.data() + vb;
                uint16_t *oq = out.qlen.data() + vb, *orl = out.rlen.data() + vb;
                uint8_t *oc = out.chars.data() + cb;
                const size_t nv = S.n_var, nc = S.n_chars;
                team.run(n_tasks, [&](size_t t) { // (out of pinned memory into the result's own arrays, the lengths apart on the way)
                    const size_t a = t * piece, b = std::min(nv, a + piece);
                    std::memcpy(op + a, dpos + a, (b - a) * 4);
                    for (size_t v = a; v < b; v++) {
                        oq[v] = (uint16_t)(dlens[v] & 0xFFFFu);
                        orl[v] = (uint16_t)(dlens[v] >> 16);
                    }
                    const size_t ca = nc * a / nv, cbb = nc * b / nv; // (the characters by the same shares: any split will do)
                    std::memcpy(oc + ca, dchars + ca, cbb - ca);
                });
            }
            const uint64_t base = var_offsets[sl.s0];
            for (size_t s = 0; s < S.ns; s++) var_offsets[sl.s0 + s + 1] = base + vfirst[s + 1];
        } else {
            // some sites are the host's (a sequence with bytes that are no bases or of 2^20 bases and more, a row whose k-mer crosses a
            // path start, what the reference panics on): resolved here as in the slow route, then merged by query position with what
            // the device made of the sequence's other sites
            n_host_sites += S.n_host;
            const SiteRec *hrecs = S.o_hrecs.as<SiteRec>();
            const uint8_t *hwin = S.o_hwin.as<uint8_t>();
            std::vector<uint32_t> order(S.n_host);
            for (uint32_t x = 0; x < S.n_host; x++) order[x] = x;
            std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                return hrecs[x].seq != hrecs[y].seq ? hrecs[x].seq < hrecs[y].seq : hrecs[x].i < hrecs[y].i;
            });
            RunAutomaton sam;
            std::vector<uint32_t> dq(k), dr(k);
            std::vector<uint8_t> qk(k), rk_spelled;
            size_t dv = 0, dc = 0; // the device's variants / characters taken so far
            auto device_upto = [&](size_t v_end) { // the device's variants [dv, v_end) as they are
                for (; dv < v_end; dv++) {
                    const uint32_t l = dlens[dv], n = (l & 0xFFFFu) + (l >> 16);
                    out.push(dpos[dv], l & 0xFFFFu, l >> 16);
                    out.chars.append(dchars + dc, dchars + dc + n);
                    dc += n;
                }
            };
            const uint64_t base = var_offsets[sl.s0];
            size_t h = 0;
            for (size_t ls = 0; ls < S.ns; ls++) {
                if (h < S.n_host && hrecs[order[h]].seq == ls) {
                    const uint8_t *seq = concat + offsets[sl.s0 + ls];
                    const size_t len = (size_t)(offsets[sl.s0 + ls + 1] - offsets[sl.s0 + ls]);
                    sam.build(seq, len, k, revcomp);
                    const size_t v_end = vfirst[ls + 1];
                    for (; h < S.n_host && hrecs[order[h]].seq == ls; h++) {
                        const SiteRec &r = hrecs[order[h]];
                        const uint8_t *w = hwin + (size_t)order[h] * stride;
                        for (uint32_t t = 0; t < k; t++) { // (as in call_slow: the query-side k-mer and its walk from the first pass's values)
                            const int64_t pos = (int64_t)r.j - (int64_t)(k - 1u) + t;
                            if (pos < 0) { qk[t] = '$'; dr[t] = 0; continue; }
                            qk[t] = seq[pos];
                            dr[t] = std::min<uint32_t>(w[t], (uint32_t)std::min<int64_t>(t + 1u, pos + 1));
                        }
                        const uint8_t *rk = w + kpad;
                        if (w[2u * kpad]) {
                            nav.access_kmer(r.lo, rk_spelled);
                            rk = rk_spelled.data();
                        }
                        sam.depths(rk, k, dq.data());
                        size_t qf, qt, rf, rt;
                        if (!kbo::resolve_variant_ranges(qk.data(), rk, dq.data(), dr.data(), k, d, qf, qt, rf, rt)) continue;
                        while (dv < v_end && dpos[dv] < r.i) device_upto(dv + 1); // the device's variants in front of this one
                        out.push(r.i, (uint32_t)(qt - qf), (uint32_t)(rt - rf));
                        out.chars.append(qk.data() + qf, qk.data() + qt);
                        out.chars.append(rk + rf, rk + rt);
                    }
                    device_upto(v_end);
                } else
                    device_upto(vfirst[ls + 1]);
                var_offsets[sl.s0 + ls + 1] = base + (out.pos.size() - vb);
            }
        }
        S.state = 0;
        t_take += since(t0);
    };

    var_offsets[0] = 0;
    for (size_t i = 0; i < slabs.size(); i++) {
        CallSlot &S = slots[i & 1], &other = slots[(i & 1) ^ 1];
        take(S);           // slab i - 2 (its downloads were asked for one round ago)
        enqueue(S, slabs[i]);
        downloads(other);  // slab i - 1: its words, then its downloads - beside slab i's kernels
    }
    take(slots[slabs.size() & 1]);
    take(slots[(slabs.size() & 1) ^ 1]);
    if (clk.on)
        std::fprintf(stderr, "[kbo timing] call: %zu slabs: staging %.1f ms, enqueue %.1f, waiting for the device %.1f, taking results %.1f, slow route %.1f (%zu slabs); %zu sites resolved on the host\n",
                     slabs.size(), t_stage, t_enq, t_wait, t_take, t_slow, n_slow, n_host_sites);
}

// both entry points: checks, threshold, the route
void call_batch_flat_impl(kbo_index_t *query_idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, const kbo_call_opts *opts,
                          FlatCalls &out, uint64_t *var_offsets)
{
    KBO_REQUIRE(query_idx && var_offsets, KBO_E_BAD_ARG, "null argument");
    check_batch(concat, offsets, n_seqs);
    kbo_call_opts o;
    if (opts) o = *opts; else kbo_call_opts_default(&o);
    KBO_REQUIRE(o.sbwt_build_opts.k == query_idx->host.k, KBO_E_K_MISMATCH, "assert!(sbwt_ref.k() == sbwt_query.k()) (lib.rs:559)");
    KBO_REQUIRE(!query_idx->sharded(), KBO_E_UNSUPPORTED,
                "intervals and the call mode need the rows of one index; this handle is a sharded index");
    const uint32_t k = query_idx->host.k;
    const size_t d = random_match_threshold(k, query_idx->host.n_kmers, 4, o.max_error_prob); // variant_calling.rs:260
    CallClock clk;
    // the second pass's values per site from the device (call_second_kernels.hip), where its q-mers can be at most as long as the
    // threshold and positions fit its tables - and then the variants themselves (call_emit_kernels.hip)
    static const int env_second = std::getenv("KBO_CALL_DEVICE_SECOND") ? std::atoi(std::getenv("KBO_CALL_DEVICE_SECOND")) : 1; // experiments
    const bool env_slow = g_call_device_emit.load() == 0; // (tests, A / B: rounds 3 - 5's route for every slab)
    const uint32_t second_q = (env_second && d >= 6 && d <= 255 && k <= 255 && k >= 2) ? (uint32_t)std::min<size_t>(12, d) : 0u;
    var_offsets[0] = 0;
    if (second_q && !env_slow && n_seqs > 0) call_fast(query_idx, concat, offsets, n_seqs, o, d, second_q, out, var_offsets, clk);
    else {
        std::vector<uint64_t> cnt(n_seqs);
        call_slow(query_idx, concat, offsets, n_seqs, o, d, second_q, out, cnt.data());
        for (size_t s = 0; s < n_seqs; s++) var_offsets[s + 1] = var_offsets[s] + cnt[s];
    }
    clk.lap("device passes + results");
}

} // namespace

extern "C" int kbo_call_batch_flat(kbo_index_t *query_idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                                   const kbo_call_opts *opts, kbo_call_flat *result, uint64_t *var_offsets)
{
    return guarded([&] {
        KBO_REQUIRE(result, KBO_E_BAD_ARG, "null argument");
        std::memset(result, 0, sizeof(*result));
        FlatCalls fc;
        call_batch_flat_impl(query_idx, concat, offsets, n_seqs, opts, fc, var_offsets);
        // (the result's arrays are the ones the slabs' variants were put into: nothing is copied again)
        const size_t nv = fc.pos.size(), nc = fc.chars.size();
        fc.pos.reserve(1);
        fc.qlen.reserve(1);
        fc.rlen.reserve(1);
        fc.chars.reserve(1);
        result->n_variants = nv;
        result->n_chars = nc;
        result->query_pos = fc.pos.release();
        result->query_len = fc.qlen.release();
        result->ref_len = fc.rlen.release();
        result->chars = fc.chars.release();
    });
}

extern "C" int kbo_set_call_device_emit(int on)
{
    g_call_device_emit = on < 0 || on > 2 ? 1 : on;
    return KBO_OK;
}

extern "C" void kbo_call_flat_free(kbo_call_flat *result)
{
    if (!result) return;
    std::free(result->query_pos);
    std::free(result->query_len);
    std::free(result->ref_len);
    std::free(result->chars);
    std::memset(result, 0, sizeof(*result));
}

extern "C" int kbo_call_batch(kbo_index_t *query_idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                              const kbo_call_opts *opts, kbo_variant **out, uint64_t *var_offsets)
{
    return guarded([&] {
        KBO_REQUIRE(out, KBO_E_BAD_ARG, "null argument");
        *out = nullptr;
        FlatCalls fc;
        call_batch_flat_impl(query_idx, concat, offsets, n_seqs, opts, fc, var_offsets);
        CallClock clk;
        // ---- the reference's records (variant_calling.rs:8-26): one allocation, the records, then the characters
        const size_t nv = fc.pos.size(), nc = fc.chars.size();
        const size_t head = std::max<size_t>(1, nv) * sizeof(kbo_variant);
        uint8_t *mem = static_cast<uint8_t *>(std::malloc(head + nc + 1));
        if (!mem) throw std::bad_alloc();
        kbo_variant *rec = reinterpret_cast<kbo_variant *>(mem);
        if (nc) HostTeam::get().copy(mem + head, fc.chars.data(), nc);
        // (where a variant's characters start: a running sum, by blocks)
        const size_t piece = 1u << 16, nb = (nv + piece - 1) / piece;
        std::vector<size_t> c0(nb + 1, 0);
        HostTeam::get().run(nb, [&](size_t t) {
            size_t sum = 0;
            for (size_t v = t * piece; v < std::min(nv, (t + 1) * piece); v++) sum += (size_t)fc.qlen[v] + fc.rlen[v];
            c0[t + 1] = sum;
        });
        for (size_t t = 0; t < nb; t++) c0[t + 1] += c0[t];
        HostTeam::get().run(nb, [&](size_t t) {
            const uint8_t *cp = mem + head + c0[t];
            for (size_t v = t * piece; v < std::min(nv, (t + 1) * piece); v++) {
                const uint32_t ql = fc.qlen[v], rl = fc.rlen[v];
                rec[v].query_pos = fc.pos[v];
                rec[v].query_chars = cp;
                rec[v].query_len = ql;
                rec[v].ref_chars = cp + ql;
                rec[v].ref_len = rl;
                cp += ql + rl;
            }
        });
        *out = rec;
        clk.lap("the reference's records");
    });
}

// test hook (kbo_hip_tuning.h): the depths RunAutomaton answers, for checking against a real one-sequence index on the CPU
extern "C" int kbo_run_automaton_depths(const uint8_t *seq, size_t len, uint32_t k, int add_revcomp, const uint8_t *kmers,
                                        size_t n_kmers, uint32_t *depths_out)
{
    return guarded([&] {
        KBO_REQUIRE(seq && kmers && depths_out && k > 0 && k <= 255, KBO_E_BAD_ARG, "null argument or k outside 1..255");
        RunAutomaton sam;
        sam.build(seq, len, k, add_revcomp != 0);
        for (size_t x = 0; x < n_kmers; x++) sam.depths(kmers + x * k, k, depths_out + x * k);
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
