// kbo_capi.cpp — the extern "C" boundary declared in include/kbo_hip.h.
//
// Host logic only: argument checks mirroring the reference's asserts, index ownership,
// device-memory plumbing and kernel launches.  All matching-statistics / derandomize /
// translate compute happens in ms_kernels.hip; nothing here falls back to the CPU.
#include "../../include/kbo_hip.h"

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
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "host_util.hpp"
#include "kernels.hpp"
#include "refine.hpp"
#include "sbwt_index.hpp"

namespace {

using namespace kbo_host;

struct DevCopy {
    DevBuf arena; // rank blocks of A,C,G,T | null block | contraction entries (32-bit build)
    DevBuf ent;   // contraction entries as their own allocation (big build)
    uint64_t n_blocks = 0;
    bool big = false;
    uint32_t pair_off = 0; // arena index of the two-base extension blocks, 0 = none
};

int g_waves_per_cu = 0;
std::vector<int> g_devices; // devices the host batch entry points spread slabs over (empty = current)
bool g_force_big = false;
uint64_t g_pair_min_rows = 24ull << 20; // indexes with at least this many rows get two-base blocks on the device // tests: use the 64-bit-offset entry layout regardless of size

int current_device()
{
    int dev = -1;
    HIP_OK(hipGetDevice(&dev));
    return dev;
}

} // namespace

struct kbo_index {
    kbo::HostIndex host;
    std::mutex mu;
    std::map<int, DevCopy *> dev;
    uint64_t rank_bytes = 0, lcs_bytes = 0;
    ~kbo_index()
    {
        for (auto &kv : dev) delete kv.second;
    }
};

namespace {

kbo::DevIndexView device_view(kbo_index *idx, int device)
{
    std::lock_guard<std::mutex> g(idx->mu);
    auto it = idx->dev.find(device);
    if (it == idx->dev.end()) {
        KBO_REQUIRE(idx->host.n_sets < 0xFFFFFFF0ull, KBO_E_UNSUPPORTED,
                    "n_sets >= 2^32: 64-bit device layout not built yet");
        // two-base extension blocks: worth their 2.7 B/row once the one-base blocks stop fitting L2
        // (the walk is then bound by line fills, and a two-base step needs one instead of two)
        const size_t est_rank = (idx->host.n_sets / 96 + 2) * 64, est_ent = (idx->host.n_sets + 2) * 12;
        const bool want_pairs = idx->host.n_sets >= g_pair_min_rows && !g_force_big &&
                                est_rank * 5 + est_ent + 64 < 0xFFFFFFF0ull;
        kbo::DeviceLayout lay;
        kbo::make_device_layout(idx->host, lay, want_pairs);
        int prev = current_device();
        if (prev != device) HIP_OK(hipSetDevice(device));
        DevCopy *dc = new DevCopy();
        try {
            const size_t per = lay.n_blocks * 16;
            // arena = rank blocks of A,C,G,T | one all-zero "null" block | contraction entries.
            // When that exceeds the 32-bit offset range (n_sets * 12 B of entries >= ~4 GiB) the
            // entries get their own allocation and 64-bit offsets ("big" kernels).
            const size_t ent_bytes = lay.ent.size() * sizeof(uint32_t);
            const size_t rank_bytes = per * 4 + 16;
            KBO_REQUIRE(rank_bytes < 0xFFFFFFF0ull, KBO_E_UNSUPPORTED, "rank blocks >= 4 GiB");
            dc->big = g_force_big || rank_bytes + ent_bytes >= 0xFFFFFFF0ull;
            const size_t pair_bytes = dc->big ? 0 : lay.pair.size() * sizeof(uint32_t);
            const size_t base_bytes = ((dc->big ? rank_bytes : rank_bytes + ent_bytes) + 15) / 16 * 16;
            const size_t arena_bytes = base_bytes + pair_bytes;
            dc->arena.alloc(arena_bytes);
            HIP_OK(hipMemset(dc->arena.p, 0, arena_bytes));
            for (int c = 0; c < 4; c++)
                HIP_OK(hipMemcpy(dc->arena.as<uint8_t>() + per * c, lay.rank[c].data(), per,
                                 hipMemcpyHostToDevice));
            if (dc->big) {
                dc->ent.alloc(ent_bytes + 16);
                HIP_OK(hipMemcpy(dc->ent.p, lay.ent.data(), ent_bytes, hipMemcpyHostToDevice));
            } else {
                HIP_OK(hipMemcpy(dc->arena.as<uint8_t>() + rank_bytes, lay.ent.data(), ent_bytes,
                                 hipMemcpyHostToDevice));
            }
            if (pair_bytes) {
                HIP_OK(hipMemcpy(dc->arena.as<uint8_t>() + base_bytes, lay.pair.data(), pair_bytes, hipMemcpyHostToDevice));
                dc->pair_off = (uint32_t)(base_bytes / 16);
            }
            dc->n_blocks = lay.n_blocks;
            idx->rank_bytes = per * 4;
            idx->lcs_bytes = ent_bytes;
        } catch (...) {
            delete dc;
            if (prev != device) (void)hipSetDevice(prev);
            throw;
        }
        if (prev != device) HIP_OK(hipSetDevice(prev));
        it = idx->dev.emplace(device, dc).first;
    }
    DevCopy *dc = it->second;
    kbo::DevIndexView v;
    v.arena = dc->arena.as<uint4>();
    v.n_blocks = (uint32_t)dc->n_blocks;
    v.lcs_off = (uint32_t)(dc->n_blocks * 4 + 1);
    v.pair_off = dc->pair_off;
    v.ent = dc->big ? dc->ent.as<uint8_t>() : nullptr;
    v.big = dc->big ? 1u : 0u;
    v.n = (uint32_t)idx->host.n_sets;
    v.k = idx->host.k;
    return v;
}

// upper bound on resident walk waves: CUs x waves per CU (default 32 = 8 per SIMD)
int walk_max_waves()
{
    int dev = current_device();
    int cus = 0;
    HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int per = g_waves_per_cu > 0 ? g_waves_per_cu : 32;
    return std::max(1, cus) * per;
}

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

// ---- work decomposition ------------------------------------------------------------------
// Reads become one item each.  Longer sequences are cut into chunks that restart the walk
// k-1 bases upstream from the empty state (MS depends only on the last k bases).
uint32_t max_len(const uint64_t *offsets, size_t n_seqs)
{
    uint64_t m = 0;
    for (size_t s = 0; s < n_seqs; s++) m = std::max(m, offsets[s + 1] - offsets[s]);
    return (uint32_t)std::min<uint64_t>(m, 0xFFFFFFFFu);
}

// emitted bases per chunk: aim for >= ~1M items when the input allows it, 256..4096 bases
// (every chunk after the first re-walks k-1 warm-up bases); a batch that already has enough
// sequences to fill the device is only cut where a sequence is very long
uint64_t walk_chunk(uint64_t total, size_t n_seqs, uint32_t k)
{
    uint64_t chunk = n_seqs >= (1u << 19) ? 4096 : std::min<uint64_t>(4096, std::max<uint64_t>(256, total >> 20));
    return std::max<uint64_t>(chunk, 4ull * k);
}

void make_items_host(const uint64_t *offsets, size_t n_seqs, uint32_t k, std::vector<kbo::WalkItem> &items)
{
    const uint64_t total = offsets[n_seqs] - offsets[0];
    const uint64_t chunk = walk_chunk(total, n_seqs, k);
    items.clear();
    for (size_t s = 0; s < n_seqs; s++) {
        const uint64_t b = offsets[s], e = offsets[s + 1];
        for (uint64_t c0 = b; c0 < e; c0 += chunk) {
            const uint64_t c1 = std::min(e, c0 + chunk);
            const uint64_t warm = std::min<uint64_t>(c0 - b, k > 0 ? k - 1 : 0);
            kbo::WalkItem it;
            it.start = c0 - warm;
            it.len = (uint32_t)(c1 - c0 + warm);
            it.warm = (uint32_t)warm;
            items.push_back(it);
        }
    }
}

void check_batch(const void *concat, const uint64_t *offsets, size_t n_seqs)
{
    KBO_REQUIRE(concat && offsets, KBO_E_BAD_ARG, "null concat/offsets");
    KBO_REQUIRE(n_seqs > 0, KBO_E_EMPTY_QUERY, "no sequences");
    KBO_REQUIRE(n_seqs < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "more than 2^32-1 sequences per call");
    for (size_t s = 0; s < n_seqs; s++) {
        KBO_REQUIRE(offsets[s + 1] >= offsets[s], KBO_E_BAD_ARG, "offsets not monotone");
        KBO_REQUIRE(offsets[s + 1] > offsets[s], KBO_E_EMPTY_QUERY,
                    "empty query (index.rs:248 assert!(!query.is_empty()))");
        KBO_REQUIRE(offsets[s + 1] - offsets[s] < 0xFFFFFFFFull, KBO_E_UNSUPPORTED,
                    "sequence longer than 2^32-1");
    }
    KBO_REQUIRE(offsets[0] == 0, KBO_E_BAD_ARG, "offsets[0] must be 0");
}

// KBO_TIMING=1 in the environment prints a phase breakdown of the host batch entry points to stderr
struct PhaseClock {
    bool on = std::getenv("KBO_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[kbo timing] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

struct BatchOnDevice {
    DevBuf q, off, items, ms, lo, hi;
    uint64_t total = 0;
    void release()
    {
        for (DevBuf *b : {&q, &off, &items, &ms, &lo, &hi}) b->release();
    }
};

// upload + A1 over a host batch (asynchronous on `stream`); leaves ms (and lo/hi) on the device.
// `items_keep` must stay alive until the stream has been synchronised.
void enqueue_walk_host(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                       bool want_ival, BatchOnDevice &B, std::vector<kbo::WalkItem> &items_keep, hipStream_t stream,
                       uint32_t longest = 0 /* longest sequence if the caller knows it */,
                       hipStream_t copy_stream = nullptr /* uploads go here when given ... */,
                       hipEvent_t copied = nullptr /* ... and `stream` waits for this event */)
{
    KBO_REQUIRE(idx->host.k <= 255, KBO_E_UNSUPPORTED, "k > 255");
    const int dev = current_device();
    kbo::DevIndexView view = device_view(idx, dev);
    const uint64_t total = offsets[n_seqs];
    B.total = total;
    // reads (nothing to chunk): the item list is derived from the offsets on the device;
    // otherwise it is built here (chunks with k-1 warm-up bases) and uploaded
    const uint64_t chunk = walk_chunk(total, n_seqs, idx->host.k);
    const bool device_items = (longest ? longest : max_len(offsets, n_seqs)) <= chunk;
    size_t n_items = n_seqs;
    if (!device_items) {
        make_items_host(offsets, n_seqs, idx->host.k, items_keep);
        n_items = items_keep.size();
    }
    KBO_REQUIRE(n_items < (1ull << 28), KBO_E_UNSUPPORTED, "more than 2^28 work items per launch");
    KBO_REQUIRE(total < 0xFFFFFF00ull, KBO_E_UNSUPPORTED, "4 GiB or more of query in one launch");

    const size_t padded = ((total + 15) / 16) * 16 + 16;
    B.q.ensure(padded);
    B.off.ensure((n_seqs + 1) * sizeof(uint64_t));
    B.items.ensure(n_items * sizeof(kbo::WalkItem));
    B.ms.ensure(padded);
    if (want_ival) {
        B.lo.ensure(total * sizeof(uint32_t));
        B.hi.ensure(total * sizeof(uint32_t));
    }
    hipStream_t up = copy_stream ? copy_stream : stream;
    HIP_OK(hipMemcpyAsync(B.q.p, concat, total, hipMemcpyHostToDevice, up));
    HIP_OK(hipMemcpyAsync(B.off.p, offsets, (n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
    if (!device_items)
        HIP_OK(hipMemcpyAsync(B.items.p, items_keep.data(), items_keep.size() * sizeof(kbo::WalkItem),
                              hipMemcpyHostToDevice, up));
    if (copy_stream) {
        HIP_OK(hipEventRecord(copied, copy_stream));
        HIP_OK(hipStreamWaitEvent(stream, copied, 0));
    }
    if (device_items) HIP_OK(kbo::launch_make_items(B.off.as<uint64_t>(), (uint32_t)n_seqs, B.items.as<kbo::WalkItem>(), stream));
    kbo::WalkArgs a;
    a.ix = view;
    a.q = B.q.as<uint8_t>();
    a.q_bytes = total;
    a.items = B.items.as<kbo::WalkItem>();
    a.n_items = (uint32_t)n_items;
    a.rounds = 0;
    a.d_out = B.ms.as<uint8_t>();
    a.lo_out = want_ival ? B.lo.as<uint32_t>() : nullptr;
    a.hi_out = want_ival ? B.hi.as<uint32_t>() : nullptr;
    HIP_OK(kbo::launch_ms_walk(a, walk_max_waves(), stream));
}

void run_walk_host(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                   bool want_ival, BatchOnDevice &B, hipStream_t stream)
{
    check_batch(concat, offsets, n_seqs);
    std::vector<kbo::WalkItem> items;
    enqueue_walk_host(idx, concat, offsets, n_seqs, want_ival, B, items, stream);
    HIP_OK(hipStreamSynchronize(stream)); // the items vector must outlive the async copy
}

// ---- slabs: a host batch is processed in pieces of at most g_slab_bytes of query so that
// (a) one launch stays below the 32-bit offset limits and (b) the H2D copy of slab i+1 and
// the D2H copy of slab i-1 overlap the kernels of slab i (two streams, user buffers pinned
// in place with hipHostRegister when that succeeds).
size_t g_slab_bytes = 32ull << 20; // tools/bench_host.py: best of 8..128 MiB on the C2 reads

struct Slab {
    size_t s0, s1;   // sequences [s0, s1)
    uint64_t b0, b1; // bases [b0, b1)
};

std::vector<Slab> make_slabs(const uint64_t *offsets, size_t n_seqs, size_t max_bytes)
{
    std::vector<Slab> slabs;
    size_t s0 = 0;
    while (s0 < n_seqs) {
        // last s1 with offsets[s1] - offsets[s0] <= max_bytes (at least one sequence per slab)
        size_t s1 = std::upper_bound(offsets + s0 + 1, offsets + n_seqs + 1, offsets[s0] + max_bytes) - offsets - 1;
        s1 = std::max(s1, s0 + 1);
        slabs.push_back(Slab{s0, s1, offsets[s0], offsets[s1]});
        s0 = s1;
    }
    return slabs;
}

// one pass over the offsets of a batch: order, emptiness, shortest and longest sequence
struct OffsetScan {
    bool monotone = true;
    uint64_t shortest = ~0ull, longest = 0;
};
OffsetScan scan_offsets(const uint64_t *offsets, size_t n_seqs)
{
    const size_t piece = 1u << 18;
    const size_t n_tasks = (n_seqs + piece - 1) / piece;
    std::vector<OffsetScan> part(n_tasks);
    HostTeam::get().run(n_tasks, [&](size_t t) {
        OffsetScan r;
        const size_t a = t * piece, b = std::min(n_seqs, a + piece);
        for (size_t s = a; s < b; s++) {
            r.monotone &= offsets[s + 1] >= offsets[s];
            const uint64_t len = offsets[s + 1] - offsets[s];
            r.shortest = std::min(r.shortest, len);
            r.longest = std::max(r.longest, len);
        }
        part[t] = r;
    });
    OffsetScan r;
    for (const OffsetScan &x : part) {
        r.monotone &= x.monotone;
        r.shortest = std::min(r.shortest, x.shortest);
        r.longest = std::max(r.longest, x.longest);
    }
    return r;
}

bool is_pinned_host(const void *ptr) // memory the DMA engines can reach without staging
{
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// ---- per-device scratch of the host batch entry points, kept between calls: slabs rotate
// through kHostSlots slots, each with its own stream, device buffers and pinned staging, so
// that the staging copy + H2D of slab i+1 and the D2H + copy-out of slab i-1 overlap the
// kernels of slab i.
constexpr int kHostSlots = 4;
struct HostSlot {
    BatchOnDevice B;
    DevBuf chars;
    PinBuf in, out, off;
    std::vector<kbo::WalkItem> items;
    hipEvent_t copied = nullptr, computed = nullptr, done = nullptr;
    bool busy = false;     // a slab is in flight in this slot
    uint64_t out_b0 = 0, out_bytes = 0;
    // run-length output (kbo_find_batch): per-sequence first-run indices + block sums, the records,
    // the number of runs (device word and its pinned copy), what the slab holds
    DevBuf rle_scratch, rles, rle_total, dt_work;
    PinBuf rle_total_pin, rle_first_pin;
    size_t rle_capacity = 0, slab_id = 0, n_seqs = 0;
    uint32_t longest = 0;
};
struct HostCtx {
    int dev = 0;
    HostSlot slot[kHostSlots];
    // one stream per stage, so that every stage runs one slab at a time, in order, next to the
    // other two stages: upload (copy engine), kernels, download (copy kernel)
    hipStream_t st_up = nullptr, st_run = nullptr, st_down = nullptr;
    explicit HostCtx(int d) : dev(d)
    {
        for (hipStream_t *st : {&st_up, &st_run, &st_down}) HIP_OK(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
        for (HostSlot &S : slot)
            for (hipEvent_t *e : {&S.copied, &S.computed, &S.done}) HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    ~HostCtx()
    {
        int prev = 0;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(dev);
        for (hipStream_t st : {st_up, st_run, st_down})
            if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        for (HostSlot &S : slot)
            for (hipEvent_t e : {S.copied, S.computed, S.done})
                if (e) (void)hipEventDestroy(e);
        for (HostSlot &S : slot) { // buffers belong to `dev`
            S.B.release();
            for (DevBuf *b : {&S.chars, &S.rle_scratch, &S.rles, &S.rle_total, &S.dt_work}) b->release();
        }
        (void)hipSetDevice(prev);
    }
};
std::mutex g_ctx_mu;
// leaked on purpose: destroying streams from a static destructor would run after the HIP runtime is gone
std::vector<std::unique_ptr<HostCtx>> &g_ctx_pool = *new std::vector<std::unique_ptr<HostCtx>>();

struct CtxLease { // takes a context of the device out of the pool (or makes one), puts it back
    std::unique_ptr<HostCtx> ctx;
    explicit CtxLease(int dev)
    {
        {
            std::lock_guard<std::mutex> g(g_ctx_mu);
            for (size_t i = 0; i < g_ctx_pool.size(); i++)
                if (g_ctx_pool[i]->dev == dev) {
                    ctx = std::move(g_ctx_pool[i]);
                    g_ctx_pool.erase(g_ctx_pool.begin() + i);
                    break;
                }
        }
        if (!ctx) ctx.reset(new HostCtx(dev));
    }
    ~CtxLease()
    {
        bool busy = false; // an error may have left work in flight
        for (HostSlot &S : ctx->slot) {
            busy |= S.busy;
            S.busy = false;
        }
        if (busy)
            for (hipStream_t st : {ctx->st_up, ctx->st_run, ctx->st_down}) (void)hipStreamSynchronize(st);
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_ctx_pool.push_back(std::move(ctx));
    }
};

void check_len_threshold(const uint64_t *offsets, size_t n_seqs, size_t k, size_t threshold)
{
    KBO_REQUIRE(k > 0, KBO_E_BAD_ARG, "k > 0 (derandomize.rs:274)");
    KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (derandomize.rs:275, translate.rs:269)");
    for (size_t s = 0; s < n_seqs; s++)
        KBO_REQUIRE(offsets[s + 1] - offsets[s] > 2, KBO_E_LEN_LE_2,
                    "len > 2 (derandomize.rs:276, translate.rs:270)");
}

// A5+A6 over a batch whose offsets are known on the host: reads -> LDS kernel, medium
// sequences -> one lane each, very long sequences -> chunked scan (one at a time).
void derand_translate_host_offsets(const uint8_t *d_ms, const uint64_t *d_off, const uint64_t *offsets, size_t n_seqs,
                                   uint32_t k, uint32_t threshold, const uint8_t *d_ref, uint8_t *d_chars,
                                   int32_t *d_derand, hipStream_t stream, uint32_t longest = 0,
                                   DevBuf *piece_work = nullptr /* lets long reads / contigs be split into pieces */)
{
    const uint32_t mx = longest ? longest : max_len(offsets, n_seqs);
    void *work = nullptr;
    size_t work_bytes = 0;
    if (piece_work && mx > 480 && !d_derand) {
        work_bytes = kbo::derand_piece_work_bytes((uint32_t)n_seqs, offsets[n_seqs]);
        piece_work->ensure(work_bytes);
        work = piece_work->p;
    }
    HIP_OK(kbo::launch_derand_translate(d_ms, d_off, (uint32_t)n_seqs, k, threshold, d_ref, d_chars, d_derand, mx,
                                        kbo::kLongSeq, stream, offsets[n_seqs], work, work_bytes));
    if (mx <= kbo::kLongSeq) return;
    size_t need = 0;
    for (size_t s = 0; s < n_seqs; s++) {
        const uint64_t len = offsets[s + 1] - offsets[s];
        if (len > kbo::kLongSeq) need = std::max(need, kbo::derand_long_scratch_bytes(len, k, threshold));
    }
    DevBuf scratch(need);
    for (size_t s = 0; s < n_seqs; s++) {
        const uint64_t b = offsets[s], len = offsets[s + 1] - offsets[s];
        if (len <= kbo::kLongSeq) continue;
        HIP_OK(kbo::launch_derand_long(d_ms + b, (uint32_t)len, k, threshold, d_ref ? d_ref + b : nullptr, d_chars + b,
                                       d_derand ? d_derand + b : nullptr, scratch.p, stream));
    }
    HIP_OK(hipStreamSynchronize(stream)); // scratch is released on return
}

// device run-length records are seven u32; the API's kbo_rle has the reference's usize fields
constexpr size_t kRleWords = 7;
void widen_rles(kbo_rle *dst, const uint32_t *src, size_t n, HostTeam &team)
{
    const size_t piece = 1u << 14;
    team.run((n + piece - 1) / piece, [&](size_t t) {
        const size_t a = t * piece, b = std::min(n, a + piece);
        for (size_t q = a; q < b; q++) {
            const uint32_t *r = src + q * kRleWords;
            dst[q] = kbo_rle{r[0], r[1], r[2], r[3], r[4], r[5], r[6]};
        }
    });
}

// Where kbo_find_batch collects format::run_lengths_gapped of every slab (computed on the device
// from the slab's characters, which then never leave it)
struct RleSink {
    size_t max_gap_len = 0;
    uint64_t *rle_offsets = nullptr; // caller's n_seqs + 1 entries
    // one device: slabs complete in order, so their records go straight into the result array
    kbo_rle *all = nullptr;
    size_t all_cap = 0, all_used = 0;
    // several devices: slabs complete out of order, kept per slab and put together at the end
    std::vector<std::vector<kbo_rle>> runs;
    std::vector<std::vector<uint32_t>> first; // index of the first run of each sequence of the slab, +1 entry
    ~RleSink() { std::free(all); }
};

// kbo::matches over a batch (lib.rs:618-627); optional relative_to_ref (lib.rs:756-757); with a sink
// the characters are turned into run lengths on the device instead of being downloaded (lib.rs:816-820)
void matches_batch_impl(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                        double max_error_prob, bool format, uint8_t *chars_out, RleSink *sink = nullptr)
{
    KBO_REQUIRE(idx && (chars_out || sink), KBO_E_BAD_ARG, "null argument");
    PhaseClock clk;
    const size_t k = idx->host.k;
    const size_t threshold = random_match_threshold(k, idx->host.n_kmers, 4, max_error_prob); // lib.rs:620
    KBO_REQUIRE(concat && offsets, KBO_E_BAD_ARG, "null concat/offsets");
    KBO_REQUIRE(n_seqs > 0, KBO_E_EMPTY_QUERY, "no sequences");
    KBO_REQUIRE(n_seqs < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "more than 2^32-1 sequences per call");
    KBO_REQUIRE(offsets[0] == 0, KBO_E_BAD_ARG, "offsets[0] must be 0");
    const OffsetScan scan = scan_offsets(offsets, n_seqs);
    KBO_REQUIRE(scan.monotone, KBO_E_BAD_ARG, "offsets not monotone");
    KBO_REQUIRE(scan.shortest > 0, KBO_E_EMPTY_QUERY, "empty query (index.rs:248 assert!(!query.is_empty()))");
    KBO_REQUIRE(scan.longest < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "sequence longer than 2^32-1");
    KBO_REQUIRE(k > 0, KBO_E_BAD_ARG, "k > 0 (derandomize.rs:274)");
    KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (derandomize.rs:275, translate.rs:269)");
    KBO_REQUIRE(scan.shortest > 2, KBO_E_LEN_LE_2, "len > 2 (derandomize.rs:276, translate.rs:270)");
    clk.lap("argument checks");
    const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, g_slab_bytes);
    // user buffers the DMA engines can reach directly are used in place, pageable ones are staged
    const bool in_pinned = is_pinned_host(concat), out_pinned = sink || is_pinned_host(chars_out);
    if (sink) {
        sink->runs.assign(slabs.size(), {});
        sink->first.assign(slabs.size(), {});
    }
    const bool sink_direct = sink && (g_devices.size() <= 1 || slabs.size() <= 1);
    if (sink_direct) { // room for 2 runs per sequence to start with (untouched pages cost nothing)
        sink->all_cap = 2 * n_seqs + 1024;
        sink->all = static_cast<kbo_rle *>(std::malloc(sink->all_cap * sizeof(kbo_rle)));
        if (!sink->all) throw std::bad_alloc();
        sink->rle_offsets[0] = 0;
    }
    clk.lap("slab list");
    // one worker per device (index replicated on each, slabs dealt round-robin, disjoint output
    // slices: no exchange between devices); a single device runs on the calling thread
    std::vector<int> devices = g_devices;
    if (devices.empty()) devices.push_back(current_device());
    const size_t nd = std::min(devices.size(), std::max<size_t>(1, slabs.size()));
    HostTeam &team = HostTeam::get();
    auto worker = [&](size_t w) {
        HIP_OK(hipSetDevice(devices[w]));
        CtxLease lease(devices[w]);
        HostCtx &C = *lease.ctx;
        // The calling thread stages and submits slabs; a second thread completes them in
        // submission order (waits for the slab's event, copies the staged output to the user
        // buffer), so the two host copies of a slab never queue behind each other.
        std::mutex mu;
        std::condition_variable cv;
        size_t submitted = 0, drained = 0;
        bool stop = false;
        int drain_code = KBO_OK;
        std::string drain_error;
        std::thread drainer([&] {
            try {
                HIP_OK(hipSetDevice(devices[w]));
                const size_t none = ~size_t(0);
                // run lengths: the number of records of a slab is known once its kernels are done, so the
                // download is issued here; it is issued for the next slab before the previous one is
                // copied out, so that the copy engine and the host copy work on different slabs
                auto start = [&](size_t turn) {
                    if (!sink) return;
                    HostSlot &S = C.slot[turn % kHostSlots];
                    HIP_OK(hipEventSynchronize(S.computed));
                    const uint32_t total = *S.rle_total_pin.as<uint32_t>();
                    if (total > S.rle_capacity) { // more runs than the speculative emit had room for
                        S.rle_capacity = (size_t)total + total / 4 + 16;
                        S.rles.ensure(S.rle_capacity * kRleWords * sizeof(uint32_t));
                        HIP_OK(kbo::launch_rle_emit(S.chars.as<uint8_t>(), S.B.off.as<uint64_t>(), (uint32_t)S.n_seqs,
                                                    (uint32_t)std::min<size_t>(sink->max_gap_len, 0xFFFFFFFFu),
                                                    S.rle_scratch.as<uint32_t>(), S.rles.as<uint32_t>(),
                                                    (uint32_t)S.rle_capacity, C.st_down, S.longest));
                    }
                    const size_t words = kbo::chunk_items_scratch_words((uint32_t)S.n_seqs);
                    S.out.ensure(std::max<size_t>(16, (size_t)total * kRleWords * sizeof(uint32_t)));
                    S.rle_first_pin.ensure(words * sizeof(uint32_t));
                    if (total)
                        HIP_OK(hipMemcpyAsync(S.out.p, S.rles.p, (size_t)total * kRleWords * sizeof(uint32_t),
                                              hipMemcpyDeviceToHost, C.st_down));
                    HIP_OK(hipMemcpyAsync(S.rle_first_pin.p, S.rle_scratch.p, words * sizeof(uint32_t), hipMemcpyDeviceToHost, C.st_down));
                    HIP_OK(hipEventRecord(S.done, C.st_down));
                    S.out_bytes = total; // records
                };
                auto finish = [&](size_t turn) {
                    HostSlot &S = C.slot[turn % kHostSlots];
                    HIP_OK(hipEventSynchronize(S.done));
                    if (!sink) {
                        if (!out_pinned) HostTeam::out().copy(chars_out + S.out_b0, S.out.p, S.out_bytes);
                    } else {
                        const size_t total = S.out_bytes;
                        const uint32_t *local = S.rle_first_pin.as<uint32_t>(), *sums = local + S.n_seqs + 1;
                        if (sink_direct) {
                            if (sink->all_used + total > sink->all_cap) {
                                const size_t cap = (sink->all_used + total) * 2;
                                kbo_rle *p = static_cast<kbo_rle *>(std::realloc(sink->all, cap * sizeof(kbo_rle)));
                                if (!p) throw std::bad_alloc();
                                sink->all = p;
                                sink->all_cap = cap;
                            }
                            const size_t base = sink->all_used, s0 = slabs[S.slab_id].s0, ns_slab = S.n_seqs;
                            widen_rles(sink->all + base, S.out.as<uint32_t>(), total, HostTeam::out());
                            const size_t piece = 1u << 15;
                            HostTeam::out().run((ns_slab + piece - 1) / piece, [&](size_t t) {
                                const size_t a = t * piece + 1, b = std::min(ns_slab, a + piece - 1);
                                for (size_t q = a; q <= b; q++) sink->rle_offsets[s0 + q] = base + sums[q / 1024] + local[q];
                            });
                            sink->all_used += total;
                        } else {
                            std::vector<kbo_rle> &runs = sink->runs[S.slab_id];
                            runs.resize(total);
                            widen_rles(runs.data(), S.out.as<uint32_t>(), total, HostTeam::out());
                            std::vector<uint32_t> &first = sink->first[S.slab_id];
                            first.resize(S.n_seqs + 1);
                            for (size_t q = 0; q <= S.n_seqs; q++) first[q] = sums[q / 1024] + local[q];
                        }
                    }
                    S.busy = false;
                    {
                        std::lock_guard<std::mutex> g(mu);
                        drained++;
                    }
                    cv.notify_all();
                };
                size_t started = 0, pending = none;
                for (;;) {
                    bool can_start;
                    {
                        std::unique_lock<std::mutex> g(mu);
                        cv.wait(g, [&] { return started < submitted || pending != none || stop; });
                        can_start = started < submitted;
                        if (!can_start && pending == none) return;
                    }
                    const size_t prev = pending;
                    pending = none;
                    if (can_start) {
                        start(started);
                        pending = started++;
                    }
                    if (prev != none) finish(prev);
                }
            } catch (const KboError &e) {
                std::lock_guard<std::mutex> g(mu);
                drain_code = e.code;
                drain_error = e.what();
                drained = ~size_t(0) / 2; // releases the submitting thread
                cv.notify_all();
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> g(mu);
                drain_code = KBO_E_HIP;
                drain_error = e.what();
                drained = ~size_t(0) / 2;
                cv.notify_all();
            }
        });
        auto join_drainer = [&] {
            {
                std::lock_guard<std::mutex> g(mu);
                stop = true;
            }
            cv.notify_all();
            if (drainer.joinable()) drainer.join();
        };
        try {
            size_t turn = 0;
            for (size_t i = w; i < slabs.size(); i += nd, turn++) {
                const Slab &sl = slabs[i];
                HostSlot &S = C.slot[turn % kHostSlots];
                {
                    std::unique_lock<std::mutex> g(mu);
                    cv.wait(g, [&] { return turn < drained + kHostSlots; }); // the slot is free again
                    if (drain_code != KBO_OK) break;
                }
                if (w == 0) clk.lap("  wait for a free slot");
                const size_t ns = sl.s1 - sl.s0;
                const uint64_t bytes = sl.b1 - sl.b0;
                // stage: slab-relative offsets (and the longest sequence of the slab), query bytes
                S.off.ensure((ns + 1) * sizeof(uint64_t));
                uint64_t *off = S.off.as<uint64_t>();
                const size_t piece = 1u << 15, n_tasks = (ns + 1 + piece - 1) / piece;
                std::vector<uint64_t> longest(n_tasks, 0);
                team.run(n_tasks, [&](size_t t) {
                    const size_t a = t * piece, b = std::min(ns + 1, a + piece);
                    uint64_t m = 0;
                    for (size_t j = a; j < b; j++) {
                        off[j] = offsets[sl.s0 + j] - sl.b0;
                        if (j < ns) m = std::max(m, offsets[sl.s0 + j + 1] - offsets[sl.s0 + j]);
                    }
                    longest[t] = m;
                });
                const uint32_t mx = (uint32_t)*std::max_element(longest.begin(), longest.end());
                const uint8_t *src = concat + sl.b0;
                if (!in_pinned) {
                    S.in.ensure(bytes);
                    team.copy(S.in.p, src, bytes);
                    src = S.in.as<uint8_t>();
                }
                if (w == 0) clk.lap("  offsets + copy in");
                enqueue_walk_host(idx, src, off, ns, false, S.B, S.items, C.st_run, mx, C.st_up, S.copied);
                if (sink) {
                    // characters stay on the device; run lengths are counted, scanned and (speculatively, into
                    // the room the slot has) emitted right behind A5/A6; the completing thread downloads them
                    S.chars.ensure(((S.B.total + 15) / 16) * 16 + 32);
                    derand_translate_host_offsets(S.B.ms.as<uint8_t>(), S.B.off.as<uint64_t>(), off, ns, (uint32_t)k,
                                                  (uint32_t)threshold, nullptr, S.chars.as<uint8_t>(), nullptr, C.st_run, mx, &S.dt_work);
                    const uint32_t gap = (uint32_t)std::min<size_t>(sink->max_gap_len, 0xFFFFFFFFu);
                    S.rle_scratch.ensure(kbo::chunk_items_scratch_words((uint32_t)ns) * sizeof(uint32_t));
                    S.rle_total.ensure(16);
                    S.rle_total_pin.ensure(16);
                    if (S.rle_capacity < 2 * ns + 16) {
                        S.rle_capacity = 2 * ns + 16;
                        S.rles.ensure(S.rle_capacity * kRleWords * sizeof(uint32_t));
                    }
                    HIP_OK(kbo::launch_rle_count(S.chars.as<uint8_t>(), S.B.off.as<uint64_t>(), (uint32_t)ns, gap,
                                                 S.rle_scratch.as<uint32_t>(), S.rle_total.as<uint32_t>(), C.st_run, mx));
                    HIP_OK(hipMemcpyAsync(S.rle_total_pin.p, S.rle_total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, C.st_run));
                    HIP_OK(kbo::launch_rle_emit(S.chars.as<uint8_t>(), S.B.off.as<uint64_t>(), (uint32_t)ns, gap,
                                                S.rle_scratch.as<uint32_t>(), S.rles.as<uint32_t>(), (uint32_t)S.rle_capacity,
                                                C.st_run, mx));
                    S.longest = mx;
                    HIP_OK(hipEventRecord(S.computed, C.st_run));
                    HIP_OK(hipStreamWaitEvent(C.st_down, S.computed, 0));
                    S.slab_id = i;
                    S.n_seqs = ns;
                } else {
                    // D2H leg: hipMemcpyAsync on the download stream.  With one stream per stage the copy
                    // engines carry both directions at once (tools/bench_host.py: 37-40 Gbp/s host->host;
                    // a small kernel storing into pinned memory, or A5/A6 storing there themselves, gave
                    // 28 and 26 Gbp/s).
                    uint8_t *dst = chars_out + sl.b0;
                    if (!out_pinned) {
                        S.out.ensure(bytes + 32);
                        dst = S.out.as<uint8_t>();
                    }
                    S.chars.ensure(((S.B.total + 15) / 16) * 16 + 16);
                    derand_translate_host_offsets(S.B.ms.as<uint8_t>(), S.B.off.as<uint64_t>(), off, ns, (uint32_t)k,
                                                  (uint32_t)threshold, format ? S.B.q.as<uint8_t>() : nullptr,
                                                  S.chars.as<uint8_t>(), nullptr, C.st_run, mx, &S.dt_work);
                    HIP_OK(hipEventRecord(S.computed, C.st_run));
                    HIP_OK(hipStreamWaitEvent(C.st_down, S.computed, 0));
                    HIP_OK(hipMemcpyAsync(dst, S.chars.p, bytes, hipMemcpyDeviceToHost, C.st_down));
                    HIP_OK(hipEventRecord(S.done, C.st_down));
                }
                S.busy = true;
                S.out_b0 = sl.b0;
                S.out_bytes = bytes;
                {
                    std::lock_guard<std::mutex> g(mu);
                    submitted++;
                }
                cv.notify_all();
                if (w == 0) clk.lap("  enqueue");
            }
        } catch (...) {
            join_drainer();
            throw;
        }
        join_drainer();
        if (drain_code != KBO_OK) throw KboError(drain_code, drain_error);
        if (w == 0) clk.lap("drain");
    };
    if (nd == 1) {
        const int prev = current_device();
        worker(0);
        if (prev != devices[0]) HIP_OK(hipSetDevice(prev));
        return;
    }
    std::vector<std::thread> threads;
    std::vector<std::string> errors(nd);
    std::vector<int> codes(nd, KBO_OK);
    for (size_t w = 0; w < nd; w++)
        threads.emplace_back([&, w] {
            try {
                worker(w);
            } catch (const KboError &e) {
                codes[w] = e.code;
                errors[w] = e.what();
            } catch (const std::exception &e) {
                codes[w] = KBO_E_HIP;
                errors[w] = e.what();
            }
        });
    for (auto &t : threads) t.join();
    for (size_t w = 0; w < nd; w++)
        if (codes[w] != KBO_OK) throw KboError(codes[w], errors[w]);
}

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

// format.rs:143-193, statement for statement (sequential, variable-length output: host)
void run_lengths_gapped_impl(const uint8_t *aln, size_t len, size_t max_gap_len, std::vector<kbo_rle> &out)
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
                // the reference indexes aln[i - 1] unguarded (format.rs:175) and would panic
                // on an 'R' at position 0; guarded here
                rle.jumps += (aln[i] == 'R' && i > 0 && aln[i - 1] == 'R');
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
        kbo_index *idx = new kbo_index();
        try {
            kbo::build_host_index(seqs, lens, n_seqs, p, idx->host);
        } catch (...) {
            delete idx;
            throw;
        }
        *out = idx;
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
        *out = idx;
    });
}

int kbo_index_export_parts(const kbo_index_t *idx, uint64_t *const rows[4], uint64_t C[4], uint8_t *lcs)
{
    return guarded([&] {
        KBO_REQUIRE(idx && rows && C && lcs, KBO_E_BAD_ARG, "null argument");
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

int kbo_index_save(const kbo_index_t *idx, const char *path)
{
    int rc = guarded([&] {
        KBO_REQUIRE(idx && path, KBO_E_BAD_ARG, "null argument");
        kbo::save_host_index(idx->host, path);
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
            kbo::load_host_index(path, idx->host);
        } catch (...) {
            delete idx;
            throw;
        }
        *out = idx;
    });
    return rc == KBO_E_BAD_ARG && path && out ? KBO_E_IO : rc;
}

int kbo_index_to_device(kbo_index_t *idx, int device)
{
    return guarded([&] {
        KBO_REQUIRE(idx, KBO_E_BAD_ARG, "null index");
        (void)device_view(idx, device < 0 ? current_device() : device);
    });
}

int kbo_index_device_bytes(const kbo_index_t *idx, uint64_t *rank_bytes, uint64_t *lcs_bytes)
{
    return guarded([&] {
        KBO_REQUIRE(idx, KBO_E_BAD_ARG, "null index");
        const uint64_t nb = idx->host.n_sets / kbo::kRankRowsPerBlock + 2;
        if (rank_bytes) *rank_bytes = nb * 16 * 4;
        if (lcs_bytes) *lcs_bytes = (3 * (idx->host.n_sets + 1) + 4) * sizeof(uint32_t);
    });
}

uint64_t kbo_index_device_pair_bytes(const kbo_index_t *idx)
{
    if (!idx || idx->host.n_sets < g_pair_min_rows || g_force_big) return 0;
    return (idx->host.n_sets / kbo::kRankRowsPerBlock + 2) * 16 * 16;
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
    return guarded([&] {
        KBO_REQUIRE(idx && d_out, KBO_E_BAD_ARG, "null argument");
        KBO_REQUIRE((lo_out == nullptr) == (hi_out == nullptr), KBO_E_BAD_ARG, "lo/hi must come together");
        check_batch(concat, offsets, n_seqs);
        hipStream_t stream = nullptr;
        // intervals cost 8 more bytes per base on the device: smaller slabs
        const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, lo_out ? g_slab_bytes / 4 : g_slab_bytes);
        std::vector<uint64_t> off;
        for (const Slab &sl : slabs) {
            const size_t ns = sl.s1 - sl.s0;
            off.resize(ns + 1);
            for (size_t j = 0; j <= ns; j++) off[j] = offsets[sl.s0 + j] - sl.b0;
            BatchOnDevice B;
            std::vector<kbo::WalkItem> items;
            enqueue_walk_host(idx, concat + sl.b0, off.data(), ns, lo_out != nullptr, B, items, stream);
            HIP_OK(hipMemcpyAsync(d_out + sl.b0, B.ms.p, B.total, hipMemcpyDeviceToHost, stream));
            if (lo_out) {
                HIP_OK(hipMemcpyAsync(lo_out + sl.b0, B.lo.p, B.total * 4, hipMemcpyDeviceToHost, stream));
                HIP_OK(hipMemcpyAsync(hi_out + sl.b0, B.hi.p, B.total * 4, hipMemcpyDeviceToHost, stream));
            }
            HIP_OK(hipStreamSynchronize(stream));
        }
    });
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
        run_lengths_gapped_impl(aln, len, max_gap_len, v);
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

int kbo_find_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                   const kbo_find_opts *opts, kbo_rle **rles, uint64_t *rle_offsets)
{
    return guarded([&] {
        KBO_REQUIRE(idx && rles && rle_offsets, KBO_E_BAD_ARG, "null argument");
        kbo_find_opts o;
        if (opts) o = *opts; else kbo_find_opts_default(&o);
        // lib.rs:815-820: matches, then run_lengths_gapped per sequence; both on the device, slab by slab
        RleSink sink;
        sink.max_gap_len = o.max_gap_len;
        sink.rle_offsets = rle_offsets;
        matches_batch_impl(idx, concat, offsets, n_seqs, o.max_error_prob, false, nullptr, &sink);
        if (sink.all) { // one device: the records are already in place
            *rles = sink.all;
            sink.all = nullptr;
            return;
        }
        const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, g_slab_bytes);
        std::vector<uint64_t> base(slabs.size() + 1, 0);
        for (size_t i = 0; i < slabs.size(); i++) base[i + 1] = base[i] + sink.runs[i].size();
        kbo_rle *all = static_cast<kbo_rle *>(std::malloc(std::max<uint64_t>(1, base.back()) * sizeof(kbo_rle)));
        if (!all) throw std::bad_alloc();
        rle_offsets[0] = 0;
        HostTeam::get().run(slabs.size(), [&](size_t i) {
            if (!sink.runs[i].empty()) std::memcpy(all + base[i], sink.runs[i].data(), sink.runs[i].size() * sizeof(kbo_rle));
            const size_t ns = slabs[i].s1 - slabs[i].s0;
            for (size_t q = 1; q <= ns; q++) rle_offsets[slabs[i].s0 + q] = base[i] + sink.first[i][q];
        });
        *rles = all;
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
    size_t bytes;
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
    return w;
}
} // namespace

size_t kbo_work_bytes(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k)
{
    return dev_work(n_seqs, total_bases, max_seq_len, k).bytes;
}

int kbo_ms_batch_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs,
                     uint64_t total_bases, size_t max_seq_len, uint8_t *d_ms_out, uint32_t *d_lo_out,
                     uint32_t *d_hi_out, void *d_work, size_t work_bytes, void *stream)
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
        kbo::DevIndexView view = device_view(idx, current_device());
        kbo::WalkItem *items = static_cast<kbo::WalkItem *>(d_work);
        // reads: one item per sequence; batches that hold (or may hold) long sequences: chunks
        const DevWork w = dev_work(n_seqs, total_bases, max_seq_len, idx->host.k);
        KBO_REQUIRE(work_bytes >= w.bytes, KBO_E_BAD_ARG, "d_work is smaller than kbo_work_bytes() for this batch");
        KBO_REQUIRE(total_bases / w.chunk + n_seqs < (1ull << 28), KBO_E_UNSUPPORTED, "more than 2^28 work items per launch");
        if (w.chunked) {
            uint32_t *scratch = reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(d_work) +
                                                            ((size_t)w.n_slots * sizeof(kbo::WalkItem) + 15) / 16 * 16);
            HIP_OK(kbo::launch_make_chunk_items(d_offsets, (uint32_t)n_seqs, w.chunk, idx->host.k, w.n_slots, items, scratch, s));
        } else {
            HIP_OK(kbo::launch_make_items(d_offsets, (uint32_t)n_seqs, items, s));
        }
        kbo::WalkArgs a;
        a.ix = view;
        a.q = d_concat;
        a.q_bytes = total_bases;
        a.items = items;
        a.n_items = w.n_slots;
        a.rounds = 0;
        a.d_out = d_ms_out;
        a.lo_out = d_lo_out;
        a.hi_out = d_hi_out;
        HIP_OK(kbo::launch_ms_walk(a, walk_max_waves(), s));
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
        HIP_OK(hipGetDeviceCount(&count));
        for (int i = 0; i < n; i++) KBO_REQUIRE(devices[i] >= 0 && devices[i] < count, KBO_E_BAD_ARG, "no such device");
        g_devices.assign(devices, devices + n);
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
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_ctx_pool.clear();
    });
}

int kbo_set_pair_steps(uint64_t min_rows, int min_depth)
{
    g_pair_min_rows = min_rows; // applies to device copies made after the call
    if (min_depth >= 0) kbo::set_pair_min_depth(min_depth);
    return KBO_OK;
}

int kbo_set_force_big_layout(int on)
{
    g_force_big = on != 0; // applies to device copies made after the call
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

int kbo_set_walk_waves_per_cu(int waves_per_cu)
{
    g_waves_per_cu = waves_per_cu > 0 ? waves_per_cu : 0;
    return KBO_OK;
}

} // extern "C"
