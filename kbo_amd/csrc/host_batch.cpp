// host_batch.cpp — host batches: work decomposition, slabs, pinned staging with helper threads, the
// three-stage (upload / kernels / download) pipeline with pooled per-device scratch, and the
// run-length sink of kbo_find_batch.  No compute here: kernels live in the *_kernels.hip files.
#include "capi_internal.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

namespace kbo_host {

std::vector<int> g_devices;
std::mutex g_devices_mu;
std::vector<int> devices_snapshot()
{
    std::lock_guard<std::mutex> g(g_devices_mu);
    return g_devices;
}
std::vector<int> devices_for(kbo_index *idx)
{
    if (idx) {
        std::lock_guard<std::mutex> g(idx->mu);
        if (idx->opts.n_devices >= 0) return idx->opts.devices;
    }
    return devices_snapshot();
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

// emitted bases per chunk: aim for about 130 k items when the input allows it, 256..4096 bases
// (every chunk after the first re-walks k-1 warm-up bases); a batch that already has enough
// sequences to fill the device is only cut where a sequence is very long
uint64_t walk_chunk(uint64_t total, size_t n_seqs, uint32_t k)
{
    // (measured on 100 Mbp of 10 kbp reads, call mode: 192 / 256 / 384 / 512 / 768 / 1024 / 1536 / 2048 bases per chunk ->
    // 39.9 / 41.9 / 42.9 / 43.6 / 46.0 / 42.5 / 32.1 / 26.0 Gbp/s: about 130 k items, a quarter of the lanes, is the best
    // trade between the k warm-up bases every chunk re-walks and the number of chains in flight)
    // Not above 768: the plan-guided walk keeps at most 29 mismatches of an item (13 for reads), which chunks of 800 bases
    // at 1 % substitutions rarely exceed and chunks of 4000 always do (500 Mbp of 10 kbp reads, call mode: 46.5 Gbp/s with
    // chunks of 3814 bases, every one of them walked as an item without a plan); 30 warm-up bases per 768 cost the plain
    // walk 4 %.
    (void)n_seqs;
    // (a multiple of 64: tools/dbg_call_chunk.py - with chunks of 457 bases, what a slab of 60 Mbp used to get, the call mode of the
    // plan-guided walk gave different sites from run to run; 448, 460, 464, 300, 1000 are exact.  Slabs were 32 MiB until round 4,
    // whose chunks are 256 bases, so nothing ever ran that way; chunk boundaries now stay 4-byte aligned relative to the sequence)
    uint64_t chunk = std::min<uint64_t>(768, std::max<uint64_t>(256, (total >> 17) & ~63ull));
    static const int env_chunk = std::getenv("KBO_WALK_CHUNK") ? std::atoi(std::getenv("KBO_WALK_CHUNK")) : 0; // experiments
    if (env_chunk > 0) chunk = ((uint64_t)env_chunk + 3u) & ~3ull;
    return std::max<uint64_t>(chunk, 4ull * k);
}

// call = true: items for the call mode of the walk (k warm-up bases so that the MS value in front of the first owned
// base is exact up to min(., k); up to k more bases behind the chunk, walked only to finish the search to the right)
void make_items_host(const uint64_t *offsets, size_t n_seqs, uint32_t k, std::vector<kbo::WalkItem> &items, bool call = false)
{
    const uint64_t total = offsets[n_seqs] - offsets[0];
    const uint64_t chunk = walk_chunk(total, n_seqs, k);
    KBO_REQUIRE(!call || (chunk & 3u) == 0, KBO_E_BAD_ARG, "call mode: chunks of a multiple of four bases (kernels.hpp launch_make_chunk_items)");
    items.clear();
    for (size_t s = 0; s < n_seqs; s++) {
        const uint64_t b = offsets[s], e = offsets[s + 1];
        for (uint64_t c0 = b; c0 < e; c0 += chunk) {
            const uint64_t c1 = std::min(e, c0 + chunk);
            const uint64_t warm = std::min<uint64_t>(c0 - b, call ? k : (k > 0 ? k - 1 : 0));
            const uint64_t tail = call ? std::min<uint64_t>(e - c1, k) : 0;
            kbo::WalkItem it;
            it.start = c0 - warm;
            it.len = (uint32_t)(c1 - c0 + warm + tail);
            it.warm = (uint32_t)warm | ((uint32_t)tail << 16);
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


// upload + A1 over a host batch (asynchronous on `stream`); leaves ms (and lo/hi) on the device.
// `items_keep` must stay alive until the stream has been synchronised.
void enqueue_walk_host(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                       bool want_ival, BatchOnDevice &B, std::vector<kbo::WalkItem> &items_keep, hipStream_t stream,
                       uint32_t longest, hipStream_t copy_stream, hipEvent_t copied, const CallSink *call, const PackedIn *packed, FusedMap *map)
{
    KBO_REQUIRE(idx->host.k <= 255, KBO_E_UNSUPPORTED, "k > 255");
    const int dev = current_device();
    const std::vector<kbo_index *> shards = shards_of(idx); // (a sharded index: every shard is walked, the maximum kept)
    KBO_REQUIRE(shards.size() == 1 || (!want_ival && !call), KBO_E_UNSUPPORTED,
                "intervals and the call mode need the rows of one index; this handle is a sharded index");
    const uint64_t total = offsets[n_seqs];
    B.total = total;
    // reads (nothing to chunk): the item list is derived from the offsets on the device;
    // otherwise it is built here (chunks with k-1 warm-up bases) and uploaded
    const uint64_t chunk = walk_chunk(total, n_seqs, idx->host.k);
    const uint32_t longest_seq = longest ? longest : max_len(offsets, n_seqs);
    const bool device_items = longest_seq <= chunk;
    size_t n_items = n_seqs;
    if (!device_items) {
        make_items_host(offsets, n_seqs, idx->host.k, items_keep, call != nullptr);
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
    const bool uniform = packed && packed->uniform_len != 0;
    if (packed) { // a quarter of the bytes: 2-bit words + the non-ACGT list; unpacked into B.q below
        B.packed.ensure(packed->n_words * 4 + 16);
        HIP_OK(hipMemcpyAsync(B.packed.p, packed->words, packed->n_words * 4, hipMemcpyHostToDevice, up));
        if (packed->n_exc) {
            B.exc_pos.ensure(packed->n_exc * 8);
            B.exc_byte.ensure(packed->n_exc);
            HIP_OK(hipMemcpyAsync(B.exc_pos.p, packed->exc_pos, packed->n_exc * 8, hipMemcpyHostToDevice, up));
            HIP_OK(hipMemcpyAsync(B.exc_byte.p, packed->exc_byte, packed->n_exc, hipMemcpyHostToDevice, up));
        }
    } else {
        HIP_OK(hipMemcpyAsync(B.q.p, concat, total, hipMemcpyHostToDevice, up));
    }
    if (!uniform) HIP_OK(hipMemcpyAsync(B.off.p, offsets, (n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, up));
    if (!device_items)
        HIP_OK(hipMemcpyAsync(B.items.p, items_keep.data(), items_keep.size() * sizeof(kbo::WalkItem),
                              hipMemcpyHostToDevice, up));
    if (copy_stream) {
        HIP_OK(hipEventRecord(copied, copy_stream));
        HIP_OK(hipStreamWaitEvent(stream, copied, 0));
    }
    if (packed) {
        if (uniform) HIP_OK(kbo::launch_uniform_offsets(B.off.as<uint64_t>(), (uint32_t)n_seqs, packed->uniform_len, stream));
        else {
            B.pscr.ensure(kbo::chunk_items_scratch_words((uint32_t)n_seqs) * sizeof(uint32_t));
            HIP_OK(kbo::launch_packed_prefix(B.off.as<uint64_t>(), (uint32_t)n_seqs, B.pscr.as<uint32_t>(), stream));
        }
    }
    // the bytes of a packed batch: only when something needs them (the one kernel takes the words as they are)
    const uint32_t wps = uniform ? (packed->uniform_len + 15u) / 16u : 0u;
    bool have_bytes = !packed;
    auto need_bytes = [&] {
        if (have_bytes) return;
        have_bytes = true;
        HIP_OK(kbo::launch_unpack2(B.packed.as<uint32_t>(), (uint32_t)packed->n_words, B.off.as<uint64_t>(), (uint32_t)n_seqs, wps,
                                   uniform ? nullptr : B.pscr.as<uint32_t>(), B.q.as<uint8_t>(), stream));
        HIP_OK(kbo::launch_exceptions(B.exc_pos.as<uint64_t>(), B.exc_byte.as<uint8_t>(), (uint32_t)packed->n_exc, packed->base,
                                      B.q.as<uint8_t>(), stream));
    };
    static const int env_native = std::getenv("KBO_PACKED_NATIVE") ? std::atoi(std::getenv("KBO_PACKED_NATIVE")) : 1; // experiments
    // (off: beside the copies and the next slabs' kernels the pass finishes late and the downloads wait for it - 600 Mbp host to host,
    // packed 118 -> 95 Gbp/s, bytes 39 -> 40 Gbp/s with a tail stream of the highest priority, 78 / 26 Gbp/s with an ordinary one)
    static const int env_tail = std::getenv("KBO_HOST_TAIL") ? std::atoi(std::getenv("KBO_HOST_TAIL")) : 0; // experiments
    // the stream the second pass of the one-kernel route goes to: the caller's tail stream behind the kernel (kbo_capi.cpp
    // map_batch_dev_impl has the pieces' size), or the kernel's own
    auto second_pass_stream = [&](kbo::WalkArgs &a) -> hipStream_t {
        if (!map || !map->tail || !map->fence || !env_tail) return stream;
        HIP_OK(hipEventRecord(map->fence, stream));
        HIP_OK(hipStreamWaitEvent(map->tail, map->fence, 0));
        a.redo_piece = 32u;
        map->results = map->tail;
        return map->tail;
    };
    if (device_items) HIP_OK(kbo::launch_make_items(B.off.as<uint64_t>(), (uint32_t)n_seqs, B.items.as<kbo::WalkItem>(), stream));
    for (size_t sh = 0; sh < shards.size(); sh++) {
        DevCopy::PlanState *plan_state = nullptr;
        const kbo::DevIndexView view = device_view(shards[sh], dev, &plan_state, total);
        if (sh > 0) B.ms_shard.ensure(padded);
        kbo::WalkArgs a{};
        a.ix = view;
        a.q = B.q.as<uint8_t>();
        a.q_bytes = total;
        a.items = B.items.as<kbo::WalkItem>();
        a.n_items = (uint32_t)n_items;
        a.rounds = 0;
        a.d_out = sh == 0 ? B.ms.as<uint8_t>() : B.ms_shard.as<uint8_t>();
        a.lo_out = want_ival ? B.lo.as<uint32_t>() : nullptr;
        a.hi_out = want_ival ? B.hi.as<uint32_t>() : nullptr;
        a.call_sites = call ? static_cast<uint4 *>(call->d_sites) : nullptr;
        a.call_counts = call ? call->d_counts : nullptr;
        a.call_cap = call ? call->cap_per_list : 0;
        a.call_thr = call ? call->threshold : 0;
        // (no item is longer than this: chunk + k - 1 warm-up bases; call mode: k warm-up + up to k borrowed bases)
        a.max_item_len = device_items ? longest_seq : (uint32_t)std::min<uint64_t>(chunk + (call ? 2ull * idx->host.k : idx->host.k), 0xFFFFFFFFu);
        if (view.pc_text && !want_ival) B.plan.ensure(kbo::plan_work_bytes(n_items, total));
        attach_plan(a, view.pc_text && !want_ival ? B.plan.p : nullptr, plan_state);
        if (map && shards.size() == 1 && device_items && !call) { // kbo::matches / map over reads: the one kernel where it applies
            a.chars_out = map->d_chars;
            a.map_thr = map->threshold;
            a.map_fmt = map->format ? 1u : 0u;
            a.map_want_ms = 0;
            if (packed && env_native && a.gitems && kbo::map_reads_packed_applies(a, map->d_packed_out != nullptr)) {
                // packed-native: the words go into the kernel as they are and (kbo_matches_batch_packed) the characters leave it
                // as words; only the reads it leaves to the plain walk get their bytes, and their characters are packed behind it
                a.qp = B.packed.as<uint32_t>();
                a.qp_wps = wps;
                a.qp_data = uniform ? nullptr : B.pscr.as<uint32_t>();
                a.qp_sums = uniform ? nullptr : B.pscr.as<uint32_t>() + n_seqs + 1u;
                a.packed_out = map->d_packed_out;
                if (packed->n_exc) {
                    B.exc_flag.ensure(n_seqs + 16);
                    HIP_OK(hipMemsetAsync(B.exc_flag.p, 0, n_seqs, stream));
                    HIP_OK(kbo::launch_flag_exceptions(B.exc_pos.as<uint64_t>(), (uint32_t)packed->n_exc, packed->base, B.off.as<uint64_t>(),
                                                       (uint32_t)n_seqs, B.exc_flag.as<uint8_t>(), stream));
                    a.qp_exc = B.exc_flag.as<uint8_t>();
                }
                const bool count_runs = map->run_counts && !map->format && !map->d_packed_out;
                if (count_runs) a.run_counts = map->run_counts;
                HIP_OK(kbo::launch_map_reads(a, stream));
                hipStream_t ts = second_pass_stream(a);
                HIP_OK(kbo::launch_unpack_flagged(a.qp, B.off.as<uint64_t>(), (uint32_t)n_seqs, wps, a.qp_data, a.redo, B.q.as<uint8_t>(), ts));
                HIP_OK(kbo::launch_exceptions(B.exc_pos.as<uint64_t>(), B.exc_byte.as<uint8_t>(), (uint32_t)packed->n_exc, packed->base,
                                              B.q.as<uint8_t>(), ts));
                HIP_OK(kbo::launch_redo_pass(a, ts));
                HIP_OK(kbo::launch_derand_flagged(B.ms.as<uint8_t>(), B.off.as<uint64_t>(), (uint32_t)n_seqs, idx->host.k, map->threshold,
                                                  map->format ? B.q.as<uint8_t>() : nullptr, map->d_chars, a.redo, longest_seq, ts,
                                                  count_runs ? map->run_counts : nullptr));
                if (map->d_packed_out) {
                    HIP_OK(kbo::launch_pack_flagged(map->d_chars, B.off.as<uint64_t>(), (uint32_t)n_seqs, wps, a.qp_data, a.redo, map->d_packed_out, ts));
                    map->packed_done = true;
                }
                if (count_runs) map->counted = true;
                plan_after_launch(a, ts, plan_state);
                map->done = true;
                return;
            }
            need_bytes();
            if (a.gitems && kbo::map_reads_applies(a)) {
                const bool count_runs = map->run_counts && !map->format && kbo::map_reads_direct(a);
                if (count_runs) a.run_counts = map->run_counts;
                HIP_OK(kbo::launch_map_reads(a, stream));
                hipStream_t ts = second_pass_stream(a);
                a.seq_off = B.off.as<uint64_t>(); // (item s is sequence s, whole: finish_reads_kernel reads the offsets)
                if (kbo::map_reads_finish_applies(a)) {
                    HIP_OK(kbo::launch_map_reads_finish(a, ts)); // the reads the kernel listed: walk, derandomize + translate, characters, runs
                } else {
                    a.seq_off = nullptr;
                    HIP_OK(kbo::launch_redo_pass(a, ts));
                    HIP_OK(kbo::launch_derand_flagged(B.ms.as<uint8_t>(), B.off.as<uint64_t>(), (uint32_t)n_seqs, idx->host.k, map->threshold,
                                                      map->format ? B.q.as<uint8_t>() : nullptr, map->d_chars, a.redo, longest_seq, ts,
                                                      count_runs ? map->run_counts : nullptr)); // (the flagged reads' runs counted on the way)
                }
                if (count_runs) map->counted = true;
                plan_after_launch(a, ts, plan_state);
                map->done = true;
                return;
            }
        }
        if (map && shards.size() == 1 && !call && !want_ival && longest_seq > 160u && kbo::map_long_applies(view, map->threshold)) {
            // kbo::matches / map / find over sequences of more than 160 bases - contigs, whole reference sequences, long reads:
            // one wave per piece of a sequence (long_kernels.hip), its flagged pieces by the plain walk + the literal recurrences
            const size_t wb = kbo::long_work_bytes(n_seqs, total, idx->host.k);
            if (wb) {
                need_bytes();
                B.longw.ensure(wb);
                kbo::LongArgs la{};
                HIP_OK(kbo::launch_map_long(view, B.q.as<uint8_t>(), B.off.as<uint64_t>(), (uint32_t)n_seqs, total, map->threshold, map->format,
                                            map->d_chars, B.longw.p, stream, la, g_plan_stats.load()));
                HIP_OK(kbo::launch_map_long_redo(la, B.ms.as<uint8_t>(), stream));
                map->done = true;
                return;
            }
        }
        need_bytes();
        HIP_OK(kbo::launch_ms_walk(a, walk_max_waves(), stream));
        plan_after_launch(a, stream, plan_state);
        // the depth against the union of the shards is the maximum of the depths against each (capi_internal.hpp)
        if (sh > 0) HIP_OK(kbo::launch_max_bytes(B.ms.as<uint8_t>(), B.ms_shard.as<uint8_t>(), total, stream));
    }
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
std::atomic<int> g_host_in_place{std::getenv("KBO_HOST_INPLACE") ? std::atoi(std::getenv("KBO_HOST_INPLACE")) : 0};
std::atomic<size_t> g_slab_bytes{16ull << 20}; // tools/bench_host.py, 600 Mbp of C2 reads through the one kernel: 16 / 24 / 32 / 64 MiB 39 / 36 / 32-39 / 31 Gbp/s (bytes), 121 / 89 / 104 Gbp/s (packed)


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
    // the last slab in halves and quarters: what the pipeline cannot hide is the last slab's way through it (upload, kernels,
    // download, copy out - 1.1 ms of a 4.8 ms call over 600 Mbp of packed reads), and that way is shorter for a smaller slab
    static const int env_taper = std::getenv("KBO_SLAB_TAPER") ? std::atoi(std::getenv("KBO_SLAB_TAPER")) : 1; // experiments
    if (env_taper && slabs.size() >= 3) {
        const Slab last = slabs.back();
        const size_t n = last.s1 - last.s0;
        if (n >= 64) {
            slabs.pop_back();
            const size_t cut[4] = {last.s0, last.s0 + n / 2, last.s0 + n / 2 + n / 4, last.s1};
            for (int i = 0; i < 3; i++) slabs.push_back(Slab{cut[i], cut[i + 1], offsets[cut[i]], offsets[cut[i + 1]]});
        }
    }
    return slabs;
}

// Slabs of a packed batch: four times the bases of a byte slab (the same bytes over PCIe).  Large on purpose: the guided
// walk has a fixed cost of about 0.19 ms per launch whatever the slab holds (its longest chain of units; a slab of 56 k reads
// spends 0.4 ms in kernels, 21 Gbp/s, one of 894 k reads 1.1 ms, 122 Gbp/s), so slabs that start small and grow - tried, to
// shorten the pipeline's unhidden first upload and last download - lose more than they hide (tools/bench_host.py PACKED=1,
// 600 Mbp: equal slabs of 32 / 64 / 128 MiB of bases 58 / 72 / 81 Gbp/s, ramped 8 .. 128 MiB 69).
size_t slab_bytes_for(const kbo_index *idx)
{
    const size_t v = idx ? idx->opts.slab_bytes.load() : 0;
    return v ? v : g_slab_bytes.load();
}
size_t packed_slab_bytes(const kbo_index *idx) { return std::min<size_t>(4 * slab_bytes_for(idx), 0xC0000000ull); }

// one pass over the offsets of a batch: order, emptiness, shortest and longest sequence
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
    // (off by default: on the MI355X boxes here the copies between a caller's large pinned buffers and the device run at 28 GB/s
    // each way, those between the slots' small staging buffers and the device at 38 - the host team's copies included: 600 Mbp
    // of pinned reads 27.7 in place against 34.9 Gbp/s staged, packed 104 against 123.  kbo_set_host_in_place(1) / KBO_HOST_INPLACE=1)
    if (!g_host_in_place.load()) return false;
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
    PinBuf in, out, off, lo_pin, hi_pin;
    std::vector<kbo::WalkItem> items;
    hipEvent_t copied = nullptr, computed = nullptr, done = nullptr, fence = nullptr;
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
    hipStream_t st_tail = nullptr; // the second pass of the one-kernel route: beside the next slab's kernel
    explicit HostCtx(int d) : dev(d)
    {
        for (hipStream_t *st : {&st_up, &st_run, &st_down}) HIP_OK(hipStreamCreateWithFlags(st, hipStreamNonBlocking));
        if (std::getenv("KBO_HOST_TAIL") && std::atoi(std::getenv("KBO_HOST_TAIL")) != 0) { // (experiment: enqueue_walk_host; every stream takes part of a hardware queue)
            int pr_lo = 0, pr_hi = 0; // (the second pass is a chain of dependent look-ups of a few waves: it goes first wherever it can)
            HIP_OK(hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi));
            HIP_OK(hipStreamCreateWithPriority(&st_tail, hipStreamNonBlocking, pr_hi));
        }
        for (HostSlot &S : slot)
            for (hipEvent_t *e : {&S.copied, &S.computed, &S.done, &S.fence}) HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    ~HostCtx()
    {
        int prev = 0;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(dev);
        for (hipStream_t st : {st_up, st_run, st_down, st_tail})
            if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
        for (HostSlot &S : slot)
            for (hipEvent_t e : {S.copied, S.computed, S.done, S.fence})
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
            for (hipStream_t st : {ctx->st_up, ctx->st_run, ctx->st_down, ctx->st_tail})
                if (st) (void)hipStreamSynchronize(st);
        // keep at most kPooledPerDevice contexts per device (each holds ~0.8 GB of device and ~0.3 GB of
        // pinned memory at the default slab size); the scratch of further concurrent callers is freed
        std::unique_lock<std::mutex> g(g_ctx_mu);
        size_t same = 0;
        for (const auto &c : g_ctx_pool) same += c->dev == ctx->dev;
        if (same < kPooledPerDevice) {
            g_ctx_pool.push_back(std::move(ctx));
            return;
        }
        g.unlock();
        ctx.reset(); // ~HostCtx switches to its device and back
    }
    static constexpr size_t kPooledPerDevice = 2;
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
                                   int32_t *d_derand, hipStream_t stream, uint32_t longest, DevBuf *piece_work)
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


// kbo::matches over a batch (lib.rs:618-627); optional relative_to_ref (lib.rs:756-757); with a sink
// the characters are turned into run lengths on the device instead of being downloaded (lib.rs:816-820)
namespace {

// what a batch call hands to its per-device workers
struct BatchJob {
    kbo_index *idx;
    const uint8_t *concat;
    const uint64_t *offsets;
    uint32_t k, threshold;
    bool format;         // apply format::relative_to_ref
    uint8_t *chars_out;  // nullptr when a sink takes run lengths instead, or in ms mode
    RleSink *sink;
    uint8_t *ms_out = nullptr;             // ms mode: A1 only, the MS values come back ...
    uint32_t *lo_out = nullptr, *hi_out = nullptr; // ... with the intervals when these are given
    bool sink_direct;    // one worker: records go straight into sink->all
    bool in_pinned, out_pinned; // user buffers the DMA engines reach directly are used in place
    const std::vector<Slab> *slabs;
    PhaseClock *clk;     // phase timing (worker 0 only)
    // packed mode (kbo_matches_batch_packed / kbo_find_batch_packed): 2-bit words in, 2-bit words (or run lengths) out
    const PackedBatch *packed = nullptr;
    uint32_t *packed_out = nullptr;
    const uint64_t *pw = nullptr;   // first word of every sequence (n_seqs + 1), nullptr when ...
    uint32_t uniform_len = 0;       // ... all sequences have this many bases
    uint64_t word_of(size_t s) const { return pw ? pw[s] : (uint64_t)s * ((uniform_len + 15u) / 16u); }
};

// One device's share of a batch: slabs `first`, `first + stride`, ... rotate through the slots of a
// leased HostCtx.  The calling thread stages and submits slabs; a second thread completes them in
// submission order (downloads, copies the staged output to the user's memory), so the two host
// copies of a slab never queue behind each other.
class SlabWorker {
public:
    SlabWorker(const BatchJob &job, int device, size_t first, size_t stride, bool timed)
        : job_(job), device_(device), first_(first), stride_(stride), timed_(timed)
    {
    }

    void run()
    {
        HIP_OK(hipSetDevice(device_));
        CtxLease lease(device_);
        C_ = lease.ctx.get();
        std::thread drainer([this] { drain_loop(); });
        auto join_drainer = [&] {
            {
                std::lock_guard<std::mutex> g(mu_);
                stop_ = true;
            }
            cv_.notify_all();
            if (drainer.joinable()) drainer.join();
        };
        try {
            size_t turn = 0;
            for (size_t i = first_; i < job_.slabs->size(); i += stride_, turn++) {
                {
                    std::unique_lock<std::mutex> g(mu_);
                    cv_.wait(g, [&] { return turn < drained_ + kHostSlots; }); // the slot is free again
                    if (drain_code_ != KBO_OK) break;
                }
                lap("  wait for a free slot");
                submit(turn, i);
                {
                    std::lock_guard<std::mutex> g(mu_);
                    submitted_++;
                }
                cv_.notify_all();
                lap("  enqueue");
            }
        } catch (...) {
            join_drainer();
            throw;
        }
        join_drainer();
        if (drain_code_ != KBO_OK) throw KboError(drain_code_, drain_error_);
        lap("drain");
    }

private:
    void lap(const char *what)
    {
        if (timed_) job_.clk->lap(what);
    }
    HostSlot &slot(size_t turn) { return C_->slot[turn % kHostSlots]; }

    // ---- submitting thread: stage the slab, enqueue upload, kernels and (characters) the download
    void submit(size_t turn, size_t slab_id)
    {
        const Slab &sl = (*job_.slabs)[slab_id];
        HostSlot &S = slot(turn);
        HostCtx &C = *C_;
        HostTeam &team = HostTeam::get();
        const size_t ns = sl.s1 - sl.s0;
        const uint64_t bytes = sl.b1 - sl.b0;
        S.busy = true; // from the first enqueue on: if anything below throws, ~CtxLease drains the streams before the
                       // context (and the caller's pinned buffers the copies read) can be reused
        // stage: slab-relative offsets (and the longest sequence of the slab), query bytes
        S.off.ensure((ns + 1) * sizeof(uint64_t));
        uint64_t *off = S.off.as<uint64_t>();
        const uint64_t *offsets = job_.offsets;
        uint32_t mx = 0;
        if (job_.packed && job_.uniform_len && job_.uniform_len <= 255u) { // (reads: below every chunk length, walk_chunk() >= 256)
            // equally long reads, packed: the offsets are made on the device and nothing below reads more of the host
            // copy than its last entry (one item per read, A5/A6 by the longest length)
            off[0] = 0;
            off[ns] = bytes;
            mx = job_.uniform_len;
        } else {
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
            mx = (uint32_t)*std::max_element(longest.begin(), longest.end());
        }
        const uint8_t *src = job_.concat ? job_.concat + sl.b0 : nullptr;
        PackedIn pin{};
        const uint64_t w0 = job_.packed ? job_.word_of(sl.s0) : 0, w1 = job_.packed ? job_.word_of(sl.s1) : 0;
        if (job_.packed) {
            pin.words = job_.packed->words + w0;
            pin.n_words = (size_t)(w1 - w0);
            if (!job_.in_pinned) {
                S.in.ensure(pin.n_words * 4 + 16);
                team.copy(S.in.p, pin.words, pin.n_words * 4);
                pin.words = S.in.as<uint32_t>();
            }
            const uint64_t *e0 = std::lower_bound(job_.packed->exc_pos, job_.packed->exc_pos + job_.packed->n_exc, sl.b0);
            const uint64_t *e1 = std::lower_bound(e0, job_.packed->exc_pos + job_.packed->n_exc, sl.b1);
            pin.exc_pos = e0;
            pin.exc_byte = job_.packed->exc_byte + (e0 - job_.packed->exc_pos);
            pin.n_exc = (size_t)(e1 - e0);
            pin.base = sl.b0;
            pin.uniform_len = job_.uniform_len;
        } else if (!job_.in_pinned) {
            S.in.ensure(bytes);
            team.copy(S.in.p, src, bytes);
            src = S.in.as<uint8_t>();
        }
        lap("  offsets + copy in");
        FusedMap fm{nullptr, job_.threshold, job_.format && !job_.sink};
        fm.tail = C.st_tail;
        fm.fence = S.fence;
        if (job_.sink && job_.sink->max_gap_len == 0) { // kbo::find, FindOpts' default: the one kernel counts the runs itself
            S.rle_scratch.ensure(kbo::chunk_items_scratch_words((uint32_t)ns) * sizeof(uint32_t));
            fm.run_counts = S.rle_scratch.as<uint32_t>();
        }
        if (!job_.ms_out) { // kbo::matches / map / find: the characters' buffer first, so that the one kernel can write into it
            S.chars.ensure(((bytes + 15) / 16) * 16 + 32);
            fm.d_chars = S.chars.as<uint8_t>();
            if (job_.packed_out && !job_.sink) { // (the words' buffer as well: the packed-native kernel writes them itself)
                S.B.packed_out.ensure((size_t)(w1 - w0) * 4 + 16);
                fm.d_packed_out = S.B.packed_out.as<uint32_t>();
            }
        }
        enqueue_walk_host(job_.idx, src, off, ns, job_.lo_out != nullptr, S.B, S.items, C.st_run, mx, C.st_up, S.copied, nullptr,
                          job_.packed ? &pin : nullptr, job_.ms_out ? nullptr : &fm);
        // (the one-kernel route with its second pass on the tail stream: what follows the characters follows them there)
        hipStream_t st_res = (fm.done && fm.results) ? fm.results : C.st_run;
        if (job_.ms_out) { // A1 only: MS values (and intervals) straight back
            HIP_OK(hipEventRecord(S.computed, C.st_run));
            HIP_OK(hipStreamWaitEvent(C.st_down, S.computed, 0));
            uint8_t *dst = job_.ms_out + sl.b0;
            uint32_t *dlo = job_.lo_out ? job_.lo_out + sl.b0 : nullptr, *dhi = job_.hi_out ? job_.hi_out + sl.b0 : nullptr;
            if (!job_.out_pinned) {
                S.out.ensure(bytes + 32);
                dst = S.out.as<uint8_t>();
                if (dlo) {
                    S.lo_pin.ensure(bytes * sizeof(uint32_t));
                    S.hi_pin.ensure(bytes * sizeof(uint32_t));
                    dlo = S.lo_pin.as<uint32_t>();
                    dhi = S.hi_pin.as<uint32_t>();
                }
            }
            HIP_OK(hipMemcpyAsync(dst, S.B.ms.p, bytes, hipMemcpyDeviceToHost, C.st_down));
            if (dlo) {
                HIP_OK(hipMemcpyAsync(dlo, S.B.lo.p, bytes * sizeof(uint32_t), hipMemcpyDeviceToHost, C.st_down));
                HIP_OK(hipMemcpyAsync(dhi, S.B.hi.p, bytes * sizeof(uint32_t), hipMemcpyDeviceToHost, C.st_down));
            }
            HIP_OK(hipEventRecord(S.done, C.st_down));
        } else if (job_.sink) {
            // characters stay on the device; run lengths are counted, scanned and (speculatively, into
            // the room the slot has) emitted right behind A5/A6; the completing thread downloads them
            if (!fm.done)
                derand_translate_host_offsets(S.B.ms.as<uint8_t>(), S.B.off.as<uint64_t>(), off, ns, job_.k, job_.threshold,
                                              nullptr, S.chars.as<uint8_t>(), nullptr, st_res, mx, &S.dt_work);
            const uint32_t gap = (uint32_t)std::min<size_t>(job_.sink->max_gap_len, 0xFFFFFFFFu);
            S.rle_scratch.ensure(kbo::chunk_items_scratch_words((uint32_t)ns) * sizeof(uint32_t));
            S.rle_total.ensure(16);
            S.rle_total_pin.ensure(16);
            if (S.rle_capacity < 2 * ns + 16) {
                S.rle_capacity = 2 * ns + 16;
                S.rles.ensure(S.rle_capacity * kRleWords * sizeof(uint32_t));
            }
            if (fm.done && fm.counted) // (counts are there: scan + total)
                HIP_OK(kbo::launch_rle_scan_counts((uint32_t)ns, S.rle_scratch.as<uint32_t>(), S.rle_total.as<uint32_t>(), st_res));
            else
                HIP_OK(kbo::launch_rle_count(S.chars.as<uint8_t>(), S.B.off.as<uint64_t>(), (uint32_t)ns, gap,
                                             S.rle_scratch.as<uint32_t>(), S.rle_total.as<uint32_t>(), st_res, mx, true));
            HIP_OK(hipMemcpyAsync(S.rle_total_pin.p, S.rle_total.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st_res));
            HIP_OK(kbo::launch_rle_emit(S.chars.as<uint8_t>(), S.B.off.as<uint64_t>(), (uint32_t)ns, gap,
                                        S.rle_scratch.as<uint32_t>(), S.rles.as<uint32_t>(), (uint32_t)S.rle_capacity,
                                        st_res, mx, true)); // (the characters are the kernels' own: M - X R)
            S.longest = mx;
            HIP_OK(hipEventRecord(S.computed, st_res));
            HIP_OK(hipStreamWaitEvent(C.st_down, S.computed, 0));
            S.slab_id = slab_id;
            S.n_seqs = ns;
        } else {
            // D2H leg: hipMemcpyAsync on the download stream.  With one stream per stage the copy
            // engines carry both directions at once (tools/bench_host.py: 37-40 Gbp/s host->host;
            // a small kernel storing into pinned memory, or A5/A6 storing there themselves, gave
            // 28 and 26 Gbp/s).
            uint8_t *dst = job_.packed_out ? nullptr : job_.chars_out + sl.b0;
            if (!job_.out_pinned && !job_.packed_out) {
                S.out.ensure(bytes + 32);
                dst = S.out.as<uint8_t>();
            }
            if (!fm.done)
                derand_translate_host_offsets(S.B.ms.as<uint8_t>(), S.B.off.as<uint64_t>(), off, ns, job_.k, job_.threshold,
                                              job_.format ? S.B.q.as<uint8_t>() : nullptr, S.chars.as<uint8_t>(), nullptr,
                                              st_res, mx, &S.dt_work);
            if (job_.packed_out) { // the characters leave as 2-bit words: a quarter of the bytes
                const size_t nw = (size_t)(w1 - w0);
                S.B.packed_out.ensure(nw * 4 + 16);
                if (!fm.packed_done)
                    HIP_OK(kbo::launch_pack2(S.chars.as<uint8_t>(), (uint32_t)nw, S.B.off.as<uint64_t>(), (uint32_t)ns,
                                             job_.uniform_len ? (job_.uniform_len + 15u) / 16u : 0u,
                                             job_.uniform_len ? nullptr : S.B.pscr.as<uint32_t>(), S.B.packed_out.as<uint32_t>(), st_res));
                HIP_OK(hipEventRecord(S.computed, st_res));
                HIP_OK(hipStreamWaitEvent(C.st_down, S.computed, 0));
                uint8_t *pdst = reinterpret_cast<uint8_t *>(job_.packed_out + w0);
                if (!job_.out_pinned) {
                    S.out.ensure(nw * 4 + 32);
                    pdst = S.out.as<uint8_t>();
                }
                HIP_OK(hipMemcpyAsync(pdst, S.B.packed_out.p, nw * 4, hipMemcpyDeviceToHost, C.st_down));
                HIP_OK(hipEventRecord(S.done, C.st_down));
                S.busy = true;
                S.out_b0 = w0 * 4; // (finish() copies out_bytes bytes to chars_out + out_b0: chars_out is the packed buffer here)
                S.out_bytes = nw * 4;
                return;
            }
            HIP_OK(hipEventRecord(S.computed, st_res));
            HIP_OK(hipStreamWaitEvent(C.st_down, S.computed, 0));
            HIP_OK(hipMemcpyAsync(dst, S.chars.p, bytes, hipMemcpyDeviceToHost, C.st_down));
            HIP_OK(hipEventRecord(S.done, C.st_down));
        }
        S.busy = true;
        S.out_b0 = sl.b0;
        S.out_bytes = bytes;
    }

    // ---- completing thread
    // run lengths: the number of records of a slab is known once its kernels are done, so the download
    // is issued here; it is issued for the next slab before the previous one is copied out, so that the
    // copy engine and the host copy work on different slabs
    void start_download(size_t turn)
    {
        if (!job_.sink) return;
        HostSlot &S = slot(turn);
        HostCtx &C = *C_;
        HIP_OK(hipEventSynchronize(S.computed));
        const uint32_t total = *S.rle_total_pin.as<uint32_t>();
        if (total > S.rle_capacity) { // more runs than the speculative emit had room for
            S.rle_capacity = (size_t)total + total / 4 + 16;
            S.rles.ensure(S.rle_capacity * kRleWords * sizeof(uint32_t));
            HIP_OK(kbo::launch_rle_emit(S.chars.as<uint8_t>(), S.B.off.as<uint64_t>(), (uint32_t)S.n_seqs,
                                        (uint32_t)std::min<size_t>(job_.sink->max_gap_len, 0xFFFFFFFFu),
                                        S.rle_scratch.as<uint32_t>(), S.rles.as<uint32_t>(), (uint32_t)S.rle_capacity,
                                        C.st_down, S.longest, true));
        }
        const size_t words = kbo::chunk_items_scratch_words((uint32_t)S.n_seqs);
        S.out.ensure(std::max<size_t>(16, (size_t)total * kRleWords * sizeof(uint32_t)));
        S.rle_first_pin.ensure(words * sizeof(uint32_t));
        if (total)
            HIP_OK(hipMemcpyAsync(S.out.p, S.rles.p, (size_t)total * kRleWords * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                  C.st_down));
        HIP_OK(hipMemcpyAsync(S.rle_first_pin.p, S.rle_scratch.p, words * sizeof(uint32_t), hipMemcpyDeviceToHost, C.st_down));
        HIP_OK(hipEventRecord(S.done, C.st_down));
        S.out_bytes = total; // records
    }

    void finish(size_t turn)
    {
        HostSlot &S = slot(turn);
        RleSink *sink = job_.sink;
        HIP_OK(hipEventSynchronize(S.done));
        if (job_.ms_out) {
            if (!job_.out_pinned) {
                HostTeam::out().copy(job_.ms_out + S.out_b0, S.out.p, S.out_bytes);
                if (job_.lo_out) {
                    HostTeam::out().copy(job_.lo_out + S.out_b0, S.lo_pin.p, S.out_bytes * sizeof(uint32_t));
                    HostTeam::out().copy(job_.hi_out + S.out_b0, S.hi_pin.p, S.out_bytes * sizeof(uint32_t));
                }
            }
        } else if (!sink) {
            if (!job_.out_pinned) HostTeam::out().copy(job_.chars_out + S.out_b0, S.out.p, S.out_bytes);
        } else {
            const size_t total = S.out_bytes;
            const uint32_t *local = S.rle_first_pin.as<uint32_t>(), *sums = local + S.n_seqs + 1;
            if (job_.sink_direct) {
                if (sink->all_used + total > sink->all_cap && !sink->caller_owns) {
                    const size_t cap = (sink->all_used + total) * 2;
                    if (sink->compact) {
                        uint32_t *p = static_cast<uint32_t *>(std::realloc(sink->all32, cap * kRleWords * sizeof(uint32_t)));
                        if (!p) throw std::bad_alloc();
                        sink->all32 = p;
                    } else {
                        kbo_rle *p = static_cast<kbo_rle *>(std::realloc(sink->all, cap * sizeof(kbo_rle)));
                        if (!p) throw std::bad_alloc();
                        sink->all = p;
                    }
                    sink->all_cap = cap;
                }
                const size_t base = sink->all_used, s0 = (*job_.slabs)[S.slab_id].s0, ns_slab = S.n_seqs;
                if (sink->compact) // the device's records as they are
                    HostTeam::out().copy(sink->all32 + base * kRleWords, S.out.p, total * kRleWords * sizeof(uint32_t));
                else if (base + total <= sink->all_cap) // a caller's buffer that is too small only gets the count
                    widen_rles(sink->all + base, S.out.as<uint32_t>(), total, HostTeam::out());
                const size_t piece = 1u << 15;
                HostTeam::out().run((ns_slab + piece - 1) / piece, [&](size_t t) {
                    const size_t a = t * piece + 1, b = std::min(ns_slab, a + piece - 1);
                    for (size_t q = a; q <= b; q++) sink->rle_offsets[s0 + q] = base + sums[q / 1024] + local[q];
                });
                sink->all_used += total;
            } else {
                if (sink->compact) {
                    sink->runs32[S.slab_id].assign(S.out.as<uint32_t>(), S.out.as<uint32_t>() + total * kRleWords);
                } else {
                    std::vector<kbo_rle> &runs = sink->runs[S.slab_id];
                    runs.resize(total);
                    widen_rles(runs.data(), S.out.as<uint32_t>(), total, HostTeam::out());
                }
                std::vector<uint32_t> &first = sink->first[S.slab_id];
                first.resize(S.n_seqs + 1);
                for (size_t q = 0; q <= S.n_seqs; q++) first[q] = sums[q / 1024] + local[q];
            }
        }
        S.busy = false;
        {
            std::lock_guard<std::mutex> g(mu_);
            drained_++;
        }
        cv_.notify_all();
    }

    void drain_loop()
    {
        auto fail = [&](int code, const char *what) {
            std::lock_guard<std::mutex> g(mu_);
            drain_code_ = code;
            drain_error_ = what;
            drained_ = ~size_t(0) / 2; // releases the submitting thread
            cv_.notify_all();
        };
        try {
            HIP_OK(hipSetDevice(device_));
            const size_t none = ~size_t(0);
            size_t started = 0, pending = none;
            for (;;) {
                bool can_start;
                {
                    std::unique_lock<std::mutex> g(mu_);
                    cv_.wait(g, [&] { return started < submitted_ || pending != none || stop_; });
                    can_start = started < submitted_;
                    if (!can_start && pending == none) return;
                }
                const size_t prev = pending;
                pending = none;
                if (can_start) {
                    start_download(started);
                    pending = started++;
                }
                if (prev != none) finish(prev);
            }
        } catch (const KboError &e) {
            fail(e.code, e.what());
        } catch (const std::bad_alloc &) {
            fail(KBO_E_NOMEM, "out of host memory");
        } catch (const std::exception &e) {
            fail(KBO_E_HIP, e.what());
        }
    }

    const BatchJob &job_;
    const int device_;
    const size_t first_, stride_;
    const bool timed_;
    HostCtx *C_ = nullptr;
    std::mutex mu_;
    std::condition_variable cv_;
    size_t submitted_ = 0, drained_ = 0;
    bool stop_ = false;
    int drain_code_ = KBO_OK;
    std::string drain_error_;
};

// one worker per device (index replicated on each, slabs dealt round-robin, disjoint output slices: no
// exchange between devices); a single device runs on the calling thread
void run_on_devices(const BatchJob &job, const std::vector<int> &devices, size_t nd)
{
    if (nd == 1) {
        const int prev = current_device();
        struct Restore { // also when run() throws
            int prev, used;
            ~Restore() { if (prev != used) (void)hipSetDevice(prev); }
        } restore{prev, devices[0]};
        SlabWorker(job, devices[0], 0, 1, true).run();
        return;
    }
    std::vector<std::thread> threads;
    std::vector<std::string> errors(nd);
    std::vector<int> codes(nd, KBO_OK);
    for (size_t w = 0; w < nd; w++)
        threads.emplace_back([&, w] {
            try {
                SlabWorker(job, devices[w], w, nd, w == 0).run();
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

} // namespace

void matches_batch_impl(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                        double max_error_prob, bool format, uint8_t *chars_out, RleSink *sink)
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
    const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, slab_bytes_for(idx));
    std::vector<int> devices = devices_for(idx);
    if (devices.empty()) devices.push_back(current_device());
    const size_t nd = std::min(devices.size(), std::max<size_t>(1, slabs.size()));
    BatchJob job;
    job.idx = idx;
    job.concat = concat;
    job.offsets = offsets;
    job.k = (uint32_t)k;
    job.threshold = (uint32_t)threshold;
    job.format = format;
    job.chars_out = chars_out;
    job.sink = sink;
    job.sink_direct = sink && nd == 1;
    job.in_pinned = is_pinned_host(concat);
    job.out_pinned = sink || is_pinned_host(chars_out);
    job.slabs = &slabs;
    job.clk = &clk;
    if (sink) {
        sink->runs.assign(slabs.size(), {});
        sink->first.assign(slabs.size(), {});
    }
    if (job.sink_direct && !sink->caller_owns) { // room for 2 runs per sequence to start with (untouched pages cost nothing)
        sink->all_cap = 2 * n_seqs + 1024;
        sink->all = static_cast<kbo_rle *>(std::malloc(sink->all_cap * sizeof(kbo_rle)));
        if (!sink->all) throw std::bad_alloc();
    }
    if (sink) sink->direct = job.sink_direct;
    if (job.sink_direct) sink->rle_offsets[0] = 0;
    clk.lap("slab list");
    run_on_devices(job, devices, nd);
}

// kbo::matches / kbo::find over a batch of 2-bit packed reads: the same pipeline, a quarter of the bytes over PCIe each way
void matches_batch_packed_impl(kbo_index *idx, const PackedBatch &in, const uint64_t *offsets, size_t n_seqs, double max_error_prob,
                               uint32_t *packed_out, RleSink *sink)
{
    KBO_REQUIRE(idx && in.words && (packed_out || sink), KBO_E_BAD_ARG, "null argument");
    KBO_REQUIRE(in.n_exc == 0 || (in.exc_pos && in.exc_byte), KBO_E_BAD_ARG, "null exception list");
    PhaseClock clk;
    const size_t k = idx->host.k;
    const size_t threshold = random_match_threshold(k, idx->host.n_kmers, 4, max_error_prob); // lib.rs:620
    KBO_REQUIRE(offsets, KBO_E_BAD_ARG, "null offsets");
    KBO_REQUIRE(n_seqs > 0, KBO_E_EMPTY_QUERY, "no sequences");
    KBO_REQUIRE(n_seqs < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "more than 2^32-1 sequences per call");
    KBO_REQUIRE(offsets[0] == 0, KBO_E_BAD_ARG, "offsets[0] must be 0");
    const OffsetScan scan = scan_offsets(offsets, n_seqs);
    KBO_REQUIRE(scan.monotone, KBO_E_BAD_ARG, "offsets not monotone");
    KBO_REQUIRE(scan.shortest > 0, KBO_E_EMPTY_QUERY, "empty query (index.rs:248 assert!(!query.is_empty()))");
    KBO_REQUIRE(scan.longest < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "sequence longer than 2^32-1");
    KBO_REQUIRE(threshold > 1, KBO_E_THRESHOLD_LE_1, "threshold > 1 (derandomize.rs:275, translate.rs:269)");
    KBO_REQUIRE(scan.shortest > 2, KBO_E_LEN_LE_2, "len > 2 (derandomize.rs:276, translate.rs:270)");
    for (size_t x = 0; x < in.n_exc; x++) // (ascending, inside the batch: the slabs cut the list by binary search)
        KBO_REQUIRE(in.exc_pos[x] < offsets[n_seqs] && (x == 0 || in.exc_pos[x] > in.exc_pos[x - 1]), KBO_E_BAD_ARG,
                    "exception positions must ascend and lie inside the batch");
    clk.lap("argument checks");
    const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, packed_slab_bytes(idx));
    std::vector<int> devices = devices_for(idx);
    if (devices.empty()) devices.push_back(current_device());
    const size_t nd = std::min(devices.size(), std::max<size_t>(1, slabs.size()));
    std::vector<uint64_t> pw;
    const bool uniform = scan.shortest == scan.longest;
    if (!uniform) { // first word of every sequence
        pw.resize(n_seqs + 1);
        pw[0] = 0;
        for (size_t s = 0; s < n_seqs; s++) pw[s + 1] = pw[s] + (offsets[s + 1] - offsets[s] + 15) / 16;
    }
    BatchJob job;
    job.idx = idx;
    job.concat = nullptr;
    job.offsets = offsets;
    job.k = (uint32_t)k;
    job.threshold = (uint32_t)threshold;
    job.format = false;
    job.chars_out = reinterpret_cast<uint8_t *>(packed_out); // (finish() copies bytes: the packed words of a slab)
    job.packed_out = sink ? nullptr : packed_out;
    job.sink = sink;
    job.sink_direct = sink && nd == 1;
    job.packed = &in;
    job.pw = uniform ? nullptr : pw.data();
    job.uniform_len = uniform ? (uint32_t)scan.longest : 0u;
    job.in_pinned = is_pinned_host(in.words);
    job.out_pinned = sink || is_pinned_host(packed_out);
    job.slabs = &slabs;
    job.clk = &clk;
    if (sink) {
        sink->runs.assign(slabs.size(), {});
        sink->runs32.assign(slabs.size(), {});
        sink->first.assign(slabs.size(), {});
    }
    if (job.sink_direct && !sink->caller_owns) {
        sink->all_cap = 2 * n_seqs + 1024;
        if (sink->compact) {
            sink->all32 = static_cast<uint32_t *>(std::malloc(sink->all_cap * kRleWords * sizeof(uint32_t)));
            if (!sink->all32) throw std::bad_alloc();
        } else {
            sink->all = static_cast<kbo_rle *>(std::malloc(sink->all_cap * sizeof(kbo_rle)));
            if (!sink->all) throw std::bad_alloc();
        }
    }
    if (sink) sink->direct = job.sink_direct;
    if (job.sink_direct) sink->rle_offsets[0] = 0;
    clk.lap("slab list");
    run_on_devices(job, devices, nd);
}

// A1 over a host batch: MS values (and intervals) only
void ms_batch_impl(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, uint8_t *d_out,
                   uint32_t *lo_out, uint32_t *hi_out)
{
    KBO_REQUIRE(idx && d_out, KBO_E_BAD_ARG, "null argument");
    KBO_REQUIRE((lo_out == nullptr) == (hi_out == nullptr), KBO_E_BAD_ARG, "lo/hi must come together");
    KBO_REQUIRE(concat && offsets, KBO_E_BAD_ARG, "null concat/offsets");
    KBO_REQUIRE(n_seqs > 0, KBO_E_EMPTY_QUERY, "no sequences");
    KBO_REQUIRE(n_seqs < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "more than 2^32-1 sequences per call");
    KBO_REQUIRE(offsets[0] == 0, KBO_E_BAD_ARG, "offsets[0] must be 0");
    const OffsetScan scan = scan_offsets(offsets, n_seqs);
    KBO_REQUIRE(scan.monotone, KBO_E_BAD_ARG, "offsets not monotone");
    KBO_REQUIRE(scan.shortest > 0, KBO_E_EMPTY_QUERY, "empty query (index.rs:248 assert!(!query.is_empty()))");
    KBO_REQUIRE(scan.longest < 0xFFFFFFFFull, KBO_E_UNSUPPORTED, "sequence longer than 2^32-1");
    PhaseClock clk;
    // intervals cost 8 more bytes per base on the device and on the way back: smaller slabs
    const std::vector<Slab> slabs = make_slabs(offsets, n_seqs, lo_out ? std::max<size_t>(1u << 16, slab_bytes_for(idx) / 4) : slab_bytes_for(idx));
    std::vector<int> devices = devices_for(idx);
    if (devices.empty()) devices.push_back(current_device());
    const size_t nd = std::min(devices.size(), std::max<size_t>(1, slabs.size()));
    BatchJob job;
    job.idx = idx;
    job.concat = concat;
    job.offsets = offsets;
    job.k = idx->host.k;
    job.threshold = 0;
    job.format = false;
    job.chars_out = nullptr;
    job.sink = nullptr;
    job.sink_direct = false;
    job.ms_out = d_out;
    job.lo_out = lo_out;
    job.hi_out = hi_out;
    job.in_pinned = is_pinned_host(concat);
    job.out_pinned = is_pinned_host(d_out) && (!lo_out || (is_pinned_host(lo_out) && is_pinned_host(hi_out)));
    job.slabs = &slabs;
    job.clk = &clk;
    run_on_devices(job, devices, nd);
}

void release_host_scratch()
{
    {
        std::lock_guard<std::mutex> g(g_ctx_mu);
        g_ctx_pool.clear();
    }
    // the calling thread's own caches (kbo_call / kbo_call_batch keep a transient-index arena and a small batch's device
    // buffers per host thread; pool threads free theirs when they exit)
    release_transient_arena();
}

} // namespace kbo_host
