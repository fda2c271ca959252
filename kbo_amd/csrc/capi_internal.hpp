// capi_internal.hpp — declarations shared by the translation units behind include/kbo_hip.h:
//   device_index.cpp  the index handle and its per-device copies,
//   host_batch.cpp    host batches: slabs, staging, the three-stage pipeline, run-length sink,
//   kbo_capi.cpp      the extern "C" entry points.
#pragma once
#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "host_util.hpp"
#include "kernels.hpp"
#include "refine.hpp"
#include "sbwt_index.hpp"

namespace kbo_host {

struct DevCopy {
    // (an index that serves one small batch borrows its arena from a buffer the calling thread keeps: kbo::call builds
    // one such index per sequence, and a hipMalloc / hipFree pair per sequence serialises every thread of the process)
    bool arena_borrowed = false;
    static bool &transient_arena_in_use()
    {
        static thread_local bool in_use = false;
        return in_use;
    }
    DevBuf arena; // rank blocks of A,C,G,T | null block | contraction entries (32-bit build) | two-base blocks
    DevBuf ent;   // contraction entries as their own allocation (big build)
    uint64_t n_blocks = 0;
    bool big = false;
    uint32_t pair_off = 0; // arena index of the two-base extension blocks, 0 = none
    DevBuf seed_tab;                 // intervals of all strings of seed_d bases (plan_kernel's seeds)
    uint32_t seed_d = 0;
    DevBuf dtab;                     // depth table (dtab_kernels.hip): 4^dtab_order bytes
    uint32_t dtab_order = 0;
    bool dtab_grouped = false;
    DevBuf anchor;                   // anchors of the depth table (kernels.hpp DevIndexView::anchor): 2^anchor_bits slots of 8 bytes
    uint32_t anchor_bits = 0;
    DevBuf fat;                      // recovery lines of the guided walk (sbwt_index.hpp)
    uint32_t fat_null = 0;
    DevBuf pc_text, pc_pos, pc_node; // path cover (sbwt_index.hpp PathCover), empty when the plan-guided walk is off
    DevBuf dfilt;                    // ... and the filter in front of the depth table (dfilt_bases bases per string; small indexes)
    uint32_t dfilt_bases = 0;
    DevBuf pc_tm, seed_pos;          // map_reads_kernel's 2-bit text + marks and its table of seed positions (copies with a depth table)
    // what making this copy cost (kbo_index_device_layout): seconds of host work / device builds / uploads, by part
    bool plan_built = false;         // path cover, recovery lines, tables: made by the first use that pays for them (device_index.cpp)
    bool plan_failed = false;        // ... or could not be made (no memory): not tried again on every call; the copy walks plainly
    uint64_t bases_seen = 0;         // bases of the batches that asked for this copy so far (guarded by the index's mutex)
    struct Setup {
        double layout_s = 0, cover_s = 0, lines_s = 0, seed_s = 0, dtab_s = 0, upload_s = 0;
        uint64_t rank_bytes = 0, entry_bytes = 0, pair_bytes = 0, cover_bytes = 0, lines_bytes = 0, seed_bytes = 0, dtab_bytes = 0,
                 anchor_bytes = 0;
    } setup;
    // Plan hold-off of THIS copy (one index on one device): a batch whose reads differ too much from the index gives the
    // plan up on the device; the host learns of it one launch late (asynchronous 8-byte copy into `bailed`, pinned, never
    // waited for) and then skips planning for the next kPlanHoldoff launches over this copy - and over no other.
    struct PlanState {
        std::atomic<int> holdoff{0};
        std::atomic<uint32_t> epoch{0};  // kbo_set_plan(1, ..) generation this state was last reset for
        uint32_t *bailed = nullptr;      // pinned: [0] plan given up, [1] a walk ended by its no-progress guard
        std::atomic<uint32_t> bails{0};  // launches that gave the plan up (inspection / tests)
    } plan;
    ~DevCopy()
    {
        if (plan.bailed) { // (a launch's 8-byte copy into it may still be in flight: plan_after_launch never waits for it)
            int prev = -1;
            (void)hipGetDevice(&prev);
            if (arena.dev >= 0 && arena.dev != prev) (void)hipSetDevice(arena.dev);
            (void)hipDeviceSynchronize();
            if (arena.dev >= 0 && arena.dev != prev && prev >= 0) (void)hipSetDevice(prev);
            (void)hipHostFree(plan.bailed);
        }
        if (arena_borrowed) {
            arena.p = nullptr;
            arena.cap = 0;
            transient_arena_in_use() = false;
        }
    }
};

} // namespace kbo_host

// per-handle options (kbo_index_set_opts, kbo_hip.h): what the process-wide knobs of kbo_hip_tuning.h set for every index, set
// for this one; kOptInherit = follow the process-wide value
constexpr int kOptInherit = INT32_MIN;
struct HandleOpts {
    std::atomic<int> plan{kOptInherit};                // 0 = plain walk only, 1 = plan structures + planned launches
    std::atomic<int> depth_table{kOptInherit};         // as kbo_set_depth_table: 0 by index size, < 0 none, else the order
    std::atomic<int> depth_table_anchors{kOptInherit}; // as kbo_set_depth_table_anchors
    std::atomic<size_t> slab_bytes{0};                 // 0 = inherit
    int n_devices = -1;                                // -1 = inherit (guarded by kbo_index::mu, with `devices`)
    std::vector<int> devices;
};

struct kbo_index {
    kbo::HostIndex host;
    std::mutex mu;
    HandleOpts opts;
    std::map<int, kbo_host::DevCopy *> dev;
    uint64_t rank_bytes = 0, lcs_bytes = 0, plan_bytes = 0;
    bool transient = false; // an index that serves one small batch (kbo::call builds one per sequence): no path cover
    // the path cover of the plan-guided walk once it has been computed (or read from an index file), guarded by `mu`:
    // every device copy uploads it from here, kbo_index_save writes it (laying it out is a 26 s pointer chase per 10^8 rows)
    std::unique_ptr<kbo::PathCover> cover;
    // A SHARDED index (kbo_capi.cpp build_sharded): an index whose rows would not fit 32-bit row numbers (a human genome
    // with its reverse complements: 6.2 * 10^9 rows) is built as several ordinary indexes over disjoint parts of the input -
    // groups of sequences, forward and reverse-complement strands apart - and this handle only holds them: host.k,
    // host.n_kmers (distinct k-mers of the union: what the threshold needs) and host.n_sets (rows over all shards) are set,
    // host.rows / host.lcs are empty.  The depth of the walk against the union index is the maximum of the depths against
    // the shards, so everything that only needs depths (matches / map without refinement / find / ms without intervals)
    // walks every shard and keeps the maximum; what needs rows of the union (intervals, call, fill_gaps, export) is refused.
    std::vector<std::unique_ptr<kbo_index>> shards;
    bool sharded() const { return !shards.empty(); }
    ~kbo_index()
    {
        for (auto &kv : dev) delete kv.second;
    }
};

namespace kbo_host {

// ---- tuning state (set through the kbo_set_* entry points)
extern std::atomic<int> g_waves_per_cu;           // walk: resident waves per CU, 0 = default (32)
extern std::vector<int> g_devices;   // guarded by g_devices_mu: read it through devices_snapshot()
extern std::mutex g_devices_mu;
std::vector<int> devices_snapshot();
// the same knobs as one index sees them (its own options first: HandleOpts)
bool plan_enabled(const kbo_index *idx);
int depth_table_setting(const kbo_index *idx);
int depth_table_anchor_setting(const kbo_index *idx);
size_t slab_bytes_for(const kbo_index *idx);
std::vector<int> devices_for(kbo_index *idx);
//   // devices the host batch entry points spread slabs over (empty = current)
extern std::atomic<bool> g_force_big;             // tests: use the 64-bit-offset entry layout regardless of size
extern std::atomic<uint64_t> g_pair_min_rows;     // indexes with at least this many rows get two-base blocks on the device
extern std::atomic<int> g_host_in_place;         // host batches use a caller's pinned buffers in place instead of staging them (kbo_set_host_in_place)
extern std::atomic<size_t> g_slab_bytes;          // host batches are cut into slabs of at most this many query bytes
extern std::atomic<int> g_plan_cap_div;           // tests: the unit array gets 1/this of its normal capacity
extern std::atomic<int> g_index_shards;           // tests: kbo_index_build makes at least this many shards (0 = by size)
extern std::atomic<bool> g_plan_stats;            // launches of the plan-guided stage count their own work (kbo_set_plan_stats)
extern std::atomic<int> g_depth_table_anchors;    // ... with anchors: -1 by margin, 0 no, 1 yes
extern std::atomic<int> g_depth_table;            // depth table of new device copies: 0 = by index size, < 0 none, else its order
extern std::atomic<int> g_seed_table_depth;       // tests: bases per seed-table entry of new device copies (0 = by index size)
extern std::atomic<bool> g_plan_enabled;          // device copies carry a path cover and MS-only batches take the plan-guided walk
extern std::atomic<uint64_t> g_plan_table_budget; // bytes a copy's tables may take (0 = half of the free device memory)
extern std::atomic<int64_t> g_plan_lazy_bases;    // bases through a copy before it builds its plan structures (-1 = by index size)

// ---- device_index.cpp
int current_device();
// the indexes a walk goes over: the shards of a sharded handle, else the handle itself
inline std::vector<kbo_index *> shards_of(kbo_index *idx)
{
    std::vector<kbo_index *> v;
    if (idx->sharded()) for (auto &s : idx->shards) v.push_back(s.get());
    else v.push_back(idx);
    return v;
}
// throws KBO_E_UNSUPPORTED for a sharded handle: `what` needs the rows of ONE index
void require_unsharded(const kbo_index *idx, const char *what);
// uploads the index on first use; *plan (optional) receives the copy's plan hold-off state.  work_bases: the bases of the batch
// that asks (a copy makes its plan structures - cover, tables - once the bases it has seen pay for them); prepare: make them now
kbo::DevIndexView device_view(kbo_index *idx, int device, DevCopy::PlanState **plan = nullptr, uint64_t work_bases = 0,
                              bool prepare = false);
int walk_max_waves();                                     // upper bound on resident walk waves: CUs x waves per CU
// points a.gitems / a.glist into `plan_work` (>= kbo::plan_work_bytes(n_items) bytes, 16-byte aligned) when the index
// view carries a path cover and the launch wants MS values only; otherwise leaves them null (plain walk)
void attach_plan(kbo::WalkArgs &a, void *plan_work, DevCopy::PlanState *ps);
void plan_reset_holdoff(); // every copy plans its next launch again
// after launch_ms_walk: lets the host learn whether the plan paid (and whether a walk was cut short by its guard)
void plan_after_launch(const kbo::WalkArgs &a, hipStream_t stream, DevCopy::PlanState *ps);

// ---- A3 (kbo_capi.cpp): derandomize.rs:91-145
double log_rm_max_cdf(size_t t, size_t alphabet_size, size_t n_kmers);
size_t random_match_threshold(size_t k, size_t n_kmers, size_t alphabet_size, double p);

// ---- host_batch.cpp
struct BatchOnDevice {
    DevBuf q, off, items, ms, lo, hi, plan;
    DevBuf longw; // work of the kernel for sequences of more than 160 bases (long_kernels.hip)
    DevBuf ms_shard; // sharded indexes: the MS values of one further shard, folded into `ms` by maximum
    DevBuf packed, pscr, exc_pos, exc_byte, packed_out, exc_flag; // (exc_flag: one byte per read of a packed-native launch)
    // packed entry points: 2-bit words in, scanned words per sequence, non-ACGT list, 2-bit words out
    uint64_t total = 0;
    void release()
    {
        for (DevBuf *b : {&q, &off, &items, &ms, &lo, &hi, &plan, &longw, &packed, &pscr, &exc_pos, &exc_byte, &packed_out, &exc_flag, &ms_shard}) b->release();
    }
};
// a slab of a packed batch (pack_kernels.hip: sequence s = ceil(len / 16) u32 words, 2 bits per base): what
// enqueue_walk_host uploads and unpacks into B.q instead of uploading bytes
struct PackedIn {
    const uint32_t *words;      // the slab's words (host; pinned or staged by the caller)
    size_t n_words;
    const uint64_t *exc_pos;    // non-ACGT bases of the slab: positions in the whole batch's base coordinates, ascending ...
    const uint8_t *exc_byte;    // ... and their bytes
    size_t n_exc;
    uint64_t base;              // the slab's first base in those coordinates
    uint32_t uniform_len;       // != 0: every sequence of the slab has this many bases (offsets are made on the device)
};
struct Slab {
    size_t s0, s1;   // sequences [s0, s1)
    uint64_t b0, b1; // bases [b0, b1)
};
struct OffsetScan { // one pass over the offsets of a batch: order, shortest and longest sequence
    bool monotone = true;
    uint64_t shortest = ~0ull, longest = 0;
};
// Where kbo_find_batch collects format::run_lengths_gapped of every slab (computed on the device
// from the slab's characters, which then never leave it)
struct RleSink {
    size_t max_gap_len = 0;
    uint64_t *rle_offsets = nullptr; // caller's n_seqs + 1 entries
    // one device: slabs complete in order, so their records go straight into the result array
    kbo_rle *all = nullptr;
    size_t all_cap = 0, all_used = 0;
    bool caller_owns = false; // `all` is the caller's buffer of all_cap records: never grown; all_used keeps counting
    bool direct = false;      // set by matches_batch_impl: one worker, records went straight into `all`
    // compact = true (kbo_find_batch_packed): the records stay the seven u32 the device writes (kbo_rle32), nothing is
    // widened: all32 / runs32 take the place of all / runs (all_cap and all_used count records either way)
    bool compact = false;
    uint32_t *all32 = nullptr;
    std::vector<std::vector<uint32_t>> runs32;
    // several devices: slabs complete out of order, kept per slab and put together at the end
    std::vector<std::vector<kbo_rle>> runs;
    std::vector<std::vector<uint32_t>> first; // index of the first run of each sequence of the slab, +1 entry
    ~RleSink()
    {
        if (!caller_owns) std::free(all);
        std::free(all32);
    }
};
constexpr size_t kRleWords = 7; // device run-length records are seven u32; kbo_rle has the reference's usize fields

uint32_t max_len(const uint64_t *offsets, size_t n_seqs);
uint64_t walk_chunk(uint64_t total, size_t n_seqs, uint32_t k);
void check_batch(const void *concat, const uint64_t *offsets, size_t n_seqs);
void check_len_threshold(const uint64_t *offsets, size_t n_seqs, size_t k, size_t threshold);
OffsetScan scan_offsets(const uint64_t *offsets, size_t n_seqs);
std::vector<Slab> make_slabs(const uint64_t *offsets, size_t n_seqs, size_t max_bytes);
size_t packed_slab_bytes(const kbo_index *idx); // bases per slab of a packed batch
// upload + A1 over a host batch (asynchronous on `stream`); leaves ms (and lo/hi) on the device.
// `items_keep` must stay alive until the stream has been synchronised.
// call mode of the walk (kernels.hpp WalkArgs::call_*): where the sites go
struct CallSink {
    void *d_sites;        // kCallSegs lists of cap_per_list 16-byte records
    uint32_t *d_counts;   // kCallSegs counters 64 bytes apart + the overflow counter behind them, zeroed by the caller
    uint32_t cap_per_list;
    uint32_t threshold;
};
// what the caller wants behind the walk (kbo::matches / map): when the slab is a batch of reads over a copy with a depth table,
// enqueue_walk_host runs map_reads_kernel (map_kernels.hip) - characters straight into d_chars, no MS values in memory - and
// sets `done`; otherwise it leaves the MS values in B.ms as ever and the caller runs A5 / A6
struct FusedMap {
    uint8_t *d_chars;   // >= total + 16 bytes
    uint32_t threshold;
    bool format;        // + format::relative_to_ref
    bool done = false;
    // a packed batch whose characters leave packed as well: where their words go (nw words + 16 bytes); packed_done = the kernel
    // wrote them there itself (its packed-native form), else the characters are in d_chars as for any batch
    uint32_t *d_packed_out = nullptr;
    bool packed_done = false;
    // a second stream for the second pass (the plain walk of the reads the kernel leaves), so that the next slab's kernel need
    // not wait for it, and an event to order it behind the kernel; `results` = the stream the slab's characters are complete on
    hipStream_t tail = nullptr;
    hipEvent_t fence = nullptr;
    hipStream_t results = nullptr;
    // kbo::find with max_gap_len = 0: where the number of runs of every sequence goes (rle scratch, n_seqs + 1 words); counted = the
    // one kernel (and, for the reads of its second pass, derand_flagged_kernel) filled it: scan + emit are what is left
    uint32_t *run_counts = nullptr;
    bool counted = false;
};
void enqueue_walk_host(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, bool want_ival,
                       BatchOnDevice &B, std::vector<kbo::WalkItem> &items_keep, hipStream_t stream,
                       uint32_t longest = 0 /* longest sequence if the caller knows it */,
                       hipStream_t copy_stream = nullptr /* uploads go here when given ... */,
                       hipEvent_t copied = nullptr /* ... and `stream` waits for this event */,
                       const CallSink *call = nullptr /* call mode: MS values + sites, no intervals */,
                       const PackedIn *packed = nullptr /* the queries arrive 2-bit packed (concat is not read) */,
                       FusedMap *map = nullptr /* kbo::matches / map: the one kernel where it applies */);
void run_walk_host(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, bool want_ival,
                   BatchOnDevice &B, hipStream_t stream);
// A5+A6 over a batch whose offsets are known on the host
void derand_translate_host_offsets(const uint8_t *d_ms, const uint64_t *d_off, const uint64_t *offsets, size_t n_seqs,
                                   uint32_t k, uint32_t threshold, const uint8_t *d_ref, uint8_t *d_chars,
                                   int32_t *d_derand, hipStream_t stream, uint32_t longest = 0,
                                   DevBuf *piece_work = nullptr /* lets long reads / contigs be split into pieces */);
void widen_rles(kbo_rle *dst, const uint32_t *src, size_t n, HostTeam &team);
// kbo::matches over a batch (lib.rs:618-627); optional relative_to_ref (lib.rs:756-757); with a sink the
// characters are turned into run lengths on the device instead of being downloaded (lib.rs:816-820)
void matches_batch_impl(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                        double max_error_prob, bool format, uint8_t *chars_out, RleSink *sink = nullptr);
// the same over 2-bit packed reads (pack_kernels.hip layout) with the non-ACGT bases in a side list; the characters come
// back 2-bit packed as well (M, -, X, R = 0 .. 3) or, with a sink, as run lengths
struct PackedBatch {
    const uint32_t *words;
    const uint64_t *exc_pos;
    const uint8_t *exc_byte;
    size_t n_exc;
};
void matches_batch_packed_impl(kbo_index *idx, const PackedBatch &in, const uint64_t *offsets, size_t n_seqs, double max_error_prob,
                               uint32_t *packed_out, RleSink *sink = nullptr);
// A1 over a host batch: MS values, and intervals when lo/hi are given
void ms_batch_impl(kbo_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, uint8_t *d_out,
                   uint32_t *lo_out, uint32_t *hi_out);
void release_host_scratch(); // frees the pooled per-device scratch of the host batch entry points + the calling thread's caches
void release_transient_arena();     // device_index.cpp: the calling thread's arena for transient indexes (when not in use)

} // namespace kbo_host
