// device_index.cpp — the per-device copies of an index handle (rank blocks, contraction entries,
// two-base blocks) and the knobs that shape them.
#include "capi_internal.hpp"

#include <cmath>

#include <atomic>
#include <chrono>
#include <cstdlib>

namespace kbo_host {

// tuning state: set through the kbo_set_* entry points, possibly while other threads are inside calls
std::atomic<int> g_waves_per_cu{0};
std::atomic<bool> g_force_big{false};
std::atomic<uint64_t> g_pair_min_rows{24ull << 20};
std::atomic<bool> g_plan_enabled{true};
std::atomic<int> g_plan_cap_div{1};
std::atomic<int> g_seed_table_depth{0};
std::atomic<int> g_depth_table{0}; // depth table of device copies made from now on: 0 = by index size, < 0 = none, else its order
std::atomic<int> g_depth_table_anchors{-1}; // ... with anchors: -1 = by the table's margin over log4(rows), 0 = no, 1 = yes

bool plan_enabled(const kbo_index *idx)
{
    const int v = idx ? idx->opts.plan.load() : kOptInherit;
    return v == kOptInherit ? g_plan_enabled.load() : v != 0;
}
int depth_table_setting(const kbo_index *idx)
{
    const int v = idx ? idx->opts.depth_table.load() : kOptInherit;
    return v == kOptInherit ? g_depth_table.load() : v;
}
int depth_table_anchor_setting(const kbo_index *idx)
{
    const int v = idx ? idx->opts.depth_table_anchors.load() : kOptInherit;
    return v == kOptInherit ? g_depth_table_anchors.load() : v;
}
std::atomic<bool> g_plan_stats{false};
std::atomic<uint64_t> g_plan_table_budget{0}; // bytes a copy's seed + depth tables may take while they are built: 0 = half of what is free
std::atomic<int64_t> g_plan_lazy_bases{-1};   // bases through a copy before it builds its plan structures: -1 = by index size, 0 = at once
std::atomic<int> g_index_shards{0};

int current_device()
{
    int dev = -1;
    HIP_OK(hipGetDevice(&dev));
    return dev;
}

// the arena transient indexes borrow (DevCopy::arena_borrowed): one per host thread, reallocated when the thread has
// moved to another device (DevBuf::ensure), freed by kbo_release_scratch() for the calling thread and at thread exit
static DevBuf &transient_arena()
{
    static thread_local DevBuf t_arena;
    return t_arena;
}
void release_transient_arena()
{
    if (!DevCopy::transient_arena_in_use()) transient_arena().release();
}

void require_unsharded(const kbo_index *idx, const char *what)
{
    if (idx && idx->sharded())
        throw KboError(KBO_E_UNSUPPORTED, std::string(what) + " needs the rows of one index; this handle is a sharded index (its rows "
                       "would not fit 32-bit row numbers): only depth-only batches (matches, map without refinement, find, ms "
                       "without intervals) are supported on it");
}

namespace {
using clk = std::chrono::steady_clock;
double since(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

// The plan structures of a copy (path cover, recovery lines, seed table(s), depth table, 2-bit text): everything that is sized
// by log4(rows) or needs the host's pointer chase.  Called with idx->mu held and the copy's device current.
void build_plan_structures(kbo_index *idx, DevCopy *dc)
{
    clk::time_point t0 = clk::now();
    // (device builds run on a stream of their own, so that a first use inside the slab pipeline does not serialise against
    // every blocking stream of the device: only this stream is waited for)
    hipStream_t bs = nullptr;
    HIP_OK(hipStreamCreateWithFlags(&bs, hipStreamNonBlocking));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamDestroy(s); }
    } guard{bs};
    if (!dc->plan.bailed) {
        HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&dc->plan.bailed), 64, hipHostMallocDefault));
        dc->plan.bailed[0] = dc->plan.bailed[1] = 0;
    }
    static_assert(kbo::PathCover::kPad == kbo::kPlanPad, "text padding");
    bool made_here = false;
    if (!idx->cover) { // (idx->mu is held)
        // laid out on the device, straight into the copy's buffers (cover_kernels.hip: the same layout position for position; the host's
        // pointer chase was 26 s per 10^8 rows), and kept on the host for index files and further copies; rows left without a position
        // (cycles no head leads into) or KBO_DEVICE_COVER=0: the host's construction
        static const int env_dev_cover = std::getenv("KBO_DEVICE_COVER") ? std::atoi(std::getenv("KBO_DEVICE_COVER")) : 1;
        const uint64_t n_rows = idx->host.n_sets;
        if (env_dev_cover != 0 && n_rows > 0) {
            const size_t text_bytes = (size_t)n_rows + 2 * kbo::PathCover::kPad;
            dc->pc_text.alloc(text_bytes + 16);
            dc->pc_pos.alloc((size_t)n_rows * 4 + 16);
            dc->pc_node.alloc((size_t)n_rows * 4 + 16);
            HIP_OK(hipMemsetAsync(dc->pc_text.p, 0, text_bytes + 16, bs));
            HIP_OK(hipMemsetAsync(dc->pc_node.p, 0, (size_t)n_rows * 4 + 16, bs));
            const uint32_t *d_ent = dc->big ? dc->ent.as<uint32_t>()
                                            : reinterpret_cast<const uint32_t *>(dc->arena.as<uint8_t>() + dc->n_blocks * 64 + 16);
            bool ok = false;
            HIP_OK(kbo::build_path_cover_device(dc->arena.as<uint4>(), d_ent, n_rows, (uint32_t)dc->n_blocks, idx->host.k, idx->host.C,
                                                dc->pc_text.as<uint8_t>() + kbo::PathCover::kPad, dc->pc_pos.as<uint32_t>(), dc->pc_node.as<uint32_t>(), bs, &ok));
            if (ok) {
                idx->cover.reset(new kbo::PathCover());
                idx->cover->text.resize(text_bytes);
                idx->cover->pos.resize(n_rows);
                idx->cover->node_at.resize(n_rows);
                HIP_OK(hipMemcpy(idx->cover->text.data(), dc->pc_text.p, text_bytes, hipMemcpyDeviceToHost));
                HIP_OK(hipMemcpy(idx->cover->pos.data(), dc->pc_pos.p, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
                HIP_OK(hipMemcpy(idx->cover->node_at.data(), dc->pc_node.p, (size_t)n_rows * 4, hipMemcpyDeviceToHost));
                made_here = true;
            }
        }
        if (!made_here) {
            idx->cover.reset(new kbo::PathCover());
            kbo::make_path_cover(idx->host, *idx->cover);
        }
    }
    dc->setup.cover_s = since(t0);
    t0 = clk::now();
    const kbo::PathCover &pc = *idx->cover;
    if (!made_here) {
        dc->pc_text.alloc(pc.text.size() + 16);
        dc->pc_pos.alloc(pc.pos.size() * 4 + 16);
        dc->pc_node.alloc(pc.node_at.size() * 4 + 16);
        HIP_OK(hipMemcpy(dc->pc_text.p, pc.text.data(), pc.text.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dc->pc_pos.p, pc.pos.data(), pc.pos.size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dc->pc_node.p, pc.node_at.data(), pc.node_at.size() * 4, hipMemcpyHostToDevice));
    }
    idx->plan_bytes = pc.text.size() + pc.pos.size() * 4 + pc.node_at.size() * 4;
    dc->setup.cover_bytes = idx->plan_bytes;
    dc->setup.upload_s += since(t0);
    {
        t0 = clk::now();
        std::vector<uint8_t> lines;
        kbo::make_recovery_lines(idx->host, lines);
        dc->fat.alloc(lines.size() + 16);
        HIP_OK(hipMemcpy(dc->fat.p, lines.data(), lines.size(), hipMemcpyHostToDevice));
        dc->fat_null = (uint32_t)(lines.size() / 128 - 1);
        idx->plan_bytes += lines.size();
        dc->setup.lines_bytes = lines.size();
        dc->setup.lines_s = since(t0);
    }
    // ---- the depth table (dtab_kernels.hip): for every string of `order` bases the longest suffix of it that is a suffix of a
    // row.  order = log4(rows) + 3.2, rounded up (1.3 % of the stretches behind mismatches run deeper than log4(rows) + 4, 5 %
    // deeper than + 3, 17 % deeper than + 2: those reads take the plain walk), at most 17 and k: 15 bases at C2 (4 GiB
    // grouped), 17 at C3 / C4 (64 GiB grouped); none where 17 bases are less than log4(rows) + 1.9 (from 1.2 * 10^9 rows on: the
    // guided walk over recovery lines stays).  Measured, A1: C2 15 / 16 bases 0.625 / 0.638 ms (guided walk 0.915); C3 per
    // 10 M reads 17 grouped / 17 plain / 16 grouped / 16 plain 7.35 / 8.88 / 9.18 / 9.86 ms (guided walk 10.15).
    const double lg = std::log2((double)std::max<uint64_t>(idx->host.n_sets, 4)) / 2.0;
    int order = std::min<int>({(int)std::ceil(lg + 3.2), 17, (int)idx->host.k});
    // (a margin of 1.9 .. 2.9 bases - a 1 Gbp index - pays with anchors only; below 3.75 bases the reads the table leaves to
    // the plain walk are many enough for the anchors to pay - C4, margin 3.05: A1 per 100 M reads 76.7 -> 67.1 ms; C3, 3.7:
    // 6.80 -> 6.55; C2, 3.9: 0.62 -> 0.66, slower - and only on indexes of 24 Mi rows and more)
    if ((double)order < lg + 1.9 && order < (int)idx->host.k) order = 0;
    const int set = depth_table_setting(idx);
    if (set < 0) order = 0;
    else if (set > 0) order = std::min<int>({set, 17, (int)idx->host.k});
    if (const char *e = std::getenv("KBO_DEPTH_TABLE")) // experiments
        order = std::max(0, std::min<int>({std::atoi(e), 17, (int)idx->host.k}));
    // ---- what it may take: the budget of the copy's tables (kbo_set_plan_table_budget; 0 = half of what is free now - the
    // batches need the rest).  The grouped layout (three consecutive bases share a 64-byte line: a third of the fills, four
    // times the bytes) when it fits, else the plain one, else a table of fewer bases (while it still reaches log4(rows) + 1.9),
    // else none: a copy never fails, and never holds more than it was allowed, because of its tables.  Forced orders alike.
    static const int env_grp = std::getenv("KBO_DEPTH_TABLE_GROUPED") ? std::atoi(std::getenv("KBO_DEPTH_TABLE_GROUPED")) : -1; // experiments
    bool grouped = order >= 4 && env_grp != 0;
    const uint64_t cap = idx->host.n_sets + 16;
    {
        size_t free_b = 0, total_b = 0;
        uint64_t budget = g_plan_table_budget.load();
        if (budget == 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = free_b / 2;
        const size_t tmp_bytes = kbo::dtab_tmp_bytes(cap);
        auto need = [&](int o, bool grp) { // peak during the build: the plain table, the frontier, the grouped copy, the seed table and its
                                           // positions, the anchor hash (whether or not this copy gets one), the 2-bit text and the filter
            return (uint64_t)kbo::dtab_bytes((uint32_t)o, false) + tmp_bytes + (grp ? kbo::dtab_bytes((uint32_t)o, true) : 0) + (((uint64_t)12) << (2u * std::min(o, 14))) +
                   ((uint64_t)8 << kbo::dtab_anchor_bits(idx->host.n_sets, (uint32_t)o)) + idx->host.n_sets + (4u << 20);
        };
        while (order > 0 && budget != 0 && need(order, grouped) > budget) {
            if (grouped) grouped = false;
            else {
                order--;
                grouped = order >= 4 && env_grp != 0;
                if ((double)order < lg + 1.9 && set <= 0) order = 0;
            }
        }
    }
    const bool thin_margin = order > 0 && (double)order < lg + 3.75 && order < (int)idx->host.k && idx->host.n_sets >= (24u << 20);
    // ---- the seed table: the interval of every string of D bases, so that a seed starts D bases deep.  Without a depth
    // table: 10 bases for indexes that can use them (8 MiB), 8 for small ones, none below k = 8; 12 bases / 128 MiB from 32 Mi
    // rows, 13 / 512 MiB from 512 Mi rows (the extensions they save are line fills there), built on the host.  With one: as
    // many bases as a seed must be deep before its row is trusted (log4(rows) + 3: 14 at C2), at most 14 (2 GiB) - a unique
    // string of that many bases IS the seed - built on the device with the depth table.
    uint32_t D = idx->host.k >= 10 && idx->host.n_sets >= (1u << 20) ? 10u : (idx->host.k >= 8 ? 8u : 0u);
    if (D == 10 && idx->host.k >= 13 && idx->host.n_sets >= (512u << 20)) D = 13;
    else if (D == 10 && idx->host.k >= 12 && idx->host.n_sets >= (32u << 20)) D = 12;
    if (order > 0) D = std::min<uint32_t>({(uint32_t)std::lround(lg) + 3u, 14u, idx->host.k, (uint32_t)order});
    if (const int forced = g_seed_table_depth.load()) D = std::min<uint32_t>({(uint32_t)forced, 14u, idx->host.k}); // tests
    if (const char *e = std::getenv("KBO_PLAN_SEED_D")) // experiments
        D = std::min<uint32_t>({(uint32_t)std::max(0, std::atoi(e)), 14u, idx->host.k});
    const bool seed_on_device = order > 0 && D > 0 && (int)D <= order;
    if (seed_on_device) {
        dc->seed_tab.alloc(((size_t)8 << (2u * D)) + 64);
        dc->seed_d = D;
        idx->plan_bytes += (size_t)8 << (2u * D);
    } else if (D) {
        t0 = clk::now();
        const kbo::HostNav nav(idx->host);
        std::vector<uint32_t> cur{0u, (uint32_t)idx->host.n_sets}, nxt; // {l, r} pairs, level by level
        for (uint32_t t = 0; t < D; t++) {
            nxt.resize(cur.size() * 4);
            for (size_t p = 0; p < cur.size() / 2; p++) {
                const uint32_t l = cur[2 * p], r = cur[2 * p + 1];
                for (int c = 0; c < 4; c++) {
                    uint32_t l2 = 0, r2 = 0;
                    if (l < r) {
                        l2 = (uint32_t)(idx->host.C[c] + nav.rank(c, l));
                        r2 = (uint32_t)(idx->host.C[c] + nav.rank(c, r));
                    }
                    nxt[2 * (4 * p + c)] = l2;
                    nxt[2 * (4 * p + c) + 1] = r2;
                }
            }
            cur.swap(nxt);
        }
        dc->seed_tab.alloc(cur.size() * 4);
        HIP_OK(hipMemcpy(dc->seed_tab.p, cur.data(), cur.size() * 4, hipMemcpyHostToDevice));
        dc->seed_d = D;
        idx->plan_bytes += cur.size() * 4;
        dc->setup.seed_s = since(t0);
    }
    if (order > 0) {
        t0 = clk::now();
        DevBuf tmp, plain;
        plain.alloc(kbo::dtab_bytes((uint32_t)order, false) + 64);
        tmp.alloc(kbo::dtab_tmp_bytes(cap));
        kbo::DevIndexView bv{};
        bv.arena = dc->arena.as<uint4>();
        bv.n_blocks = (uint32_t)dc->n_blocks;
        bv.n = (uint32_t)idx->host.n_sets;
        bv.k = idx->host.k;
        // anchors: the strings of `order` bases that are the suffix of one row only, with that row's place in the path cover -
        // what a base deeper than the table knows is read off (dtab_kernels.hip)
        const uint32_t abits = kbo::dtab_anchor_bits(idx->host.n_sets, (uint32_t)order);
        static const int env_anchor = std::getenv("KBO_DEPTH_TABLE_ANCHORS") ? std::atoi(std::getenv("KBO_DEPTH_TABLE_ANCHORS")) : -1; // experiments
        const int anch_set = env_anchor >= 0 ? env_anchor : depth_table_anchor_setting(idx);
        // (small indexes as well - where the filter in front of the table applies: the one kernel then prices a window that is present by
        // chance by its exact depth instead of leaving the read to the second pass: 1.9 % -> 0.8 % of the reads at C2, second pass
        // 0.187 -> 0.141 ms; 128 MB at 5 * 10^6 rows)
        const bool small_index = 2ull * idx->host.n_sets <= (1ull << 24);
        const bool want_anchors = anch_set > 0 || (anch_set < 0 && (thin_margin || small_index));
        if (want_anchors && order < (int)idx->host.k) {
            dc->anchor.alloc(((size_t)1 << abits) * 8 + 64);
            dc->anchor_bits = abits;
            bv.pc_pos = dc->pc_pos.as<uint32_t>();
        }
        // the filter in front of the table (map_kernels.hip): one bit per string of F bases, F the shortest that leaves at most half
        // of the strings present, where that is a table the L2 keeps (<= 12 bases: 2 MB) and shorter than the depth table's own.
        // At C2 it keeps four table look-ups in five from the table - L2 misses 8.68 M -> 6.61 M, fetched bytes - 28 %.  The kernel
        // alone is no faster for it (0.251 against 0.248 ms: it is not bound by its fills, LABNOTES round 4), but beside another
        // batch's second pass it is: two batches in flight 0.332 -> 0.308 ms per batch.  KBO_DEPTH_FILTER=0: none, n: that many bases
        static const int env_filter = std::getenv("KBO_DEPTH_FILTER") ? std::atoi(std::getenv("KBO_DEPTH_FILTER")) : -1; // experiments
        // (round 5, C3: a filter of 14 bases - 32 MB, the Infinity Cache's rather than an L2's - by KBO_DEPTH_FILTER=14: 348 against 354
        // Gbp/s without; larger indexes keep none)
        // (round 6: the one kernel at 24 resident waves a CU is bound by its fabric traffic on large indexes - 12.7 GB per 1.5 Gbases at C3,
        // 5.9 TB/s - and a filter the Infinity Cache keeps pays there too: C3 with 14 / 15 bases (32 / 128 MB) 536 / 575 against 500 Gbp/s,
        // the kernel alone 2.02 / 1.85 against 2.20 ms.  Indexes beyond the L2's 12 bases take 15: a window in eleven passes at 10^8 rows)
        uint32_t fb = 0;
        for (uint32_t f = 6; f <= 15u; f++)
            if (2ull * idx->host.n_sets <= (1ull << (2u * f))) { fb = f; break; }
        if (fb > 12u) fb = 15u;
        if (env_filter >= 0) fb = (uint32_t)env_filter;
        if (fb < 6u || fb > 15u || fb + 2u > (uint32_t)order) fb = 0;
        if (fb) {
            dc->dfilt.alloc((((size_t)1 << (2u * fb)) / 8u) + 64);
            dc->dfilt_bases = fb;
        }
        HIP_OK(kbo::build_depth_table(bv, (uint32_t)order, plain.as<uint8_t>(), tmp.p, cap, bs,
                                      dc->anchor_bits ? dc->anchor.as<uint64_t>() : nullptr, dc->anchor_bits,
                                      seed_on_device ? dc->seed_tab.as<uint2>() : nullptr, seed_on_device ? dc->seed_d : 0u,
                                      fb ? dc->dfilt.as<uint32_t>() : nullptr, fb));
        if (fb) idx->plan_bytes += ((size_t)1 << (2u * fb)) / 8u;
        if (dc->anchor_bits) idx->plan_bytes += ((size_t)1 << abits) * 8;
        if (grouped) {
            tmp.release();
            dc->dtab.alloc(kbo::dtab_bytes((uint32_t)order, true) + 64);
            HIP_OK(kbo::regroup_depth_table(plain.as<uint8_t>(), (uint32_t)order, dc->dtab.as<uint8_t>(), bs));
        } else {
            std::swap(dc->dtab.p, plain.p);
            std::swap(dc->dtab.cap, plain.cap);
            std::swap(dc->dtab.dev, plain.dev);
        }
        dc->dtab_order = (uint32_t)order;
        dc->dtab_grouped = grouped;
        idx->plan_bytes += kbo::dtab_bytes((uint32_t)order, grouped);
        HIP_OK(hipStreamSynchronize(bs));
        dc->setup.dtab_s = since(t0);
        dc->setup.dtab_bytes = kbo::dtab_bytes((uint32_t)order, grouped) + (dc->dfilt_bases ? ((uint64_t)1 << (2u * dc->dfilt_bases)) / 8u : 0u);
        dc->setup.anchor_bytes = dc->anchor_bits ? ((uint64_t)1 << dc->anchor_bits) * 8 : 0;
    }
    dc->setup.seed_bytes = dc->seed_d ? (uint64_t)8 << (2u * dc->seed_d) : 0;
    // what map_reads_kernel reads (map_kernels.hip): the text as 2-bit digits with its path-start marks (0.5 B per row) and the
    // seed table as text positions (4 B per string of seed_d bases), both made on the device
    if (dc->dtab_order >= 4 && dc->seed_d >= 4) {
        t0 = clk::now();
        const uint64_t units = kbo::pack_text_units(idx->host.n_sets);
        dc->pc_tm.alloc(units * 8 + 64);
        HIP_OK(kbo::launch_pack_text(dc->pc_text.as<uint8_t>(), pc.text.size(), dc->pc_tm.as<uint2>(), units, bs));
        dc->seed_pos.alloc(((size_t)4 << (2u * dc->seed_d)) + 64);
        HIP_OK(kbo::launch_seed_pos(dc->seed_tab.as<uint2>(), dc->pc_pos.as<uint32_t>(), dc->seed_pos.as<uint32_t>(), dc->seed_d, bs));
        HIP_OK(hipStreamSynchronize(bs));
        dc->setup.cover_bytes += units * 8;
        dc->setup.seed_bytes += (uint64_t)4 << (2u * dc->seed_d);
        idx->plan_bytes += units * 8 + ((size_t)4 << (2u * dc->seed_d));
        dc->setup.seed_s += since(t0);
    }
    dc->plan_built = true;
}

// bases a copy must have seen - or be about to see - before its plan structures pay for themselves: making them costs the
// host's pointer chase over the rows (unless the handle carries a cover already) and the device builds, about 50 ns + 15 ns per
// row on the GPU box (0.08 + 0.02 s at 5 * 10^6 rows, 8.3 + 3.5 s at 2.5 * 10^8), and with them a base costs about 7 ps less
// (plain walk ~ 100 Gbp/s, the one kernel ~ 320; tools/bench_setup.py has the table).  Indexes whose structures take under
// a tenth of a second get them at once.
uint64_t plan_break_even_bases(const kbo_index *idx)
{
    const int64_t set = g_plan_lazy_bases.load();
    if (set >= 0) return (uint64_t)set;
    const double rows = (double)idx->host.n_sets;
    const double setup_s = rows * ((idx->cover ? 0.0 : 50e-9) + 15e-9);
    if (setup_s < 0.1) return 0;
    return (uint64_t)(setup_s / 7e-12);
}
} // namespace

kbo::DevIndexView device_view(kbo_index *idx, int device, DevCopy::PlanState **plan, uint64_t work_bases, bool prepare)
{
    require_unsharded(idx, "this operation");
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
        clk::time_point t0 = clk::now();
        // rank blocks, contraction entries and two-base blocks are made on the device from the index's own arrays (layout_kernels.hip:
        // 1.5 bytes a row over PCIe instead of 12.7; the host's single-threaded make_device_layout was 40 s of a 54 s copy at 3 * 10^9
        // rows); KBO_DEVICE_LAYOUT=0: the host's, uploaded - the same bytes (kbo_index_layout_check)
        static const int env_dev_layout = std::getenv("KBO_DEVICE_LAYOUT") ? std::atoi(std::getenv("KBO_DEVICE_LAYOUT")) : 1;
        const bool on_device = env_dev_layout != 0;
        kbo::DeviceLayout lay;
        if (!on_device) kbo::make_device_layout(idx->host, lay, want_pairs);
        const uint64_t n_rows = idx->host.n_sets;
        const uint64_t lay_n_blocks = n_rows / kbo::kRankRowsPerBlock + 2;
        const size_t lay_ent_words = 3 * (size_t)(n_rows + 1) + 4, lay_pair_words = want_pairs ? (size_t)16 * lay_n_blocks * 4 : 0;
        int prev = current_device();
        if (prev != device) HIP_OK(hipSetDevice(device));
        DevCopy *dc = new DevCopy();
        dc->setup.layout_s = since(t0);
        try {
            t0 = clk::now();
            const size_t per = lay_n_blocks * 16;
            // arena = rank blocks of A,C,G,T | one all-zero "null" block | contraction entries.
            // When that exceeds the 32-bit offset range (n_sets * 12 B of entries >= ~4 GiB) the
            // entries get their own allocation and 64-bit offsets ("big" kernels).
            const size_t ent_bytes = lay_ent_words * sizeof(uint32_t);
            const size_t rank_bytes = per * 4 + 16;
            KBO_REQUIRE(rank_bytes < 0xFFFFFFF0ull, KBO_E_UNSUPPORTED, "rank blocks >= 4 GiB");
            dc->big = g_force_big || rank_bytes + ent_bytes >= 0xFFFFFFF0ull;
            const size_t pair_bytes = dc->big ? 0 : lay_pair_words * sizeof(uint32_t);
            const size_t base_bytes = ((dc->big ? rank_bytes : rank_bytes + ent_bytes) + 15) / 16 * 16;
            const size_t arena_bytes = base_bytes + pair_bytes;
            if (idx->transient && !dc->big && !DevCopy::transient_arena_in_use()) {
                DevBuf &t_arena = transient_arena(); // (the current device is `device` here)
                t_arena.ensure(arena_bytes);
                dc->arena.p = t_arena.p;
                dc->arena_borrowed = true;
                DevCopy::transient_arena_in_use() = true;
            } else {
                dc->arena.alloc(arena_bytes);
            }
            HIP_OK(hipMemset(dc->arena.p, 0, arena_bytes));
            if (dc->big) dc->ent.alloc(ent_bytes + 16);
            uint8_t *d_ent = dc->big ? dc->ent.as<uint8_t>() : dc->arena.as<uint8_t>() + rank_bytes;
            if (on_device) {
                const size_t n_words = (size_t)((n_rows + 63) / 64);
                DevBuf d_rows(4 * n_words * 8 + 64), d_lcs((size_t)n_rows + 64), d_scr(kbo::device_layout_scratch_bytes(n_rows));
                const uint64_t *rows_dev[4];
                for (int c = 0; c < 4; c++) {
                    HIP_OK(hipMemcpy(d_rows.as<uint8_t>() + (size_t)c * n_words * 8, idx->host.rows[c].data(), n_words * 8, hipMemcpyHostToDevice));
                    rows_dev[c] = reinterpret_cast<const uint64_t *>(d_rows.as<uint8_t>() + (size_t)c * n_words * 8);
                }
                if (n_rows) HIP_OK(hipMemcpy(d_lcs.p, idx->host.lcs.data(), (size_t)n_rows, hipMemcpyHostToDevice));
                if (dc->big) HIP_OK(hipMemset(dc->ent.p, 0, ent_bytes + 16));
                HIP_OK(kbo::build_device_layout(rows_dev, n_words, d_lcs.as<uint8_t>(), n_rows, idx->host.C, (uint32_t)lay_n_blocks,
                                                dc->arena.as<uint4>(), reinterpret_cast<uint32_t *>(d_ent),
                                                pair_bytes ? reinterpret_cast<uint4 *>(dc->arena.as<uint8_t>() + base_bytes) : nullptr, d_scr.p, nullptr));
            } else {
                for (int c = 0; c < 4; c++)
                    HIP_OK(hipMemcpy(dc->arena.as<uint8_t>() + per * c, lay.rank[c].data(), per, hipMemcpyHostToDevice));
                HIP_OK(hipMemcpy(d_ent, lay.ent.data(), ent_bytes, hipMemcpyHostToDevice));
                if (pair_bytes) HIP_OK(hipMemcpy(dc->arena.as<uint8_t>() + base_bytes, lay.pair.data(), pair_bytes, hipMemcpyHostToDevice));
            }
            if (pair_bytes) dc->pair_off = (uint32_t)(base_bytes / 16);
            dc->n_blocks = lay_n_blocks;
            idx->rank_bytes = per * 4;
            idx->lcs_bytes = ent_bytes;
            dc->setup.rank_bytes = rank_bytes;
            dc->setup.entry_bytes = ent_bytes;
            dc->setup.pair_bytes = pair_bytes;
            dc->setup.upload_s = since(t0);
        } catch (...) {
            delete dc;
            if (prev != device) (void)hipSetDevice(prev);
            throw;
        }
        if (prev != device) HIP_OK(hipSetDevice(prev));
        it = idx->dev.emplace(device, dc).first;
    }
    DevCopy *dc = it->second;
    // ---- the plan structures: at once when asked for (kbo_index_to_device) or cheap, else once the copy has seen the bases that
    // pay for them (plan_break_even_bases); a copy that may not hold them walks plainly, with the same results
    const bool plan_on = plan_enabled(idx); // (this index's own option first: kbo_index_set_opts)
    // (an explicit kbo_index_to_device after a failed build - memory may have been freed since - tries again and reports what happens)
    if (prepare && dc->plan_failed && !dc->plan_built) dc->plan_failed = false;
    if (!dc->plan_built && !dc->plan_failed && plan_on && !idx->transient) {
        dc->bases_seen += work_bases;
        if (prepare || dc->bases_seen >= plan_break_even_bases(idx)) {
            int prev = current_device();
            if (prev != device) HIP_OK(hipSetDevice(device));
            try {
                build_plan_structures(idx, dc);
            } catch (const KboError &e) {
                // A copy never fails because of its plan structures: what was made so far is released (a half-built set - a seed
                // table's depth without its table, a text without its positions - must not reach a launch), the copy is marked so
                // that the build is not paid again by every later call, and it walks plainly - with the same results.  An explicit
                // kbo_index_to_device still reports the failure.
                for (DevBuf *b : {&dc->pc_text, &dc->pc_pos, &dc->pc_node, &dc->fat, &dc->seed_tab, &dc->dtab, &dc->anchor, &dc->dfilt, &dc->pc_tm, &dc->seed_pos})
                    b->release();
                dc->seed_d = dc->dtab_order = dc->anchor_bits = dc->dfilt_bases = 0;
                dc->dtab_grouped = false;
                dc->fat_null = 0;
                dc->plan_failed = true;
                (void)hipGetLastError();
                if (prev != device) (void)hipSetDevice(prev);
                if (prepare) throw;
                last_error() = std::string("plan structures not built (the copy walks plainly): ") + e.what();
            } catch (...) {
                if (prev != device) (void)hipSetDevice(prev);
                throw;
            }
            if (prev != device && !dc->plan_failed) HIP_OK(hipSetDevice(prev));
        }
    }
    if (plan) *plan = &dc->plan;
    kbo::DevIndexView v;
    v.arena = dc->arena.as<uint4>();
    v.n_blocks = (uint32_t)dc->n_blocks;
    v.lcs_off = (uint32_t)(dc->n_blocks * 4 + 1);
    v.pair_off = dc->pair_off;
    v.ent = dc->big ? dc->ent.as<uint8_t>() : nullptr;
    v.big = dc->big ? 1u : 0u;
    v.n = (uint32_t)idx->host.n_sets;
    v.k = idx->host.k;
    v.pc_text = (plan_on && dc->pc_text.p) ? dc->pc_text.as<uint8_t>() + kbo::kPlanPad : nullptr; // (no text, no planned launch: attach_plan)
    v.pc_pos = dc->pc_pos.as<uint32_t>();
    v.seed_tab = dc->seed_d ? dc->seed_tab.as<uint2>() : nullptr;
    v.seed_d = dc->seed_d;
    const bool use_tab = dc->dtab_order != 0 && depth_table_setting(idx) >= 0; // (set to -1 since the copy was made: launches ignore it)
    v.dtab = use_tab ? dc->dtab.as<uint8_t>() : nullptr;
    v.dtab_order = use_tab ? dc->dtab_order : 0u;
    v.dtab_grouped = dc->dtab_grouped ? 1u : 0u;
    v.anchor = (use_tab && dc->anchor_bits) ? dc->anchor.as<uint64_t>() : nullptr;
    v.anchor_bits = dc->anchor_bits;
    v.fat = dc->fat.p ? dc->fat.as<uint8_t>() : nullptr;
    v.fat_null = dc->fat_null;
    v.pc_node = dc->pc_node.as<uint32_t>();
    v.pc_tm = (use_tab && dc->pc_tm.p) ? dc->pc_tm.as<uint2>() : nullptr;
    v.dfilt = (use_tab && dc->dfilt_bases) ? dc->dfilt.as<uint32_t>() : nullptr;
    v.dfilt_bases = v.dfilt ? dc->dfilt_bases : 0u;
    v.seed_pos = (use_tab && dc->seed_pos.p) ? dc->seed_pos.as<uint32_t>() : nullptr;
    for (int c = 0; c < 4; c++) v.C[c] = (uint32_t)idx->host.C[c];
    v.C[4] = v.n;
    return v;
}

// Batches whose reads differ too much from the index give the plan up on the device (plan_emit_kernel), after
// having paid for the plan kernel.  The host learns about it one launch late (an asynchronous 8-byte copy into pinned
// memory, never waited for) and then skips planning for the next kPlanHoldoff launches OVER THAT COPY of that index
// (DevCopy::PlanState): other indexes, and the same index on other devices, keep planning.
namespace {
constexpr int kPlanHoldoff = 16;
std::atomic<uint32_t> g_plan_epoch{1}; // bumped by kbo_set_plan(1, ..): every copy forgets its hold-off
} // namespace

void plan_reset_holdoff() { g_plan_epoch.fetch_add(1); }

void plan_after_launch(const kbo::WalkArgs &a, hipStream_t stream, DevCopy::PlanState *ps)
{
    if (!a.gitems || !a.qctl || !ps || !ps->bailed) return;
    (void)hipMemcpyAsync(ps->bailed, a.qctl + 2, 8, hipMemcpyDeviceToHost, stream);
}

void attach_plan(kbo::WalkArgs &a, void *plan_work, DevCopy::PlanState *ps)
{
    a.gitems = nullptr;
    a.glist = nullptr;
    a.ucount = a.usums = a.qctl = a.pstats = nullptr;
    a.redo = nullptr;
    a.units = nullptr;
    a.n_items_dev = nullptr;
    // (a.call_* are set by the caller before attach_plan)
    static const int env_pc = std::getenv("KBO_PLAN_CALL") ? std::atoi(std::getenv("KBO_PLAN_CALL")) : 1; // experiments: 0 = call mode never plans
    if (a.call_sites && !env_pc) return;
    a.unit_bail = 0;
    a.unit_cap = a.plan_dmin = a.plan_cap = a.plan_gap = a.plan_chunk = 0;
    if (!plan_work || !a.ix.pc_text || (a.lo_out && a.hi_out) || a.n_items == 0) return;
    if (ps) {
        const uint32_t epoch = g_plan_epoch.load();
        if (ps->epoch.exchange(epoch) != epoch) { // an explicit kbo_set_plan(1, ..) since: plan the next launch
            ps->holdoff.store(0);
            if (ps->bailed) *reinterpret_cast<volatile uint32_t *>(ps->bailed) = 0;
        }
        if (ps->bailed && *reinterpret_cast<volatile uint32_t *>(ps->bailed)) {
            *reinterpret_cast<volatile uint32_t *>(ps->bailed) = 0;
            ps->holdoff.store(kPlanHoldoff);
            ps->bails.fetch_add(1);
        }
        if (ps->holdoff.load() > 0) {
            ps->holdoff.fetch_sub(1);
            return;
        }
    }
    uint8_t *w = static_cast<uint8_t *>(plan_work);
    const kbo::PlanLayout L = kbo::plan_layout(a.n_items, a.q_bytes);
    a.gitems = reinterpret_cast<kbo::GuidedItem *>(w + L.gitems);
    a.unit_cap = L.unit_cap;
    a.redo_cap = 2u * a.unit_cap; // (>= every item cut into pieces of 64 bases: redo_collect_kernel)
    a.units = reinterpret_cast<kbo::WalkUnit *>(w + L.units);
    const int cap_div = g_plan_cap_div.load();
    if (cap_div > 1) a.unit_cap = std::max<uint32_t>(1u, a.unit_cap / (uint32_t)cap_div); // (tests: force the overflow path)
    a.glist = reinterpret_cast<uint16_t *>(w + L.glist);
    a.ucount = reinterpret_cast<uint32_t *>(w + L.ucount);
    a.usums = reinterpret_cast<uint32_t *>(w + L.usums);
    a.qctl = reinterpret_cast<uint32_t *>(w + L.qctl);
    a.pstats = g_plan_stats.load() ? reinterpret_cast<uint32_t *>(w + L.pstats) : nullptr; // (instrumentation: kbo_set_plan_stats)
    a.redo = w + L.redo;
}

// upper bound on resident walk waves: CUs x waves per CU (default 32 = 8 per SIMD)
int walk_max_waves()
{
    int dev = current_device();
    int cus = 0;
    HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int wpc = g_waves_per_cu.load();
    int per = wpc > 0 ? wpc : 32;
    return std::max(1, cus) * per;
}

} // namespace kbo_host
