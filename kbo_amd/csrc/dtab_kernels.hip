// dtab_kernels.hip — gfx950 (MI355X, CDNA4): the DEPTH TABLE form of the plan-guided A1 (sbwt::StreamingIndex::
// matching_statistics, called at reference index.rs:251-252) for batches that ask for the MS values only.
//
// The k-bounded matching statistic of base i is the length of the longest suffix of query[..i] that is a suffix of a row of
// the index, capped at k - a property of the STRING query[i-L+1 .. i] alone.  For an index of n rows a random string is
// present up to about log4(n) characters, so a table over ALL strings of `order` = log4(n) + 3..4 characters
//     dtab[key of the last `order` bases] = longest suffix of those bases that is present  (+ which one-base extensions to
//                                            the left are present when all of them are)
// answers MS(i) with ONE independent byte look-up wherever the true value is at most `order` - which is everywhere the
// plan-guided walk has to walk: the 11 - 16 bases behind a mismatch against the plan's diagonal, where the match is a random
// one through the mismatching base.  (plan_kernels.hip's walk spends a chain of ~36 dependent look-ups and ~22 line fills on
// such a stretch; this form spends `order + 1` independent ones.)
//
//   dtab_expand_kernel / dtab_scatter_kernel / dtab_ext_kernel   build the table on the device, level by level, from the
//                            index's own rank blocks: the present strings of s characters (a sparse frontier of at most n
//                            {key, interval} records) come from those of s - 1 by one extend-right step each
//   dtab_resolve_kernel      behind plan_kernel: one group of 16 / 32 lanes per work item, one lane per base behind a
//                            mismatch of the item's list; every lane looks its base up, the group finds where the stretch is
//                            back on the diagonal (longest present suffix == bases since the mismatch: nothing longer exists
//                            from there to the next mismatch, plan_kernel's predicted values are exact) and writes the values
//                            in front of that; a value the table cannot tell (deeper than `order`) flags the item for the
//                            plain walk (redo pass), as does an item without a diagonal
//
// Exactness does not depend on the plan: the values written come from the table (exact by construction), the values kept
// are min(k, bases since the last mismatch) on a stretch that equals the path-cover text (so that suffix is present) and
// behind a base where the table proved that nothing longer is.
//
// Integer / byte work only: no MFMA.  HBM-bound by independent random byte gathers (one 64-byte sector each).
#include "device_util.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>

namespace kbo {
namespace {

struct alignas(16) DtabRec { // a present string: its key (bases as 2-bit digits, the newest base least significant) and rows
    uint64_t key;
    uint32_t l, r;
};

// frontier of level s from level s - 1: four lanes per record, one per appended base
__global__ __launch_bounds__(256) void dtab_expand_kernel(DevIndexView ix, const DtabRec *in, uint32_t n_in, DtabRec *out, uint32_t cap,
                                                          uint32_t *count)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t e = (uint32_t)(t >> 2), c = (uint32_t)(t & 3u);
    bool ok = false;
    DtabRec o{0, 0, 0};
    if (e < n_in) {
        const DtabRec rec = in[e];
        const uint8_t *arena = reinterpret_cast<const uint8_t *>(ix.arena);
        const uint32_t bl = div96(rec.l), br = div96(rec.r);
        const uint4 xA = ld16(arena, (c * ix.n_blocks + bl) << 4), xB = ld16(arena, (c * ix.n_blocks + br) << 4);
        o.l = rank_eval(xA, rec.l - bl * kRankRows);
        o.r = rank_eval(xB, rec.r - br * kRankRows);
        o.key = (rec.key << 2) | c;
        ok = o.l < o.r;
    }
    const uint64_t m = __ballot(ok);
    if (m == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(m));
    base = __shfl(base, 0);
    if (ok) {
        const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (slot < cap) out[slot] = o;
    }
}

// T[key] = s for the present strings of s characters (value 0x80 on the last level: "all `order` bases, extension bits follow")
__global__ __launch_bounds__(256) void dtab_scatter_kernel(const DtabRec *recs, uint32_t n, uint8_t value, uint8_t *tab)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) tab[recs[t].key] = value;
}

// plan_kernel's seed table: {l, r} of the present strings of seed_d characters (the table is zeroed first: l >= r = absent)
__global__ __launch_bounds__(256) void dtab_seed_kernel(const DtabRec *recs, uint32_t n, uint2 *seed)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) seed[recs[t].key] = make_uint2(recs[t].l, recs[t].r);
}

// anchors: the strings of `order` characters that are the suffix of exactly one row -> hash slot {key tag, text position of the row}
__global__ __launch_bounds__(256) void dtab_anchor_kernel(const DtabRec *recs, uint32_t n, const uint32_t *pc_pos, uint64_t *slots, uint32_t bits)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const DtabRec rec = recs[t];
    if (rec.r != rec.l + 1u) return;
    const uint64_t mask = ((uint64_t)1 << bits) - 1ull;
    const uint64_t v = ((uint64_t)((uint32_t)rec.key + 1u) << 32) | pc_pos[rec.l];
    uint64_t h = (rec.key * 0x9E3779B97F4A7C15ull) >> (64u - bits);
    for (;;) { // (twice as many slots as strings: always ends)
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long *>(slots + h), 0ull, (unsigned long long)v);
        if (old == 0ull) return;
        h = (h + 1u) & mask;
    }
}

// the present strings of order + 1 characters: bit (oldest base) of the entry of their last `order` bases
__global__ __launch_bounds__(256) void dtab_ext_kernel(const DtabRec *recs, uint32_t n, uint32_t order, uint8_t *tab)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const uint64_t key = recs[t].key, y = key & ((1ull << (2u * order)) - 1ull);
    const uint32_t c = (uint32_t)(key >> (2u * order)) & 3u;
    atomicOr(reinterpret_cast<uint32_t *>(tab + (y & ~3ull)), (1u << c) << (8u * (uint32_t)(y & 3ull)));
}

// The table the look-ups go to is GROUPED: the windows of three consecutive bases i0, i0 + 1, i0 + 2 (i0 a multiple of 3 in
// the item) share their middle order - 2 bases (the "core"), and the 48 entries that differ in the two bases around it
//     slot  0 .. 15   window of i0     = 2 older bases . core            (slot = the older bases)
//     slot 16 .. 31   window of i0 + 1 = 1 older base . core . 1 newer   (slot = 16 + 4 older + newer)
//     slot 32 .. 47   window of i0 + 2 = core . 2 newer bases            (slot = 32 + the newer bases)
// sit in the 64 bytes of that core: the three look-ups of a triple are ONE line fill (the stage is bound by the number of
// fills: DESIGN.md section 4.2).  4^(order - 2) cores x 64 bytes = four times the plain table.
__global__ __launch_bounds__(256) void dtab_regroup_kernel(const uint8_t *plain, uint32_t order, uint8_t *grouped)
{
    const uint64_t core = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; // one lane per core: its 48 entries, one 64-byte line
    const uint32_t cb = 2u * (order - 2u);
    if (core >> cb) return;
    uint32_t w[16];
#pragma unroll
    for (uint32_t q = 0; q < 12; q++) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            const uint32_t slot = 4u * q + b;
            uint64_t key;
            if (slot < 16u) key = ((uint64_t)slot << cb) | core;
            else if (slot < 32u) key = ((uint64_t)((slot - 16u) >> 2) << (cb + 2u)) | (core << 2) | (slot & 3u);
            else key = (core << 4) | (slot - 32u);
            v |= (uint32_t)plain[key] << (8u * b);
        }
        w[q] = v;
    }
    w[12] = w[13] = w[14] = w[15] = 0;
    uint4 *o = reinterpret_cast<uint4 *>(grouped + (core << 6));
    o[0] = make_uint4(w[0], w[1], w[2], w[3]);
    o[1] = make_uint4(w[4], w[5], w[6], w[7]);
    o[2] = make_uint4(w[8], w[9], w[10], w[11]);
    o[3] = make_uint4(0, 0, 0, 0);
}

// the value of base i of an item (start = its offset in the query buffer) as far as the table knows: false = L, true = deeper
// than the table tells (L == kDtabUnknown then: no window at all, the first bytes of the buffer)
template <bool STATS>
__device__ __forceinline__ bool dtab_look(const WalkArgs &a, uint32_t start, uint32_t i, uint32_t &L, uint32_t &st_look)
{
    const uint32_t order = a.ix.dtab_order, k = a.ix.k;
    const uint64_t end1 = (uint64_t)start + i + 1u; // one past the base, in the query buffer
    L = kDtabUnknown;
    if (end1 < 32u) return true; // (no 32 bytes in front of them; no anchor either)
    L = 0;
    const uint8_t *p = a.q + (end1 - 32u);
    uint4 hi4, lo4;
    __builtin_memcpy(&lo4, p, 16);      // the older 16 bases
    __builtin_memcpy(&hi4, p + 16, 16); // the newer 16 bases
    uint32_t c_old, v_old, c_new, v_new;
    pack16(lo4, c_old, v_old);
    pack16(hi4, c_new, v_new);
    const uint64_t code = ((uint64_t)c_old << 32) | c_new;
    // consecutive bases ending at the newest byte (bit 15 of v_new), then into the older block
    const uint32_t run_new = (uint32_t)__clz((int)~(v_new << 16));
    const uint32_t run_old = (uint32_t)__clz((int)~(v_old << 16));
    uint32_t v = run_new < 16u ? run_new : 16u + min(run_old, 16u);
    v = min(v, i + 1u);
    const uint64_t key = code & ((1ull << (2u * order)) - 1ull);
    const uint32_t byte = a.ix.dtab_grouped ? a.ix.dtab[dtab_grouped_addr(key, i % 3u, order)] : a.ix.dtab[key];
    if (STATS) st_look++;
    if (!(byte & 0x80u)) {
        L = min(byte, v);
        return false;
    }
    // all `order` bases of the window are a suffix of a row
    if (v <= order || order >= k) {
        L = min(v, order);
        return false;
    }
    if (!((byte >> ((uint32_t)(code >> (2u * order)) & 3u)) & 1u)) {
        L = order;
        return false;
    }
    return true;
}
// ... and off the path-cover text when the window is an anchor (L == kDtabUnknown on entry: no window, stays unknown)
template <bool STATS>
__device__ __forceinline__ bool dtab_anchor_at(const WalkArgs &a, uint32_t start, uint32_t i, uint32_t &L, uint32_t &st_anch)
{
    if (L == kDtabUnknown) return true;
    const uint8_t *qi = a.q + (uint64_t)start + i;
    L = dtab_anchor_depth(a.ix, i + 1u, [qi](uint32_t t) -> uint32_t { return qi[-(int64_t)t]; });
    if (STATS) st_anch++;
    return L == kDtabUnknown;
}

// -------------------------------------------------------------------------------------------------------------
// dtab_resolve_kernel<GW>: GW (16 or 32 >= order + 1) lanes per work item.  For every mismatch m of the item's list, lane j
// has base i = m + j (up to the next mismatch, the item's end, and order + 1 bases):
//     v    = bases of the window that count: consecutive A/C/G/T ending at i, inside the item (i + 1)
//     L    = longest present suffix of those: min(T, v); when all `order` bases of the window are present and v > order,
//            `order` if the string with one more base to the left is absent, UNKNOWN (the item is flagged) if it is present
//     conv = L <= j: the longest present suffix is the j bases behind the mismatch -> from here to the next mismatch the
//            predicted values min(k, bases since the mismatch) are exact (a longer suffix further on would contain the absent
//            string query[m .. i])
// lanes in front of the first conv write L.  j = order always decides (L <= order), so order + 1 lanes are enough.  A stretch
// with an UNKNOWN in front of its conv writes nothing and flags the item; the item's other stretches are treated as if it
// were not (what they write is right, and the plain walk writes it again).
// STATS: counts its look-ups, written values and flagged items (kPlanStat*), pinned by the CPU model.
template <int GW, bool STATS>
__global__ __launch_bounds__(256) void dtab_resolve_kernel(WalkArgs a)
{
    const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / GW, j = threadIdx.x & (GW - 1);
    const uint32_t sub = (threadIdx.x & 63u) / GW; // group inside the wave
    const uint32_t order = a.ix.dtab_order, k = a.ix.k;
    uint32_t st_look = 0, st_written = 0, st_flag = 0, st_anch = 0;
    bool flag = false;
    const bool have = gid < a.n_items;
    uint4 g = make_uint4(0, 0, 0, kPlanNone << 24);
    if (have) g = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.gitems) + (size_t)gid * 16u);
    const uint32_t start = g.x, len = g.z & 0xFFFFu, mm0 = g.w & 0xFFFFu, warm = (g.w >> 16) & 0xFFu, n_mm = g.w >> 24;
    uint32_t n_loop = 0;
    bool no_plan = false; // no seed, no diagonal: every base of the item from the table
    if (have && len != 0) {
        if (n_mm == kPlanNone) no_plan = true;
        else if (n_mm > a.plan_list + 1u) flag = true; // more mismatches than the list holds (a wrong diagonal): the plain walk
        else n_loop = n_mm;
    }
    auto look = [&](uint32_t i, uint32_t &L) -> bool { return dtab_look<STATS>(a, start, i, L, st_look); };
    auto anchor = [&](uint32_t i, uint32_t &L) -> bool { return dtab_anchor_at<STATS>(a, start, i, L, st_anch); };
    // the list: lane t holds mismatch t (t <= 28 < GW * 2: two per lane when GW = 16)
    const uint16_t *list = a.glist + (size_t)gid * a.plan_list;
    uint32_t mine0 = kPlanInf, mine1 = kPlanInf;
    if (j < n_loop) mine0 = j == 0 ? mm0 : (uint32_t)list[j - 1u];
    if (GW == 16 && j + 16u < n_loop) mine1 = (uint32_t)list[j + 15u];
    const uint64_t gmask = (GW == 64 ? ~0ull : ((1ull << GW) - 1ull)) << (sub * GW);
    auto wave_max = [&](uint32_t v) { // (wave-uniform trip counts)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
        return v;
    };
    const uint32_t max_loop = wave_max(n_loop);
    for (uint32_t t = 0; t < max_loop; t++) {
        const int src = (int)(sub * GW + (GW == 16 ? (t & 15u) : t));
        const uint32_t m_a = __shfl(mine0, src), m_b = __shfl(mine1, src);
        const uint32_t m = (GW == 16 && t >= 16u) ? m_b : m_a;
        const int srcn = (int)(sub * GW + (GW == 16 ? ((t + 1u) & 15u) : t + 1u));
        const uint32_t n_a = __shfl(mine0, srcn & 63), n_b = __shfl(mine1, srcn & 63);
        uint32_t nxt = (GW == 16 && t + 1u >= 16u) ? n_b : n_a;
        if (t + 1u >= n_loop) nxt = len;
        const uint32_t i = m + j;
        const bool act = t < n_loop && j <= order && i < nxt && i < len; // (every mismatch on its own, flagged item or not)
        uint32_t L = 0;
        bool sat = false;
        if (act) sat = look(i, L);
        const bool conv = act && !sat && L <= j;
        const uint64_t bc = __ballot(conv) & gmask;
        const uint32_t first_conv = bc ? (uint32_t)__ffsll((long long)bc) - 1u - sub * GW : (uint32_t)GW;
        // the bases in front of the first conv that the table cannot tell: the anchors (their values exceed `order`: no conv)
        if (sat && j < first_conv) sat = anchor(i, L);
        const uint64_t bs = __ballot(sat) & gmask;
        const uint32_t first_sat = bs ? (uint32_t)__ffsll((long long)bs) - 1u - sub * GW : (uint32_t)GW;
        if (first_sat < first_conv) flag = true; // (group-uniform: every lane of the group sees the same ballots)
        else if (act && j <= first_conv && i >= warm) {
            a.d_out[(uint64_t)start + i] = (uint8_t)min(L, k);
            if (STATS) st_written++;
        }
    }
    // items without a seed (unrelated reads, the other strand): all their bases, GW at a time -
    // every value on its own; one the table (and the anchors) cannot tell sends the item to the plain walk
    const uint32_t n_blk = no_plan ? (len - min(len, warm) + GW - 1u) / GW : 0u, max_blk = wave_max(n_blk);
    for (uint32_t b = 0; b < max_blk; b++) {
        const uint32_t i = warm + b * GW + j;
        const bool act = b < n_blk && i < len;
        uint32_t L = 0;
        bool sat = false;
        if (act) sat = look(i, L);
        if (STATS && sat && L != kDtabUnknown) st_anch++; // (deeper than the table knows: no anchors for items without a plan)
        if (act && !sat) {
            a.d_out[(uint64_t)start + i] = (uint8_t)min(L, k);
            if (STATS) st_written++;
        }
        if (__ballot(sat) & gmask) flag = true;
    }

    if (flag && j == 0) {
        a.redo[gid] = 1;
        if (STATS) st_flag++;
    }
    // items flagged by this wave -> qctl[4] (redo_collect_kernel gives the plan up when most of a launch is flagged), items
    // without a plan -> qctl[5] (the same: a batch of them is cheaper to walk plainly)
    const uint64_t fm = __ballot(flag && j == 0), nm = __ballot(no_plan && j == 0);
    if ((threadIdx.x & 63u) == 0) {
        if (fm) atomicAdd(a.qctl + 4, (uint32_t)__popcll(fm));
        if (nm) atomicAdd(a.qctl + 5, (uint32_t)__popcll(nm));
    }
    if (STATS) plan_stats_add(a.pstats, kPlanStatTabLookups, st_look, kPlanStatTabWritten, st_written, kPlanStatTabFlagged, st_flag, kPlanStatTabAnchored, st_anch);
}

// -------------------------------------------------------------------------------------------------------------
// dtab_stretch_kernel: the stretches the fused plan_kernel could not finish from the table alone (a base deeper than the
// table knows in front of the stretch's end), with the anchors.  plan_kernel leaves them in a per-wave list (DtabStretchBlock:
// it would otherwise wait, all 64 lanes, for the two dependent loads of one lane's anchor); here 32 lanes take one stretch,
// one base each, like dtab_resolve_kernel: what the table tells, the anchors for the bases in front of the first end, the
// values written when all of them are known - else the item is flagged (once) for the plain walk.
template <bool STATS>
__global__ __launch_bounds__(64) void dtab_stretch_kernel(WalkArgs a, uint32_t n_waves)
{
    const uint32_t w = blockIdx.x; // the plan_kernel wave whose list this is
    if (w >= n_waves) return;
    const uint8_t *blk = reinterpret_cast<const uint8_t *>(a.units) + (size_t)w * kDtabStretchBlockBytes;
    const uint32_t count = min(*reinterpret_cast<const uint32_t *>(blk), kDtabStretchCap);
    if (count == 0) return;
    const uint32_t lane = threadIdx.x, sub = lane >> 5, j = lane & 31u;
    const uint32_t order = a.ix.dtab_order, k = a.ix.k;
    const uint64_t gmask = 0xFFFFFFFFull << (32u * sub);
    uint32_t st_written = 0, st_flag = 0, st_anch = 0, st_dummy = 0;
    for (uint32_t e0 = 0; e0 < count; e0 += 2u) {
        const uint32_t e = e0 + sub;
        const bool have = e < count;
        uint4 ent = make_uint4(0, 0, 0, 0);
        if (have) ent = *reinterpret_cast<const uint4 *>(blk + 16u + (size_t)e * 16u);
        const uint32_t item = ent.x, start = ent.y, m = ent.z & 0xFFFFu, nxt = ent.z >> 16, len = ent.w & 0xFFFFu, warm = ent.w >> 16;
        const uint32_t i = m + j;
        const bool act = have && j <= order && i < nxt && i < len;
        uint32_t L = 0;
        bool sat = false;
        if (act) sat = dtab_look<false>(a, start, i, L, st_dummy); // (the look-ups were counted by plan_kernel)
        const bool conv = act && !sat && L <= j;
        const uint64_t bc = __ballot(conv) & gmask;
        const uint32_t first_conv = bc ? (uint32_t)__ffsll((long long)bc) - 1u - 32u * sub : 32u;
        if (sat && j < first_conv) sat = dtab_anchor_at<STATS>(a, start, i, L, st_anch);
        const uint64_t bs = __ballot(sat) & gmask;
        const uint32_t first_sat = bs ? (uint32_t)__ffsll((long long)bs) - 1u - 32u * sub : 32u;
        if (first_sat < first_conv) { // still unknown: the item takes the plain walk (flagged and counted once)
            if (have && j == 0) {
                uint32_t *word = reinterpret_cast<uint32_t *>(a.redo + (item & ~3u));
                const uint32_t bit = 1u << (8u * (item & 3u));
                const uint32_t old = atomicOr(word, bit);
                if (!(old & (0xFFu << (8u * (item & 3u))))) {
                    atomicAdd(a.qctl + 4, 1u);
                    if (STATS) st_flag++;
                }
            }
        } else if (act && j <= first_conv && i >= warm) {
            a.d_out[(uint64_t)start + i] = (uint8_t)min(L, k);
            if (STATS) st_written++;
        }
    }
    if (STATS) plan_stats_add(a.pstats, kPlanStatTabWritten, st_written, kPlanStatTabFlagged, st_flag, kPlanStatTabAnchored, st_anch, 0, 0);
}

} // namespace

// ---- host side -------------------------------------------------------------------------------------------------
// Builds the table of `order` bases for the index behind `ix` into d_tab (4^order bytes).  d_tmp: two frontiers of
// frontier_cap records each (dtab_tmp_bytes).  Synchronous (one small copy per level).
size_t dtab_tmp_bytes(uint64_t frontier_cap) { return 2 * (size_t)frontier_cap * sizeof(DtabRec) + 64; }

// plain table (4^order bytes at d_plain) -> grouped table (dtab_bytes(order, true) bytes at d_grouped)
hipError_t regroup_depth_table(const uint8_t *d_plain, uint32_t order, uint8_t *d_grouped, hipStream_t stream)
{
    if (order < 3u) return hipErrorInvalidValue;
    const uint64_t cores = (uint64_t)1 << (2u * (order - 2u));
    hipLaunchKernelGGL(dtab_regroup_kernel, dim3((uint32_t)((cores + 255) / 256)), dim3(256), 0, stream, d_plain, order, d_grouped);
    const hipError_t e = hipStreamSynchronize(stream);
    return e != hipSuccess ? e : hipGetLastError();
}

namespace {
// the table of level F as one bit per string: "these F bases are a suffix of a row" (map_kernels.hip's filter)
__global__ __launch_bounds__(256) void dtab_filter_kernel(const uint8_t *__restrict__ tab, uint32_t F, uint32_t *__restrict__ out, uint64_t n_words)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint4 a = *reinterpret_cast<const uint4 *>(tab + 32u * w), b = *reinterpret_cast<const uint4 *>(tab + 32u * w + 16u);
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t bits = 0;
#pragma unroll
    for (uint32_t t = 0; t < 32u; t++) bits |= (((v[t >> 2] >> (8u * (t & 3u))) & 0xFFu) == F ? 1u : 0u) << t;
    out[w] = bits;
}
} // namespace

hipError_t build_depth_table(const DevIndexView &ix, uint32_t order, uint8_t *d_tab, void *d_tmp, uint64_t frontier_cap, hipStream_t stream,
                             uint64_t *d_anchor, uint32_t anchor_bits, uint2 *d_seed, uint32_t seed_d, uint32_t *d_filter, uint32_t filter_bases)
{
    if (d_filter && (filter_bases < 3u || filter_bases >= order)) return hipErrorInvalidValue;
    if (d_seed) {
        if (seed_d == 0 || seed_d > order || seed_d > 14u) return hipErrorInvalidValue;
        const hipError_t es = hipMemsetAsync(d_seed, 0, (size_t)8 << (2u * seed_d), stream);
        if (es != hipSuccess) return es;
    }
    if (d_anchor) {
        if (!ix.pc_pos || anchor_bits < 4u || anchor_bits > 40u) return hipErrorInvalidValue;
        const hipError_t ea = hipMemsetAsync(d_anchor, 0, ((size_t)1 << anchor_bits) * 8, stream);
        if (ea != hipSuccess) return ea;
    }
    if (order == 0 || order > 17u || order > ix.k) return hipErrorInvalidValue;
    DtabRec *fa = reinterpret_cast<DtabRec *>(d_tmp), *fb = fa + frontier_cap;
    uint32_t *count = reinterpret_cast<uint32_t *>(fb + frontier_cap);
    const DtabRec root{0, 0, ix.n};
    hipError_t e = hipMemcpyAsync(fa, &root, sizeof(root), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(d_tab, 0, 1, stream); // level 0: the empty string, depth 0
    if (e != hipSuccess) return e;
    uint32_t n_in = 1;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(frontier_cap, 0xFFFFFFF0ull);
    const uint32_t levels = order < ix.k ? order + 1u : order; // (order == k: depths are capped at k, no extension bits)
    for (uint32_t s = 1; s <= levels; s++) {
        e = hipMemsetAsync(count, 0, 4, stream);
        if (e != hipSuccess) return e;
        const uint64_t lanes = 4ull * n_in;
        hipLaunchKernelGGL(dtab_expand_kernel, dim3((uint32_t)((lanes + 255) / 256)), dim3(256), 0, stream, ix, fa, n_in, fb, cap, count);
        uint32_t n_out = 0;
        e = hipMemcpyAsync(&n_out, count, 4, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (n_out > cap) return hipErrorOutOfMemory; // (cannot happen with frontier_cap >= n: distinct strings <= rows)
        if (s <= order) {
            // T_s[c . Y] = T_{s-1}[Y] unless c . Y itself is present: quadrant 0 is T_{s-1} already, copied to the other three
            const size_t quad = (size_t)1 << (2u * (s - 1u));
            for (int c = 1; c < 4; c++) {
                e = hipMemcpyAsync(d_tab + quad * c, d_tab, quad, hipMemcpyDeviceToDevice, stream);
                if (e != hipSuccess) return e;
            }
            if (n_out)
                hipLaunchKernelGGL(dtab_scatter_kernel, dim3((n_out + 255) / 256), dim3(256), 0, stream, fb, n_out,
                                   (uint8_t)(s == order ? 0x80u : s), d_tab);
            if (d_filter && s == filter_bases) { // (the table of this level is complete: T[key] == s <=> the string is present)
                const uint64_t n_words = ((uint64_t)1 << (2u * s)) / 32u;
                hipLaunchKernelGGL(dtab_filter_kernel, dim3((uint32_t)((n_words + 255) / 256)), dim3(256), 0, stream, d_tab, s, d_filter, n_words);
            }
            if (n_out && s == seed_d && d_seed) hipLaunchKernelGGL(dtab_seed_kernel, dim3((n_out + 255) / 256), dim3(256), 0, stream, fb, n_out, d_seed);
            if (n_out && s == order && d_anchor)
                hipLaunchKernelGGL(dtab_anchor_kernel, dim3((n_out + 255) / 256), dim3(256), 0, stream, fb, n_out, ix.pc_pos, d_anchor, anchor_bits);
        } else if (n_out) {
            hipLaunchKernelGGL(dtab_ext_kernel, dim3((n_out + 255) / 256), dim3(256), 0, stream, fb, n_out, order, d_tab);
        }
        std::swap(fa, fb);
        n_in = n_out;
        if (n_in == 0) { // nothing this long: the deeper levels are copies of this one
            if (d_filter && s < filter_bases) { // (no string of filter_bases bases is present)
                e = hipMemsetAsync(d_filter, 0, ((size_t)1 << (2u * filter_bases)) / 8u, stream);
                if (e != hipSuccess) return e;
            }
            for (uint32_t s2 = s + 1; s2 <= order; s2++) {
                const size_t quad = (size_t)1 << (2u * (s2 - 1u));
                for (int c = 1; c < 4; c++) {
                    e = hipMemcpyAsync(d_tab + quad * c, d_tab, quad, hipMemcpyDeviceToDevice, stream);
                    if (e != hipSuccess) return e;
                }
            }
            break;
        }
    }
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    return hipGetLastError();
}

// behind the fused plan_kernel: the stretches it left to the anchors (n_waves = its waves)
hipError_t launch_dtab_stretches(const WalkArgs &a, uint32_t n_waves, hipStream_t stream)
{
    if (n_waves == 0) return hipSuccess;
    if (a.pstats) hipLaunchKernelGGL((dtab_stretch_kernel<true>), dim3(n_waves), dim3(64), 0, stream, a, n_waves);
    else hipLaunchKernelGGL((dtab_stretch_kernel<false>), dim3(n_waves), dim3(64), 0, stream, a, n_waves);
    return hipGetLastError();
}

// behind plan_kernel (launch_plan_table in plan_kernels.hip): the items' stretches behind their mismatches from the table
hipError_t launch_dtab_resolve(const WalkArgs &a, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    const bool wide = a.ix.dtab_order + 1u > 16u;
    const uint32_t gw = wide ? 32u : 16u;
    const uint64_t lanes = (uint64_t)a.n_items * gw;
    const dim3 grid((uint32_t)((lanes + 255) / 256)), block(256);
    if (wide) {
        if (a.pstats) hipLaunchKernelGGL((dtab_resolve_kernel<32, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((dtab_resolve_kernel<32, false>), grid, block, 0, stream, a);
    } else {
        if (a.pstats) hipLaunchKernelGGL((dtab_resolve_kernel<16, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((dtab_resolve_kernel<16, false>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

} // namespace kbo
