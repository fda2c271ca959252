// walk_kernels.hip — gfx950 (MI355X, CDNA4): A1, the k-bounded matching-statistics walk, and the item lists it consumes.
//
//   ms_walk_kernel           A1  sbwt::StreamingIndex::matching_statistics
//                                (called at reference index.rs:251-252)
//   make_items_kernel, make_chunk_items_kernel   offsets -> work items (reads / chunked long sequences)
// (A5+A6 live in derand_kernels.hip, format::run_lengths_gapped in rle_kernels.hip.)
//
// Integer / bit work only: no MFMA.  Wavefront = 64 lanes.
//
// Walk mapping: ONE LANE PER QUERY.  Every step of the walk is a dependent chain
// (interval -> two rank blocks -> new interval), so throughput comes from the number
// of independent chains in flight, not from lanes cooperating on one rank: with the
// 16-byte rank block a rank is one aligned load + three masked popcounts, and 64 lanes
// x 8 waves/SIMD x 1024 SIMDs = 524,288 chains hide the L2/MALL/HBM latency.  A lane is
// a small state machine; every iteration every lane issues exactly two 16-byte loads
// from the index arena (rank blocks or LCS windows) through one load site, then all
// lanes consume.  Divergence costs only the per-state post-processing.
#include "device_util.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace kbo {
namespace {

// per-lane flag bits of the walk
enum : uint32_t {
    F_QF = 1u,    // fetch the next query word (serviced in the hot path)
    F_HAVE = 2u,  // next item (descriptor + first two query words) is prefetched
    F_PF = 4u,    // next item's descriptor is in flight, its query words not yet requested
    // bits >= F_BLOCK take the lane out of the hot path until the rare block has run
    F_CON = 8u,   // contracting: loads contraction entries instead of rank blocks
    F_NOPAIR = 16u, // the two-base step failed for the current base: take it alone
    F_DONE = 32u, // finished its item, wants the next one
    F_FIN = 64u,  // no items left
    F_BLOCK = 32u
};

// -------------------------------------------------------------------------------------
// A1.  Semantics:
//   for each base c:  Ic = extend_right(I, c)
//                     while d > 0 && Ic empty:  I = contract_left(I, d-1); d -= 1; Ic = extend_right(I, c)
//                     if Ic non-empty: I = Ic; d = min(d+1, k)
//                     emit (d, I)
// A non-ACGT base reads the all-zero "null" rank block, so its extension is empty at every
// depth and the contraction machinery takes it down to the root (d = 0, I = [0,n)), which is
// what the loop above does.
//
// The kernel is bound by the L2 request pipeline on L2-resident indexes and by line fills beyond
// (DESIGN.md section 6), so it is one loop with
//   a hot path (every iteration, lanes not blocked): two 16-byte loads per lane - rank blocks
//             for extending lanes, contraction entries for contracting ones, two-base blocks
//             for PAIR lanes - then the arithmetic of all three kinds written with selects,
//             accept / emit / advance to the next base;
//   a rare block (entered when >= rare_batch lanes are blocked, or every rare_mask+1-th
//             iteration if any is): switching to the prefetched next item, requesting the
//             prefetch after that, exit test.
// One lane per work item; lane j of wave w walks items w*64*rounds + j + 64*t.  All
// offsets are 32-bit (one launch covers < 4 GiB of query and an index arena < 4 GiB for
// the 32-bit build), so every access is SGPR base + 32-bit VGPR offset.
// KBO_NO_TARGETS=1 (compile time) drops the nearest-set-bit targets of the contraction: a lane
// then climbs one level and re-tries; measured -2 % walk time at 1 % substitutions, +4.5 % at 5 %.
#ifndef KBO_NO_TARGETS
#define KBO_NO_TARGETS 0
#endif

// Appends the MS value of base i of the item to its output stream (16-byte item-relative blocks:
// output byte e = i - warm).  fin_e: base i is the last one of the item.
__device__ __forceinline__ void emit_ms(uint8_t *d_out, uint32_t start, uint32_t warm, uint32_t i, bool fin_e,
                                        uint32_t dval, uint32_t &ocur, uint4 &oblk)
{
    const uint32_t e = i - warm;
    ocur |= dval << ((e & 3u) * 8u);
    if ((e & 3u) == 3u || fin_e) { // word complete (or item ends): move it into the block
        const uint32_t w = (e >> 2) & 3u;
        oblk.x = w == 0 ? ocur : oblk.x;
        oblk.y = w == 1 ? ocur : oblk.y;
        oblk.z = w == 2 ? ocur : oblk.z;
        oblk.w = w == 3 ? ocur : oblk.w;
        ocur = 0;
        if ((e & 15u) == 15u) { // full block: one unaligned 16-byte store
            st16u(d_out, start + warm + (e & ~15u), oblk);
        } else if (fin_e) { // tail of the item: words, then bytes
            st_partial(d_out + (start + warm + (e & ~15u)), oblk, (e & 15u) + 1u);
        }
    }
}

// PAIR: the index carries two-base extension blocks (DevIndexView::pair_off) and lanes that are deep
// in a match extend by two bases per iteration: pair_extend(I, c1 c2) = extend(extend(I, c1), c2)
// exactly (same block format, bit i of D_{c1c2} = B_c1[i] & B_c2[C[c1] + rank_c1(i)]), and a
// non-empty result means neither step needed a contraction, so the two emitted values are
// min(d+1, k) and min(d+2, k).  An empty result says nothing: the lane falls back to single steps
// for that base.  Halves the line fills per base where the index does not fit L2.
// CALL: the lane also runs the breakpoint scan of call_variants on the values it produces: a base whose MS value drops
// from >= t to < t is a breakpoint; the first base within k to its right with MS >= t and a single-row interval resolves
// it (and every other breakpoint still waiting).  Up to four breakpoints wait per lane; a fifth within k bases sets the
// overflow counter (call_counts[16 * kCallSegs]) and the host falls back to the stand-alone scan for the batch.
template <bool IVAL, bool BIG, bool PAIR, bool CALL = false>
__global__ __launch_bounds__(256) void ms_walk_kernel(WalkArgs a)
{
    const uint32_t n = a.ix.n, k = a.ix.k;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint8_t *arena = reinterpret_cast<const uint8_t *>(a.ix.arena);
    const uint8_t *qb = a.q;
    const uint8_t *itb = reinterpret_cast<const uint8_t *>(a.items);
    const uint32_t q_end = (uint32_t)a.q_bytes;
    const uint32_t nblk = a.ix.n_blocks;
    const uint32_t null_blk = 4u * nblk; // all-zero rank block (non-ACGT bases)
    const uint32_t ent_byte0 = a.ix.lcs_off << 4; // arena byte offset of contraction entry 0
    const uint32_t pair_blk0 = a.ix.pair_off;     // arena index of the first two-base block (PAIR)

    // this lane's items: first, first + 64, ...
    const uint32_t n_items = a.n_items_dev ? min(*a.n_items_dev, a.n_items) : a.n_items; // (redo pass: counted on the device)
    // (redo pass: as many items per lane as the launch needs to cover the list - one, unless the list is long)
    const uint32_t lanes_all = gridDim.x * blockDim.x;
    const uint32_t rounds = a.n_items_dev ? max(1u, (n_items + lanes_all - 1u) / lanes_all) : a.rounds;
    const uint64_t first64 = (uint64_t)wave * 64u * rounds + lane;
    uint32_t next_item = (first64 < n_items && lane < a.lane_limit) ? (uint32_t)first64 : n_items;
    uint32_t left = 0; // items whose descriptor has not been requested yet
    if (next_item < n_items) left = min(rounds, (n_items - 1u - next_item) / 64u + 1u);

    uint32_t flags = left ? F_DONE : F_FIN;
    uint32_t l = 0, r = n, d = 0, m = 0, cb = 0; // m: contraction targets known (rare block only)
    uint32_t pcb = 0; // PAIR: first two-base block of (current base, next base), 0 = no pair step here
    uint32_t pmin = 0; // PAIR: depth from which pair steps are tried: 0 until the item's first failure (a read
                       // is expected to match from its first base), pair_min_d afterwards (random matches)
    uint32_t tgt_l = 0, tgt_r = 0; // contraction targets (rare block only)
    // Query and output are streamed in 16-byte blocks RELATIVE TO THE ITEM (unaligned global
    // accesses): i = base index inside the item; block i>>4, word (i>>2)&3, byte i&3.
    uint32_t i = 0, len = 0, warm = 0, start = 0;
    uint32_t tail = 0, dprev = 0, np = 0, pend0 = 0, pend1 = 0, pend2 = 0, pend3 = 0; // CALL: see above
    uint4 qblk = make_uint4(0, 0, 0, 0), qnxt = make_uint4(0, 0, 0, 0); // current / next query block
    uint32_t qcur = 0;                                                  // current query word
    uint4 oblk = make_uint4(0, 0, 0, 0);                                // output block being filled
    uint32_t ocur = 0;                                                  // output word being filled
    uint4 nit = make_uint4(0, 0, 0, 0); // prefetched descriptor of the next item
    uint4 nq0 = make_uint4(0, 0, 0, 0); // and its first query block

    if (left) {
        nit = ld16(itb, next_item * 16u);
        next_item += 64;
        left--;
        flags |= F_PF;
    }

    uint32_t dbg_rare = 0, dbg_con = 0, dbg_iter = 0, dbg_ext = 0, dbg_fail = 0, dbg_wait = 0, dbg_fin = 0;
    for (;;) {
        // ================================ rare block ================================
        // entered every rare_period iterations: the hot loop below carries no item bookkeeping at
        // all, and a lane that finishes its item waits at most rare_period - 1 iterations
        {
            dbg_rare++;
            // ---- request the first query block of the next item once its descriptor is here
            if (flags & F_PF) {
                nq0 = ld16u(qb, nit.x);
                flags = (flags & ~F_PF) | F_HAVE;
            }
            // ---- switch finished lanes to their prefetched item
            if (flags & F_DONE) {
                if (flags & F_HAVE) {
                    start = nit.x;
                    len = nit.z;
                    warm = nit.w & 0xFFFFu;
                    if (CALL) {
                        tail = nit.w >> 16;
                        np = 0;
                        dprev = 0;
                    }
                    i = 0;
                    qblk = nq0;
                    qcur = qblk.x;
                    const uint32_t c = decode_base(qcur & 0xFFu);
                    cb = c < 4u ? c * nblk : null_blk;
                    if (PAIR) {
                        const uint32_t c2 = len > 1u ? decode_base((qcur >> 8) & 0xFFu) : 4u;
                        pcb = (c < 4u && c2 < 4u) ? pair_blk0 + (c * 4u + c2) * nblk : 0u;
                        pmin = 0;
                    }
                    l = 0;
                    r = n;
                    d = 0;
                    ocur = 0;
                    flags = len ? F_QF : F_DONE; // fetch block 1 right away; empty items are skipped
                    if (left) {
                        nit = ld16(itb, next_item * 16u);
                                        next_item += 64;
                        left--;
                        flags |= F_PF;
                    }
                } else {
                    flags = F_FIN;
                }
            }
            if (__ballot(flags != F_FIN) == 0) break;
        }

#pragma unroll 1
        for (uint32_t it = 0; it < a.rare_period; it++) {
        dbg_iter++;
#ifdef KBO_WALK_DEBUG
        dbg_wait += (flags & F_DONE) && !(flags & F_FIN) ? 1u : 0u; // finished an item, waiting for the bookkeeping visit
        dbg_fin += (flags & F_FIN) ? 1u : 0u;                       // out of items, waiting for the wave to end
#endif
        // ================================= hot path =================================
        // A lane is either extending (two rank blocks) or contracting (two contraction
        // entries {lcs, psv, nsv}); both kinds of load go through the same two load sites and
        // are consumed one wait later, so contracting lanes never stall the wave.
        //
        // Contraction (bit-identical to the reference's d-1, d-2, ... loop):
        //  (i) contract_left(I, t) leaves I unchanged for t > m = max(lcs[l], lcs[r]) and at
        //      t = m moves exactly the side(s) whose boundary value is m, to psv[l] / nsv[r];
        //  (ii) the extension stays empty until the interval reaches the nearest set bit of
        //      B_c below l (row tgt_l) or passes the nearest one at/after r (row tgt_r - 1).
        // So a failing lane climbs one LCS-interval-tree level per iteration until (ii) holds
        // and then extends successfully at exactly the depth where the reference's loop stops.
        // If a nearest bit lies outside the loaded rank block, it climbs a single level and
        // re-tries the extension.
        if (flags < F_BLOCK) {
            // (written with selects rather than branches where the body is a few instructions:
            // the SIMD issues one instruction of ANY kind per 4 cycles, so exec-mask bookkeeping
            // around a short divergent body costs as much as the body)
            const bool con = (flags & F_CON) != 0;
            const uint32_t bl = div96(l), br = div96(r);
            const uint32_t bmask = cb == null_blk ? 0u : ~0u;
            // (call mode: two-base steps only while no breakpoint is waiting for its match - the interval after the first
            // base of a pair is never computed)
            const bool pair_try = PAIR && !con && pcb != 0u && !(flags & F_NOPAIR) && d >= pmin &&
                                  (!CALL || np == 0u || d + 2u < a.call_thr); // (neither base of the pair can be the match then)
            const uint32_t xb = pair_try ? pcb : cb; // first block of the bit-vector this lane ranks in
            const uint32_t rkA = (xb + (bl & bmask)) << 4, rkB = (xb + (br & bmask)) << 4;
            uint4 xA, xB;
            if (BIG) { // entries live in their own region, 64-bit offsets (n_sets * 12 B >= 4 GiB)
                const uint8_t *pA = con ? a.ix.ent + (uint64_t)l * 12u : arena + rkA;
                const uint8_t *pB = con ? a.ix.ent + (uint64_t)r * 12u : arena + rkB;
                __builtin_memcpy(&xA, pA, 16);
                __builtin_memcpy(&xB, pB, 16);
            } else {
                const uint32_t enA = ent_byte0 + ((l + (l << 1)) << 2), enB = ent_byte0 + ((r + (r << 1)) << 2);
                xA = ld16u(arena, con ? enA : rkA);
                xB = ld16u(arena, con ? enB : rkB);
            }
            if (flags & F_QF) { // the query block after the current one (reads <= 16 bytes past the item)
                qnxt = ld16u(qb, min(start + (i & ~15u) + 16u, q_end)); // stays within the 16-byte slack
                flags &= ~F_QF;
            }
            // ---- contracting lanes: one level up the LCS interval tree
            const uint32_t lv = max(xA.x, xB.x);
            const bool root = lv == 0;
            const uint32_t cl = root ? 0u : (xA.x == lv ? xA.y : l);
            const uint32_t cr = root ? n : (xB.x == lv ? xB.z : r);
            const bool cstop = root || !m || cl <= tgt_l || cr >= tgt_r;
            // ---- extending lanes
            const uint32_t ol = l - bl * kRankRows, orr = r - br * kRankRows;
            const uint32_t l2 = rank_eval(xA, ol), r2 = rank_eval(xB, orr);
            const bool ok = !con && l2 < r2;
            const uint32_t d_one = min(d + 1, k);
            const uint32_t d_ext = pair_try ? min(d + 2, k) : d_one;
            const bool accept = !con && (l2 < r2 || (d == 0 && !pair_try));
            const bool fail = !con && !accept && !pair_try; // a failed pair step proves nothing
            if (PAIR) {
                flags = (pair_try && !ok) ? (flags | F_NOPAIR) : ((accept && !pair_try) ? (flags & ~F_NOPAIR) : flags);
                pmin = (fail || (pair_try && !ok)) ? a.pair_min_d : pmin;
            }
            if (con) dbg_con++;
#ifdef KBO_WALK_DEBUG
            dbg_ext += accept ? 1u : 0u; dbg_fail += fail ? 1u : 0u;
#endif
            // nearest set bits of B_c around [l, r), used when the extension failed (searched only
            // inside the 32-bit word that holds the position: set bits are a few rows apart, and a
            // miss merely costs one extra extension attempt)
            uint32_t dl = 0, dr = 0;
#if !KBO_NO_TARGETS
            {
                const uint32_t wsel = ol >> 5, pb = ol & 31u;
                const uint32_t W = wsel == 0 ? xA.y : (wsel == 1 ? xA.z : xA.w);
                const uint32_t below = W & ((1u << pb) - 1u);
                dl = below ? pb - (31u - (uint32_t)__clz((int)below)) : 0u;
            }
            {
                const uint32_t wsel = orr >> 5, pb = orr & 31u;
                const uint32_t W = wsel == 0 ? xB.y : (wsel == 1 ? xB.z : xB.w);
                const uint32_t above = W & (~0u << pb);
                dr = above ? (uint32_t)__ffs((int)above) - pb : 0u; // (bit index - pb) + 1 rows to pass
            }
#endif
            m = fail ? ((dl && dr) ? 1u : 0u) : m;
            tgt_l = fail ? l - dl : tgt_l; // row of the nearest set bit below l
            tgt_r = fail ? r + dr : tgt_r; // one past the nearest set bit at/after r
            // ---- new state
            l = con ? cl : (ok ? l2 : l);
            r = con ? cr : (ok ? r2 : r);
            d = con ? lv : (ok ? d_ext : d);
            flags = (con && cstop) ? (flags & ~F_CON) : (fail ? (flags | F_CON) : flags);
            if (CALL && accept) { // (a two-base step arrives here with no breakpoint waiting and d rising: nothing to do)
                const uint32_t k1 = k;
                // breakpoint at a base this item owns (not the first base of a sequence: the reference's loop starts at 1)
                if (i >= warm && i > 0 && i < len - tail && d < dprev && dprev >= a.call_thr && d < a.call_thr) {
                    // drop the ones whose window has passed, then queue this one
                    while (np && pend0 + k1 < i) { pend0 = pend1; pend1 = pend2; pend2 = pend3; np--; }
                    if (np == 4u) atomicAdd(a.call_counts + 16u * kCallSegs, 1u);
                    else {
                        if (np == 0) pend0 = i; else if (np == 1) pend1 = i; else if (np == 2) pend2 = i; else pend3 = i;
                        np++;
                    }
                } else if (np && d >= a.call_thr && r == l + 1u) {
                    // the first unique match to the right of every waiting breakpoint: a site for those within k of it
                    const uint32_t seg = wave % kCallSegs;
                    for (uint32_t x = 0; x < np; x++) {
                        const uint32_t bp = x == 0 ? pend0 : x == 1 ? pend1 : x == 2 ? pend2 : pend3;
                        if (i <= bp + k1) {
                            const uint32_t slot = atomicAdd(a.call_counts + seg * 16u, 1u);
                            if (slot < a.call_cap) a.call_sites[(size_t)seg * a.call_cap + slot] = make_uint4(start + bp, start + i, l, 0u);
                        }
                    }
                    np = 0;
                }
                dprev = d;
            }
            // call mode: an item whose breakpoints are all resolved need not walk the bases it borrowed from the next chunk
            const uint32_t end_at = (CALL && np == 0u) ? len - tail : len;
            if (accept) {
                if (i >= warm) { // emit: output byte e = i - warm of this item
                    if (IVAL) {
                        a.lo_out[start + i] = l;
                        a.hi_out[start + i] = r;
                    }
                    emit_ms(a.d_out, start, warm, i, i + 1 >= end_at && !(PAIR && pair_try), pair_try ? d_one : d, ocur, oblk);
                }
                if (PAIR && pair_try) { // second base of the pair (same query word, never the item's first)
                    i++;
                    if (i >= warm) emit_ms(a.d_out, start, warm, i, i + 1 >= end_at, d, ocur, oblk);
                }
                i++;
                const bool fin = i >= end_at;
                const bool newblk = (i & 15u) == 0;
                qblk.x = newblk ? qnxt.x : qblk.x;
                qblk.y = newblk ? qnxt.y : qblk.y;
                qblk.z = newblk ? qnxt.z : qblk.z;
                qblk.w = newblk ? qnxt.w : qblk.w;
                flags |= fin ? F_DONE : (newblk ? F_QF : 0u);
                const uint32_t w = (i >> 2) & 3u;
                const uint32_t lo = (w & 1u) ? qblk.y : qblk.x, hi = (w & 1u) ? qblk.w : qblk.z;
                qcur = (w & 2u) ? hi : lo;
                const uint32_t c = decode_base((qcur >> ((i & 3u) * 8u)) & 0xFFu);
                cb = c < 4u ? c * nblk : null_blk;
                if (PAIR) { // the base after it, when it sits in the same query word
                    const uint32_t c2 = ((i & 3u) != 3u && i + 1u < len)
                                            ? decode_base((qcur >> ((i & 3u) * 8u + 8u)) & 0xFFu) : 4u;
                    pcb = (c < 4u && c2 < 4u) ? pair_blk0 + (c * 4u + c2) * nblk : 0u;
                }
            }
        }
        } // hot loop
    }
#ifdef KBO_WALK_DEBUG
    if (lane == 0 && a.lo_out == nullptr && a.hi_out != nullptr) { // debug: hi_out doubles as counter sink
        atomicAdd(a.hi_out + 0, dbg_iter);
        atomicAdd(a.hi_out + 1, dbg_rare);
        atomicAdd(a.hi_out + 2, dbg_con);
        atomicAdd(a.hi_out + 3, 1u);
    }
    if (a.lo_out == nullptr && a.hi_out != nullptr) { // per-lane totals
        atomicAdd(a.hi_out + 4, dbg_ext);
        atomicAdd(a.hi_out + 5, dbg_fail);
        atomicAdd(a.hi_out + 6, dbg_con);
        atomicAdd(a.hi_out + 7, dbg_wait);
        atomicAdd(a.hi_out + 8, dbg_fin);
    }
#endif
    (void)dbg_wait; (void)dbg_fin; (void)dbg_rare; (void)dbg_con; (void)dbg_iter; (void)lane; (void)dbg_ext; (void)dbg_fail;
}

__global__ void make_items_kernel(const uint64_t *__restrict__ off, uint32_t n_seqs,
                                  WalkItem *__restrict__ items)
{
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    uint64_t b = off[s], e = off[s + 1];
    WalkItem it;
    it.start = b;
    it.len = (uint32_t)(e - b);
    it.warm = 0;
    items[s] = it;
}

// ---- items for batches with long sequences, built on the device -------------------------
// Sequence s is cut into ceil(len / chunk) items; every item after the first re-walks k-1
// warm-up bases (the MS of a base depends only on the k bases ending at it).  counts ->
// exclusive prefix sums (two-level scan, 1024 values per block) -> one lane per item slot,
// which finds its sequence by binary search; slots beyond the last item become empty items.

__global__ void chunk_count_kernel(const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t chunk,
                                   uint32_t *__restrict__ counts)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seqs) return;
    counts[s] = s < n_seqs ? (uint32_t)((off[s + 1] - off[s] + chunk - 1) / chunk) : 0u;
}

__global__ void make_chunk_items_kernel(const uint64_t *__restrict__ off, const uint32_t *__restrict__ local,
                                        const uint32_t *__restrict__ sums, uint32_t n_seqs, uint32_t chunk,
                                        uint32_t k, uint32_t n_slots, WalkItem *__restrict__ items, uint32_t call)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_slots) return;
    auto first_item = [&](uint32_t s) { return sums[s / kScanBlock] + local[s]; }; // s in [0, n_seqs]
    WalkItem it;
    it.start = 0;
    it.len = 0;
    it.warm = 0;
    if (t < first_item(n_seqs)) {
        uint32_t lo = 0, hi = n_seqs; // largest s with first_item(s) <= t (empty sequences own no item)
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (first_item(mid) <= t) lo = mid;
            else hi = mid;
        }
        const uint64_t b = off[lo], len = off[lo + 1] - b;
        const uint64_t c0 = (uint64_t)(t - first_item(lo)) * chunk;
        // (call mode: k warm-up bases and up to k bases of the next chunk, see WalkItem)
        const uint64_t warm = min(c0, (uint64_t)(call ? k : (k > 0 ? k - 1 : 0)));
        const uint64_t body = min((uint64_t)chunk, len - c0);
        const uint64_t tail = call ? min((uint64_t)k, len - c0 - body) : 0;
        it.start = b + c0 - warm;
        it.len = (uint32_t)(body + warm + tail);
        it.warm = (uint32_t)warm | ((uint32_t)tail << 16);
    }
    items[t] = it;
}

} // namespace

hipError_t launch_make_items(const uint64_t *d_offsets, uint32_t n_seqs, WalkItem *d_items,
                             hipStream_t stream)
{
    if (n_seqs == 0) return hipSuccess;
    hipLaunchKernelGGL(make_items_kernel, dim3((n_seqs + 255) / 256), dim3(256), 0, stream, d_offsets,
                       n_seqs, d_items);
    return hipGetLastError();
}

size_t chunk_items_scratch_words(uint32_t n_seqs) { return (size_t)n_seqs + 1 + ((size_t)n_seqs + 1 + kScanBlock - 1) / kScanBlock; }

hipError_t launch_make_chunk_items(const uint64_t *d_offsets, uint32_t n_seqs, uint32_t chunk, uint32_t k,
                                   uint32_t n_slots, WalkItem *d_items, uint32_t *d_scratch, hipStream_t stream, bool call)
{
    if (n_seqs == 0 || n_slots == 0) return hipSuccess;
    // (call mode: the chunks of a sequence must start a multiple of four bases apart - host_batch.cpp walk_chunk rounds to 64 -: with
    // chunks of 457 bases the call mode of the plan-guided walk gave different sites from run to run, tools/dbg_call_chunk.py; the
    // cause inside plan_kernel's call mode is not found, so the precondition is checked where the items are made)
    if (call && (chunk & 3u)) return hipErrorInvalidValue;
    const uint32_t n = n_seqs + 1;
    uint32_t *local = d_scratch, *sums = d_scratch + n;
    hipLaunchKernelGGL(chunk_count_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_offsets, n_seqs, chunk, local);
    const hipError_t es = launch_scan(local, n, sums, stream);
    if (es != hipSuccess) return es;
    hipLaunchKernelGGL(make_chunk_items_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, stream, d_offsets, local, sums,
                       n_seqs, chunk, k, n_slots, d_items, call ? 1u : 0u);
    return hipGetLastError();
}

// tuning state (kbo_set_* entry points; may change while other threads launch: atomics, read once per launch)
std::atomic<int> g_walk_threads{kWalkThreads};
std::atomic<int> g_rare_period{8}; // tuned on C2 (tools/sweep_walk.py RARE=1)
std::atomic<int> g_walk_lane_limit{64}, g_walk_dummy_lds{0}; // experiments (kbo_set_walk_experiment): see DESIGN.md section 6
void set_walk_experiment(int lane_limit, int dummy_lds_bytes)
{
    g_walk_lane_limit = std::max(1, std::min(64, lane_limit));
    g_walk_dummy_lds = std::max(0, std::min(64 << 10, dummy_lds_bytes));
}
// guided walk: resident waves per CU, in 32nds of the plain walk's (so that kbo_set_walk_waves_per_cu scales both); 0 = by
// the kernel: 8 (rank blocks + entries) or 12 (recovery lines).  Few: a lane comes back to the line of its last iteration
// (failed extension -> contraction -> retry) and finds it in L2 only while the lines of all lanes in flight fit there -
// 8 waves x 64 lanes x 32 CUs x 128 B = 2 of an XCD's 4 MiB - and even lanes that never come back run into each other's
// fills beyond that
std::atomic<int> g_guided_waves_32nds{0};
// guided walk over the recovery lines (plan_kernels.hip): -1 = where the rank blocks are far beyond L2 (the two-base
// steps' threshold), 0 / 1 = never / always
std::atomic<int> g_guided_fat{-1};
void set_guided_walk(int waves_per_cu, int recovery_lines)
{
    if (waves_per_cu >= 0) g_guided_waves_32nds = std::min(32, waves_per_cu);
    if (recovery_lines >= -1) g_guided_fat = recovery_lines < 0 ? -1 : (recovery_lines != 0 ? 1 : 0);
}
bool guided_uses_recovery_lines(const WalkArgs &a)
{
    static const int env_fat = std::getenv("KBO_PLAN_FAT") ? std::atoi(std::getenv("KBO_PLAN_FAT")) : -1; // experiments
    const int f = env_fat >= 0 ? env_fat : g_guided_fat.load();
    return a.ix.fat != nullptr && (f < 0 ? a.ix.n >= (24u << 20) : f != 0);
}
std::atomic<int> g_pair_min_depth{16};                 // two-base steps only from matches at least this deep
void set_pair_min_depth(int d) { g_pair_min_depth = d < 0 ? 0 : d; }
void set_walk_rare(int period) { g_rare_period = std::max(1, std::min(1024, period)); }
void set_walk_threads(int t) { g_walk_threads = (t == 64 || t == 128 || t == 256) ? t : kWalkThreads; }

// the redo pass: items a unit flagged (its successor's start state was a wrong guess) - or, in table mode, items the table
// could not resolve - walked plainly, in full; redo_collect_kernel has listed them where the units were, one item per lane
static hipError_t launch_redo_walk(WalkArgs a, hipStream_t stream)
{
    const uint32_t n_orig = a.n_items;
    a.items = reinterpret_cast<const WalkItem *>(a.units); // (the unit array is free by now: see redo_collect_kernel)
    a.n_items = a.redo_cap;
    a.n_items_dev = a.qctl + 1;
    a.gitems = nullptr;
    a.rounds = 1;
    a.rare_period = (uint32_t)g_rare_period.load();
    a.lane_limit = (uint32_t)g_walk_lane_limit.load();
    a.pair_min_d = (uint32_t)g_pair_min_depth.load();
#ifdef KBO_WALK_DEBUG
    a.hi_out = nullptr; // (counters build: the counter sink belongs to the guided kernel)
#endif
    // (one item per lane, so that a few hundred flagged items spread over waves; nearly all of these waves find no
    // item and leave at once - workgroups of four waves: a quarter of the dispatches)
    const uint32_t rwaves = (n_orig + 63u) / 64u;
    const dim3 rgrid((rwaves + 3u) / 4u), rblock(256);
    if (a.call_sites) { // (call mode: the flagged items' scan with their walk)
        if (a.ix.big) hipLaunchKernelGGL((ms_walk_kernel<false, true, false, true>), rgrid, rblock, 0, stream, a);
        else hipLaunchKernelGGL((ms_walk_kernel<false, false, false, true>), rgrid, rblock, 0, stream, a);
    } else if (a.ix.big) hipLaunchKernelGGL((ms_walk_kernel<false, true, false>), rgrid, rblock, 0, stream, a);
    else hipLaunchKernelGGL((ms_walk_kernel<false, false, false>), rgrid, rblock, 0, stream, a);
    return hipGetLastError();
}

// the plain walk over a list whose length is on the device (long_kernels.hip: the sub-items of the flagged pieces)
hipError_t launch_walk_list(WalkArgs a, const WalkItem *d_list, uint32_t cap, const uint32_t *d_count, uint32_t lanes, hipStream_t stream)
{
    if (cap == 0 || lanes == 0) return hipSuccess;
    a.items = d_list;
    a.n_items = cap;
    a.n_items_dev = d_count;
    a.gitems = nullptr;
    a.call_sites = nullptr;
    a.lo_out = a.hi_out = nullptr;
    a.rounds = 1;
    a.rare_period = (uint32_t)g_rare_period.load();
    a.lane_limit = (uint32_t)g_walk_lane_limit.load();
    a.pair_min_d = (uint32_t)g_pair_min_depth.load();
    const uint32_t waves = (lanes + 63u) / 64u;
    const dim3 grid((waves + 3u) / 4u), block(256);
    if (a.ix.big) hipLaunchKernelGGL((ms_walk_kernel<false, true, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((ms_walk_kernel<false, false, false>), grid, block, 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_redo_collect(const WalkArgs &a, hipStream_t stream); // plan_kernels.hip
hipError_t launch_redo_pass(WalkArgs a, hipStream_t stream)
{
    const hipError_t e = launch_redo_collect(a, stream);
    if (e != hipSuccess) return e;
    return launch_redo_walk(a, stream);
}

hipError_t launch_ms_walk(WalkArgs a, int max_waves, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    // every lane walks `rounds` items; the grid is sized so that lanes * rounds covers the
    // items with as little slack as possible (one 64-lane wave per workgroup)
    const uint64_t lanes = (uint64_t)std::max(1, max_waves) * 64u;
    a.rounds = (uint32_t)((a.n_items + lanes - 1) / lanes);
    a.rare_period = (uint32_t)g_rare_period.load();
    a.lane_limit = (uint32_t)g_walk_lane_limit.load();
    const uint32_t lds = (uint32_t)g_walk_dummy_lds.load(); // (experiment: LDS the kernel does not touch, to cap the occupancy)
    const uint64_t per_wave = 64ull * a.rounds;
    const uint32_t waves = (uint32_t)((a.n_items + per_wave - 1) / per_wave);
    const uint32_t threads = (uint32_t)g_walk_threads.load();
    const uint32_t wpb = threads / 64;
    const dim3 grid((waves + wpb - 1) / wpb), block(threads);
    const bool ival = a.lo_out && a.hi_out;
    a.pair_min_d = (uint32_t)g_pair_min_depth.load();
    if (a.gitems && a.glist && a.ix.pc_text && !ival) { // MS values only, index with a path cover: plan, then guided walk
        hipError_t e = hipSuccess;
        // the stretches behind mismatches from the depth table (dtab_kernels.hip): no units.  On large indexes only for reads,
        // whose look-ups plan_kernel does itself; items that cannot be staged there keep units + the guided walk over recovery
        // lines (the stand-alone resolve kernel is no faster than that walk: 4.3 against 3.9 ms per 5 M reads at C3)
        const bool reads = a.max_item_len != 0 && a.max_item_len <= 160u;
        if (a.ix.dtab && !a.call_sites && (reads || a.ix.n < (24u << 20))) {
            e = launch_plan_table(a, stream);
            if (e != hipSuccess) return e;
        } else {
            e = launch_plan(a, stream);
            if (e != hipSuccess) return e;
            // the guided kernel takes its units off a queue: a fixed number of resident waves, but no more than there are
            // chunks of 64 units
            static const int env_gw = std::getenv("KBO_GUIDED_WAVES") ? std::atoi(std::getenv("KBO_GUIDED_WAVES")) : 0; // experiments: 32nds
            const int gw_set = env_gw > 0 ? env_gw : g_guided_waves_32nds.load();
            const int gw = gw_set > 0 ? gw_set : (guided_uses_recovery_lines(a) ? 12 : 8);
            const uint32_t gwaves = (uint32_t)std::min<uint64_t>((uint64_t)std::max(1, max_waves * gw / 32),
                                                                 ((uint64_t)a.unit_cap + 63) / 64);
            e = launch_ms_walk_guided(a, (gwaves + wpb - 1) / wpb, threads, stream);
            if (e != hipSuccess) return e;
        }
        return launch_redo_walk(a, stream);
    }
    if (a.call_sites) { // call mode: MS values + the breakpoint scan, no intervals written
        if (a.ix.big) hipLaunchKernelGGL((ms_walk_kernel<false, true, false, true>), grid, block, lds, stream, a);
        else if (a.ix.pair_off) hipLaunchKernelGGL((ms_walk_kernel<false, false, true, true>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((ms_walk_kernel<false, false, false, true>), grid, block, lds, stream, a);
        return hipGetLastError();
    }
    if (a.ix.big) {
        if (ival) hipLaunchKernelGGL((ms_walk_kernel<true, true, false>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((ms_walk_kernel<false, true, false>), grid, block, lds, stream, a);
    } else {
        if (ival) hipLaunchKernelGGL((ms_walk_kernel<true, false, false>), grid, block, lds, stream, a);
        else if (a.ix.pair_off) hipLaunchKernelGGL((ms_walk_kernel<false, false, true>), grid, block, lds, stream, a);
        else hipLaunchKernelGGL((ms_walk_kernel<false, false, false>), grid, block, lds, stream, a);
    }
    return hipGetLastError();
}

} // namespace kbo
