// ms_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the k-bounded matching-statistics path.
//
//   ms_walk_kernel           A1  sbwt::StreamingIndex::matching_statistics
//                                (called at reference index.rs:251-252)
//   derand_translate_kernel  A5+A6  derandomize_ms_vec (derandomize.rs:269-288) fused with
//                                translate_ms_vec (translate.rs:263-293) and, optionally,
//                                format::relative_to_ref (format.rs:266-287)
//   translate_kernel         A6 alone (stencil form)
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
#include "kernels.hpp"

#include <algorithm>

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

// Rank inside one 16-byte block { base, w0, w1, w2 }: base + popcount of the o lowest of
// the 96 row bits, 0 <= o < 96.  One 64-bit shift builds the "bits to drop" masks of all
// three words: X = ~0 << (o mod 64) is the drop mask of (w0,w1) when o < 64 and of w2
// when o >= 64.
__device__ __forceinline__ uint32_t rank_eval(const uint4 &b, uint32_t o)
{
    const uint64_t X = ~0ull << (o & 63u);
    const uint32_t xl = (uint32_t)X, xh = (uint32_t)(X >> 32);
    const bool big = o >= 64u;
    const uint32_t d0 = big ? 0u : xl, d1 = big ? 0u : xh, d2 = big ? xl : ~0u;
    return b.x + __popc(b.y & ~d0) + __popc(b.z & ~d1) + __popc(b.w & ~d2);
}

__device__ __forceinline__ uint32_t div96(uint32_t i) { return __umulhi(i, 0xAAAAAAABu) >> 6; }

// 'A','C','G','T' -> 0..3, anything else -> 4 (sbwt's DNA alphabet is exactly ACGT)
__device__ __forceinline__ uint32_t decode_base(uint32_t ch)
{
    uint32_t c = ((ch >> 1) & 3u) ^ ((ch >> 2) & 1u);
    uint32_t back = (0x54474341u >> (8 * c)) & 0xFFu;
    return back == ch ? c : 4u;
}

__device__ __forceinline__ uint4 ld16(const uint8_t *base, uint32_t byte_off)
{
    return *reinterpret_cast<const uint4 *>(base + byte_off);
}
// unaligned 16-byte load (gfx950 global loads accept any byte address)
__device__ __forceinline__ uint4 ld16u(const uint8_t *base, uint32_t byte_off)
{
    uint4 v;
    __builtin_memcpy(&v, base + byte_off, 16);
    return v;
}
// unaligned stores
#ifndef KBO_NT_STORE
#define KBO_NT_STORE 1 // streaming stores: -4.5 % walk time on C2 (outputs are never re-read here)
#endif
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ void st16u(uint8_t *base, uint32_t byte_off, const uint4 &v)
{
#if KBO_NT_STORE
    u32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t *>(base + byte_off));
#else
    __builtin_memcpy(base + byte_off, &v, 16);
#endif
}
__device__ __forceinline__ void st4u(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// store the first nb (1..15) bytes of a 16-byte block: whole words, then the trailing bytes
__device__ __forceinline__ void st_partial(uint8_t *o, const uint4 &v, uint32_t nb)
{
#define KBO_ST_WORD(J, W)                                                                         \
    if (nb >= 4u * (J) + 4u) st4u(o + 4u * (J), (W));                                             \
    else {                                                                                        \
        if (nb > 4u * (J) + 0u) o[4u * (J) + 0u] = (uint8_t)((W));                                 \
        if (nb > 4u * (J) + 1u) o[4u * (J) + 1u] = (uint8_t)((W) >> 8);                            \
        if (nb > 4u * (J) + 2u) o[4u * (J) + 2u] = (uint8_t)((W) >> 16);                           \
    }
    KBO_ST_WORD(0u, v.x)
    KBO_ST_WORD(1u, v.y)
    KBO_ST_WORD(2u, v.z)
    KBO_ST_WORD(3u, v.w)
#undef KBO_ST_WORD
}

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
// The kernel is instruction-issue bound on L2-resident indexes and line-fill bound beyond
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
template <bool IVAL, bool BIG, bool PAIR>
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
    const uint64_t first64 = (uint64_t)wave * 64u * a.rounds + lane;
    uint32_t next_item = first64 < a.n_items ? (uint32_t)first64 : a.n_items;
    uint32_t left = 0; // items whose descriptor has not been requested yet
    if (next_item < a.n_items) left = min(a.rounds, (a.n_items - 1u - next_item) / 64u + 1u);

    uint32_t flags = left ? F_DONE : F_FIN;
    uint32_t l = 0, r = n, d = 0, m = 0, cb = 0; // m: contraction targets known (rare block only)
    uint32_t pcb = 0; // PAIR: first two-base block of (current base, next base), 0 = no pair step here
    uint32_t pmin = 0; // PAIR: depth from which pair steps are tried: 0 until the item's first failure (a read
                       // is expected to match from its first base), pair_min_d afterwards (random matches)
    uint32_t tgt_l = 0, tgt_r = 0; // contraction targets (rare block only)
    // Query and output are streamed in 16-byte blocks RELATIVE TO THE ITEM (unaligned global
    // accesses): i = base index inside the item; block i>>4, word (i>>2)&3, byte i&3.
    uint32_t i = 0, len = 0, warm = 0, start = 0;
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

    uint32_t dbg_rare = 0, dbg_con = 0, dbg_iter = 0, dbg_ext = 0, dbg_fail = 0;
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
                    warm = nit.w;
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
            const bool pair_try = PAIR && !con && pcb != 0u && !(flags & F_NOPAIR) && d >= pmin;
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
            if (accept) {
                if (i >= warm) { // emit: output byte e = i - warm of this item
                    if (IVAL) {
                        a.lo_out[start + i] = l;
                        a.hi_out[start + i] = r;
                    }
                    emit_ms(a.d_out, start, warm, i, i + 1 == len, pair_try ? d_one : d, ocur, oblk);
                }
                if (PAIR && pair_try) { // second base of the pair (same query word, never the item's first)
                    i++;
                    if (i >= warm) emit_ms(a.d_out, start, warm, i, i + 1 == len, d, ocur, oblk);
                }
                i++;
                const bool fin = i == len;
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
    }
#endif
    (void)dbg_rare; (void)dbg_con; (void)dbg_iter; (void)lane; (void)dbg_ext; (void)dbg_fail;
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
constexpr uint32_t kScanBlock = 1024;

__global__ void chunk_count_kernel(const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t chunk,
                                   uint32_t *__restrict__ counts)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seqs) return;
    counts[s] = s < n_seqs ? (uint32_t)((off[s + 1] - off[s] + chunk - 1) / chunk) : 0u;
}

// exclusive scan of `per` consecutive values per thread, 256 or 1024 threads per block, in place;
// sums[blockIdx.x] = total of the block's values (when sums != nullptr)
__global__ void scan_kernel(uint32_t *__restrict__ data, uint32_t n, uint32_t per, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t sh[1024];
    const uint32_t t = threadIdx.x, nt = blockDim.x;
    const uint64_t base = ((uint64_t)blockIdx.x * nt + t) * per;
    uint32_t local = 0;
    for (uint32_t j = 0; j < per; j++)
        if (base + j < n) local += data[base + j];
    sh[t] = local;
    __syncthreads();
    for (uint32_t step = 1; step < nt; step <<= 1) { // Hillis-Steele inclusive scan of the thread sums
        const uint32_t v = t >= step ? sh[t - step] : 0u;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    uint32_t run = sh[t] - local; // exclusive prefix of this thread inside the block
    for (uint32_t j = 0; j < per; j++)
        if (base + j < n) {
            const uint32_t v = data[base + j];
            data[base + j] = run;
            run += v;
        }
    if (sums && t == nt - 1) sums[blockIdx.x] = sh[t];
}

__global__ void make_chunk_items_kernel(const uint64_t *__restrict__ off, const uint32_t *__restrict__ local,
                                        const uint32_t *__restrict__ sums, uint32_t n_seqs, uint32_t chunk,
                                        uint32_t k, uint32_t n_slots, WalkItem *__restrict__ items)
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
        const uint64_t warm = min(c0, (uint64_t)(k > 0 ? k - 1 : 0));
        it.start = b + c0 - warm;
        it.len = (uint32_t)(min((uint64_t)chunk, len - c0) + warm);
        it.warm = (uint32_t)warm;
    }
    items[t] = it;
}

// -------------------------------------------------------------------------------------
// A5 + A6.  derandomize_ms_vec is a right-to-left recurrence (derandomize.rs:282-285):
//     out[len-1] = noisy[len-1] > t ? noisy[len-1] : 0
//     out[i]     = noisy[i]==k ? k : (noisy[i] > t && out[i+1] < noisy[i]) ? noisy[i] : out[i+1]-1
// translate_ms_vec's sequential 'R','R' look-ahead (translate.rs:275-290) has the closed
// form (checked against the literal oracle by the tests):
//     condR(p)  = x[p] > t && 0 < x[p+1] < t
//     res[pos]  = 'R'                                  if 2 <= pos < len-1 && condR(pos-1)
//               = translate_ms_val(x[pos], next, prev).0   otherwise
//     next = pos < len-1 ? x[pos+1] : x[pos];   prev = pos > 1 ? x[pos-1] : k   (note pos > 1)
// so one right-to-left pass with a three-value window produces the characters.
__device__ __forceinline__ uint32_t translate_char(int xm, int xc, int xn, uint32_t rel, uint32_t len, int K, int T)
{
    // branch-free; 0 < v < T is written (unsigned)(v - 1) < (unsigned)(T - 1), 2 <= rel <= len-2 as
    // (rel - 2) < (len - 3) (len >= 3)
    const uint32_t Tm1 = (uint32_t)(T - 1);
    const int prev = rel > 1u ? xm : K;
    const int next = rel < len - 1u ? xn : xc;
    const bool inherits = (rel - 2u) < (len - 3u) && xm > T && (uint32_t)(xc - 1) < Tm1;
    const bool own = xc > T && (uint32_t)(next - 1) < Tm1;
    const uint32_t plain = xc <= 0 ? ((next == 1 && prev > 0) ? (uint32_t)'X' : (uint32_t)'-') : (uint32_t)'M';
    return (inherits || own) ? (uint32_t)'R' : plain;
}

// byte J (compile-time 0..15) of a 16-byte block held in registers
template <int J> __device__ __forceinline__ uint32_t blk_byte(const uint4 &v)
{
    const uint32_t w = (J >> 2) == 0 ? v.x : (J >> 2) == 1 ? v.y : (J >> 2) == 2 ? v.z : v.w;
    return (w >> ((J & 3) * 8)) & 0xFFu;
}
template <int J> __device__ __forceinline__ void blk_or_byte(uint4 &v, uint32_t x)
{
    const uint32_t sh = x << ((J & 3) * 8);
    if ((J >> 2) == 0) v.x |= sh;
    else if ((J >> 2) == 1) v.y |= sh;
    else if ((J >> 2) == 2) v.z |= sh;
    else v.w |= sh;
}

struct DtState {
    int x_cur, x_next, x_prev;
};

// one position of the right-to-left pass; J = byte inside the current 16-byte block
template <int J>
__device__ __forceinline__ void dt_step(DtState &st, const uint4 &cur, const uint4 &below, const uint4 &rcur,
                                        uint4 &oblk, uint32_t p, uint32_t len, int K, int T, bool fmt,
                                        int32_t *derand_out_p)
{
    if (p >= len) return; // only in the topmost block
    if (p == len - 1) {   // derandomize.rs:282
        const int a = (int)blk_byte<J>(cur);
        st.x_cur = a > T ? a : 0;
        st.x_next = st.x_cur;
    }
    if (p > 0) { // x[p-1] from noisy[p-1] and x[p] (derandomize.rs:233-246)
        const int a = (int)(J > 0 ? blk_byte<(J + 15) & 15>(cur) : blk_byte<15>(below));
        st.x_prev = (a == K) ? K : ((a > T && st.x_cur < a) ? a : st.x_cur - 1);
    }
    uint32_t ch = translate_char(st.x_prev, st.x_cur, st.x_next, p, len, K, T);
    if (fmt) // format::relative_to_ref: M,R keep the reference base, X and '-' become '-'
        ch = (ch == 'M' || ch == 'R') ? blk_byte<J>(rcur) : (uint32_t)'-';
    blk_or_byte<J>(oblk, ch);
    if (derand_out_p) *derand_out_p = st.x_cur;
    st.x_next = st.x_cur;
    st.x_cur = st.x_prev;
}

// the same for a position with 2 <= p <= len-2 whose block lies wholly inside the sequence and is
// not its first: no position tests at all (most blocks of a long sequence)
template <int J>
__device__ __forceinline__ void dt_step_mid(DtState &st, const uint4 &cur, const uint4 &below, const uint4 &rcur,
                                            uint4 &oblk, int K, int T, bool fmt)
{
    const int a = (int)(J > 0 ? blk_byte<(J + 15) & 15>(cur) : blk_byte<15>(below));
    st.x_prev = (a == K) ? K : ((a > T && st.x_cur < a) ? a : st.x_cur - 1);
    const uint32_t Tm1 = (uint32_t)(T - 1);
    const bool is_r = (st.x_prev > T && (uint32_t)(st.x_cur - 1) < Tm1) || (st.x_cur > T && (uint32_t)(st.x_next - 1) < Tm1);
    const uint32_t plain = st.x_cur <= 0 ? ((st.x_next == 1 && st.x_prev > 0) ? (uint32_t)'X' : (uint32_t)'-') : (uint32_t)'M';
    uint32_t ch = is_r ? (uint32_t)'R' : plain;
    if (fmt) ch = (ch == 'M' || ch == 'R') ? blk_byte<J>(rcur) : (uint32_t)'-';
    blk_or_byte<J>(oblk, ch);
    st.x_next = st.x_cur;
    st.x_cur = st.x_prev;
}

#define KBO_DT_BLOCK_GUARDED(DOUT)                                                                                    \
    {                                                                                                                \
        KBO_DT(15, DOUT) KBO_DT(14, DOUT) KBO_DT(13, DOUT) KBO_DT(12, DOUT) KBO_DT(11, DOUT) KBO_DT(10, DOUT)        \
        KBO_DT(9, DOUT) KBO_DT(8, DOUT) KBO_DT(7, DOUT) KBO_DT(6, DOUT) KBO_DT(5, DOUT) KBO_DT(4, DOUT)              \
        KBO_DT(3, DOUT) KBO_DT(2, DOUT) KBO_DT(1, DOUT) KBO_DT(0, DOUT)                                              \
    }
#define KBO_DT(J, DOUT) dt_step<J>(st, cur, below, rcur, oblk, p0 + J, len, K, T, fmt, (DOUT) ? (DOUT) + p0 + J : nullptr);
#define KBO_DT_BLOCK_MID                                                                                             \
    {                                                                                                                \
        KBO_DM(15) KBO_DM(14) KBO_DM(13) KBO_DM(12) KBO_DM(11) KBO_DM(10) KBO_DM(9) KBO_DM(8)                        \
        KBO_DM(7) KBO_DM(6) KBO_DM(5) KBO_DM(4) KBO_DM(3) KBO_DM(2) KBO_DM(1) KBO_DM(0)                              \
    }
#define KBO_DM(J) dt_step_mid<J>(st, cur, below, rcur, oblk, K, T, fmt);

// One lane per sequence, right to left, one 16-byte block (relative to the sequence start,
// unaligned global accesses) at a time with the block below it already in flight; inside a
// block the 16 positions are unrolled so every byte access is a constant bit-field.
__global__ __launch_bounds__(256) void derand_translate_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t k,
    uint32_t t, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out, int32_t *__restrict__ derand_out,
    uint32_t max_len, const uint32_t *__restrict__ only)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    if (only && !only[s]) return; // second pass of the piece-wise path: flagged sequences only
    const uint64_t b = off[s], e = off[s + 1];
    const uint32_t len = (uint32_t)(e - b);
    if (len < 3) return;       // the host side rejects these (derandomize.rs:276)
    if (len > max_len) return; // longer sequences take the chunked path (launch_derand_long)
    const int K = (int)k, T = (int)t;
    const uint8_t *msb = ms + b;
    const bool fmt = ref != nullptr;
    const uint8_t *refb = fmt ? ref + b : msb;
    uint8_t *outb = out + b;
    int32_t *dout = derand_out ? derand_out + b : nullptr;

    const uint32_t nblk = (len + 15u) >> 4;
    uint4 cur = ld16u(msb, 16u * (nblk - 1u));
    uint4 rcur = fmt ? ld16u(refb, 16u * (nblk - 1u)) : make_uint4(0, 0, 0, 0);
    DtState st{0, 0, K};
    for (uint32_t bi = nblk; bi-- > 0;) {
        uint4 below = cur, rbelow = rcur;
        if (bi > 0) {
            below = ld16u(msb, 16u * (bi - 1u));
            if (fmt) rbelow = ld16u(refb, 16u * (bi - 1u));
        }
        uint4 oblk = make_uint4(0, 0, 0, 0);
        const uint32_t p0 = 16u * bi;
        if (bi >= 1u && p0 + 17u <= len && !dout) KBO_DT_BLOCK_MID
        else KBO_DT_BLOCK_GUARDED(dout)
        if (p0 + 16u <= len) st16u(outb, p0, oblk);
        else st_partial(outb + p0, oblk, len - p0); // topmost, partial block of the sequence
        cur = below;
        rcur = rbelow;
    }
}

// ---- LDS-staged variant for batches of short sequences (reads) -------------------------
// One wave per workgroup handles 64 consecutive sequences, whose bytes are contiguous in the
// concatenated buffers: the wave copies that span HBM -> LDS with coalesced 16-byte accesses,
// every lane runs the right-to-left pass over its own sequence inside LDS (bytes in place:
// MS value in, character out), and the wave copies the span back out, applying
// format::relative_to_ref on the way when a reference is given.  Global traffic is fully
// coalesced (the per-lane kernel above issues one 16-byte request per lane instead).
__device__ __forceinline__ uint32_t fmt_word(uint32_t ch, uint32_t rf)
{ // per byte: ch in {'M','R'} ? rf : '-'
    const uint32_t xm = ch ^ 0x4D4D4D4Du, xr = ch ^ 0x52525252u; // zero byte where equal
    const uint32_t zm = ~(((xm & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xm | 0x7F7F7F7Fu);
    const uint32_t zr = ~(((xr & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xr | 0x7F7F7F7Fu);
    const uint32_t hi = zm | zr;                 // 0x80 in matching bytes
    const uint32_t mask = (hi >> 7) * 0xFFu;     // 0xFF in matching bytes
    return (rf & mask) | (0x2D2D2D2Du & ~mask);
}

// SKEW: the LDS image gets 4 bytes of padding after every 128 bytes.  Lanes touch position p of their own
// sequence in the same step, so with sequences whose common length is a multiple of 32 bytes the flat image
// puts 8..64 lanes on one bank (reads of 128 or 256 bases: 3.2x slower); the padding spreads them.
template <bool SKEW>
__global__ __launch_bounds__(64) void derand_translate_lds_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t k,
    uint32_t t, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t s0 = blockIdx.x * 64u;
    const uint32_t s = s0 + lane;
    const uint32_t s_end = min(s0 + 64u, n_seqs);
    const uint64_t base = off[s0];
    const uint32_t span = (uint32_t)(off[s_end] - base);
    if (span > lds_bytes) return; // cannot happen: the host sizes lds_bytes from the longest sequence
    const int K = (int)k, T = (int)t;
    auto at = [&](uint32_t x) -> uint8_t & { return lds[SKEW ? x + ((x >> 7) << 2) : x]; };

    for (uint32_t o = lane * 16u; o < span; o += 1024u) { // stage in (reads <= 15 B past the span)
        const uint4 v = ld16u(ms + base, o);
        if (SKEW) { // a 16-byte chunk never straddles a 128-byte granule, but it is only 4-byte aligned
            uint32_t *d = reinterpret_cast<uint32_t *>(&at(o));
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
            *reinterpret_cast<uint4 *>(lds + o) = v;
        }
    }
    __syncthreads();

    if (s < n_seqs) {
        const uint32_t b = (uint32_t)(off[s] - base), len = (uint32_t)(off[s + 1] - off[s]);
        if (len >= 3) {
            // Branch-free pass with the window (x_prev, x_cur, x_next) = x[p-1], x[p], x[p+1]:
            //   in(v) = 0 < v < T, written (unsigned)(v - 1) < (unsigned)(T - 1);
            //   'R' at p  <=>  (x[p-1] > T && in(x[p]) && 2 <= p < len-1)  ||  (x[p] > T && in(next));
            // the two ends (p = len-1, where next = x[p]; p = 1 and 0, where prev = K and no 'R' is
            // inherited from below) are peeled so that the middle of the sequence carries no
            // position tests.  gt_* / in_* flags move down the window with the values.
            const uint32_t Tm1 = (uint32_t)(T - 1);
            auto step = [&](int a, int x_cur) { return (a == K) ? K : ((a > T && x_cur < a) ? a : x_cur - 1); };
            auto plain = [&](int x_cur, int next, int prev) -> uint32_t { // translate.rs:180-216 without the 'R' cases
                return x_cur <= 0 ? ((next == 1 && prev > 0) ? (uint32_t)'X' : (uint32_t)'-') : (uint32_t)'M';
            };
            int a = at(b + len - 1);
            int x_cur = a > T ? a : 0; // derandomize.rs:282
            int a_below = at(b + len - 2);
            int x_prev = step(a_below, x_cur);
            // p = len-1: next = x_cur itself; inherits 'R' from below (len-1 >= 2 always holds, but
            // the rule needs pos < len-1, so it does not apply here)
            bool in_cur = (uint32_t)(x_cur - 1) < Tm1, gt_cur = x_cur > T;
            at(b + len - 1) = (uint8_t)((gt_cur && in_cur) ? (uint32_t)'R' : plain(x_cur, x_cur, x_prev));
            int x_next = x_cur;
            bool in_next = in_cur;
            x_cur = x_prev;
            gt_cur = x_cur > T;
            in_cur = (uint32_t)(x_cur - 1) < Tm1;
            a_below = at(b + len - 3);
            for (uint32_t p = len - 2; p >= 2; p--) { // middle: 2 <= p <= len-2
                x_prev = step(a_below, x_cur);
                a_below = at(b + p - 2); // p >= 2
                const bool gt_prev = x_prev > T;
                const bool is_r = (gt_prev && in_cur) || (gt_cur && in_next);
                at(b + p) = (uint8_t)(is_r ? (uint32_t)'R' : plain(x_cur, x_next, x_prev));
                x_next = x_cur;
                in_next = in_cur;
                x_cur = x_prev;
                gt_cur = gt_prev;
                in_cur = (uint32_t)(x_cur - 1) < Tm1;
            }
            // p = 1: prev = K (translate.rs:277 tests pos > 1), no 'R' inherited (needs pos >= 2)
            x_prev = step(a_below, x_cur); // a_below == at(b + 0)
            at(b + 1) = (uint8_t)((gt_cur && in_next) ? (uint32_t)'R' : plain(x_cur, x_next, K));
            // p = 0
            x_next = x_cur;
            in_next = in_cur;
            x_cur = x_prev;
            at(b + 0) = (uint8_t)((x_cur > T && in_next) ? (uint32_t)'R' : plain(x_cur, x_next, K));
        }
    }
    __syncthreads();

    for (uint32_t o = lane * 16u; o < span; o += 1024u) { // stage out
        uint4 c;
        if (SKEW) {
            const uint32_t *d = reinterpret_cast<const uint32_t *>(&at(o));
            c = make_uint4(d[0], d[1], d[2], d[3]);
        } else {
            c = *reinterpret_cast<const uint4 *>(lds + o);
        }
        if (ref) {
            const uint4 rf = ld16u(ref + base, o);
            c.x = fmt_word(c.x, rf.x);
            c.y = fmt_word(c.y, rf.y);
            c.z = fmt_word(c.z, rf.z);
            c.w = fmt_word(c.w, rf.w);
        }
        if (o + 16u <= span) st16u(out + base, o, c);
        else st_partial(out + base + o, c, span - o);
    }
}

// ---- piece-wise, LDS-staged variant for batches of long reads / contigs -----------------
// One lane per piece of kDtPiece positions of a sequence, so that a few thousand sequences of
// 10 kbp still fill the device.  The recurrence runs right to left, so a piece needs x at its
// upper end: the lane looks for the nearest position at or above the piece's end whose value is
// known without context - a hard reset (noisy == k gives x = k whatever follows,
// derandomize.rs:235-238) or the sequence's last position (derandomize.rs:282) - and runs the
// recurrence from there down to the piece.  Exact whenever such a position lies within
// kDtLookahead positions (inside matches every position is a reset); otherwise the sequence is
// flagged and redone by one lane in a second launch (adversarial inputs: long stretches without a
// single full-length match).  One wave takes 64 consecutive pieces (a contiguous span of <= 16 KB
// of the concatenated buffers) and stages it through LDS like the read kernel above: coalesced
// 16-byte traffic (one 16-byte request per lane and block, measured first, is bound by the
// L2-miss request rate at 8x the bytes).  Phase 1: every lane reads what it
// needs from OUTSIDE its piece - the look-ahead up to the nearest hard reset and the value just
// below the piece - and derives its start state; barrier; phase 2: it overwrites its piece in
// place (MS value in, character out); barrier; the span is copied out, relative_to_ref applied
// on the way.  Pieces that give up (no reset in reach) leave their MS bytes in place and flag the
// sequence, which the per-lane kernel redoes afterwards.
// 132 = 33 words: consecutive lanes' pieces start one LDS bank apart (a 256-byte stride puts all 64
// lanes of a step on the same bank: measured 0.67 ms against 0.44 ms for 260 on 200 Mbp); 10 KB
// of LDS per wave keeps 4 waves per SIMD resident
constexpr uint32_t kDtPiece = 132, kDtLookahead = 1024;
constexpr uint32_t kDtBehind = 16; // staged bytes below the span (the value just below the first piece)

__global__ __launch_bounds__(64) void derand_translate_piece_lds_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint64_t total,
    const WalkItem *__restrict__ pieces, uint32_t n_pieces, uint32_t k, uint32_t t, const uint8_t *__restrict__ ref,
    uint8_t *__restrict__ out, uint32_t max_len, uint32_t *__restrict__ redo)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint32_t span_len_sh;
    const uint32_t lane = threadIdx.x;
    const uint32_t first = blockIdx.x * 64u;
    WalkItem it;
    it.start = 0;
    it.len = 0;
    it.warm = 0;
    if (first + lane < n_pieces) it = pieces[first + lane];
    const uint64_t span_base = pieces[first].start; // wave-uniform; slots behind the last piece are empty
    if (lane == 0) span_len_sh = 0;
    __syncthreads();
    if (it.len) atomicMax(&span_len_sh, (uint32_t)(it.start + it.len - span_base));
    __syncthreads();
    const uint32_t span_len = span_len_sh;
    if (span_len == 0) return;
    const uint64_t stage_lo = span_base >= kDtBehind ? span_base - kDtBehind : 0;
    const uint32_t head = (uint32_t)(span_base - stage_lo); // 0 or 16
    const uint32_t stage_len = (uint32_t)(min(span_base + span_len + kDtLookahead, total) - stage_lo);
    for (uint32_t o = lane * 16u; o < stage_len; o += 1024u) // stage in (reads <= 15 B past the batch: buffers are padded)
        *reinterpret_cast<uint4 *>(lds + o) = ld16u(ms + stage_lo, o);
    __syncthreads();

    const int K = (int)k, T = (int)t;
    uint32_t len = 0, c0 = 0, c1 = 0;
    uint8_t *row = lds; // LDS address of position 0 of the lane's sequence (may lie below lds: only [c0-1, ..) is touched)
    int x_cur = 0, x_next = 0, a_under = 0;
    bool active = false;
    // the sequence that holds the span's first byte (wave-uniform search), then, per lane, the one that
    // holds its piece: at most 63 sequences further on
    uint32_t s_first = 0;
    {
        uint32_t hi = n_seqs;
        while (hi - s_first > 1) {
            const uint32_t mid = s_first + (hi - s_first) / 2;
            if (off[mid] <= span_base) s_first = mid;
            else hi = mid;
        }
    }
    if (it.len) {
        uint32_t lo = s_first, hi = min(n_seqs, s_first + 64u);
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (off[mid] <= it.start) lo = mid;
            else hi = mid;
        }
        const uint64_t b = off[lo];
        len = (uint32_t)(off[lo + 1] - b);
        c0 = (uint32_t)(it.start - b);
        c1 = c0 + it.len;
        row = lds + (int64_t)(b - stage_lo);
        active = len >= 3 && len <= max_len;
        if (active) {
            a_under = c0 > 0 ? row[c0 - 1u] : 0;
            if (c1 < len) { // x[c1 - 1], x[c1] from the nearest context-free position at or above c1
                const uint32_t last = len - 1u, stop = min(last, c1 + kDtLookahead - 1u);
                uint32_t p = c1;
                while (p < stop && row[p] != (uint8_t)K) p++;
                const int a = row[p];
                if (p != last && a != K) { // nothing context-free in reach
                    redo[lo] = 1u;
                    active = false;
                } else {
                    int x = p == last ? (a > T ? a : 0) : K, xn = x;
                    for (uint32_t q = p; q-- > c1 - 1u;) {
                        const int aq = row[q];
                        xn = x;
                        x = (aq == K) ? K : ((aq > T && x < aq) ? aq : x - 1);
                    }
                    x_cur = x;
                    x_next = xn;
                }
            } else { // the piece holds the sequence's last position (derandomize.rs:282)
                const int a = row[len - 1u];
                x_cur = a > T ? a : 0;
                x_next = x_cur;
            }
        }
    }
    __syncthreads();
    if (active) {
        int a = c1 - 1u > c0 ? (int)row[c1 - 2u] : a_under; // noisy[p - 1] for p = c1 - 1, fetched one step ahead
        for (uint32_t p = c1; p-- > c0;) { // (a variant without position tests for interior pieces was slower:
            const int a_here = a;         //  its carried flags cost more mask bookkeeping than the tests)
            a = p > c0 + 1u ? (int)row[p - 2u] : a_under; // for the next step (unused after the last one)
            const int x_prev = p > 0 ? ((a_here == K) ? K : ((a_here > T && x_cur < a_here) ? a_here : x_cur - 1)) : K;
            row[p] = (uint8_t)translate_char(x_prev, x_cur, x_next, p, len, K, T);
            x_next = x_cur;
            x_cur = x_prev;
        }
    }
    __syncthreads();
    if ((head & 15u) == 0) {
        for (uint32_t o = lane * 16u; o < span_len; o += 1024u) { // stage out
            uint4 c = *reinterpret_cast<const uint4 *>(lds + head + o);
            if (ref) {
                const uint4 rf = ld16u(ref + span_base, o);
                c.x = fmt_word(c.x, rf.x);
                c.y = fmt_word(c.y, rf.y);
                c.z = fmt_word(c.z, rf.z);
                c.w = fmt_word(c.w, rf.w);
            }
            if (o + 16u <= span_len) st16u(out + span_base, o, c);
            else st_partial(out + span_base + o, c, span_len - o);
        }
    } else { // a span that starts within the first 16 bytes of the batch but not at byte 0: byte by byte
        for (uint32_t o = lane; o < span_len; o += 64u) {
            uint32_t ch = lds[head + o];
            if (ref) ch = (ch == 'M' || ch == 'R') ? ref[span_base + o] : (uint32_t)'-';
            out[span_base + o] = (uint8_t)ch;
        }
    }
}

// ---- one very long sequence: chunked derandomize ---------------------------------------
// The recurrence x[i] = f(noisy[i], x[i+1]) only ever compares x with values in (t, k], so
// for decisions x matters through the finite state  S(x) = x > t ? x - t : 0  (k - t + 1
// states); a value <= t keeps counting down exactly until some position "fires" (reset
// noisy == k, anchor noisy > t && x < noisy, or the last position's own rule).  Every chunk
// of positions is therefore a function  x_in -> x_out  described by one entry per state:
//     fired ? (exact x_out) : x_in - chunk_len
// Tables compose exactly, which gives a three-level scan: per-chunk tables (parallel),
// per-group tables (parallel over groups x states), one short sequential pass over the
// groups, inputs per chunk (parallel over groups), and the final per-chunk pass that emits
// characters (parallel).  Bit-identical to the sequential loop.
constexpr uint32_t kDlChunk = 128;  // positions per chunk
constexpr uint32_t kDlGroup = 128;  // chunks per group
constexpr int32_t kDlPass = (int32_t)0x80000000; // table value: "not fired, x_out = x_in - len"

__device__ __forceinline__ int dl_step(int a, int x, uint32_t p, uint32_t len, int K, int T, bool &fired)
{
    if (p == len - 1) { fired = true; return a > T ? a : 0; }   // derandomize.rs:282
    if (a == K) { fired = true; return K; }                      // derandomize.rs:235-238
    if (a > T && x < a) { fired = true; return a; }              // derandomize.rs:240-244
    return x - 1;
}
__device__ __forceinline__ uint32_t dl_state(int x, int T) { return x > T ? (uint32_t)(x - T) : 0u; }
__device__ __forceinline__ int dl_apply(int32_t entry, int x_in, uint32_t span) { return entry == kDlPass ? x_in - (int)span : entry; }

// lane = (chunk, state)
__global__ void dl_chunk_tables_kernel(const uint8_t *__restrict__ ms, uint32_t len, uint32_t k, uint32_t t,
                                       uint32_t n_chunks, uint32_t n_states, int32_t *__restrict__ t1)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_chunks * n_states) return;
    const uint32_t c = id / n_states, st = id % n_states;
    const int K = (int)k, T = (int)t;
    const uint32_t p0 = c * kDlChunk, p1 = min(len, p0 + kDlChunk);
    int x = T + (int)st; // representative of the state (st == 0: any value <= t)
    bool fired = false;
    for (uint32_t p = p1; p-- > p0;) x = dl_step(ms[p], x, p, len, K, T, fired);
    t1[id] = fired ? x : kDlPass;
}

// lane = (group, state): compose the chunk tables of the group, right to left
__global__ void dl_group_tables_kernel(const int32_t *__restrict__ t1, uint32_t len, uint32_t t, uint32_t n_chunks,
                                       uint32_t n_groups, uint32_t n_states, int32_t *__restrict__ t2)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_groups * n_states) return;
    const uint32_t g = id / n_states, st = id % n_states;
    const int T = (int)t;
    const uint32_t c0 = g * kDlGroup, c1 = min(n_chunks, c0 + kDlGroup);
    int x = T + (int)st;
    bool fired = false;
    for (uint32_t c = c1; c-- > c0;) {
        const uint32_t span = min(len, (c + 1) * kDlChunk) - c * kDlChunk;
        const int32_t e = t1[c * n_states + dl_state(x, T)];
        fired = fired || e != kDlPass;
        x = dl_apply(e, x, span);
    }
    t2[id] = fired ? x : kDlPass;
}

// one lane: exact x entering every group (from its right)
__global__ void dl_top_kernel(const int32_t *__restrict__ t2, uint32_t len, uint32_t t, uint32_t n_groups,
                              uint32_t n_states, int32_t *__restrict__ g_in)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int T = (int)t;
    int x = 0; // irrelevant: the last position's rule ignores it
    for (uint32_t g = n_groups; g-- > 0;) {
        g_in[g] = x;
        const uint32_t p0 = g * kDlGroup * kDlChunk, p1 = min(len, p0 + kDlGroup * kDlChunk);
        x = dl_apply(t2[g * n_states + dl_state(x, T)], x, p1 - p0);
    }
}

// lane = group: exact x entering every chunk of the group
__global__ void dl_chunk_inputs_kernel(const int32_t *__restrict__ t1, const int32_t *__restrict__ g_in, uint32_t len,
                                       uint32_t t, uint32_t n_chunks, uint32_t n_groups, uint32_t n_states,
                                       int32_t *__restrict__ c_in)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int T = (int)t;
    const uint32_t c0 = g * kDlGroup, c1 = min(n_chunks, c0 + kDlGroup);
    int x = g_in[g];
    for (uint32_t c = c1; c-- > c0;) {
        c_in[c] = x;
        const uint32_t span = min(len, (c + 1) * kDlChunk) - c * kDlChunk;
        x = dl_apply(t1[c * n_states + dl_state(x, T)], x, span);
    }
}

// lane = chunk: final pass with the exact incoming value; emits characters (and values)
__global__ void dl_emit_kernel(const uint8_t *__restrict__ ms, const int32_t *__restrict__ c_in, uint32_t len, uint32_t k,
                               uint32_t t, uint32_t n_chunks, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out,
                               int32_t *__restrict__ derand_out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int K = (int)k, T = (int)t;
    const uint32_t p0 = c * kDlChunk, p1 = min(len, p0 + kDlChunk);
    int x_next = c_in[c]; // x[p1] (unused when p1 == len)
    bool fired = false;
    int x_cur = dl_step(ms[p1 - 1], x_next, p1 - 1, len, K, T, fired);
    if (p1 == len) x_next = x_cur;
    for (uint32_t p = p1; p-- > p0;) {
        int x_prev = K;
        if (p > 0) x_prev = dl_step(ms[p - 1], x_cur, p - 1, len, K, T, fired);
        uint32_t ch = translate_char(x_prev, x_cur, x_next, p, len, K, T);
        if (ref) ch = (ch == 'M' || ch == 'R') ? ref[p] : (uint32_t)'-';
        out[p] = (uint8_t)ch;
        if (derand_out) derand_out[p] = x_cur;
        x_next = x_cur;
        x_cur = x_prev;
    }
}

__global__ void translate_kernel(const int32_t *__restrict__ x, uint64_t len, uint32_t k, uint32_t t,
                                 uint8_t *__restrict__ out)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= len) return;
    const int xc = x[p];
    const int xm = p > 0 ? x[p - 1] : 0;
    const int xn = p + 1 < len ? x[p + 1] : xc;
    out[p] = (uint8_t)translate_char(xm, xc, xn, (uint32_t)p, (uint32_t)len, (int)k, (int)t); // len < 2^32 (checked by the caller)
}

// ---- format::run_lengths_gapped (format.rs:143-193) on the device -------------------------
// One lane per sequence, left to right over its characters in 16-byte blocks.  The reference's
// two nested loops become one step per character with the state below: outside a run a
// character other than '-' / ' ' opens one (and is then processed as the run's first
// character); inside a run a ' ' closes it without being consumed, anything else updates the
// counters and may close the run (gap longer than max_gap_len, or a gap at the very end), in
// which case the gap that closed it is taken back out of the counters.  The kernel runs twice:
// COUNT (runs per sequence) and, after an exclusive scan of the counts, EMIT (records of seven
// u32 {start, end, matches, mismatches, jumps, gap_bases, gap_opens}; the host widens them).
struct RleState {
    uint32_t in_run, start, end, matches, mismatches, jumps, gap_bases, gap_opens, gap_run, in_gap, prev, n_out;
};

template <bool EMIT>
__device__ __forceinline__ void rle_close(RleState &st, uint32_t *__restrict__ out, uint32_t first, uint32_t capacity)
{
    if (EMIT) {
        const uint32_t slot = first + st.n_out;
        if (slot < capacity) {
            uint32_t *o = out + (uint64_t)slot * 7u;
            o[0] = st.start; o[1] = st.end; o[2] = st.matches; o[3] = st.mismatches;
            o[4] = st.jumps; o[5] = st.gap_bases; o[6] = st.gap_opens;
        }
    }
    st.n_out++;
    st.in_run = 0;
}

template <bool EMIT>
__device__ __forceinline__ void rle_step(RleState &st, uint32_t c, uint32_t i, uint32_t len, uint32_t max_gap_len,
                                         uint32_t *__restrict__ out, uint32_t first, uint32_t capacity)
{
    if (i >= len) return; // only in the last block
    if (st.in_run && c == ' ') rle_close<EMIT>(st, out, first, capacity); // format.rs:154: the blank is not consumed
    if (!st.in_run && c != '-' && c != ' ') { // format.rs:148-152: a run starts here
        st.in_run = 1;
        st.start = i;
        st.end = st.matches = st.mismatches = st.jumps = st.gap_bases = st.gap_opens = st.gap_run = st.in_gap = 0;
    }
    if (st.in_run) { // format.rs:155-188
        const bool true_gap = c == '-';
        if (true_gap && !st.in_gap) {
            st.in_gap = 1;
            st.gap_opens++;
            st.gap_run = 0;
        }
        if (!true_gap) st.in_gap = 0;
        const bool is_match = c == 'M' || c == 'R' || c == 'I';
        const bool is_gap = true_gap || c == 'D';
        st.matches += is_match;
        st.gap_bases += is_gap;
        st.mismatches += (!is_match && !is_gap);
        st.end = (is_match || !is_gap) ? i + 1u : st.end;
        st.jumps += (c == 'R' && i > 0 && st.prev == 'R'); // aln[i-1] is unguarded in the reference (format.rs:175)
        st.gap_run += true_gap;
        if (st.gap_run > max_gap_len || (is_gap && i + 1u == len && st.gap_opens > 0)) {
            st.gap_opens -= 1;
            st.gap_bases -= st.gap_run;
            rle_close<EMIT>(st, out, first, capacity);
        }
    }
    st.prev = c;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void rle_kernel(const uint8_t *__restrict__ chars, const uint64_t *__restrict__ off,
                                                  uint32_t n_seqs, uint32_t max_gap_len, uint32_t *__restrict__ counts,
                                                  const uint32_t *__restrict__ sums, uint32_t *__restrict__ out,
                                                  uint32_t capacity)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    const uint64_t b = off[s];
    const uint32_t len = (uint32_t)(off[s + 1] - b);
    const uint8_t *row = chars + b;
    const uint32_t first = EMIT ? sums[s / kScanBlock] + counts[s] : 0u; // exclusive prefix after the scan
    RleState st;
    st.in_run = st.start = st.end = st.matches = st.mismatches = st.jumps = st.gap_bases = st.gap_opens = 0;
    st.gap_run = st.in_gap = st.prev = st.n_out = 0;
    const uint32_t nblk = (len + 15u) / 16u;
    uint4 cur = nblk ? ld16u(row, 0) : make_uint4(0, 0, 0, 0); // reads <= 15 bytes past the sequence (buffers are padded)
    for (uint32_t bi = 0; bi < nblk; bi++) {
        uint4 nxt = cur;
        if (bi + 1 < nblk) nxt = ld16u(row, 16u * (bi + 1u));
        const uint32_t p0 = 16u * bi;
#define KBO_RLE(J) rle_step<EMIT>(st, blk_byte<J>(cur), p0 + J, len, max_gap_len, out, first, capacity);
        KBO_RLE(0) KBO_RLE(1) KBO_RLE(2) KBO_RLE(3) KBO_RLE(4) KBO_RLE(5) KBO_RLE(6) KBO_RLE(7)
        KBO_RLE(8) KBO_RLE(9) KBO_RLE(10) KBO_RLE(11) KBO_RLE(12) KBO_RLE(13) KBO_RLE(14) KBO_RLE(15)
#undef KBO_RLE
        cur = nxt;
    }
    if (st.in_run) rle_close<EMIT>(st, out, first, capacity); // format.rs:189 after the inner loop ran off the end
    if (!EMIT) counts[s] = st.n_out;
}

// max_gap_len == 0 (FindOpts' default) for batches of reads (<= 480 characters), one wave per 64
// consecutive sequences staged through LDS (SKEW as in the A5/A6 kernel): every '-' closes the run it would open a gap
// in, so runs are the maximal stretches without '-' and ' ', and a run's record is a handful of
// class counts over its stretch.  Each lane classifies 16 characters of its LDS row into bit masks
// and then walks the run boundaries inside them (ffs) instead of stepping the state machine once
// per character; same records as rle_step (format.rs:143-193 with max_gap_len = 0).  (The general
// state machine staged through LDS measured slower than one lane per sequence: 0.65 against 0.49 ms on C2.)
template <bool EMIT, bool SKEW>
__global__ __launch_bounds__(64) void rle0_lds_kernel(const uint8_t *__restrict__ chars, const uint64_t *__restrict__ off,
                                                      uint32_t n_seqs, uint32_t *__restrict__ counts,
                                                      const uint32_t *__restrict__ sums, uint32_t *__restrict__ out,
                                                      uint32_t capacity, uint32_t lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x;
    const uint32_t s0 = blockIdx.x * 64u;
    const uint32_t s = s0 + lane;
    const uint32_t s_end = min(s0 + 64u, n_seqs);
    const uint64_t base = off[s0];
    const uint32_t span = (uint32_t)(off[s_end] - base);
    if (span > lds_bytes) return; // cannot happen: the host sizes lds_bytes from the longest sequence
    auto at = [&](uint32_t x) -> uint8_t & { return lds[SKEW ? x + ((x >> 7) << 2) : x]; };
    for (uint32_t o = lane * 16u; o < span; o += 1024u) { // stage in (reads <= 15 B past the span)
        const uint4 v = ld16u(chars + base, o);
        if (SKEW) {
            uint32_t *d = reinterpret_cast<uint32_t *>(&at(o));
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
            *reinterpret_cast<uint4 *>(lds + o) = v;
        }
    }
    __syncthreads();
    if (s >= n_seqs) return;
    const uint32_t b = (uint32_t)(off[s] - base), len = (uint32_t)(off[s + 1] - off[s]);
    const uint32_t first = EMIT ? sums[s / kScanBlock] + counts[s] : 0u;
    uint32_t n_out = 0, in_run = 0, start = 0, end = 0, matches = 0, mismatches = 0, jumps = 0, gap_bases = 0, prev_r = 0;
    auto close = [&]() {
        if (EMIT) {
            const uint32_t slot = first + n_out;
            if (slot < capacity) {
                uint32_t *o = out + (uint64_t)slot * 7u;
                o[0] = start; o[1] = end; o[2] = matches; o[3] = mismatches; o[4] = jumps; o[5] = gap_bases; o[6] = 0u;
            }
        }
        n_out++;
        in_run = 0;
    };
    for (uint32_t i0 = 0; i0 < len; i0 += 16u) {
        const uint32_t n = min(16u, len - i0);
        uint32_t E = 0, M = 0, D = 0, R = 0; // bit j describes character i0 + j: run breaker / match / 'D' / 'R'
#pragma unroll
        for (uint32_t j = 0; j < 16u; j++) {
            if (j < n) {
                const uint32_t c = at(b + i0 + j);
                E |= (uint32_t)(c == '-' || c == ' ') << j;
                M |= (uint32_t)(c == 'M' || c == 'R' || c == 'I') << j;
                D |= (uint32_t)(c == 'D') << j;
                R |= (uint32_t)(c == 'R') << j;
            }
        }
        const uint32_t valid = n == 16u ? 0xFFFFu : (1u << n) - 1u;
        const uint32_t RR = R & ((R << 1) | prev_r); // 'R' right after an 'R' (format.rs:175)
        prev_r = (R >> 15) & 1u;                     // only meaningful when n == 16 (no block follows otherwise)
        uint32_t cur = 0;
        while (cur < n) {
            if (in_run) {
                const uint32_t ev = E & valid & (~0u << cur);
                const uint32_t e = ev ? (uint32_t)__ffs((int)ev) - 1u : n;
                const uint32_t seg = ((1u << e) - 1u) & (~0u << cur); // characters [cur, e) of the block
                matches += __popc(M & seg);
                gap_bases += __popc(D & seg);
                mismatches += __popc(~M & ~D & seg);
                jumps += __popc(RR & seg);
                const uint32_t non_d = ~D & seg; // end moves past every character that is not a gap (format.rs:172)
                if (non_d) end = i0 + (32u - (uint32_t)__clz((int)non_d));
                if (e < n) { // a '-' (its gap is taken back out: net nothing) or a ' ' ends the run
                    close();
                    cur = e + 1u;
                } else {
                    cur = n;
                }
            } else {
                const uint32_t ne = ~E & valid & (~0u << cur);
                if (!ne) break;
                cur = (uint32_t)__ffs((int)ne) - 1u;
                in_run = 1;
                start = i0 + cur;
                end = matches = mismatches = jumps = gap_bases = 0;
            }
        }
    }
    if (in_run) close();
    if (!EMIT) counts[s] = n_out;
}

template <bool EMIT>
static void launch_rle_pass(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                            uint32_t *local, const uint32_t *sums, uint32_t *d_rles, uint32_t capacity, uint32_t max_seq_len,
                            hipStream_t stream)
{
    if (max_seq_len > 0 && max_seq_len <= 480 && max_gap_len == 0) {
        const uint32_t lds_bytes = ((64u * max_seq_len + 15u) / 16u) * 16u + 16u;
        if (max_seq_len % 32u == 0)
            hipLaunchKernelGGL((rle0_lds_kernel<EMIT, true>), dim3((n_seqs + 63) / 64), dim3(64), lds_bytes + lds_bytes / 32u + 16u,
                               stream, d_chars, d_offsets, n_seqs, local, sums, d_rles, capacity, lds_bytes);
        else
            hipLaunchKernelGGL((rle0_lds_kernel<EMIT, false>), dim3((n_seqs + 63) / 64), dim3(64), lds_bytes, stream, d_chars,
                               d_offsets, n_seqs, local, sums, d_rles, capacity, lds_bytes);
    } else {
        hipLaunchKernelGGL((rle_kernel<EMIT>), dim3((n_seqs + 255) / 256), dim3(256), 0, stream, d_chars, d_offsets, n_seqs,
                           max_gap_len, local, sums, d_rles, capacity);
    }
}

// total number of runs (the scan's grand total) -> one word the host can read after the stream
__global__ void rle_total_kernel(const uint32_t *__restrict__ local, const uint32_t *__restrict__ sums, uint32_t n_seqs,
                                 uint32_t *__restrict__ total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = sums[n_seqs / kScanBlock] + local[n_seqs];
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

// format::run_lengths_gapped over a batch: counts -> exclusive scan (counts[n_seqs] = total slot) -> emit.
// d_scratch: chunk_items_scratch_words(n_seqs) u32 (per-sequence first-run index, block sums);
// d_total: one u32.  Records beyond `capacity` are counted but not written (the caller re-emits).
hipError_t launch_rle_count(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                            uint32_t *d_scratch, uint32_t *d_total, hipStream_t stream, uint32_t max_seq_len)
{
    if (n_seqs == 0) return hipSuccess;
    const uint32_t n = n_seqs + 1, nb = (n + kScanBlock - 1) / kScanBlock;
    uint32_t *local = d_scratch, *sums = d_scratch + n;
    const hipError_t e = hipMemsetAsync(local + n_seqs, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    launch_rle_pass<false>(d_chars, d_offsets, n_seqs, max_gap_len, local, nullptr, nullptr, 0u, max_seq_len, stream);
    hipLaunchKernelGGL(scan_kernel, dim3(nb), dim3(256), 0, stream, local, n, kScanBlock / 256, sums);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, sums, nb, (nb + 1023) / 1024, (uint32_t *)nullptr);
    hipLaunchKernelGGL(rle_total_kernel, dim3(1), dim3(64), 0, stream, local, sums, n_seqs, d_total);
    return hipGetLastError();
}

hipError_t launch_rle_emit(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                           uint32_t *d_scratch, uint32_t *d_rles, uint32_t capacity, hipStream_t stream,
                           uint32_t max_seq_len)
{
    if (n_seqs == 0) return hipSuccess;
    uint32_t *local = d_scratch, *sums = d_scratch + n_seqs + 1;
    launch_rle_pass<true>(d_chars, d_offsets, n_seqs, max_gap_len, local, sums, d_rles, capacity, max_seq_len, stream);
    return hipGetLastError();
}

size_t chunk_items_scratch_words(uint32_t n_seqs) { return (size_t)n_seqs + 1 + ((size_t)n_seqs + 1 + kScanBlock - 1) / kScanBlock; }

hipError_t launch_make_chunk_items(const uint64_t *d_offsets, uint32_t n_seqs, uint32_t chunk, uint32_t k,
                                   uint32_t n_slots, WalkItem *d_items, uint32_t *d_scratch, hipStream_t stream)
{
    if (n_seqs == 0 || n_slots == 0) return hipSuccess;
    const uint32_t n = n_seqs + 1, nb = (n + kScanBlock - 1) / kScanBlock;
    uint32_t *local = d_scratch, *sums = d_scratch + n;
    hipLaunchKernelGGL(chunk_count_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_offsets, n_seqs, chunk, local);
    hipLaunchKernelGGL(scan_kernel, dim3(nb), dim3(256), 0, stream, local, n, kScanBlock / 256, sums);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, sums, nb, (nb + 1023) / 1024, (uint32_t *)nullptr);
    hipLaunchKernelGGL(make_chunk_items_kernel, dim3((n_slots + 255) / 256), dim3(256), 0, stream, d_offsets, local, sums,
                       n_seqs, chunk, k, n_slots, d_items);
    return hipGetLastError();
}

int g_walk_threads = kWalkThreads;
int g_rare_period = 8; // tuned on C2 (tools/sweep_walk.py RARE=1)
int g_pair_min_depth = 16;                 // two-base steps only from matches at least this deep
void set_pair_min_depth(int d) { g_pair_min_depth = d < 0 ? 0 : d; }
void set_walk_rare(int period) { g_rare_period = std::max(1, std::min(1024, period)); }
void set_walk_threads(int t) { g_walk_threads = (t == 64 || t == 128 || t == 256) ? t : kWalkThreads; }

hipError_t launch_ms_walk(WalkArgs a, int max_waves, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    // every lane walks `rounds` items; the grid is sized so that lanes * rounds covers the
    // items with as little slack as possible (one 64-lane wave per workgroup)
    const uint64_t lanes = (uint64_t)std::max(1, max_waves) * 64u;
    a.rounds = (uint32_t)((a.n_items + lanes - 1) / lanes);
    a.rare_period = (uint32_t)g_rare_period;
    const uint64_t per_wave = 64ull * a.rounds;
    const uint32_t waves = (uint32_t)((a.n_items + per_wave - 1) / per_wave);
    const uint32_t threads = g_walk_threads;
    const uint32_t wpb = threads / 64;
    const dim3 grid((waves + wpb - 1) / wpb), block(threads);
    const bool ival = a.lo_out && a.hi_out;
    a.pair_min_d = (uint32_t)g_pair_min_depth;
    if (a.ix.big) {
        if (ival) hipLaunchKernelGGL((ms_walk_kernel<true, true, false>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((ms_walk_kernel<false, true, false>), grid, block, 0, stream, a);
    } else {
        if (ival) hipLaunchKernelGGL((ms_walk_kernel<true, false, false>), grid, block, 0, stream, a);
        else if (a.ix.pair_off) hipLaunchKernelGGL((ms_walk_kernel<false, false, true>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((ms_walk_kernel<false, false, false>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

size_t derand_piece_work_bytes(uint32_t n_seqs, uint64_t total_bases)
{
    const uint64_t slots = total_bases / kDtPiece + n_seqs;
    return (size_t)(slots * sizeof(WalkItem) + (chunk_items_scratch_words(n_seqs) + n_seqs) * sizeof(uint32_t) + 64);
}

hipError_t launch_derand_translate(const uint8_t *d_ms, const uint64_t *d_offsets, uint32_t n_seqs,
                                   uint32_t k, uint32_t threshold, const uint8_t *d_ref,
                                   uint8_t *d_chars_out, int32_t *d_derand_out, uint32_t max_seq_len,
                                   uint32_t per_lane_max_len, hipStream_t stream, uint64_t total_bases,
                                   void *d_work, size_t work_bytes)
{
    if (n_seqs == 0) return hipSuccess;
    // short sequences: LDS-staged kernel (64 sequences per wave must fit the LDS budget)
    if (max_seq_len > 0 && max_seq_len <= 480 && d_derand_out == nullptr) {
        const uint32_t lds_bytes = ((64u * max_seq_len + 15u) / 16u) * 16u + 16u;
        if (max_seq_len % 32u == 0) // e.g. reads of 128 or 256 bases: padded LDS image (see the kernel)
            hipLaunchKernelGGL((derand_translate_lds_kernel<true>), dim3((n_seqs + 63) / 64), dim3(64),
                               lds_bytes + lds_bytes / 32u + 16u, stream, d_ms, d_offsets, n_seqs, k, threshold, d_ref,
                               d_chars_out, lds_bytes);
        else
            hipLaunchKernelGGL((derand_translate_lds_kernel<false>), dim3((n_seqs + 63) / 64), dim3(64), lds_bytes, stream,
                               d_ms, d_offsets, n_seqs, k, threshold, d_ref, d_chars_out, lds_bytes);
        return hipGetLastError();
    }
    // long reads / contigs with scratch available: one lane per piece, then the flagged sequences again
    const uint64_t slots = total_bases / kDtPiece + n_seqs;
    if (d_work && d_derand_out == nullptr && total_bases > 0 && slots < (1ull << 31) &&
        work_bytes >= derand_piece_work_bytes(n_seqs, total_bases)) {
        WalkItem *pieces = static_cast<WalkItem *>(d_work);
        uint32_t *scratch = reinterpret_cast<uint32_t *>(pieces + slots);
        uint32_t *redo = scratch + chunk_items_scratch_words(n_seqs);
        hipError_t e = hipMemsetAsync(redo, 0, (size_t)n_seqs * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        e = launch_make_chunk_items(d_offsets, n_seqs, kDtPiece, 1u, (uint32_t)slots, pieces, scratch, stream);
        if (e != hipSuccess) return e;
        const uint32_t lds_bytes = kDtBehind + 64u * kDtPiece + kDtLookahead + 32u;
        hipLaunchKernelGGL(derand_translate_piece_lds_kernel, dim3((unsigned)((slots + 63) / 64)), dim3(64), lds_bytes, stream,
                           d_ms, d_offsets, n_seqs, total_bases, pieces, (uint32_t)slots, k, threshold, d_ref, d_chars_out,
                           per_lane_max_len, redo);
        hipLaunchKernelGGL(derand_translate_kernel, dim3((n_seqs + 255) / 256), dim3(256), 0, stream, d_ms, d_offsets,
                           n_seqs, k, threshold, d_ref, d_chars_out, (int32_t *)nullptr, per_lane_max_len,
                           (const uint32_t *)redo);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(derand_translate_kernel, dim3((n_seqs + 255) / 256), dim3(256), 0, stream, d_ms,
                       d_offsets, n_seqs, k, threshold, d_ref, d_chars_out, d_derand_out, per_lane_max_len,
                       (const uint32_t *)nullptr);
    return hipGetLastError();
}

size_t derand_long_scratch_bytes(uint64_t len, uint32_t k, uint32_t threshold)
{
    const uint64_t n_chunks = (len + kDlChunk - 1) / kDlChunk, n_groups = (n_chunks + kDlGroup - 1) / kDlGroup;
    const uint64_t n_states = (uint64_t)(k > threshold ? k - threshold : 0) + 1;
    return (size_t)((n_chunks * n_states + n_groups * n_states + n_groups + n_chunks) * sizeof(int32_t) + 64);
}

hipError_t launch_derand_long(const uint8_t *d_ms, uint32_t len, uint32_t k, uint32_t threshold, const uint8_t *d_ref,
                              uint8_t *d_chars_out, int32_t *d_derand_out, void *d_scratch, hipStream_t stream)
{
    if (len < 3) return hipSuccess;
    const uint32_t n_chunks = (len + kDlChunk - 1) / kDlChunk, n_groups = (n_chunks + kDlGroup - 1) / kDlGroup;
    const uint32_t n_states = (k > threshold ? k - threshold : 0) + 1;
    int32_t *t1 = static_cast<int32_t *>(d_scratch);
    int32_t *t2 = t1 + (size_t)n_chunks * n_states;
    int32_t *g_in = t2 + (size_t)n_groups * n_states;
    int32_t *c_in = g_in + n_groups;
    const uint32_t T = 256;
    hipLaunchKernelGGL(dl_chunk_tables_kernel, dim3((n_chunks * n_states + T - 1) / T), dim3(T), 0, stream, d_ms, len, k,
                       threshold, n_chunks, n_states, t1);
    hipLaunchKernelGGL(dl_group_tables_kernel, dim3((n_groups * n_states + T - 1) / T), dim3(T), 0, stream, t1, len,
                       threshold, n_chunks, n_groups, n_states, t2);
    hipLaunchKernelGGL(dl_top_kernel, dim3(1), dim3(64), 0, stream, t2, len, threshold, n_groups, n_states, g_in);
    hipLaunchKernelGGL(dl_chunk_inputs_kernel, dim3((n_groups + T - 1) / T), dim3(T), 0, stream, t1, g_in, len, threshold,
                       n_chunks, n_groups, n_states, c_in);
    hipLaunchKernelGGL(dl_emit_kernel, dim3((n_chunks + T - 1) / T), dim3(T), 0, stream, d_ms, c_in, len, k, threshold,
                       n_chunks, d_ref, d_chars_out, d_derand_out);
    return hipGetLastError();
}

hipError_t launch_translate(const int32_t *d_derand, uint64_t len, uint32_t k, uint32_t threshold,
                            uint8_t *d_chars_out, hipStream_t stream)
{
    if (len == 0) return hipSuccess;
    hipLaunchKernelGGL(translate_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, stream,
                       d_derand, len, k, threshold, d_chars_out);
    return hipGetLastError();
}

} // namespace kbo
