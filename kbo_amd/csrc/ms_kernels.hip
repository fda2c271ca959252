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
    F_CON = 8u,   // needs contract-left
    F_MK = 16u,   // contract-left depth m already known (continuing a scan)
    F_DONE = 32u, // finished its item, wants the next one
    F_FIN = 64u,  // no items left
    F_BLOCK = 8u
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

// bit 7 of every byte of x that is < m (mm = m replicated into 4 bytes); full 0..255 range
__device__ __forceinline__ uint32_t bytes_lt(uint32_t x, uint32_t mm)
{
    const uint32_t H = 0x80808080u;
    uint32_t t = (x | H) - (mm & ~H);
    uint32_t ge = ((x & ~mm) | (~(x ^ mm) & t)) & H;
    return ge ^ H;
}
__device__ __forceinline__ uint32_t pack4(uint32_t h) { return (((h >> 7) * 0x00204081u) >> 21) & 0xFu; }
// bit j set iff LCS byte j of the 16-byte window is < m
__device__ __forceinline__ uint32_t lt_mask16(const uint4 &w, uint32_t m)
{
    uint32_t mm = m * 0x01010101u;
    return pack4(bytes_lt(w.x, mm)) | (pack4(bytes_lt(w.y, mm)) << 4) |
           (pack4(bytes_lt(w.z, mm)) << 8) | (pack4(bytes_lt(w.w, mm)) << 12);
}
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
__device__ __forceinline__ void st16u(uint8_t *base, uint32_t byte_off, const uint4 &v)
{
    __builtin_memcpy(base + byte_off, &v, 16);
}
__device__ __forceinline__ void st4u(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

// -------------------------------------------------------------------------------------
// A1.  Semantics:
//   for each base c:  Ic = extend_right(I, c)
//                     while d > 0 && Ic empty:  I = contract_left(I, d-1); d -= 1; Ic = extend_right(I, c)
//                     if Ic non-empty: I = Ic; d = min(d+1, k)
//                     emit (d, I)
// Bit-identical shortcut: contract_left(I, t) leaves I unchanged for every
// t > m = max(LCS[l], LCS[r]), and extend_right of an unchanged interval is still empty,
// so the loop jumps straight to depth m (one LCS look-up + one scan) instead of stepping
// d-1, d-2, ... with a failing rank pair each time.  A non-ACGT base reads the all-zero
// "null" rank block, so its extension is empty at every depth and the same machinery
// contracts it down to the root (d = 0, I = [0,n)), which is what the loop above does.
//
// The kernel is VALU-issue bound (rocprof: SIMD issue saturated, see profiles/), so it is
// organised around a short hot path:
//   hot path  (every iteration, lanes not blocked): two rank-block loads, rank arithmetic,
//             accept / emit / advance to the next base;
//   rare block (entered when >= kRareBatch lanes are blocked, or every 4th iteration if
//             any is): contract-left via the LCS windows, switching to the prefetched next
//             item, requesting the prefetch after that, exit test.
// One lane per work item; lane j of wave w walks items w*64*rounds + j + 64*t.  All
// offsets are 32-bit (one launch covers < 4 GiB of query and an index arena < 4 GiB for
// the 32-bit build), so every access is SGPR base + 32-bit VGPR offset.
#ifndef KBO_ABLATE
#define KBO_ABLATE 0
#endif

template <bool IVAL>
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
    const uint32_t lcs_byte0 = a.ix.lcs_off << 4; // arena byte offset of LCS[0]

    // this lane's items: first, first + 64, ...
    const uint64_t first64 = (uint64_t)wave * 64u * a.rounds + lane;
    uint32_t next_item = first64 < a.n_items ? (uint32_t)first64 : a.n_items;
    uint32_t left = 0; // items whose descriptor has not been requested yet
    if (next_item < a.n_items) left = min(a.rounds, (a.n_items - 1u - next_item) / 64u + 1u);

    uint32_t flags = left ? F_DONE : F_FIN;
    uint32_t l = 0, r = n, d = 0, m = 0, cb = 0;
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

    uint32_t dbg_rare = 0, dbg_con = 0, dbg_iter = 0;
    for (uint32_t iter = 0;; iter++) {
        dbg_iter = iter;
        // ================================ rare block ================================
        const uint64_t blocked = __ballot((flags & (F_CON | F_DONE)) != 0);
        const uint64_t hot = __ballot(flags < F_BLOCK);
        if (hot == 0 || (blocked != 0 && ((uint32_t)__popcll(blocked) >= a.rare_batch || (iter & a.rare_mask) == 0))) {
            dbg_rare++;
            // ---- contract-left for every lane that asked for it (usually one pass)
            // LCS windows are UNALIGNED 16-byte loads: wl = LCS[l-15 .. l], wr = LCS[r .. r+15]
            // (the zero "null" block in front of the LCS bytes and the zero padding behind
            // them are the scan sentinels), so LCS[l] and LCS[r] sit at fixed bytes.
            while (__ballot((flags & F_CON) != 0)) {
                dbg_con++;
                if (flags & F_CON) {
                    const uint4 wl = ld16u(arena, lcs_byte0 + l - 15u);
                    const uint4 wr = ld16u(arena, lcs_byte0 + r);
                    if (!(flags & F_MK)) {
                        // Final depth in one step.  The extension stays empty until the interval
                        // reaches the nearest set bit of B_c below l (row l - dl) or at/after r
                        // (row r + dr); contract_left reaches them at depths
                        //   tL = min LCS[l-dl+1 .. l],  tR = min LCS[r .. r+dr]
                        // so the loop of the reference ends exactly at depth max(tL, tR).  When a
                        // nearest bit is outside the loaded block or window, fall back to the
                        // one-level jump m = max(LCS[l], LCS[r]) (always a valid intermediate stop).
                        const uint32_t bl = div96(l), br = div96(r);
                        const bool null_c = cb == null_blk;
                        const uint4 bA = ld16(arena, (null_c ? cb : cb + bl) << 4);
                        const uint4 bB = ld16(arena, (null_c ? cb : cb + br) << 4);
                        const uint32_t ol = l - bl * kRankRows, orr = r - br * kRankRows;
                        uint32_t dl = 1000u, dr = 1000u;
                        {
                            const uint64_t X = ~0ull << (ol & 63u);
                            const uint32_t xl = (uint32_t)X, xh = (uint32_t)(X >> 32);
                            const bool big = ol >= 64u;
                            const uint32_t y = bA.y & ~(big ? 0u : xl), z = bA.z & ~(big ? 0u : xh),
                                           w = bA.w & ~(big ? xl : ~0u);
                            int top = -1; // highest set bit strictly below ol
                            if (y) top = 31 - __clz((int)y);
                            if (z) top = 63 - __clz((int)z);
                            if (w) top = 95 - __clz((int)w);
                            if (top >= 0) dl = ol - (uint32_t)top;
                        }
                        {
                            const uint64_t X = ~0ull << (orr & 63u);
                            const uint32_t xl = (uint32_t)X, xh = (uint32_t)(X >> 32);
                            const bool big = orr >= 64u;
                            const uint32_t y = bB.y & (big ? 0u : xl), z = bB.z & (big ? 0u : xh),
                                           w = bB.w & (big ? xl : ~0u);
                            int low = -1; // lowest set bit at or above orr
                            if (w) low = 64 + __ffs((int)w) - 1;
                            if (z) low = 32 + __ffs((int)z) - 1;
                            if (y) low = __ffs((int)y) - 1;
                            if (low >= 0) dr = (uint32_t)low - orr;
                        }
                        const uint32_t lcs_l = wl.w >> 24, lcs_r = wr.x & 0xFFu;
                        if (dl <= 4u && dr <= 3u) {
                            // bytes l-3..l are wl.w (low to high); keep the top dl of them
                            const uint32_t el = wl.w | (dl == 4u ? 0u : ((1u << (8u * (4u - dl))) - 1u));
                            // bytes r..r+3 are wr.x; keep the low dr+1 of them
                            const uint32_t er = wr.x | (dr == 3u ? 0u : (~0u << (8u * (dr + 1u))));
                            const uint32_t tl = min(min(el & 0xFFu, (el >> 8) & 0xFFu), min((el >> 16) & 0xFFu, el >> 24));
                            const uint32_t tr = min(min(er & 0xFFu, (er >> 8) & 0xFFu), min((er >> 16) & 0xFFu, er >> 24));
                            m = max(tl, tr);
                        } else {
                            m = max(lcs_l, lcs_r);
                        }
                        d = m;
                        flags |= F_MK;
                    }
                    if (m == 0) {
                        l = 0;
                        r = n;
                        flags &= ~(F_CON | F_MK);
                    } else {
                        const uint32_t ml = lt_mask16(wl, m), mr = lt_mask16(wr, m);
                        // left: highest j with LCS[l-15+j] < m; right: lowest j with LCS[r+j] < m
                        l = ml ? l - (uint32_t)__clz((int)ml) + 16u : l - 16u; // l-15+(31-clz)
                        r = mr ? r + (uint32_t)__ffs((int)mr) - 1u : r + 16u;
                        if (ml && mr) flags &= ~(F_CON | F_MK);
                    }
                }
            }
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

        // ================================= hot path =================================
        if (flags < F_BLOCK) {
            const uint32_t bl = div96(l), br = div96(r);
            const bool null_c = cb == null_blk;
            const uint4 xA = ld16(arena, (null_c ? cb : cb + bl) << 4);
            const uint4 xB = ld16(arena, (null_c ? cb : cb + br) << 4);
            if (flags & F_QF) { // the query block after the current one (reads <= 16 bytes past the item)
                qnxt = ld16u(qb, min(start + (i & ~15u) + 16u, q_end)); // stays within the 16-byte slack
                flags &= ~F_QF;
            }
            const uint32_t l2 = rank_eval(xA, l - bl * kRankRows), r2 = rank_eval(xB, r - br * kRankRows);
            const bool ok = l2 < r2;
            if (ok) {
                l = l2;
                r = r2;
                d = min(d + 1, k);
            }
            if (ok || d == 0) {
                if (i >= warm) { // emit: output byte e = i - warm of this item
                    const uint32_t e = i - warm;
                    ocur |= d << ((e & 3u) * 8u);
                    if (IVAL) {
                        a.lo_out[start + i] = l;
                        a.hi_out[start + i] = r;
                    }
                    const bool fin_e = (i + 1 == len);
                    if ((e & 3u) == 3u || fin_e) { // word complete (or item ends): move it into the block
                        const uint32_t w = (e >> 2) & 3u;
                        oblk.x = w == 0 ? ocur : oblk.x;
                        oblk.y = w == 1 ? ocur : oblk.y;
                        oblk.z = w == 2 ? ocur : oblk.z;
                        oblk.w = w == 3 ? ocur : oblk.w;
                        ocur = 0;
                        if ((e & 15u) == 15u) { // full block: one unaligned 16-byte store
                            st16u(a.d_out, start + warm + (e & ~15u), oblk);
                        } else if (fin_e) { // tail of the item: words, then bytes
                            uint8_t *o = a.d_out + (start + warm + (e & ~15u));
                            const uint32_t nb = (e & 15u) + 1u; // valid bytes in the block
                            const uint32_t wv[4] = {oblk.x, oblk.y, oblk.z, oblk.w};
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++) {
                                if (nb >= 4u * j + 4u) st4u(o + 4u * j, wv[j]);
                                else {
#pragma unroll
                                    for (uint32_t b = 0; b < 3; b++)
                                        if (nb > 4u * j + b) o[4u * j + b] = (uint8_t)(wv[j] >> (8u * b));
                                }
                            }
                        }
                    }
                }
                i++;
                if (i == len) flags |= F_DONE;
                else {
                    if ((i & 3u) == 0) {
                        if ((i & 15u) == 0) {
                            qblk = qnxt;
                            flags |= F_QF;
                        }
                        const uint32_t w = (i >> 2) & 3u;
                        const uint32_t lo = (w & 1u) ? qblk.y : qblk.x, hi = (w & 1u) ? qblk.w : qblk.z;
                        qcur = (w & 2u) ? hi : lo;
                    }
                    const uint32_t c = decode_base((qcur >> ((i & 3u) * 8u)) & 0xFFu);
                    cb = c < 4u ? c * nblk : null_blk;
                }
            } else {
                flags |= F_CON;
            }
        }
    }
#ifdef KBO_WALK_DEBUG
    if (lane == 0 && a.lo_out == nullptr && a.hi_out != nullptr) { // debug: hi_out doubles as counter sink
        atomicAdd(a.hi_out + 0, dbg_iter);
        atomicAdd(a.hi_out + 1, dbg_rare);
        atomicAdd(a.hi_out + 2, dbg_con);
        atomicAdd(a.hi_out + 3, 1u);
    }
#endif
    (void)dbg_rare; (void)dbg_con; (void)dbg_iter; (void)lane;
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
__device__ __forceinline__ uint8_t translate_char(int xm, int xc, int xn, uint64_t rel, uint64_t len,
                                                  int K, int T)
{
    const int prev = rel > 1 ? xm : K;
    const int next = rel < len - 1 ? xn : xc;
    const bool r_prev = rel >= 2 && rel < len - 1 && xm > T && xc > 0 && xc < T;
    const bool r_here = xc > T && next > 0 && next < T;
    if (r_prev || r_here) return 'R';
    if (xc <= 0) return (next == 1 && prev > 0) ? 'X' : '-';
    return 'M';
}

// byte j (0..15) of a 16-byte block held in registers
__device__ __forceinline__ uint32_t blk_byte(const uint4 &v, uint32_t j)
{
    const uint32_t lo = (j & 4u) ? v.y : v.x, hi = (j & 4u) ? v.w : v.z;
    return (((j & 8u) ? hi : lo) >> ((j & 3u) * 8u)) & 0xFFu;
}
__device__ __forceinline__ void blk_set_word(uint4 &v, uint32_t w, uint32_t x)
{
    v.x = w == 0 ? x : v.x;
    v.y = w == 1 ? x : v.y;
    v.z = w == 2 ? x : v.z;
    v.w = w == 3 ? x : v.w;
}

// One lane per sequence, right to left.  All three streams (MS in, optional reference in,
// characters out) move in 16-byte blocks relative to the sequence start (unaligned global
// accesses; the input buffers carry 16 bytes of slack, see include/kbo_hip.h).
__global__ __launch_bounds__(256) void derand_translate_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t k,
    uint32_t t, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out, int32_t *__restrict__ derand_out)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    const uint64_t b = off[s], e = off[s + 1];
    const uint32_t len = (uint32_t)(e - b);
    if (len < 3) return; // the host side rejects these (derandomize.rs:276)
    const int K = (int)k, T = (int)t;
    const uint8_t *msb = ms + b;
    const uint8_t *refb = ref ? ref + b : nullptr;
    uint8_t *outb = out + b;

    uint32_t p = len - 1;
    uint4 mblk = ld16u(msb, p & ~15u), rblk = make_uint4(0, 0, 0, 0), oblk = make_uint4(0, 0, 0, 0);
    if (ref) rblk = ld16u(refb, p & ~15u);
    uint4 mnext = mblk, rnext = rblk; // block below the current one, requested one block ahead
    if (p >= 16u) {
        mnext = ld16u(msb, (p & ~15u) - 16u);
        if (ref) rnext = ld16u(refb, (p & ~15u) - 16u);
    }
    int a = (int)blk_byte(mblk, p & 15u);
    int x_cur = a > T ? a : 0, x_next = x_cur, x_prev = K; // derandomize.rs:282
    uint32_t ocur = 0;
    for (;;) {
        if (p > 0) { // x[p-1] from noisy[p-1] and x[p] (derandomize.rs:233-246)
            const uint32_t q = p - 1;
            a = (int)blk_byte((q & 15u) == 15u ? mnext : mblk, q & 15u);
            x_prev = (a == K) ? K : ((a > T && x_cur < a) ? a : x_cur - 1);
        }
        uint32_t ch = translate_char(x_prev, x_cur, x_next, p, len, K, T);
        if (ref) { // format::relative_to_ref: M,R keep the reference base, X and '-' become '-'
            const uint32_t rb = blk_byte(rblk, p & 15u);
            ch = (ch == 'M' || ch == 'R') ? rb : (uint32_t)'-';
        }
        ocur |= ch << ((p & 3u) * 8u);
        if (derand_out) derand_out[b + p] = x_cur;
        if ((p & 3u) == 0) {
            blk_set_word(oblk, (p >> 2) & 3u, ocur);
            ocur = 0;
            if ((p & 15u) == 0) { // block complete down to its first byte
                if (p + 16u <= len) st16u(outb, p, oblk);
                else { // topmost, partial block of the sequence
                    const uint32_t nb = len - p;
                    const uint32_t wv[4] = {oblk.x, oblk.y, oblk.z, oblk.w};
#pragma unroll
                    for (uint32_t j = 0; j < 4; j++) {
                        if (nb >= 4u * j + 4u) st4u(outb + p + 4u * j, wv[j]);
                        else {
#pragma unroll
                            for (uint32_t bb = 0; bb < 3; bb++)
                                if (nb > 4u * j + bb) outb[p + 4u * j + bb] = (uint8_t)(wv[j] >> (8u * bb));
                        }
                    }
                }
                if (p == 0) break;
                mblk = mnext;
                rblk = rnext;
                if (p >= 32u) {
                    mnext = ld16u(msb, p - 32u);
                    if (ref) rnext = ld16u(refb, p - 32u);
                }
            }
        }
        p--;
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
    out[p] = translate_char(xm, xc, xn, p, len, (int)k, (int)t);
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

int g_walk_threads = kWalkThreads;
int g_rare_batch = 8, g_rare_period = 4;
void set_walk_rare(int batch, int period)
{
    g_rare_batch = std::max(1, std::min(64, batch));
    int p = 1;
    while (p < period && p < 1024) p <<= 1;
    g_rare_period = p;
}
void set_walk_threads(int t) { g_walk_threads = (t == 64 || t == 128 || t == 256) ? t : kWalkThreads; }

hipError_t launch_ms_walk(WalkArgs a, int max_waves, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    // every lane walks `rounds` items; the grid is sized so that lanes * rounds covers the
    // items with as little slack as possible (one 64-lane wave per workgroup)
    const uint64_t lanes = (uint64_t)std::max(1, max_waves) * 64u;
    a.rounds = (uint32_t)((a.n_items + lanes - 1) / lanes);
    a.rare_batch = (uint32_t)g_rare_batch;
    a.rare_mask = (uint32_t)g_rare_period - 1u;
    const uint64_t per_wave = 64ull * a.rounds;
    const uint32_t waves = (uint32_t)((a.n_items + per_wave - 1) / per_wave);
    const uint32_t threads = g_walk_threads;
    const uint32_t wpb = threads / 64;
    const dim3 grid((waves + wpb - 1) / wpb), block(threads);
    if (a.lo_out && a.hi_out) hipLaunchKernelGGL(ms_walk_kernel<true>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(ms_walk_kernel<false>, grid, block, 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_derand_translate(const uint8_t *d_ms, const uint64_t *d_offsets, uint32_t n_seqs,
                                   uint32_t k, uint32_t threshold, const uint8_t *d_ref,
                                   uint8_t *d_chars_out, int32_t *d_derand_out, hipStream_t stream)
{
    if (n_seqs == 0) return hipSuccess;
    hipLaunchKernelGGL(derand_translate_kernel, dim3((n_seqs + 255) / 256), dim3(256), 0, stream, d_ms,
                       d_offsets, n_seqs, k, threshold, d_ref, d_chars_out, d_derand_out);
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
