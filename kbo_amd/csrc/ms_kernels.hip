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
// (rank blocks, LCS windows, its next work item, or its first query words), then all
// lanes consume.  Divergence costs only the per-state post-processing.
#include "kernels.hpp"

namespace kbo {
namespace {

enum : uint32_t {
    ST_EXT = 0,      // extend-right: load rank blocks of l and r
    ST_CON_NEW = 1,  // contract-left, target depth not known yet
    ST_CON_SCAN = 2, // contract-left, continuing a scan at depth m
    ST_WANT = 3,     // needs a work item
    ST_ITEM = 4,     // loading its work item
    ST_QLOAD = 5,    // loading the first query words of its item
    ST_DONE = 6
};

// dword i (0..3) of a 16-byte value, by shifts (no dynamic register indexing)
__device__ __forceinline__ uint32_t sel4(const uint4 &v, uint32_t i)
{
    const uint64_t lo = (uint64_t)v.x | ((uint64_t)v.y << 32), hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    const uint64_t h = (i & 2u) ? hi : lo;
    return (uint32_t)(h >> ((i & 1u) * 32u));
}

// ((1 << t) - 1) for t clamped to [0, 32]
__device__ __forceinline__ uint32_t low_mask(int t)
{
    t = min(max(t, 0), 32);
    return (uint32_t)((1ull << t) - 1ull);
}

// block = { C[c] + rank before the block, 96 row bits }; o = offset inside the block
__device__ __forceinline__ uint32_t rank_eval(const uint4 &b, uint32_t o)
{
    return b.x + __popc(b.y & low_mask((int)o)) + __popc(b.z & low_mask((int)o - 32)) +
           __popc(b.w & low_mask((int)o - 64));
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
__device__ __forceinline__ uint32_t byte_at(const uint4 &w, uint32_t j)
{
    const uint64_t lo = (uint64_t)w.x | ((uint64_t)w.y << 32), hi = (uint64_t)w.z | ((uint64_t)w.w << 32);
    const uint64_t h = (j & 8u) ? hi : lo;
    return (uint32_t)(h >> ((j & 7u) * 8u)) & 0xFFu;
}

// 'A','C','G','T' -> 0..3, anything else -> 4 (sbwt's DNA alphabet is exactly ACGT)
__device__ __forceinline__ uint32_t decode_base(uint32_t ch)
{
    uint32_t c = ((ch >> 1) & 3u) ^ ((ch >> 2) & 1u);
    uint32_t back = (0x54474341u >> (8 * c)) & 0xFFu;
    return back == ch ? c : 4u;
}

// -------------------------------------------------------------------------------------
// A1.  Semantics:
//   for each base c:  Ic = extend_right(I, c)
//                     while d > 0 && Ic empty:  I = contract_left(I, d-1); d -= 1; Ic = extend_right(I, c)
//                     if Ic non-empty: I = Ic; d = min(d+1, k)
//                     emit (d, I)
// Bit-identical shortcut used here: contract_left(I, t) leaves I unchanged for every
// t > m = max(LCS[l], LCS[r]), and extend_right of an unchanged interval is still empty,
// so the loop can jump straight to depth m (one LCS look-up + one scan) instead of
// stepping d-1, d-2, ... with a failing rank pair each time.
template <bool IVAL>
__global__ __launch_bounds__(kWalkThreads) void ms_walk_kernel(WalkArgs a)
{
    const uint32_t n = a.ix.n, k = a.ix.k;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves_per_block = blockDim.x >> 6;
    const uint32_t wave = blockIdx.x * waves_per_block + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * waves_per_block;
    // contiguous slice of the items per wave; lanes of the wave pull from it dynamically
    uint32_t w_next = (uint32_t)(((uint64_t)a.n_items * wave) / n_waves);
    const uint32_t w_end = (uint32_t)(((uint64_t)a.n_items * (wave + 1)) / n_waves);

    const uint4 *q16 = reinterpret_cast<const uint4 *>(a.q);
    const uint32_t *q32 = reinterpret_cast<const uint32_t *>(a.q);
    const uint64_t last_win = (a.q_bytes - 1) >> 4, last_dw = (a.q_bytes - 1) >> 2;
    uint32_t *d_out32 = reinterpret_cast<uint32_t *>(a.d_out);

    uint32_t st = ST_WANT;
    uint32_t l = 0, r = 0, d = 0, m = 0, c = 0;
    uint32_t i = 0, len = 0, warm = 0, item_idx = 0;
    uint64_t base = 0;
    uint32_t qcur = 0, qnext = 0, obuf = 0;
    bool need_fetch = false;

    for (;;) {
        // ---- hand out items (wave-local: ballots and popcounts only)
        const uint64_t want = __ballot(st == ST_WANT);
        if (want) {
            const uint32_t mine = w_next + (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
            if (st == ST_WANT) {
                if (mine < w_end) { item_idx = mine; st = ST_ITEM; }
                else st = ST_DONE;
            }
            w_next = min(w_next + (uint32_t)__popcll(want), w_end);
        }
        if (__ballot(st != ST_DONE) == 0) break;

        // ---- two 16-byte loads per lane, whatever its state
        uint32_t bl = 0, br = 0;
        const uint4 *pA = a.ix.lcs16, *pB = a.ix.lcs16;
        if (st == ST_EXT) {
            bl = div96(l);
            br = div96(r);
            const uint4 *rk = a.ix.rank + (uint64_t)(c & 3u) * a.ix.n_blocks;
            pA = rk + bl;
            pB = rk + br;
        } else if (st <= ST_CON_SCAN) {
            pA = a.ix.lcs16 + (l >> 4);
            pB = a.ix.lcs16 + (r >> 4);
        } else if (st == ST_ITEM) {
            pA = pB = reinterpret_cast<const uint4 *>(a.items + item_idx);
        } else if (st == ST_QLOAD) {
            const uint64_t w0 = base >> 4;
            pA = q16 + w0;
            pB = q16 + min(w0 + 1, last_win);
        }
        uint4 xA = make_uint4(0, 0, 0, 0), xB = make_uint4(0, 0, 0, 0);
        if (st != ST_DONE) {
            xA = *pA;
            xB = *pB;
            if (need_fetch) { // next query word, consumed >= 4 accepted bases from now
                qnext = q32[min(((base + i) >> 2) + 1, last_dw)];
                need_fetch = false;
            }
        }

        // ---- consume
        if (st == ST_EXT) {
            const uint32_t l2 = rank_eval(xA, l - bl * kRankRows), r2 = rank_eval(xB, r - br * kRankRows);
            const bool valid = c < 4u;
            const bool ok = valid && l2 < r2;
            if (ok) {
                l = l2;
                r = r2;
                d = min(d + 1, k);
            } else if (!valid) { // no row ends with a non-ACGT char: contracts down to the root
                d = 0;
                l = 0;
                r = n;
            }
            if (ok || d == 0) {
                const uint64_t pos = base + i;
                if (i >= warm) {
                    obuf |= d << ((uint32_t)(pos & 3) * 8);
                    if (IVAL) {
                        a.lo_out[pos] = l;
                        a.hi_out[pos] = r;
                    }
                }
                i++;
                const bool fin = (i == len);
                if ((pos & 3) == 3 || fin) {
                    if (i > warm) {
                        const uint64_t w0 = pos & ~3ull;
                        const uint64_t first = max(w0, base + warm);
                        if (first == w0 && (pos & 3) == 3) d_out32[pos >> 2] = obuf;
                        else
                            for (uint64_t p = first; p <= pos; p++)
                                a.d_out[p] = (uint8_t)(obuf >> ((uint32_t)(p & 3) * 8));
                    }
                    obuf = 0;
                }
                if (fin) st = ST_WANT;
                else {
                    if (((pos + 1) & 3) == 0) {
                        qcur = qnext;
                        need_fetch = true;
                    }
                    c = decode_base((qcur >> ((uint32_t)((pos + 1) & 3) * 8)) & 0xFFu);
                }
            } else {
                st = ST_CON_NEW;
            }
        } else if (st <= ST_CON_SCAN) {
            if (st == ST_CON_NEW) { // LCS[0] = 0 and LCS[n] = 0 are stored sentinels
                m = max(byte_at(xA, l & 15u), byte_at(xB, r & 15u));
                d = m;
            }
            if (m == 0) {
                l = 0;
                r = n;
                st = ST_EXT;
            } else {
                const uint32_t ml = lt_mask16(xA, m), mr = lt_mask16(xB, m);
                const uint32_t cl = ml & ((2u << (l & 15u)) - 1u);
                const uint32_t cr = mr & ~((1u << (r & 15u)) - 1u);
                l = cl ? (l & ~15u) + (31u - (uint32_t)__clz((int)cl)) : (l & ~15u) - 1u;
                r = cr ? (r & ~15u) + ((uint32_t)__ffs((int)cr) - 1u) : (r & ~15u) + 16u;
                st = (cl && cr) ? ST_EXT : ST_CON_SCAN;
            }
        } else if (st == ST_ITEM) {
            base = (uint64_t)xA.x | ((uint64_t)xA.y << 32);
            len = xA.z;
            warm = xA.w;
            st = len ? ST_QLOAD : ST_WANT;
        } else if (st == ST_QLOAD) {
            const uint32_t sub = (uint32_t)(base >> 2) & 3u;
            qcur = sel4(xA, sub);
            qnext = sub == 3 ? xB.x : sel4(xA, sub + 1);
            c = decode_base((qcur >> ((uint32_t)(base & 3) * 8)) & 0xFFu);
            i = 0;
            l = 0;
            r = n;
            d = 0;
            obuf = 0;
            need_fetch = false;
            st = ST_EXT;
        }
    }
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

__global__ __launch_bounds__(256) void derand_translate_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t k,
    uint32_t t, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out, int32_t *__restrict__ derand_out)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    const uint64_t b = off[s], e = off[s + 1], len = e - b;
    if (len < 3) return; // the host side rejects these (derandomize.rs:276)
    const int K = (int)k, T = (int)t;
    const uint32_t *ms32 = reinterpret_cast<const uint32_t *>(ms);
    const uint32_t *ref32 = reinterpret_cast<const uint32_t *>(ref);
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);

    uint64_t p = e - 1;
    uint32_t mw = ms32[p >> 2];
    uint32_t rw = ref ? ref32[p >> 2] : 0;
    int a = (int)((mw >> ((uint32_t)(p & 3) * 8)) & 0xFFu);
    int x_cur = a > T ? a : 0, x_next = x_cur, x_prev = K;
    uint32_t obuf = 0;
    for (;;) {
        if (p > b) {
            const uint64_t q = p - 1;
            if ((q & 3) == 3) mw = ms32[q >> 2];
            a = (int)((mw >> ((uint32_t)(q & 3) * 8)) & 0xFFu);
            x_prev = (a == K) ? K : ((a > T && x_cur < a) ? a : x_cur - 1);
        }
        uint32_t ch = translate_char(x_prev, x_cur, x_next, p - b, len, K, T);
        if (ref) { // format::relative_to_ref: M,R keep the reference base, X and '-' become '-'
            const uint32_t rb = (rw >> ((uint32_t)(p & 3) * 8)) & 0xFFu;
            ch = (ch == 'M' || ch == 'R') ? rb : (uint32_t)'-';
        }
        obuf |= ch << ((uint32_t)(p & 3) * 8);
        if (derand_out) derand_out[p] = x_cur;
        if ((p & 3) == 0 || p == b) {
            const uint64_t hi = min(p | 3ull, e - 1);
            if ((p & 3) == 0 && hi == (p | 3ull)) out32[p >> 2] = obuf;
            else
                for (uint64_t w = p; w <= hi; w++) out[w] = (uint8_t)(obuf >> ((uint32_t)(w & 3) * 8));
            obuf = 0;
        }
        if (p == b) break;
        p--;
        if (ref && (p & 3) == 3) rw = ref32[p >> 2];
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

hipError_t launch_ms_walk(const WalkArgs &a, int blocks, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    if (a.lo_out && a.hi_out)
        hipLaunchKernelGGL(ms_walk_kernel<true>, dim3(blocks), dim3(kWalkThreads), 0, stream, a);
    else
        hipLaunchKernelGGL(ms_walk_kernel<false>, dim3(blocks), dim3(kWalkThreads), 0, stream, a);
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
