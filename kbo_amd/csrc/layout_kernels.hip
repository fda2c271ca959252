// layout_kernels.hip — gfx950 (MI355X, CDNA4): what every walk of a device copy reads - rank blocks, contraction entries, two-base blocks
// (sbwt_index.hpp has the formats) - made ON THE DEVICE from the index's own arrays (four row bit-vectors, the LCS bytes: 1.5 bytes a
// row over PCIe instead of 12.7).  The host's make_device_layout is one thread's work - a monotone stack over 3 * 10^9 LCS values and a
// bit-by-bit gather of the blocks: 40 s of a 54 s device copy at C5's size, 4.5 of 5.8 s at C4 - and stays as the check
// (kbo_index_layout_check) and the fall-back (KBO_DEVICE_LAYOUT=0).
//
//   rank_bits_kernel    per character and block of 96 rows: the row bits out of the 64-bit words, their count
//   rank_cum_kernel     behind a scan of the counts: C[c] + rank_c(96 b) into the block's first word
//   lcs_min_kernel      minima of the LCS values per 1 024 rows, then per 1 024 of those
//   ent_kernel          per row { lcs, psv, nsv }: the nearest smaller value to either side - a scan over the neighbours (two steps on
//                       average: a value's neighbour is smaller every other time), then, for the few rows whose answer lies further out
//                       (one row in 4^j looks 4^j rows far), over the minima of blocks, of blocks of blocks, and back down
//   pair_bits_kernel    two-base blocks: bit i of pair (c1, c2) = B_c1[i] & B_c2[C[c1] + rank_c1(i)], a lane per 32 rows (the rank
//   pair_cum_kernel     blocks are there by then), their counts scanned like the rank blocks'
//
// Integer / bit work only: no MFMA.
#include "device_util.hpp"

#include <algorithm>

namespace kbo {
namespace {

constexpr uint32_t kMinBlock = 1024;

__device__ __forceinline__ uint32_t bits32(const uint64_t *__restrict__ w, uint64_t n_words, uint64_t bit0)
{ // 32 bits from bit `bit0` on (bits beyond the last word are 0)
    const uint64_t wi = bit0 >> 6;
    const uint32_t sh = (uint32_t)(bit0 & 63u);
    uint64_t lo = wi < n_words ? w[wi] : 0ull;
    uint64_t v = lo >> sh;
    if (sh > 32u) {
        const uint64_t hi = wi + 1u < n_words ? w[wi + 1u] : 0ull;
        v |= hi << (64u - sh);
    }
    return (uint32_t)v;
}

__global__ __launch_bounds__(256) void rank_bits_kernel(const uint64_t *__restrict__ rows, uint64_t n_words, uint64_t n, uint32_t n_blocks,
                                                        uint4 *__restrict__ blk, uint32_t *__restrict__ counts)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    uint32_t w[3];
#pragma unroll
    for (uint32_t t = 0; t < 3u; t++) {
        const uint64_t base = (uint64_t)b * 96u + 32u * t;
        uint32_t v = base < n ? bits32(rows, n_words, base) : 0u;
        if (base < n && n - base < 32u) v &= (1u << (uint32_t)(n - base)) - 1u; // (bits beyond row n - 1: none)
        w[t] = v;
    }
    blk[b] = make_uint4(0u, w[0], w[1], w[2]);
    counts[b] = (uint32_t)(__popc(w[0]) + __popc(w[1]) + __popc(w[2]));
}

__global__ __launch_bounds__(256) void rank_cum_kernel(uint4 *__restrict__ blk, const uint32_t *__restrict__ counts, const uint32_t *__restrict__ sums,
                                                       uint32_t n_blocks, uint32_t base)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    reinterpret_cast<uint32_t *>(blk + b)[0] = base + sums[b / kScanBlock] + counts[b];
}

// level 1: out[j] = min of lcs[1024 j .. +1024) (rows at and beyond n count as 0: the sentinel); level 2: the same over level 1
__global__ __launch_bounds__(256) void lcs_min_kernel(const uint8_t *__restrict__ lcs, uint64_t n, uint8_t *__restrict__ out, uint64_t n_out)
{
    __shared__ uint32_t sh[4];
    const uint64_t j = blockIdx.x;
    if (j >= n_out) return;
    uint32_t m = 255u;
    for (uint32_t t = threadIdx.x; t < kMinBlock; t += 256u) {
        const uint64_t i = j * kMinBlock + t;
        m = min(m, i < n ? (uint32_t)lcs[i] : 0u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[j] = (uint8_t)min(min(sh[0], sh[1]), min(sh[2], sh[3]));
}
__global__ __launch_bounds__(256) void min_of_min_kernel(const uint8_t *__restrict__ in, uint64_t n_in, uint8_t *__restrict__ out, uint64_t n_out)
{
    __shared__ uint32_t sh[4];
    const uint64_t j = blockIdx.x;
    if (j >= n_out) return;
    uint32_t m = 255u;
    for (uint32_t t = threadIdx.x; t < kMinBlock; t += 256u) {
        const uint64_t i = j * kMinBlock + t;
        if (i < n_in) m = min(m, (uint32_t)in[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[j] = (uint8_t)min(min(sh[0], sh[1]), min(sh[2], sh[3]));
}

// { lcs[i], psv[i], nsv[i] } for rows 0 .. n (the sentinel row n has LCS 0): psv = the largest j < i with lcs[j] < lcs[i] (0 when there is
// none), nsv = the smallest j > i with lcs[j] < lcs[i], the sentinel included (n when there is none) - make_device_layout's monotone stacks
__global__ __launch_bounds__(256) void ent_kernel(const uint8_t *__restrict__ lcs, uint64_t n, const uint8_t *__restrict__ m1, uint64_t n1,
                                                  const uint8_t *__restrict__ m2, uint64_t n2, uint32_t *__restrict__ ent)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    auto at = [&](uint64_t x) -> uint32_t { return x < n ? (uint32_t)lcs[x] : 0u; };
    const uint32_t v = at(i);
    uint64_t psv = 0, nsv = n;
    if (v != 0u) {
        // ---- to the left: inside the row's block of 1024, then blocks by their minima, then blocks of blocks
        bool found = false;
        const uint64_t b1 = i / kMinBlock;
        for (uint64_t j = i; j-- > b1 * kMinBlock;)
            if (at(j) < v) { psv = j; found = true; break; }
        if (!found && b1 > 0) {
            uint64_t blk = ~0ull; // the block of 1024 rows that holds the answer
            const uint64_t b2 = b1 / kMinBlock;
            for (uint64_t j = b1; j-- > b2 * kMinBlock;)
                if ((uint32_t)m1[j] < v) { blk = j; break; }
            if (blk == ~0ull && b2 > 0) {
                uint64_t sb = ~0ull;
                for (uint64_t j = b2; j-- > 0;)
                    if ((uint32_t)m2[j] < v) { sb = j; break; }
                if (sb != ~0ull)
                    for (uint64_t j = min((sb + 1u) * kMinBlock, n1); j-- > sb * kMinBlock;)
                        if ((uint32_t)m1[j] < v) { blk = j; break; }
            }
            if (blk != ~0ull)
                for (uint64_t j = min((blk + 1u) * kMinBlock, n + 1u); j-- > blk * kMinBlock;)
                    if (at(j) < v) { psv = j; found = true; break; }
        }
        (void)found; // (row 0 has LCS 0: a row with v > 0 always finds one; psv stays 0 otherwise, as the host's)
        // ---- to the right (the sentinel row n, LCS 0, ends every search)
        found = false;
        const uint64_t e1 = min((b1 + 1u) * kMinBlock, n + 1u);
        for (uint64_t j = i + 1u; j < e1; j++)
            if (at(j) < v) { nsv = j; found = true; break; }
        if (!found) {
            uint64_t blk = ~0ull;
            const uint64_t b2 = b1 / kMinBlock, e2 = min((b2 + 1u) * kMinBlock, n1);
            for (uint64_t j = b1 + 1u; j < e2; j++)
                if ((uint32_t)m1[j] < v) { blk = j; break; }
            if (blk == ~0ull) {
                uint64_t sb = ~0ull;
                for (uint64_t j = b2 + 1u; j < n2; j++)
                    if ((uint32_t)m2[j] < v) { sb = j; break; }
                if (sb != ~0ull)
                    for (uint64_t j = sb * kMinBlock; j < min((sb + 1u) * kMinBlock, n1); j++)
                        if ((uint32_t)m1[j] < v) { blk = j; break; }
            }
            if (blk != ~0ull)
                for (uint64_t j = blk * kMinBlock; j < min((blk + 1u) * kMinBlock, n + 1u); j++)
                    if (at(j) < v) { nsv = j; break; }
        }
    }
    uint32_t *o = ent + 3u * i;
    o[0] = v;
    o[1] = (uint32_t)psv;
    o[2] = (uint32_t)nsv;
}

// two-base blocks: a lane per 32 rows [32 t, 32 t + 32) - word t % 3 of block t / 3 of all sixteen pairs
__global__ __launch_bounds__(256) void pair_bits_kernel(const uint64_t *__restrict__ r0, const uint64_t *__restrict__ r1, const uint64_t *__restrict__ r2,
                                                        const uint64_t *__restrict__ r3, uint64_t n_words, uint64_t n, uint32_t n_blocks,
                                                        const uint4 *__restrict__ rank, uint4 *__restrict__ pair)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_blocks * 3u) return;
    const uint64_t *rows[4] = {r0, r1, r2, r3};
    const uint64_t base = t * 32u;
    auto bit_at = [&](uint32_t c, uint64_t x) -> uint32_t { return x < n ? (uint32_t)((rows[c][x >> 6] >> (x & 63u)) & 1ull) : 0u; };
    uint32_t out[16];
#pragma unroll
    for (uint32_t p = 0; p < 16u; p++) out[p] = 0u;
    for (uint32_t c1 = 0; c1 < 4u; c1++) {
        uint32_t w = base < n ? bits32(rows[c1], n_words, base) : 0u;
        if (base < n && n - base < 32u) w &= (1u << (uint32_t)(n - base)) - 1u;
        for (uint32_t m = w; m; m &= m - 1u) {
            const uint32_t j = (uint32_t)__builtin_ctz(m);
            const uint64_t i = base + j;
            const uint32_t b = (uint32_t)(i / 96u);
            const uint64_t target = rank_eval(rank[(uint64_t)c1 * n_blocks + b], (uint32_t)(i - (uint64_t)b * 96u)); // C[c1] + rank_c1(i)
#pragma unroll
            for (uint32_t c2 = 0; c2 < 4u; c2++)
                if (bit_at(c2, target)) out[4u * c1 + c2] |= 1u << j;
        }
    }
    const uint32_t b = (uint32_t)(t / 3u), wsel = (uint32_t)(t % 3u);
#pragma unroll
    for (uint32_t p = 0; p < 16u; p++) reinterpret_cast<uint32_t *>(pair + (uint64_t)p * n_blocks + b)[1u + wsel] = out[p];
}

__global__ __launch_bounds__(256) void pair_count_kernel(const uint4 *__restrict__ pair, uint32_t n_blocks, uint32_t *__restrict__ counts)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_blocks) return;
    const uint4 v = pair[b];
    counts[b] = (uint32_t)(__popc(v.y) + __popc(v.z) + __popc(v.w));
}

} // namespace

size_t device_layout_scratch_bytes(uint64_t n_sets)
{
    const uint64_t n_blocks = n_sets / 96u + 2u, n1 = (n_sets + 1u + kMinBlock - 1u) / kMinBlock, n2 = (n1 + kMinBlock - 1u) / kMinBlock;
    return (size_t)(n_blocks * 4u + (n_blocks / kScanBlock + 2u) * 4u + n1 + n2 + 256u);
}

// d_rows[c]: the index's row bit-vectors (64-bit words, n_words each), d_lcs: n bytes; d_rank: 4 * n_blocks blocks (character c at
// c * n_blocks), d_ent: 3 * (n + 1) words, d_pair: 16 * n_blocks blocks or nullptr; d_scratch: device_layout_scratch_bytes(n).  C[c] as the
// index has them.  Synchronises the stream.
hipError_t build_device_layout(const uint64_t *const d_rows[4], uint64_t n_words, const uint8_t *d_lcs, uint64_t n, const uint64_t C[4],
                               uint32_t n_blocks, uint4 *d_rank, uint32_t *d_ent, uint4 *d_pair, void *d_scratch, hipStream_t stream)
{
    uint32_t *counts = static_cast<uint32_t *>(d_scratch), *sums = counts + n_blocks;
    const uint64_t n1 = (n + 1u + kMinBlock - 1u) / kMinBlock, n2 = (n1 + kMinBlock - 1u) / kMinBlock;
    uint8_t *m1 = reinterpret_cast<uint8_t *>(sums + n_blocks / kScanBlock + 2u), *m2 = m1 + n1;
    const dim3 blk256(256), grid_b((n_blocks + 255u) / 256u);
    hipError_t e;
    for (uint32_t c = 0; c < 4u; c++) {
        hipLaunchKernelGGL(rank_bits_kernel, grid_b, blk256, 0, stream, d_rows[c], n_words, n, n_blocks, d_rank + (uint64_t)c * n_blocks, counts);
        e = launch_scan(counts, n_blocks, sums, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(rank_cum_kernel, grid_b, blk256, 0, stream, d_rank + (uint64_t)c * n_blocks, counts, sums, n_blocks, (uint32_t)C[c]);
    }
    hipLaunchKernelGGL(lcs_min_kernel, dim3((uint32_t)n1), blk256, 0, stream, d_lcs, n, m1, n1);
    hipLaunchKernelGGL(min_of_min_kernel, dim3((uint32_t)n2), blk256, 0, stream, m1, n1, m2, n2);
    hipLaunchKernelGGL(ent_kernel, dim3((uint32_t)((n + 1u + 255u) / 256u)), blk256, 0, stream, d_lcs, n, m1, n1, m2, n2, d_ent);
    if (d_pair) {
        e = hipMemsetAsync(d_pair, 0, (size_t)16u * n_blocks * 16u, stream);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(pair_bits_kernel, dim3((uint32_t)(((uint64_t)n_blocks * 3u + 255u) / 256u)), blk256, 0, stream, d_rows[0], d_rows[1], d_rows[2],
                           d_rows[3], n_words, n, n_blocks, d_rank, d_pair);
        for (uint32_t p = 0; p < 16u; p++) {
            const uint32_t c1 = p / 4u, c2 = p % 4u;
            uint4 *pb = d_pair + (uint64_t)p * n_blocks;
            hipLaunchKernelGGL(pair_count_kernel, grid_b, blk256, 0, stream, pb, n_blocks, counts);
            e = launch_scan(counts, n_blocks, sums, stream);
            if (e != hipSuccess) return e;
            // base count: C[c2] + rank_c2(C[c1]) - one look-up in the finished rank blocks, on the host side of the launch
            const uint64_t at = C[c1];
            uint4 rb;
            e = hipMemcpyAsync(&rb, d_rank + (uint64_t)c2 * n_blocks + at / 96u, 16, hipMemcpyDeviceToHost, stream);
            if (e != hipSuccess) return e;
            e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            const uint32_t o = (uint32_t)(at % 96u);
            uint32_t cnt = rb.x;
            const uint32_t ws[3] = {rb.y, rb.z, rb.w};
            for (uint32_t w = 0; w < 3u; w++) {
                const uint32_t lo = 32u * w;
                if (o > lo) cnt += (uint32_t)__builtin_popcount(o - lo >= 32u ? ws[w] : ws[w] & ((1u << (o - lo)) - 1u));
            }
            hipLaunchKernelGGL(rank_cum_kernel, grid_b, blk256, 0, stream, pb, counts, sums, n_blocks, cnt);
        }
    }
    e = hipStreamSynchronize(stream);
    return e != hipSuccess ? e : hipGetLastError();
}

} // namespace kbo
