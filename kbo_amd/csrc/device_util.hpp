// device_util.hpp — device-side helpers shared by the kernel files: unaligned 16-byte accesses,
// partial block stores, and the two-level exclusive scan used for item lists and run lengths.
#pragma once
#include "kernels.hpp"

namespace kbo {
namespace {

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

// 16 ASCII bases -> 2-bit digits (first byte most significant) and a mask of the bytes that are A, C, G or T (bit t = byte t)
__device__ __forceinline__ void pack16(const uint4 &v, uint32_t &code, uint32_t &valid)
{
    code = 0;
    valid = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t x = w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w;
        uint32_t c2 = (x >> 1) & 0x03030303u;
        c2 ^= (x >> 2) & 0x01010101u;
        // the byte each code stands for, compared with the byte that is there
        uint32_t back = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) back |= ((0x54474341u >> (8u * ((c2 >> (8 * b)) & 3u))) & 0xFFu) << (8 * b);
        const uint32_t diff = back ^ x;
        const uint32_t nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u; // bit 7 of every byte that differs
        const uint32_t okb = (((nz >> 7) * 0x01020408u) >> 24) ^ 0xFu;                    // bits 0..3: bytes that are bases
        valid |= okb << (4 * w);
        const uint32_t d8 = ((c2 << 6) | (c2 >> 4) | (c2 >> 14) | (c2 >> 24)) & 0xFFu; // b0 b1 b2 b3 as 2-bit digits
        code = (code << 8) | d8;
    }
}

// the same for a block whose 16 bytes all count: the digits, and whether any byte is no base (a quarter of pack16's instructions:
// the letters the digits stand for come from one byte permute, the digits of four bytes are gathered by one multiplication, and
// no per-byte mask is made - map_kernels.hip spent 40 % of its vector instructions in pack16)
__device__ __forceinline__ void pack16_whole(const uint4 &v, uint32_t &code, bool &any_invalid)
{
    code = 0;
    uint32_t diff = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t x = w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w;
        uint32_t c2 = (x >> 1) & 0x03030303u;
        c2 ^= (x >> 2) & 0x01010101u;
        diff |= __builtin_amdgcn_perm(0u, 0x54474341u, c2) ^ x;      // "ACGT"[digit] against the byte that is there
        code = (code << 8) | ((c2 * 0x40100401u) >> 24);            // b0 b1 b2 b3 as 2-bit digits, b0 first
    }
    any_invalid = diff != 0;
}

// 16 ASCII bases -> 2-bit digits (first byte most significant); a byte that is no base gives some digit
__device__ __forceinline__ uint32_t digits16(const uint4 &v)
{
    uint32_t code = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const uint32_t x = w == 0 ? v.x : w == 1 ? v.y : w == 2 ? v.z : v.w;
        uint32_t c2 = (x >> 1) & 0x03030303u;
        c2 ^= (x >> 2) & 0x01010101u;
        code = (code << 8) | (((c2 << 6) | (c2 >> 4) | (c2 >> 14) | (c2 >> 24)) & 0xFFu);
    }
    return code;
}
// address of a depth-table entry in the grouped layout (dtab_kernels.hip): key = the window's bases, g = position mod 3
__device__ __forceinline__ uint64_t dtab_grouped_addr(uint64_t key, uint32_t g, uint32_t order)
{
    const uint32_t cb = 2u * (order - 2u); // bits of a core
    const uint64_t cm = (1ull << cb) - 1ull;
    if (g == 0) return ((key & cm) << 6) + (key >> cb);
    if (g == 1) return (((key >> 2) & cm) << 6) + 16u + ((key >> (cb + 2u)) << 2) + (key & 3u);
    return ((key >> 4) << 6) + 32u + (key & 15u);
}
__device__ __forceinline__ uint64_t dtab_grouped_addr32(uint32_t key, uint32_t g, uint32_t order) // (order <= 16: 32-bit keys)
{
    const uint32_t cb = 2u * (order - 2u), cm = (uint32_t)((1ull << cb) - 1ull);
    const uint32_t core = g == 0 ? key & cm : g == 1 ? (key >> 2) & cm : key >> 4;
    const uint32_t slot = g == 0 ? key >> cb : g == 1 ? 16u + ((key >> (cb + 2u)) << 2) + (key & 3u) : 32u + (key & 15u);
    return ((uint64_t)core << 6) + slot;
}
// The value of a base the depth table cannot tell (the string of dtab_order bases ending there is a suffix of a row, and so is
// the string with one more base): when those dtab_order bases are the suffix of exactly one row (an anchor), every longer
// suffix that is present is a suffix of THAT row, whose characters are the path-cover text in front of its position - read
// off as long as the text is one path (a 0 is a path start: unknown).  qb(t) = the base t positions in front of the one in
// question (qb(0) = itself), avail = how many there are (inside the item).  kDtabUnknown: no anchor / not one path.
constexpr uint32_t kDtabUnknown = 0xFFFFFFFFu;
template <typename GetBase>
__device__ __forceinline__ uint32_t dtab_anchor_depth(const DevIndexView &ix, uint32_t avail, GetBase qb)
{
    const uint32_t order = ix.dtab_order, k = ix.k;
    if (!ix.anchor) return kDtabUnknown;
    uint64_t key = 0;
    for (uint32_t t = order; t-- > 0;) key = (key << 2) | (decode_base(qb(t)) & 3u); // (all of them are bases: the table said so)
    const uint64_t mask = ((uint64_t)1 << ix.anchor_bits) - 1ull;
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64u - ix.anchor_bits);
    const uint32_t tag = (uint32_t)key + 1u;
    uint32_t p = 0;
    bool found = false;
    for (uint32_t probe = 0; probe < 64u; probe++) {
        const uint64_t slot = ix.anchor[h];
        if (slot == 0) break;
        if ((uint32_t)(slot >> 32) == tag) {
            p = (uint32_t)slot;
            found = true;
            break;
        }
        h = (h + 1u) & mask;
    }
    if (!found) return kDtabUnknown;
    // the text in front of p, 16 bytes at a time (kPlanPad zero bytes in front of the text: a 0 ends it as a path start does)
    for (uint32_t t0 = 0; t0 < k; t0 += 16u) {
        uint4 tv = make_uint4(0, 0, 0, 0);
        if (t0 + 15u <= p + kPlanPad) __builtin_memcpy(&tv, ix.pc_text + (int64_t)p - (int64_t)(t0 + 15u), 16); // text[p-t0-15 .. p-t0]
#pragma unroll
        for (uint32_t b = 0; b < 16; b++) {
            const uint32_t t = t0 + b;
            if (t >= k) return k;
            if (t >= avail) return t;
            const uint32_t qc = qb(t);
            if (decode_base(qc) >= 4u) return t;
            const uint32_t w = (15u - b) >> 2, sh = ((15u - b) & 3u) * 8u; // text[p - t] is byte 15 - b of the block
            const uint32_t tc = ((w == 0 ? tv.x : w == 1 ? tv.y : w == 2 ? tv.z : tv.w) >> sh) & 0xFFu;
            if (tc == 0) return kDtabUnknown;
            if (tc != qc) return t >= order ? t : kDtabUnknown; // (in front of `order`: two keys with the same 32-bit tag)
        }
    }
    return k;
}

// sum of v over the 64 lanes of the wave (all lanes must call it)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// adds the wave's totals of up to four work counters to the launch's statistics (kernels.hpp kPlanStat*): one atomic
// per counter and wave, on the slot the wave's number selects
__device__ __forceinline__ void plan_stats_add(uint32_t *pstats, uint32_t i0, uint32_t v0, uint32_t i1, uint32_t v1, uint32_t i2, uint32_t v2,
                                               uint32_t i3, uint32_t v3)
{
    v0 = wave_sum(v0);
    v1 = wave_sum(v1);
    v2 = wave_sum(v2);
    v3 = wave_sum(v3);
    if ((threadIdx.x & 63u) == 0 && pstats) {
        uint32_t *s = pstats + (((blockIdx.x * blockDim.x + threadIdx.x) >> 6) % kPlanStatSlots) * kPlanStatWords;
        if (v0) atomicAdd(s + i0, v0);
        if (v1) atomicAdd(s + i1, v1);
        if (v2) atomicAdd(s + i2, v2);
        if (v3) atomicAdd(s + i3, v3);
    }
}

constexpr uint32_t kScanBlock = 1024; // values per block of the two-level scan

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


} // namespace

// two-level exclusive scan of n u32 in place (n <= 2^28): data[i] becomes the prefix inside its block of
// kScanBlock values, sums[b] the prefix of block b; value = sums[i / kScanBlock] + data[i]
inline hipError_t launch_scan(uint32_t *data, uint32_t n, uint32_t *sums, hipStream_t stream)
{
    const uint32_t nb = (n + kScanBlock - 1) / kScanBlock;
    hipLaunchKernelGGL(scan_kernel, dim3(nb), dim3(256), 0, stream, data, n, kScanBlock / 256, sums);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, stream, sums, nb, (nb + 1023) / 1024, (uint32_t *)nullptr);
    return hipGetLastError();
}

} // namespace kbo
