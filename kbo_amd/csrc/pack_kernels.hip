// pack_kernels.hip — gfx950 (MI355X, CDNA4): 2-bit packed reads in, 2-bit packed alignments out, for the packed host
// entry points (kbo_matches_batch_packed / kbo_find_batch_packed).  The host -> host rate of the batch entry points is
// bound by PCIe at one byte per base each way; reads are 2 bits of information per base and the
// alphabet of kbo::matches is exactly { M, -, X, R } (translate.rs:180-216), so a quarter of the bytes carry the same.
//
// Packed layout (input and output alike): sequence s of len_s bases occupies ceil(len_s / 16) little-endian u32 words
// starting at word sum_{r<s} ceil(len_r / 16); base i of the sequence sits in bits 2 (i mod 16) .. +1 of word i / 16.
// Input codes A, C, G, T = 0 .. 3 (non-ACGT bytes travel in a side list and are written over the unpacked base);
// output codes M, -, X, R = 0 .. 3.  One lane per word: coalesced 4-byte accesses on the packed side, 16-byte blocks of
// one sequence on the byte side.
#include "device_util.hpp"

namespace kbo {
namespace {

// words per sequence (for the scan that gives ragged batches their word prefix)
__global__ __launch_bounds__(256) void seq_words_kernel(const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t *__restrict__ words)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seqs) return;
    words[s] = s < n_seqs ? (uint32_t)((off[s + 1] - off[s] + 15u) / 16u) : 0u;
}

// offsets of a batch of equally long sequences, made on the device (nothing to upload)
__global__ __launch_bounds__(256) void uniform_offsets_kernel(uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t len)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s <= n_seqs) off[s] = (uint64_t)s * len;
}

// word w of the packed batch -> (sequence, block of 16 bases inside it).  uniform_wps != 0: every sequence has that many
// words; otherwise binary search over the scanned word counts (prefix(e) = sums[e / kScanBlock] + data[e])
__device__ __forceinline__ void locate_word(uint32_t w, uint32_t n_seqs, uint32_t uniform_wps, const uint32_t *data, const uint32_t *sums,
                                            uint32_t &seq, uint32_t &blk)
{
    if (uniform_wps) {
        seq = w / uniform_wps;
        blk = w - seq * uniform_wps;
        return;
    }
    uint32_t s0 = 0, s1 = n_seqs; // largest s with prefix(s) <= w
    while (s1 - s0 > 1) {
        const uint32_t m = s0 + (s1 - s0) / 2;
        if (sums[m / kScanBlock] + data[m] <= w) s0 = m;
        else s1 = m;
    }
    seq = s0;
    blk = w - (sums[s0 / kScanBlock] + data[s0]);
}

// stores bytes [lo, hi) of a 16-byte block to o + lo .. o + hi (0 <= lo <= hi <= 16)
__device__ __forceinline__ void st_bytes(uint8_t *o, const uint4 &v, uint32_t lo, uint32_t hi)
{
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) { // (compile-time byte positions: no indexed copy of v in scratch)
        const uint32_t w = (t >> 2) == 0 ? v.x : (t >> 2) == 1 ? v.y : (t >> 2) == 2 ? v.z : v.w;
        if (t >= lo && t < hi) o[t] = (uint8_t)(w >> ((t & 3u) * 8u));
    }
}

// One lane per packed word.  The 64 words of a wave unpack to one contiguous stretch of the byte buffer (sequences lie
// back to back), but at a 150-byte stride nothing in it is 16-byte aligned: lanes storing their own blocks put misaligned
// and partial stores into every instruction (first version: 187 us per 75 Mbp, 0.4 TB/s).  So the stretch is put together
// in LDS and leaves in aligned 16-byte blocks, bytes only at its two ends (the plan kernel's way of writing its output).
__global__ __launch_bounds__(256) void unpack2_kernel(const uint32_t *__restrict__ packed, uint32_t n_words, const uint64_t *__restrict__ off,
                                                      uint32_t n_seqs, uint32_t uniform_wps, const uint32_t *__restrict__ data,
                                                      const uint32_t *__restrict__ sums, uint8_t *__restrict__ q)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][1024 + 48];
    uint8_t *sm = lds[threadIdx.x >> 6];
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    const bool active = w < n_words;
    uint64_t out = 0;
    uint32_t nb = 0;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (active) {
        uint32_t seq, blk;
        locate_word(w, n_seqs, uniform_wps, data, sums, seq, blk);
        const uint32_t v = packed[w];
        const uint64_t b0 = off[seq], len = off[seq + 1] - b0;
        const uint32_t first = blk * 16u;
        nb = (uint32_t)min((uint64_t)16u, len - first);
        out = b0 + first;
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const uint32_t ch = (0x54474341u >> (8u * ((v >> (2 * t)) & 3u))) & 0xFFu; // "ACGT"[code]
            const uint32_t sh = ch << ((t & 3) * 8);
            if ((t >> 2) == 0) o.x |= sh;
            else if ((t >> 2) == 1) o.y |= sh;
            else if ((t >> 2) == 2) o.z |= sh;
            else o.w |= sh;
        }
    }
    const uint64_t am = __ballot(active);
    if (am == 0) return;
    const int last = 63 - (int)__builtin_clzll(am);
    const uint32_t out_lo32 = (uint32_t)out, out_hi32 = (uint32_t)(out >> 32);
    const uint64_t lo = ((uint64_t)__shfl(out_hi32, 0) << 32) | __shfl(out_lo32, 0);
    const uint64_t hi = (((uint64_t)__shfl(out_hi32, last) << 32) | __shfl(out_lo32, last)) + __shfl(nb, last);
    const uint64_t base16 = lo & ~15ull;
    if (active) {
        uint8_t *d = sm + (uint32_t)(out - base16);
        if (nb == 16u) __builtin_memcpy(d, &o, 16);
        else st_bytes(d, o, 0, nb); // (a sequence's last block: the bytes behind it are the next lane's)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t span = (uint32_t)(hi - base16);
    for (uint32_t c = lane * 16u; c < span; c += 1024u) {
        const uint4 v = *reinterpret_cast<const uint4 *>(sm + c);
        const uint64_t g0 = base16 + c;
        if (g0 >= lo && g0 + 16u <= hi) *reinterpret_cast<uint4 *>(q + g0) = v;
        else st_bytes(q + g0, v, lo > g0 ? (uint32_t)(lo - g0) : 0u, (uint32_t)min((uint64_t)16u, hi - g0));
    }
}

// non-ACGT bytes of the input, written over the unpacked bases: pos = offset of the base in the (slab's) byte buffer
__global__ __launch_bounds__(256) void exceptions_kernel(const uint64_t *__restrict__ pos, const uint8_t *__restrict__ byte, uint32_t n,
                                                         uint64_t base, uint8_t *__restrict__ q)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x < n) q[pos[x] - base] = byte[x];
}

__global__ __launch_bounds__(256) void pack2_kernel(const uint8_t *__restrict__ chars, uint32_t n_words, const uint64_t *__restrict__ off,
                                                    uint32_t n_seqs, uint32_t uniform_wps, const uint32_t *__restrict__ data,
                                                    const uint32_t *__restrict__ sums, uint32_t *__restrict__ packed)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint32_t seq, blk;
    locate_word(w, n_seqs, uniform_wps, data, sums, seq, blk);
    const uint64_t b0 = off[seq], len = off[seq + 1] - b0;
    const uint32_t first = blk * 16u, nb = (uint32_t)min((uint64_t)16u, len - first);
    uint4 v;
    __builtin_memcpy(&v, chars + b0 + first, 16); // (reads up to 15 bytes past the sequence: inside the buffer's slack)
    uint32_t out = 0;
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const uint32_t word = (t >> 2) == 0 ? v.x : (t >> 2) == 1 ? v.y : (t >> 2) == 2 ? v.z : v.w;
        const uint32_t ch = (word >> ((t & 3) * 8)) & 0xFFu;
        const uint32_t code = ch == 'M' ? 0u : ch == '-' ? 1u : ch == 'X' ? 2u : 3u; // translate.rs:180-216: M, -, X, R
        out |= ((uint32_t)t < nb ? code : 0u) << (2 * t);
    }
    packed[w] = out;
}

// ---- packed-native batches (map_kernels.hip, IO != 0): what the few reads that take the plain walk need
// exc[s] = 1 for the read that holds listed byte x (pos = its offset in the byte layout of the slab)
__global__ __launch_bounds__(256) void flag_exceptions_kernel(const uint64_t *__restrict__ pos, uint32_t n, uint64_t base,
                                                              const uint64_t *__restrict__ off, uint32_t n_seqs, uint8_t *__restrict__ exc)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const uint64_t p = pos[x] - base;
    uint32_t s0 = 0, s1 = n_seqs; // largest s with off[s] <= p
    while (s1 - s0 > 1) {
        const uint32_t m = s0 + (s1 - s0) / 2;
        if (off[m] <= p) s0 = m;
        else s1 = m;
    }
    exc[s0] = 1;
}

__device__ __forceinline__ uint32_t first_word_of(uint32_t s, uint32_t uniform_wps, const uint32_t *data, const uint32_t *sums)
{
    return uniform_wps ? s * uniform_wps : sums[s / kScanBlock] + data[s];
}

// one lane per read looks at its flag; the wave then does its flagged reads one after the other, a lane per base (two reads
// in a hundred come here: a lane walking its own read alone was 40 us, sixteen lanes per read over all reads 30)
__global__ __launch_bounds__(256) void unpack_flagged_kernel(const uint32_t *__restrict__ packed, const uint64_t *__restrict__ off, uint32_t n_seqs,
                                                             uint32_t uniform_wps, const uint32_t *__restrict__ data, const uint32_t *__restrict__ sums,
                                                             const uint8_t *__restrict__ flags, uint8_t *__restrict__ q)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint64_t fm = __ballot(s < n_seqs && flags[s] != 0);
    while (fm) {
        const uint32_t r = s - lane + (uint32_t)__ffsll((long long)fm) - 1u;
        fm &= fm - 1ull;
        const uint64_t b0 = off[r];
        const uint32_t len = (uint32_t)(off[r + 1] - b0), w0 = first_word_of(r, uniform_wps, data, sums);
        for (uint32_t i = lane; i < len; i += 64u)
            q[b0 + i] = (uint8_t)((0x54474341u >> (8u * ((packed[w0 + (i >> 4)] >> (2u * (i & 15u))) & 3u))) & 0xFFu);
    }
}

__global__ __launch_bounds__(256) void pack_flagged_kernel(const uint8_t *__restrict__ chars, const uint64_t *__restrict__ off, uint32_t n_seqs,
                                                           uint32_t uniform_wps, const uint32_t *__restrict__ data, const uint32_t *__restrict__ sums,
                                                           const uint8_t *__restrict__ flags, uint32_t *__restrict__ packed)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint64_t fm = __ballot(s < n_seqs && flags[s] != 0);
    while (fm) {
        const uint32_t r = s - lane + (uint32_t)__ffsll((long long)fm) - 1u;
        fm &= fm - 1ull;
        const uint64_t b0 = off[r];
        const uint32_t len = (uint32_t)(off[r + 1] - b0), w0 = first_word_of(r, uniform_wps, data, sums);
        for (uint32_t i0 = 0; i0 < len; i0 += 64u) { // 64 bases = four words: sixteen lanes put a word together
            const uint32_t i = i0 + lane;
            const uint32_t ch = i < len ? chars[b0 + i] : (uint32_t)'M';
            uint32_t v = (ch == 'M' ? 0u : ch == '-' ? 1u : ch == 'X' ? 2u : 3u) << (2u * (lane & 15u)); // translate.rs:180-216: M, -, X, R
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v |= __shfl_xor(v, o);
            if ((lane & 15u) == 0 && i < len) packed[w0 + (i >> 4)] = v;
        }
    }
}

// per-byte unsigned maximum of two words
__device__ __forceinline__ uint32_t max4(uint32_t x, uint32_t y)
{
    const uint32_t H = 0x80808080u;
    // bit 7 of every byte of ge: x >= y there (the top bits decide, else the low seven)
    const uint32_t lo = ((x | H) - (y & ~H)) & H; // low seven bits: x >= y
    const uint32_t ge = ((x & ~y) | (~(x ^ y) & lo)) & H;
    const uint32_t mask = (ge >> 7) * 0xFFu;
    return (x & mask) | (y & ~mask);
}

// a = max(a, b) byte by byte, 16 bytes per lane
__global__ __launch_bounds__(256) void max_bytes_kernel(uint4 *__restrict__ a, const uint4 *__restrict__ b, uint64_t n16)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    uint4 x = a[i];
    const uint4 y = b[i];
    x.x = max4(x.x, y.x);
    x.y = max4(x.y, y.y);
    x.z = max4(x.z, y.z);
    x.w = max4(x.w, y.w);
    a[i] = x;
}

} // namespace

hipError_t launch_max_bytes(uint8_t *d_a, const uint8_t *d_b, uint64_t n, hipStream_t stream)
{
    const uint64_t n16 = (n + 15) / 16;
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(max_bytes_kernel, dim3((uint32_t)((n16 + 255) / 256)), dim3(256), 0, stream, reinterpret_cast<uint4 *>(d_a),
                       reinterpret_cast<const uint4 *>(d_b), n16);
    return hipGetLastError();
}

hipError_t launch_uniform_offsets(uint64_t *d_off, uint32_t n_seqs, uint32_t len, hipStream_t stream)
{
    hipLaunchKernelGGL(uniform_offsets_kernel, dim3((n_seqs + 256u) / 256u), dim3(256), 0, stream, d_off, n_seqs, len);
    return hipGetLastError();
}

// d_scratch: chunk_items_scratch_words(n_seqs) u32 (ragged batches only): the scanned words-per-sequence
hipError_t launch_packed_prefix(const uint64_t *d_off, uint32_t n_seqs, uint32_t *d_scratch, hipStream_t stream)
{
    hipLaunchKernelGGL(seq_words_kernel, dim3((n_seqs + 256u) / 256u), dim3(256), 0, stream, d_off, n_seqs, d_scratch);
    return launch_scan(d_scratch, n_seqs + 1u, d_scratch + n_seqs + 1u, stream);
}

hipError_t launch_unpack2(const uint32_t *d_packed, uint32_t n_words, const uint64_t *d_off, uint32_t n_seqs, uint32_t uniform_wps,
                          const uint32_t *d_scratch, uint8_t *d_q, hipStream_t stream)
{
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack2_kernel, dim3((n_words + 255u) / 256u), dim3(256), 0, stream, d_packed, n_words, d_off, n_seqs, uniform_wps,
                       d_scratch, d_scratch ? d_scratch + n_seqs + 1u : nullptr, d_q);
    return hipGetLastError();
}

hipError_t launch_exceptions(const uint64_t *d_pos, const uint8_t *d_byte, uint32_t n, uint64_t base, uint8_t *d_q, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(exceptions_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_pos, d_byte, n, base, d_q);
    return hipGetLastError();
}

hipError_t launch_flag_exceptions(const uint64_t *d_pos, uint32_t n, uint64_t base, const uint64_t *d_off, uint32_t n_seqs, uint8_t *d_exc, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(flag_exceptions_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_pos, n, base, d_off, n_seqs, d_exc);
    return hipGetLastError();
}

hipError_t launch_unpack_flagged(const uint32_t *d_packed, const uint64_t *d_off, uint32_t n_seqs, uint32_t uniform_wps, const uint32_t *d_scratch,
                                 const uint8_t *d_flags, uint8_t *d_q, hipStream_t stream)
{
    if (n_seqs == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_flagged_kernel, dim3((n_seqs + 255u) / 256u), dim3(256), 0, stream, d_packed, d_off, n_seqs,
                       uniform_wps, d_scratch, d_scratch ? d_scratch + n_seqs + 1u : nullptr, d_flags, d_q);
    return hipGetLastError();
}

hipError_t launch_pack_flagged(const uint8_t *d_chars, const uint64_t *d_off, uint32_t n_seqs, uint32_t uniform_wps, const uint32_t *d_scratch,
                               const uint8_t *d_flags, uint32_t *d_packed, hipStream_t stream)
{
    if (n_seqs == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_flagged_kernel, dim3((n_seqs + 255u) / 256u), dim3(256), 0, stream, d_chars, d_off, n_seqs,
                       uniform_wps, d_scratch, d_scratch ? d_scratch + n_seqs + 1u : nullptr, d_flags, d_packed);
    return hipGetLastError();
}

hipError_t launch_pack2(const uint8_t *d_chars, uint32_t n_words, const uint64_t *d_off, uint32_t n_seqs, uint32_t uniform_wps,
                        const uint32_t *d_scratch, uint32_t *d_packed, hipStream_t stream)
{
    if (n_words == 0) return hipSuccess;
    hipLaunchKernelGGL(pack2_kernel, dim3((n_words + 255u) / 256u), dim3(256), 0, stream, d_chars, n_words, d_off, n_seqs, uniform_wps,
                       d_scratch, d_scratch ? d_scratch + n_seqs + 1u : nullptr, d_packed);
    return hipGetLastError();
}

} // namespace kbo
