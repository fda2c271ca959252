// call_emit_kernels.hip — gfx950 (MI355X, CDNA4): the tail of kbo::call (variant_calling.rs:271-291) for the sites of a slab, on the
// device: the variants leave it as they are returned - in the order of (sequence, query position), as flat arrays - instead of a
// 16-byte record, a word and a 2 k + 16-byte window per SITE that host threads sorted, resolved and sliced (round 5: 164 bytes per
// site over PCIe into freshly pinned memory, 0.18 s of a 0.34 s call of 60 000 reads of 10 kbp).
//
// Behind call_finalize_kernel (records {sequence, i, j, row} + windows) and call_depths_kernel (one word per site: the common suffix
// of the two k-mers and the two rightmost significant peaks - all resolve_variant, variant_calling.rs:139-201, reads):
//
//   call_prefix_kernel     the kCallSegs list counters -> their prefix (the sites' numbering) without a trip to the host; overflow /
//                          give-up flags of the walk's call mode for the host to read with everything else, once per slab
//   call_hist / bucket     the sites of every sequence together (counting sort by sequence: one atomic per site, twice)
//   call_rank_kernel       ... in the order of their query position i (the order variant_calling.rs:268-291 finds them in): a site's
//                          rank = the sites of its sequence with a smaller i, counted (a 10 kbp read has 70 of them; a sequence
//                          with more than kCallMaxRank sites is left to the host, which sorts)
//   call_resolve_kernel    resolve_variant's case analysis on the three numbers (refine.cpp resolve_variant_peaks is the host's
//                          copy): does the site give a variant, and which slices of the two k-mers are its characters.  What the
//                          reference would panic on (common suffix 0, a slice beyond the k-mer) and what call_depths_kernel left
//                          (bytes that are no bases, a row whose k-mer crosses a path start) goes to a list for the host
//   call_write_kernel      behind two scans: query_pos / lengths / characters of every variant at its place
//   call_host_gather       records + windows of the host's sites, compact
//
// Integer / byte work only: no MFMA.
#include "device_util.hpp"

#include <algorithm>

namespace kbo {
namespace {

__global__ __launch_bounds__(256) void call_prefix_kernel(const uint32_t *__restrict__ counts, uint32_t seg_cap, uint32_t *__restrict__ prefix,
                                                          uint32_t *__restrict__ meta)
{
    static_assert(kCallSegs == 256, "one thread per list");
    __shared__ uint32_t sh[256];
    __shared__ uint32_t over, worst;
    const uint32_t t = threadIdx.x;
    if (t == 0) over = 0, worst = 0;
    __syncthreads();
    const uint32_t raw = counts[t * 16u], c = min(raw, seg_cap);
    if (raw > seg_cap) atomicOr(&over, 1u);
    atomicMax(&worst, raw);
    sh[t] = c;
    __syncthreads();
    for (uint32_t step = 1; step < 256u; step <<= 1) {
        const uint32_t v = t >= step ? sh[t - step] : 0u;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    prefix[t] = sh[t] - c;
    if (t == 255u) {
        prefix[256] = sh[t];
        meta[kCallMetaSites] = sh[t];
        meta[kCallMetaFlags] = over | (counts[kCallSegs * 16u] ? 2u : 0u);
        meta[kCallMetaWorst] = worst;
    }
}

__device__ __forceinline__ uint32_t scanned(const uint32_t *data, const uint32_t *sums, uint32_t i) { return sums[i / kScanBlock] + data[i]; }

__global__ __launch_bounds__(256) void call_hist_kernel(CallEmitArgs a)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= *a.n_sites) return;
    const uint32_t seq = a.recs[x].x;
    if (seq != 0xFFFFFFFFu) atomicAdd(a.seq_cnt + seq, 1u);
}

__global__ __launch_bounds__(256) void call_bucket_kernel(CallEmitArgs a)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= *a.n_sites) return;
    const uint4 rec = a.recs[x];
    if (rec.x == 0xFFFFFFFFu) return;
    const uint32_t p = scanned(a.seq_cnt, a.seq_sums, rec.x) + atomicAdd(a.seq_fill + rec.x, 1u);
    a.bucket[p] = x;
    a.bkey[p] = rec.y;
}

__global__ __launch_bounds__(256) void call_rank_kernel(CallEmitArgs a)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_valid = scanned(a.seq_cnt, a.seq_sums, a.n_seqs);
    if (p >= n_valid) return;
    const uint32_t x = a.bucket[p], seq = a.recs[x].x, key = a.bkey[p];
    const uint32_t lo = scanned(a.seq_cnt, a.seq_sums, seq), hi = scanned(a.seq_cnt, a.seq_sums, seq + 1u);
    uint32_t rank = p - lo;
    if (hi - lo <= kCallMaxRank) {
        rank = 0;
        for (uint32_t q = lo; q < hi; q++) {
            const uint32_t kq = a.bkey[q];
            rank += (kq < key || (kq == key && q < p)) ? 1u : 0u;
        }
    }
    a.sorted[lo + rank] = x;
}

// resolve_variant (variant_calling.rs:139-201) on the site's word: -> the slices as { q_from | q_len << 8 | r_from << 16 | r_len << 24 },
// kCallNoVariant for Err(ResolveVariantErr), kCallHostSite for what the host has to look at (what the reference panics on included)
__device__ __forceinline__ uint32_t resolve_word(uint32_t code, uint32_t k)
{
    if (code >> 24) return kCallHostSite;
    const uint32_t rp = code & 0xFFu, qp = (code >> 8) & 0xFFu, csl = (code >> 16) & 0xFFu;
    if (csl == 0u) return kCallHostSite; // assert!(common_suffix_len > 0)
    if (rp == 0xFFu || qp == 0xFFu) return kCallNoVariant;
    const int32_t sms = (int32_t)(k - csl);
    const int32_t query_gap = sms - (int32_t)qp - 1, ref_gap = sms - (int32_t)rp - 1;
    if (query_gap > 0 && ref_gap > 0) return (qp + 1u) | ((uint32_t)query_gap << 8) | ((rp + 1u) << 16) | ((uint32_t)ref_gap << 24);
    const int32_t qo = -query_gap, ro = -ref_gap;
    if (qo == ro) return kCallNoVariant;
    const uint32_t vlen = (uint32_t)(qo > ro ? qo - ro : ro - qo);
    if (vlen > 255u) return kCallHostSite;
    if (qo > ro) { // deletion in query
        if (rp + 1u + vlen > k) return kCallHostSite; // (the reference's slice would panic)
        return ((rp + 1u) << 16) | (vlen << 24);
    }
    if (qp + 1u + vlen > k) return kCallHostSite;
    return (qp + 1u) | (vlen << 8);
}

__global__ __launch_bounds__(256) void call_resolve_kernel(CallEmitArgs a)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_valid = scanned(a.seq_cnt, a.seq_sums, a.n_seqs);
    if (p >= n_valid) return; // (vcnt / ccnt beyond: zeroed by the caller)
    const uint32_t x = a.sorted[p], seq = a.recs[x].x;
    const uint32_t lo = scanned(a.seq_cnt, a.seq_sums, seq), hi = scanned(a.seq_cnt, a.seq_sums, seq + 1u);
    uint32_t w = hi - lo > kCallMaxRank ? kCallHostSite : resolve_word(a.codes[x], a.k);
    if (w == kCallHostSite) {
        const uint32_t slot = atomicAdd(a.meta + kCallMetaHost, 1u);
        if (slot < a.host_cap) a.host_list[slot] = x;
    }
    a.vrec[p] = w;
    const bool has = w != kCallHostSite && w != kCallNoVariant;
    a.vcnt[p] = has ? 1u : 0u;
    a.ccnt[p] = has ? ((w >> 8) & 0xFFu) + (w >> 24) : 0u;
}

__global__ __launch_bounds__(256) void call_write_kernel(CallEmitArgs a)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_valid = scanned(a.seq_cnt, a.seq_sums, a.n_seqs);
    if (p == 0) {
        a.meta[kCallMetaValid] = n_valid;
        a.meta[kCallMetaVariants] = scanned(a.vcnt, a.vsums, n_valid);
        a.meta[kCallMetaChars] = scanned(a.ccnt, a.csums, n_valid);
    }
    if (p <= a.n_seqs) a.seq_vfirst[p] = scanned(a.vcnt, a.vsums, scanned(a.seq_cnt, a.seq_sums, p)); // variants in front of sequence p
    if (p >= n_valid) return;
    const uint32_t w = a.vrec[p];
    if (w == kCallHostSite || w == kCallNoVariant) return;
    const uint32_t x = a.sorted[p];
    const uint4 rec = a.recs[x];
    const uint32_t v = scanned(a.vcnt, a.vsums, p), c = scanned(a.ccnt, a.csums, p);
    const uint32_t qf = w & 0xFFu, ql = (w >> 8) & 0xFFu, rf = (w >> 16) & 0xFFu, rl = w >> 24;
    a.out_pos[v] = rec.y;
    a.out_lens[v] = ql | (rl << 16);
    if (c + ql + rl > a.chars_cap) return; // (the host sees the total and asks again with room)
    // the query-side k-mer = the k characters of the sequence ending at j ('$' in front of it: variant_calling.rs:46-58); the row's from the window
    const uint8_t *r = a.q + a.off[rec.x];
    uint8_t *o = a.out_chars + c;
    for (uint32_t t = 0; t < ql; t++) {
        const int64_t pos = (int64_t)rec.z - (int64_t)(a.k - 1u) + qf + t;
        o[t] = pos < 0 ? (uint8_t)'$' : r[pos];
    }
    const uint8_t *rk = a.win + (size_t)x * a.stride + a.kpad;
    for (uint32_t t = 0; t < rl; t++) o[ql + t] = rk[rf + t];
}

__global__ __launch_bounds__(256) void call_host_gather_kernel(CallEmitArgs a, uint4 *__restrict__ out_recs, uint8_t *__restrict__ out_win)
{
    const uint32_t idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (idx >= min(a.meta[kCallMetaHost], a.host_cap)) return;
    const uint32_t x = a.host_list[idx];
    if (lane == 0) out_recs[idx] = a.recs[x];
    const uint32_t *src = reinterpret_cast<const uint32_t *>(a.win + (size_t)x * a.stride);
    uint32_t *dst = reinterpret_cast<uint32_t *>(out_win + (size_t)idx * a.stride);
    for (uint32_t t = lane; t < a.stride / 4u; t += 64u) dst[t] = src[t];
}

} // namespace

hipError_t launch_call_prefix(const uint32_t *d_counts, uint32_t seg_cap, uint32_t *d_prefix, uint32_t *d_meta, hipStream_t stream)
{
    hipLaunchKernelGGL(call_prefix_kernel, dim3(1), dim3(256), 0, stream, d_counts, seg_cap, d_prefix, d_meta);
    return hipGetLastError();
}

// everything behind call_depths_kernel for one slab; a.cap bounds the sites (the lists' capacity), the per-site arrays hold a.cap + 1
// words.  seq_cnt .. seq_fill (2 n_seqs + 2 words + the scan's sums) and vcnt / ccnt are zeroed here
hipError_t launch_call_emit(const CallEmitArgs &a, void *d_host_recs, uint8_t *d_host_win, hipStream_t stream)
{
    if (a.cap == 0 || a.n_seqs == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(a.seq_cnt, 0, ((size_t)a.n_seqs + 1u) * 4u, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.seq_fill, 0, (size_t)a.n_seqs * 4u, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.vcnt, 0, ((size_t)a.cap + 1u) * 4u, stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(a.ccnt, 0, ((size_t)a.cap + 1u) * 4u, stream);
    if (e != hipSuccess) return e;
    const dim3 grid((a.cap + 255u) / 256u), block(256);
    hipLaunchKernelGGL(call_hist_kernel, grid, block, 0, stream, a);
    e = launch_scan(a.seq_cnt, a.n_seqs + 1u, a.seq_sums, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(call_bucket_kernel, grid, block, 0, stream, a);
    hipLaunchKernelGGL(call_rank_kernel, grid, block, 0, stream, a);
    hipLaunchKernelGGL(call_resolve_kernel, grid, block, 0, stream, a);
    e = launch_scan(a.vcnt, a.cap + 1u, a.vsums, stream);
    if (e != hipSuccess) return e;
    e = launch_scan(a.ccnt, a.cap + 1u, a.csums, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(call_write_kernel, dim3((std::max(a.cap, a.n_seqs + 1u) + 255u) / 256u), block, 0, stream, a);
    if (a.host_cap)
        hipLaunchKernelGGL(call_host_gather_kernel, dim3((a.host_cap + 3u) / 4u), block, 0, stream, a, static_cast<uint4 *>(d_host_recs), d_host_win);
    return hipGetLastError();
}

} // namespace kbo
