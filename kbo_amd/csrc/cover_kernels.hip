// cover_kernels.hip — gfx950 (MI355X, CDNA4): the path cover of the index's de Bruijn graph (path_cover.cpp has what it is and why) laid out
// ON THE DEVICE, straight into the copy's own buffers: the host's pointer chase was the slow part of a copy's plan structures (26 s per
// 10^8 rows on the box's 16 CPUs' worth of time).  The same layout, position for position (heads in row order, each chain to its end):
//
//   cover_next_kernel     next[u] = the successor matched to row u: the r-th row of a (k-1)-suffix group takes the group's r-th edge, whose
//                         target is C[c] + rank_c(first row of the group) - one look-up in the copy's rank blocks (the host streams over the
//                         rows with four running counters)
//   cover_split_*         splitters = the heads (rows nobody points at) and every row that is a multiple of 1024, in row order (a count per
//                         1024 rows, a scan, an ordered write)
//   cover_measure_kernel  per splitter: the length of the segment that starts there and the splitter it runs into (a pointer chase of
//                         about a thousand steps - millions of them side by side)
//   (host)                one pass over the splitters - a few per thousand rows - hands every segment of a head's chain its text position
//   cover_lay_kernel      per splitter: pos / node_at / text of its segment
//   cover_left_kernel     rows without a position (they lie on cycles no head leads into): any -> the host's construction decides
//
// Integer / pointer work only: no MFMA.
#include "device_util.hpp"

#include <algorithm>
#include <vector>

namespace kbo {
namespace {

constexpr uint32_t kCoverSplit = 1024, kCoverNone = 0xFFFFFFFFu;

struct CoverArgs {
    const uint4 *rank;     // rank blocks, character c at c * n_blocks
    const uint32_t *ent;   // contraction entries: lcs of row i = ent[3 i]
    uint64_t n;
    uint32_t n_blocks, k;
    uint32_t C[4];
    uint32_t *next;        // n
    uint8_t *has_pred;     // n
};

__global__ __launch_bounds__(256) void cover_next_kernel(CoverArgs a)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    // the first row of i's (k-1)-suffix group: the nearest row at or in front of i that opens one
    uint64_t g = i;
    while (g > 0 && a.ent[3u * g] + 1u >= a.k) g--;
    const uint32_t r = (uint32_t)(i - g);
    const uint32_t b = (uint32_t)(g / 96u), o = (uint32_t)(g - (uint64_t)b * 96u);
    uint32_t seen = 0, succ = kCoverNone;
#pragma unroll
    for (uint32_t c = 0; c < 4u; c++) {
        const uint4 v = a.rank[(uint64_t)c * a.n_blocks + b];
        const uint32_t w = o < 32u ? v.y : o < 64u ? v.z : v.w;
        if ((w >> (o & 31u)) & 1u) {
            if (seen == r) succ = rank_eval(v, o); // C[c] + rank_c(g): the target of the group's edge c
            seen++;
        }
    }
    if (succ != kCoverNone && succ != (uint32_t)i && succ < a.n) { // (a self-loop - AAA -> AAA - cannot be a path step)
        a.next[i] = succ;
        a.has_pred[succ] = 1;
    }
}

// splitters per block of 1024 rows (256 threads, four rows each), then written in row order behind the scan of the counts
__device__ __forceinline__ uint32_t split_flags(const uint8_t *__restrict__ has_pred, uint64_t n, uint64_t i0)
{ // bit j: row i0 + j is a splitter (a head, or a multiple of 1024)
    uint32_t f = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4u; j++) {
        const uint64_t i = i0 + j;
        if (i < n && (!has_pred[i] || (i & (kCoverSplit - 1u)) == 0)) f |= 1u << j;
    }
    return f;
}
__global__ __launch_bounds__(256) void cover_split_count_kernel(const uint8_t *__restrict__ has_pred, uint64_t n, uint32_t *__restrict__ counts)
{
    __shared__ uint32_t sh[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kCoverSplit + 4u * threadIdx.x;
    uint32_t c = (uint32_t)__popc(split_flags(has_pred, n, i0));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63u) == 0) sh[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void cover_split_write_kernel(const uint8_t *__restrict__ has_pred, uint64_t n, const uint32_t *__restrict__ counts,
                                                                const uint32_t *__restrict__ sums, uint32_t *__restrict__ splitters,
                                                                uint8_t *__restrict__ is_head)
{
    __shared__ uint32_t wave_tot[4];
    const uint64_t i0 = (uint64_t)blockIdx.x * kCoverSplit + 4u * threadIdx.x;
    const uint32_t f = split_flags(has_pred, n, i0), mine = (uint32_t)__popc(f);
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)lane >= off) incl += t;
    }
    if (lane == 63u) wave_tot[wv] = incl;
    __syncthreads();
    uint32_t base = sums[blockIdx.x / kScanBlock] + counts[blockIdx.x] + incl - mine;
    for (uint32_t w = 0; w < wv; w++) base += wave_tot[w];
    for (uint32_t m = f; m; m &= m - 1u) {
        const uint64_t i = i0 + (uint32_t)__builtin_ctz(m);
        splitters[base] = (uint32_t)i;
        is_head[base] = has_pred[i] ? 0 : 1;
        base++;
    }
}

__global__ __launch_bounds__(256) void cover_measure_kernel(const uint32_t *__restrict__ next, const uint32_t *__restrict__ splitters, uint32_t ns,
                                                            uint32_t *__restrict__ seg_len, uint32_t *__restrict__ seg_next)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    uint32_t len = 1, u = next[splitters[s]];
    while (u != kCoverNone && (u & (kCoverSplit - 1u)) != 0) { // (a row reached through next has a predecessor: a splitter iff a multiple of 1024)
        len++;
        u = next[u];
    }
    seg_len[s] = len;
    seg_next[s] = u;
}

__global__ __launch_bounds__(256) void cover_lay_kernel(CoverArgs a, const uint32_t *__restrict__ splitters, uint32_t ns, const uint32_t *__restrict__ seg_len,
                                                        const uint32_t *__restrict__ seg_start, uint32_t *__restrict__ pos, uint32_t *__restrict__ node_at,
                                                        uint8_t *__restrict__ text)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    uint32_t q = seg_start[s];
    if (q == kCoverNone) return; // (a splitter on a cycle no head leads into)
    uint32_t u = splitters[s];
    const uint32_t len = seg_len[s];
    for (uint32_t j = 0; j < len; j++) {
        pos[u] = q;
        node_at[q] = u;
        // the label of the edge into u = the last character of its row (rows are in colex order); 0 where a path starts
        uint8_t ch = 0;
        if (a.has_pred[u]) ch = u >= a.C[3] ? (uint8_t)'T' : u >= a.C[2] ? (uint8_t)'G' : u >= a.C[1] ? (uint8_t)'C' : u >= a.C[0] ? (uint8_t)'A' : (uint8_t)0;
        text[q] = ch;
        q++;
        u = a.next[u];
    }
}

__global__ __launch_bounds__(256) void cover_left_kernel(const uint32_t *__restrict__ pos, uint64_t n, uint32_t *__restrict__ count)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool left = i < n && pos[i] == kCoverNone;
    const uint64_t m = __ballot(left);
    if (m && (threadIdx.x & 63u) == 0) atomicAdd(count, (uint32_t)__popcll(m));
}

} // namespace

// d_text: points at text position 0 of a zeroed buffer with kPlanPad bytes either side; d_pos / d_node: n words each.  *ok = false when
// rows are left without a position (cycles no head leads into) - the buffers then hold a partial layout and the host's construction
// decides.  Temporaries: 5 bytes a row + a few per thousand rows.  Synchronises the stream.
hipError_t build_path_cover_device(const uint4 *d_rank, const uint32_t *d_ent, uint64_t n, uint32_t n_blocks, uint32_t k, const uint64_t C[4],
                                   uint8_t *d_text, uint32_t *d_pos, uint32_t *d_node, hipStream_t stream, bool *ok)
{
    *ok = false;
    if (n == 0 || n >= 0xFFFFFFF0ull) return hipSuccess;
    void *tmp = nullptr;
    const uint64_t n_chunks = (n + kCoverSplit - 1u) / kCoverSplit;
    const size_t sums_words = (size_t)(n_chunks / kScanBlock + 2u);
    const size_t bytes_next = (size_t)n * 4u, bytes_pred = ((size_t)n + 15u) / 16u * 16u, bytes_counts = ((size_t)n_chunks + 1u + sums_words) * 4u + 64u;
    hipError_t e = hipMalloc(&tmp, bytes_next + bytes_pred + bytes_counts);
    if (e != hipSuccess) return e;
    struct Free {
        void *p, *q;
        ~Free() { (void)hipFree(p); if (q) (void)hipFree(q); }
    } guard{tmp, nullptr};
    CoverArgs a{};
    a.rank = d_rank;
    a.ent = d_ent;
    a.n = n;
    a.n_blocks = n_blocks;
    a.k = k;
    for (int c = 0; c < 4; c++) a.C[c] = (uint32_t)C[c];
    a.next = static_cast<uint32_t *>(tmp);
    a.has_pred = static_cast<uint8_t *>(tmp) + bytes_next;
    uint32_t *counts = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(tmp) + bytes_next + bytes_pred), *sums = counts + n_chunks + 1u;
    if ((e = hipMemsetAsync(a.next, 0xFF, bytes_next, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(a.has_pred, 0, bytes_pred, stream)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(d_pos, 0xFF, (size_t)n * 4u, stream)) != hipSuccess) return e;
    const dim3 blk(256), grid_rows((uint32_t)((n + 255u) / 256u));
    hipLaunchKernelGGL(cover_next_kernel, grid_rows, blk, 0, stream, a);
    if ((e = hipMemsetAsync(counts + n_chunks, 0, 4, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(cover_split_count_kernel, dim3((uint32_t)n_chunks), blk, 0, stream, a.has_pred, n, counts);
    if ((e = launch_scan(counts, (uint32_t)n_chunks + 1u, sums, stream)) != hipSuccess) return e;
    uint32_t tail[2] = {0, 0}; // the grand total = the scanned value of the extra entry
    if ((e = hipMemcpyAsync(&tail[0], counts + n_chunks, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(&tail[1], sums + n_chunks / kScanBlock, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    const uint32_t ns = tail[0] + tail[1];
    if (ns == 0) return hipSuccess;
    void *tmp2 = nullptr;
    const size_t w = ((size_t)ns + 3u) / 4u * 4u;
    if ((e = hipMalloc(&tmp2, w * 4u * 4u + w + 64u)) != hipSuccess) return e;
    guard.q = tmp2;
    uint32_t *splitters = static_cast<uint32_t *>(tmp2), *seg_len = splitters + w, *seg_next = seg_len + w, *seg_start = seg_next + w;
    uint8_t *is_head = reinterpret_cast<uint8_t *>(seg_start + w);
    hipLaunchKernelGGL(cover_split_write_kernel, dim3((uint32_t)n_chunks), blk, 0, stream, a.has_pred, n, counts, sums, splitters, is_head);
    hipLaunchKernelGGL(cover_measure_kernel, dim3((ns + 255u) / 256u), blk, 0, stream, a.next, splitters, ns, seg_len, seg_next);
    std::vector<uint32_t> h_split(ns), h_len(ns), h_next(ns), h_start(ns, kCoverNone);
    std::vector<uint8_t> h_head(ns);
    if ((e = hipMemcpyAsync(h_split.data(), splitters, (size_t)ns * 4u, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(h_len.data(), seg_len, (size_t)ns * 4u, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(h_next.data(), seg_next, (size_t)ns * 4u, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipMemcpyAsync(h_head.data(), is_head, ns, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    // heads in row order, each chain to its end: exactly the layout one thread following the chains would make (path_cover.cpp)
    uint64_t p = 0;
    for (uint32_t s = 0; s < ns; s++) {
        if (!h_head[s]) continue;
        size_t at = s;
        for (;;) {
            h_start[at] = (uint32_t)p;
            p += h_len[at];
            if (h_next[at] == kCoverNone) break;
            at = (size_t)(std::lower_bound(h_split.begin(), h_split.end(), h_next[at]) - h_split.begin());
            if (at >= ns || h_start[at] != kCoverNone) return hipSuccess; // (cannot happen: a chain meets itself - the host's construction decides)
        }
    }
    if (p != n) return hipSuccess; // rows on cycles: *ok stays false
    if ((e = hipMemcpyAsync(seg_start, h_start.data(), (size_t)ns * 4u, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(cover_lay_kernel, dim3((ns + 255u) / 256u), blk, 0, stream, a, splitters, ns, seg_len, seg_start, d_pos, d_node, d_text);
    if ((e = hipMemsetAsync(counts, 0, 4, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(cover_left_kernel, grid_rows, blk, 0, stream, d_pos, n, counts);
    uint32_t left = 1;
    if ((e = hipMemcpyAsync(&left, counts, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    *ok = left == 0;
    return hipGetLastError();
}

} // namespace kbo
