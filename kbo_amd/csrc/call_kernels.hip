// call_kernels.hip — gfx950 (MI355X, CDNA4): the breakpoint scan of variant_calling::call_variants
// (reference variant_calling.rs:268-273) over a whole batch, on the device, so that only the sites leave it.
//
//   for i in 1..len:  if ms[i].d < ms[i-1].d && ms[i-1].d >= t && ms[i].d < t:
//       first j in i+1 .. min(i+k+1, len) with ms[j].d >= t && interval of ms[j] is one row  ->  site (i, j, row)
//
// One lane per base of the concatenated batch: the d test is three byte compares on coalesced loads and is false
// for all but about one base per mismatch; a lane that passes finds its sequence by binary search over the offsets
// (needed for i >= 1 and for len); the scan of the at most k positions to its right is done by the whole wave, 64
// positions per step (one lane scanning alone spends a memory round trip per position).  Sites are appended to one list, one
// atomic per wave; the host sorts them by (sequence, i).  Integer work only.
#include "device_util.hpp"

#include <algorithm>

namespace kbo {
namespace {

__global__ __launch_bounds__(256) void call_sites_kernel(const uint8_t *__restrict__ d, const uint32_t *__restrict__ lo,
                                                         const uint32_t *__restrict__ hi, const uint64_t *__restrict__ off,
                                                         uint32_t n_seqs, uint64_t total, uint32_t k, uint32_t t,
                                                         uint4 *__restrict__ sites, uint32_t cap, uint32_t *__restrict__ count)
{
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    bool cand = false, hit = false;
    uint4 rec = make_uint4(0, 0, 0, 0);
    uint64_t b0 = 0, e0 = 0;
    uint32_t s0 = 0;
    if (p >= 1 && p < total) {
        const uint32_t a = d[p - 1], b = d[p];
        if (b < a && a >= t && b < t) {
            uint32_t s1 = n_seqs; // the sequence that holds p: largest s with off[s] <= p
            while (s1 - s0 > 1) {
                const uint32_t m = s0 + (s1 - s0) / 2;
                if (off[m] <= p) s0 = m;
                else s1 = m;
            }
            b0 = off[s0];
            e0 = off[s0 + 1];
            cand = p > b0; // i >= 1: p - 1 belongs to the same sequence
        }
    }
    // the search to the right, one breakpoint at a time with the whole wave: lane x looks at position p + 1 + x (+ 64, ...)
    for (uint64_t todo = __ballot(cand); todo; todo &= todo - 1) {
        const uint32_t src = (uint32_t)__builtin_ctzll(todo);
        const uint64_t pp = ((uint64_t)__shfl((uint32_t)(p >> 32), src) << 32) | __shfl((uint32_t)p, src);
        const uint64_t ee = ((uint64_t)__shfl((uint32_t)(e0 >> 32), src) << 32) | __shfl((uint32_t)e0, src);
        const uint64_t j_end = min(pp + k + 1u, ee);
        for (uint64_t j0 = pp + 1; j0 < j_end; j0 += 64) {
            const uint64_t j = j0 + lane;
            const bool ok = j < j_end && d[j] >= t && hi[j] - lo[j] == 1u;
            const uint64_t okm = __ballot(ok);
            if (okm) {
                const uint32_t first = (uint32_t)__builtin_ctzll(okm);
                const uint64_t jf = j0 + first;
                const uint32_t row = __shfl(ok ? lo[j] : 0u, first);
                if (lane == src) {
                    hit = true;
                    rec = make_uint4(s0, (uint32_t)(p - b0), (uint32_t)(jf - b0), row);
                }
                break;
            }
        }
    }
    const uint64_t mk = __ballot(hit);
    if (mk) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(mk);
        // kCallSegs lists instead of one: a single counter would serialise one returning atomic per wave with a site
        const uint32_t seg = blockIdx.x % kCallSegs, seg_cap = cap / kCallSegs;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(count + seg * 16u, (uint32_t)__popcll(mk));
        base = __shfl(base, leader);
        const uint32_t slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        if (hit && slot < seg_cap) sites[(size_t)seg * seg_cap + slot] = rec;
    }
}

// The second pass of a batch starts on the device, right behind the first one (one wave per site, in list order):
//  * the site's record becomes {sequence, i, j, row} with positions relative to the sequence (binary search over the
//    offsets; a void record of the walk - an item the redo pass scanned again - stays void: first word ~0);
//  * the inputs of resolve_variant that the device already holds are gathered: the k MS values in front of and at the
//    match j - the walk of the query-side k-mer (variant_calling.rs:275, 279) is a fresh walk over a substring of the
//    sequence, and the set of strings that are suffixes of index rows is closed under taking suffixes, so its value at
//    position t is min(MS of the whole sequence there, t + 1): no second walk - and the k characters of the matched row
//    (variant_calling.rs:276 access_kmer), read off the path cover: text[p - k + 1 .. p] for p = pos[row] spells the row as
//    long as no path starts inside (p - k + 1, p]; the first character is the label of the node at p - k + 1.  A window that
//    crosses a path start is flagged and spelled by the host (rank / select on its copy of the index).
// Window record: [0, k) MS bytes (0xFF in front of the batch's first base: the host cuts at the sequence's start anyway),
// [kpad, kpad + k) row characters, [2 kpad] flag (1 = spell the row on the host).
__global__ __launch_bounds__(256) void call_finalize_kernel(const uint4 *__restrict__ lists, const uint32_t *__restrict__ counts,
                                                            const uint32_t *__restrict__ prefix, uint32_t seg_cap, uint32_t by_walk,
                                                            const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t k, uint32_t kpad,
                                                            const uint8_t *__restrict__ ms, DevIndexView ix, uint4 *__restrict__ out_recs,
                                                            uint8_t *__restrict__ out_win, uint32_t stride)
{
    const uint32_t g = blockIdx.y, lane = threadIdx.x & 63u;
    // (the grid strides over the list: its length is only known here)
    for (uint32_t slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; slot < min(counts[g * 16u], seg_cap); slot += (gridDim.x * blockDim.x) >> 6) {
    const uint32_t x = prefix[g] + slot;
    const uint4 v = lists[(size_t)g * seg_cap + slot];
    if (by_walk && v.x == 0xFFFFFFFFu) {
        if (lane == 0) out_recs[x] = make_uint4(0xFFFFFFFFu, 0, 0, 0);
        continue;
    }
    uint32_t seq = v.x, j_off = 0, row = 0;
    uint4 rec;
    if (by_walk) { // {offset of i, offset of j, row, 0}: the sequence that holds i
        uint32_t s0 = 0, s1 = n_seqs;
        while (s1 - s0 > 1) {
            const uint32_t m = s0 + (s1 - s0) / 2;
            if (off[m] <= v.x) s0 = m;
            else s1 = m;
        }
        seq = s0;
        const uint32_t b0 = (uint32_t)off[s0];
        rec = make_uint4(seq, v.x - b0, v.y - b0, v.z);
        j_off = v.y;
        row = v.z;
    } else { // {sequence, i, j, row} from call_sites_kernel
        rec = v;
        j_off = (uint32_t)off[v.x] + v.z;
        row = v.w;
    }
    if (lane == 0) out_recs[x] = rec;
    uint8_t *w = out_win + (size_t)x * stride;
    const bool cover = ix.pc_text != nullptr;
    const int64_t p = cover ? (int64_t)ix.pc_pos[row] : 0;
    bool broken = !cover;
    for (uint32_t t = lane; t < k; t += 64u) {
        const int64_t a = (int64_t)j_off - (int64_t)(k - 1u) + t;
        w[t] = a >= 0 ? ms[a] : (uint8_t)0xFF;
        uint32_t ch = 0;
        if (cover) {
            const int64_t q = p - (int64_t)(k - 1u) + t;
            if (q >= -(int64_t)kPlanPad) ch = ix.pc_text[q];
            if (t == 0 && q >= 0) { // the node where the window begins: its own last character (the text holds edge labels)
                const uint32_t r0 = ix.pc_node[q];
                ch = r0 >= ix.C[3] ? 'T' : r0 >= ix.C[2] ? 'G' : r0 >= ix.C[1] ? 'C' : r0 >= ix.C[0] ? 'A' : '$';
            }
            broken = broken || ch == 0;
        }
        w[kpad + t] = (uint8_t)ch;
    }
    const bool any_broken = __ballot(broken) != 0;
    if (lane == 0) w[2u * kpad] = any_broken ? 1 : 0;
    }
}

} // namespace

hipError_t launch_call_finalize(const void *d_lists, const uint32_t *d_counts, const uint32_t *d_prefix, uint32_t seg_cap, uint32_t max_count,
                                bool by_walk, const uint64_t *d_off, uint32_t n_seqs, uint32_t k, const uint8_t *d_ms,
                                const DevIndexView &ix, void *d_recs, uint8_t *d_win, uint32_t stride, hipStream_t stream)
{
    if (max_count == 0) return hipSuccess;
    const uint32_t kpad = (k + 15u) / 16u * 16u;
    hipLaunchKernelGGL(call_finalize_kernel, dim3(std::min((max_count + 3u) / 4u, 32u), kCallSegs), dim3(256), 0, stream, static_cast<const uint4 *>(d_lists),
                       d_counts, d_prefix, seg_cap, by_walk ? 1u : 0u, d_off, n_seqs, k, kpad, d_ms, ix, static_cast<uint4 *>(d_recs), d_win,
                       stride);
    return hipGetLastError();
}

// d_count: kCallSegs counters 64 bytes apart, 0 before the launch; list g holds records [g * (cap / kCallSegs), ... +
// d_count[16 g]) of d_sites; a counter above cap / kCallSegs means that list overflowed (repeat with more room)
hipError_t launch_call_sites(const uint8_t *d_ms, const uint32_t *d_lo, const uint32_t *d_hi, const uint64_t *d_off,
                             uint32_t n_seqs, uint64_t total, uint32_t k, uint32_t threshold, void *d_sites, uint32_t cap,
                             uint32_t *d_count, hipStream_t stream)
{
    if (total == 0 || n_seqs == 0) return hipSuccess;
    const uint64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(call_sites_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_ms, d_lo, d_hi, d_off, n_seqs, total,
                       k, threshold, static_cast<uint4 *>(d_sites), cap, d_count);
    return hipGetLastError();
}

} // namespace kbo
