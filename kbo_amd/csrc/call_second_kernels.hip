// call_second_kernels.hip — gfx950 (MI355X, CDNA4): the second pass of kbo::call (variant_calling.rs:271-291) for the sites
// of a slab, on the device, right behind the first pass.
//
// What resolve_variant (variant_calling.rs:139-201) reads per site:
//   * the common suffix length of the query-side k-mer (the k characters ending at the match j; '$' in front of the
//     sequence, :46-58) and the matched row's k-mer (access_kmer, :276);
//   * the rightmost significant peak (:73-83) of the query-side k-mer's walk against the reference index (:279) - a function
//     of the first pass's MS values, see call_kernels.hip;
//   * the rightmost significant peak of the ROW's k-mer walked against the index of the sequence itself (:280), which the
//     reference builds per call (lib.rs:553).  The depth of that walk at position t is the length of the longest suffix of
//     kmer[0 ..= t] that is a substring of the sequence or (add_revcomp) of its reverse complement (call_batch.cpp
//     RunAutomaton has the argument: the suffixes of the rows of a one-sequence SBWT are the substrings of its ACGT runs of at
//     least k characters).  The peak test `d[i] >= thr && d[i] > d[i + 1]` only needs a depth exactly where it is at least
//     q <= thr; below that "less than q" is enough.
// So instead of an index of the sequence per call (the reference), or a suffix automaton of it per sequence on a host thread
// (round 3: 1.83 s of host time per 125 k reads of 10 kbp), the device keeps, per sequence of the slab, a hash table of the
// START positions of its q-mers (q = min(12, thr)): a depth >= q at t means the q-mer ending at t occurs in the sequence, every
// occurrence is in the table, and the occurrence's match is extended backwards base by base (forwards, against complements,
// for the reverse strand: no second copy of the sequence).  One wave per site, a lane per position of the k-mer.
//
//   call_qmer_index_kernel   one workgroup per sequence: its q-mer start positions into its table (slot = (start + 1) | 12 tag
//                            bits << 20, linear probing, atomicCAS), a flag for a sequence that holds a byte that is no base
//   call_depths_kernel       per site: the three values above as one word { rpeak | qpeak << 8 | csl << 16 | flags << 24 }
//                            (0xFF = no peak); flags bit 0 = this site is left to the host (a sequence with bytes that are no
//                            bases or without a table, a row's k-mer that crosses a path start or holds '$')
//
// Integer / byte work only: no MFMA.
#include "device_util.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace kbo {
namespace {

__device__ __forceinline__ uint32_t base_code(uint32_t ch) { return ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u; }
__device__ __forceinline__ uint32_t qmer_slot(uint32_t code, uint32_t mask) { return ((code * 0x9E3779B1u) >> 7) & mask; }
__device__ __forceinline__ uint32_t qmer_tag(uint32_t code) { return ((code * 0x85EBCA6Bu) >> 20) & 0xFFFu; }

// the q-mer start positions of one sequence into `table` (global memory or LDS; size slots, zeroed); returns "holds a byte that is no base"
template <typename TablePtr>
__device__ __forceinline__ bool index_sequence(const uint8_t *r, uint32_t len, uint32_t qlen, TablePtr table, uint32_t size)
{
    bool mine_bad = false;
    // four start positions per lane and round: the 16 bases from the first of them on as 2-bit digits, then one shift per q-mer
    const uint32_t mask = size - 1u;
    for (uint32_t st0 = 4u * threadIdx.x; st0 < len; st0 += 4u * blockDim.x) {
        // (one unaligned 16-byte load - the slab's bases are padded by 16 - and pack16 instead of 16 byte loads and compare chains)
        uint32_t code16, valid;
        pack16(ld16u(r, st0), code16, valid);
        const uint32_t inside = len - st0 >= 16u ? 0xFFFFu : (1u << (len - st0)) - 1u;
        const uint64_t digits = code16;                        // base st0 + m in bits 2 (15 - m)
        const uint32_t invalid = (~valid | ~inside) & 0xFFFFu; // bit m: base st0 + m is no base (or lies behind the sequence)
        if (~valid & inside) mine_bad = true;
        const uint32_t span = (1u << qlen) - 1u;
#pragma unroll
        for (uint32_t v4 = 0; v4 < 4u; v4++) {
            const uint32_t st = st0 + v4;
            if (st + qlen > len || ((invalid >> v4) & span)) continue;
            const uint32_t code = (uint32_t)(digits >> (2u * (16u - qlen - v4))) & ((1u << (2u * qlen)) - 1u);
            const uint32_t v = (st + 1u) | (qmer_tag(code) << 20);
            uint32_t h = qmer_slot(code, mask);
            while (atomicCAS(table + h, 0u, v) != 0u) h = (h + 1u) & mask;
        }
    }
    return mine_bad;
}

// tables in global memory (zeroed by the caller): sequences whose table does not fit the LDS
__global__ __launch_bounds__(256) void call_qmer_index_kernel(const uint8_t *__restrict__ q, const uint64_t *__restrict__ off, uint32_t n_seqs,
                                                              uint32_t qlen, const uint64_t *__restrict__ tab_off, uint32_t *__restrict__ tab,
                                                              uint8_t *__restrict__ seq_flag)
{
    const uint32_t s = blockIdx.x;
    if (s >= n_seqs) return;
    const uint64_t b0 = off[s];
    const uint32_t len = (uint32_t)(off[s + 1] - b0);
    const uint64_t t0 = tab_off[s];
    const uint32_t size = (uint32_t)(tab_off[s + 1] - t0);
    __shared__ uint32_t bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    // (no table for a sequence too long for 20-bit positions: the host does its sites)
    const bool mine_bad = size == 0 ? len != 0 : index_sequence(q + b0, len, qlen, tab + t0, size);
    if (mine_bad) bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) seq_flag[s] = bad ? 1 : 0;
}

// the same with the table put together in LDS (up to 32 Ki slots: sequences of up to 21 k bases) and written out in whole lines:
// returning atomics on global memory were 1.2 ms per slab of 1 600 reads of 10 kbp (16 M of them), the LDS's are not measurable
__global__ __launch_bounds__(256) void call_qmer_index_lds_kernel(const uint8_t *__restrict__ q, const uint64_t *__restrict__ off, uint32_t n_seqs,
                                                                  uint32_t qlen, const uint64_t *__restrict__ tab_off, uint32_t *__restrict__ tab,
                                                                  uint8_t *__restrict__ seq_flag)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_tab[];
    const uint32_t s = blockIdx.x;
    if (s >= n_seqs) return;
    const uint64_t b0 = off[s];
    const uint32_t len = (uint32_t)(off[s + 1] - b0);
    const uint64_t t0 = tab_off[s];
    const uint32_t size = (uint32_t)(tab_off[s + 1] - t0);
    __shared__ uint32_t bad;
    if (threadIdx.x == 0) bad = 0;
    for (uint32_t i = threadIdx.x * 4u; i < size; i += blockDim.x * 4u) *reinterpret_cast<uint4 *>(lds_tab + i) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const bool mine_bad = size == 0 ? len != 0 : index_sequence(q + b0, len, qlen, lds_tab, size);
    if (mine_bad) bad = 1;
    __syncthreads();
    for (uint32_t i = threadIdx.x * 4u; i < size; i += blockDim.x * 4u) *reinterpret_cast<uint4 *>(tab + t0 + i) = *reinterpret_cast<const uint4 *>(lds_tab + i);
    if (threadIdx.x == 0) seq_flag[s] = bad ? 1 : 0;
}

constexpr uint32_t kCallMaxK = 256;

// rightmost i in [0, k - 2] with v[i] >= thr && v[i] > v[i + 1] (variant_calling.rs:73-83), or 0xFF; v in LDS, one wave
__device__ __forceinline__ uint32_t rightmost_peak(const uint8_t *v, uint32_t k, uint32_t thr, uint32_t lane)
{
    for (int32_t base = (int32_t)((k - 2u) & ~63u); base >= 0; base -= 64) {
        const uint32_t i = (uint32_t)base + lane;
        const bool p = i + 1u < k && v[i] >= thr && v[i] > v[i + 1u];
        const uint64_t m = __ballot(p);
        if (m) return (uint32_t)base + 63u - (uint32_t)__builtin_clzll(m);
    }
    return 0xFFu;
}

__global__ __launch_bounds__(256) void call_depths_kernel(const uint4 *__restrict__ recs, const uint8_t *__restrict__ win, uint32_t stride,
                                                          uint32_t kpad, uint32_t n_sites, const uint8_t *__restrict__ q,
                                                          const uint64_t *__restrict__ off, uint32_t k, uint32_t thr, uint32_t qlen,
                                                          uint32_t revcomp, const uint64_t *__restrict__ tab_off, const uint32_t *__restrict__ tab,
                                                          const uint8_t *__restrict__ seq_flag, uint32_t *__restrict__ out,
                                                          const uint32_t *__restrict__ n_dev)
{
    __shared__ uint8_t lds[4][3 * kCallMaxK];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = blockIdx.x * 4u + wv;
    if (x >= (n_dev ? min(*n_dev, n_sites) : n_sites)) return;
    const uint4 rec = recs[x];
    if (rec.x == 0xFFFFFFFFu) { // a void record (an item the redo pass scanned again)
        if (lane == 0) out[x] = 0xFFFFFFFFu;
        return;
    }
    uint8_t *rk = lds[wv], *dq = rk + kCallMaxK, *dr = dq + kCallMaxK;
    const uint8_t *w = win + (size_t)x * stride;
    const uint64_t b0 = off[rec.x];
    const uint32_t len = (uint32_t)(off[rec.x + 1] - b0), j = rec.z;
    const uint8_t *r = q + b0;
    bool bad = false;
    // the row's k-mer, the query-side walk's depths (min(MS, distance to the k-mer's / the sequence's first base): call_batch.cpp)
    // and where the two k-mers differ
    uint32_t hi_diff = 0xFFFFFFFFu; // the rightmost position where the k-mers differ (for the common suffix)
    for (uint32_t t0 = 0; t0 < k; t0 += 64u) {
        const uint32_t t = t0 + lane;
        bool differ = false;
        if (t < k) {
            const uint32_t ch = w[kpad + t];
            rk[t] = (uint8_t)ch;
            bad = bad || base_code(ch) > 3u;
            const int64_t pos = (int64_t)j - (int64_t)(k - 1u) + t;
            uint32_t qc = '$', dv = 0;
            if (pos >= 0) {
                qc = r[pos];
                dv = min((uint32_t)w[t], (uint32_t)min((int64_t)(t + 1u), pos + 1));
            }
            dr[t] = (uint8_t)dv;
            differ = qc != ch;
        }
        const uint64_t m = __ballot(differ);
        if (m) hi_diff = t0 + 63u - (uint32_t)__builtin_clzll(m);
    }
    const uint32_t flag_w = w[2u * kpad];
    const uint64_t tb0 = tab_off[rec.x];
    const uint32_t size = (uint32_t)(tab_off[rec.x + 1] - tb0);
    const bool host_site = __ballot(bad) != 0 || flag_w != 0 || seq_flag[rec.x] != 0 || (size == 0 && len >= k);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (host_site) {
        if (lane == 0) out[x] = 0x01FFFFFFu;
        return;
    }
    // the depths of the row's k-mer against the sequence (and its reverse complement), exact where they are at least qlen
    const uint32_t mask = size ? size - 1u : 0u;
    const uint32_t *tb = tab + tb0;
    for (uint32_t t0 = 0; t0 < k; t0 += 64u) {
        const uint32_t t = t0 + lane;
        if (t >= k) continue;
        uint32_t best = 0;
        if (t + 1u >= qlen && len >= k) { // (a sequence shorter than k has no k-mer: its index holds nothing)
            uint32_t code = 0, rcode = 0;
            for (uint32_t m = 0; m < qlen; m++) {
                const uint32_t c = base_code(rk[t + 1u - qlen + m]);
                code = (code << 2) | c;
                rcode |= (3u - c) << (2u * m); // reverse complement: the last base's complement first
            }
            // forward strand: the q-mer ends at t; an occurrence that starts at st matches on backwards from st - 1 / t - qlen
            {
                const uint32_t tag = qmer_tag(code);
                for (uint32_t h = qmer_slot(code, mask);; h = (h + 1u) & mask) {
                    const uint32_t v = tb[h];
                    if (v == 0) break;
                    if ((v >> 20) != tag) continue;
                    const uint32_t st = (v & 0xFFFFFu) - 1u;
                    bool same = true;
                    for (uint32_t m = 0; m < qlen; m++) same = same && r[st + m] == rk[t + 1u - qlen + m];
                    if (!same) continue;
                    uint32_t L = qlen;
                    while (L <= t && L < st + qlen && r[st + qlen - 1u - L] == rk[t - L]) L++;
                    best = max(best, L);
                }
            }
            if (revcomp) { // reverse strand: the complement of rk[t - m] stands at st + m
                const uint32_t tag = qmer_tag(rcode);
                for (uint32_t h = qmer_slot(rcode, mask);; h = (h + 1u) & mask) {
                    const uint32_t v = tb[h];
                    if (v == 0) break;
                    if ((v >> 20) != tag) continue;
                    const uint32_t st = (v & 0xFFFFFu) - 1u;
                    auto comp_eq = [&](uint32_t a, uint32_t b) { return base_code(a) + base_code(b) == 3u; };
                    bool same = true;
                    for (uint32_t m = 0; m < qlen; m++) same = same && comp_eq(r[st + m], rk[t - m]);
                    if (!same) continue;
                    uint32_t L = qlen;
                    while (L <= t && st + L < len && comp_eq(r[st + L], rk[t - L])) L++;
                    best = max(best, L);
                }
            }
        }
        dq[t] = (uint8_t)min(best, k);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const uint32_t rpeak = rightmost_peak(dq, k, thr, lane), qpeak = rightmost_peak(dr, k, thr, lane);
    const uint32_t csl = hi_diff == 0xFFFFFFFFu ? k : k - 1u - hi_diff;
    if (lane == 0) out[x] = rpeak | (qpeak << 8) | (min(csl, 255u) << 16);
}


// The same for k <= 64 (what kbo is run with: 31, 51, 63): a lane per position of the k-mer as above, but an occurrence's match is not
// extended base by base by its lane - up to t dependent byte loads from the sequence per lane, 1 900 a site - but read off its DIAGONAL:
// the lanes whose q-mers were found on one diagonal of (k-mer position, sequence position) - nearly all of a site's, the k-mer being
// a copy of the sequence around the variant - share one 64-bit mask of the positions where k-mer and sequence agree on it (one
// coalesced read of k bases + a ballot), and a lane's match is the run of ones that ends at its bit.  The reverse strand the same on
// anti-diagonals (the complement of rk[u] at s - u).  Same words as call_depths_kernel (tests run both on k = 51).
__device__ __forceinline__ void depths64_site(uint32_t x, uint32_t lane, uint8_t *rk, const uint4 *__restrict__ recs, const uint8_t *__restrict__ win,
                                              uint32_t stride, uint32_t kpad, const uint8_t *__restrict__ q, const uint64_t *__restrict__ off, uint32_t k,
                                              uint32_t thr, uint32_t qlen, uint32_t revcomp, const uint64_t *__restrict__ tab_off,
                                              const uint32_t *__restrict__ tab, const uint8_t *__restrict__ seq_flag, uint32_t *__restrict__ out)
{
    const uint4 rec = recs[x];
    if (rec.x == 0xFFFFFFFFu) { // a void record (an item the redo pass scanned again)
        if (lane == 0) out[x] = 0xFFFFFFFFu;
        return;
    }
    uint8_t *dq = rk + 64, *dr = dq + 64;
    const uint8_t *w = win + (size_t)x * stride;
    const uint64_t b0 = off[rec.x];
    const uint32_t len = (uint32_t)(off[rec.x + 1] - b0), j = rec.z;
    const uint8_t *r = q + b0;
    const uint32_t t = lane;
    // the row's k-mer, the query-side walk's depths and where the two k-mers differ (as above)
    uint32_t my_ch = 0, my_code = 4u;
    bool differ = false;
    if (t < k) {
        my_ch = w[kpad + t];
        my_code = base_code(my_ch);
        rk[t] = (uint8_t)my_ch;
        const int64_t pos = (int64_t)j - (int64_t)(k - 1u) + t;
        uint32_t qc = '$', dv = 0;
        if (pos >= 0) {
            qc = r[pos];
            dv = min((uint32_t)w[t], (uint32_t)min((int64_t)(t + 1u), pos + 1));
        }
        dr[t] = (uint8_t)dv;
        differ = qc != my_ch;
    }
    const uint64_t dm = __ballot(differ);
    const uint32_t hi_diff = dm ? 63u - (uint32_t)__builtin_clzll(dm) : 0xFFFFFFFFu;
    const bool bad = __ballot(t < k && my_code > 3u) != 0;
    const uint32_t flag_w = w[2u * kpad];
    const uint64_t tb0 = tab_off[rec.x];
    const uint32_t size = (uint32_t)(tab_off[rec.x + 1] - tb0);
    const bool host_site = bad || flag_w != 0 || seq_flag[rec.x] != 0 || (size == 0 && len >= k);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (host_site) {
        if (lane == 0) out[x] = 0x01FFFFFFu;
        return;
    }
    const uint32_t mask = size ? size - 1u : 0u;
    const uint32_t *tb = tab + tb0;
    uint32_t best = 0;
    const bool looks = t < k && t + 1u >= qlen && len >= k; // (a sequence shorter than k has no k-mer: its index holds nothing)
    uint32_t code = 0, rcode = 0;
    if (looks)
        for (uint32_t m = 0; m < qlen; m++) {
            const uint32_t c = base_code(rk[t + 1u - qlen + m]);
            code = (code << 2) | c;
            rcode |= (3u - c) << (2u * m); // reverse complement: the last base's complement first
        }
    for (uint32_t strand = 0; strand < (revcomp ? 2u : 1u); strand++) {
        const uint32_t cd = strand ? rcode : code, tag = qmer_tag(cd);
        uint32_t h = qmer_slot(cd, mask);
        bool probing = looks;
        while (__ballot(probing)) {
            // every lane on to its next occurrence by tag (or the end of its chain)
            bool cand = false;
            int32_t diag = 0; // forward: sequence position of rk[0]; reverse: of rk[0]'s complement
            while (probing && !cand) {
                const uint32_t v = tb[h];
                h = (h + 1u) & mask;
                if (v == 0) probing = false;
                else if ((v >> 20) == tag) {
                    const uint32_t st = (v & 0xFFFFFu) - 1u;
                    diag = strand ? (int32_t)(st + t) : (int32_t)st - (int32_t)(t + 1u - qlen);
                    cand = true;
                }
            }
            // the occurrences found, diagonal by diagonal
            uint64_t pending = __ballot(cand);
            while (pending) {
                const uint32_t leader = (uint32_t)__builtin_ctzll(pending);
                const int32_t dg = __shfl(diag, (int)leader);
                const int64_t sp = strand ? (int64_t)dg - (int64_t)t : (int64_t)dg + (int64_t)t; // where this lane's k-mer base stands on it
                bool eq = false;
                if (t < k && sp >= 0 && sp < (int64_t)len) {
                    const uint32_t c = base_code(r[sp]);
                    eq = strand ? c + my_code == 3u : c == my_code;
                }
                const uint64_t M = __ballot(eq);
                const bool mine = cand && diag == dg;
                if (mine) {
                    const uint64_t inv = ~(M << (63u - t)); // bit 63 = position t, downwards
                    const uint32_t L = inv ? (uint32_t)__builtin_clzll(inv) : 64u;
                    if (L >= qlen) best = max(best, L); // (shorter: the tag matched another q-mer)
                }
                pending &= ~__ballot(mine);
            }
        }
    }
    if (t < k) dq[t] = (uint8_t)min(best, k);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const uint32_t rpeak = rightmost_peak(dq, k, thr, lane), qpeak = rightmost_peak(dr, k, thr, lane);
    const uint32_t csl = hi_diff == 0xFFFFFFFFu ? k : k - 1u - hi_diff;
    if (lane == 0) out[x] = rpeak | (qpeak << 8) | (min(csl, 255u) << 16);
}

// (a fixed grid that strides over the sites: their number is only known on the device - n_dev -, and a grid for the lists' capacity was
// three empty workgroups in four)
__global__ __launch_bounds__(256) void call_depths64_kernel(const uint4 *__restrict__ recs, const uint8_t *__restrict__ win, uint32_t stride,
                                                            uint32_t kpad, uint32_t n_sites, const uint8_t *__restrict__ q,
                                                            const uint64_t *__restrict__ off, uint32_t k, uint32_t thr, uint32_t qlen,
                                                            uint32_t revcomp, const uint64_t *__restrict__ tab_off, const uint32_t *__restrict__ tab,
                                                            const uint8_t *__restrict__ seq_flag, uint32_t *__restrict__ out,
                                                            const uint32_t *__restrict__ n_dev)
{
    __shared__ uint8_t lds[4][3 * 64];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t n = n_dev ? min(*n_dev, n_sites) : n_sites;
    for (uint32_t x = blockIdx.x * 4u + wv; x < n; x += gridDim.x * 4u) {
        depths64_site(x, lane, lds[wv], recs, win, stride, kpad, q, off, k, thr, qlen, revcomp, tab_off, tab, seq_flag, out);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace

// table sizes are the caller's (powers of two >= 64, tab_off[s + 1] - tab_off[s] slots for sequence s; 0 = none).  max_slots = the
// largest of them: up to 32 Ki the tables are put together in LDS (d_tab need not be zeroed), else in global memory (d_tab zeroed)
hipError_t launch_call_qmer_index(const uint8_t *d_q, const uint64_t *d_off, uint32_t n_seqs, uint32_t qlen, const uint64_t *d_tab_off,
                                  uint32_t *d_tab, uint8_t *d_seq_flag, uint64_t total_slots, uint32_t max_slots, hipStream_t stream)
{
    if (n_seqs == 0) return hipSuccess;
    if (max_slots <= 32768u) {
        const uint32_t lds = std::max(max_slots, 64u) * 4u;
        // (per device and cheap: set every time rather than remembered per process)
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(call_qmer_index_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(call_qmer_index_lds_kernel, dim3(n_seqs), dim3(256), lds, stream, d_q, d_off, n_seqs, qlen, d_tab_off, d_tab, d_seq_flag);
        return hipGetLastError();
    }
    const hipError_t e = hipMemsetAsync(d_tab, 0, total_slots * 4, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(call_qmer_index_kernel, dim3(n_seqs), dim3(256), 0, stream, d_q, d_off, n_seqs, qlen, d_tab_off, d_tab, d_seq_flag);
    return hipGetLastError();
}

hipError_t launch_call_depths(const void *d_recs, const uint8_t *d_win, uint32_t stride, uint32_t n_sites, const uint8_t *d_q,
                              const uint64_t *d_off, uint32_t k, uint32_t thr, uint32_t qlen, bool revcomp, const uint64_t *d_tab_off,
                              const uint32_t *d_tab, const uint8_t *d_seq_flag, uint32_t *d_out, hipStream_t stream, const uint32_t *d_n_sites,
                              bool per_lane)
{
    if (n_sites == 0) return hipSuccess;
    if (k > kCallMaxK - 1u || k < 2u) return hipErrorInvalidValue;
    const uint32_t kpad = (k + 15u) / 16u * 16u;
    if (k <= 64u && !per_lane) {
        hipLaunchKernelGGL(call_depths64_kernel, dim3(std::min((n_sites + 3u) / 4u, 8192u)), dim3(256), 0, stream, static_cast<const uint4 *>(d_recs), d_win, stride, kpad,
                           n_sites, d_q, d_off, k, thr, qlen, revcomp ? 1u : 0u, d_tab_off, d_tab, d_seq_flag, d_out, d_n_sites);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(call_depths_kernel, dim3((n_sites + 3u) / 4u), dim3(256), 0, stream, static_cast<const uint4 *>(d_recs), d_win, stride, kpad,
                       n_sites, d_q, d_off, k, thr, qlen, revcomp ? 1u : 0u, d_tab_off, d_tab, d_seq_flag, d_out, d_n_sites);
    return hipGetLastError();
}

} // namespace kbo
