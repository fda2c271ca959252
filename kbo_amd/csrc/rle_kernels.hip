// rle_kernels.hip — gfx950 (MI355X, CDNA4): format::run_lengths_gapped (format.rs:143-193), the tail of kbo::find,
// over a batch of translated sequences: count pass, two-level scan, emit pass (reads with max_gap_len = 0: bit-mask kernels staged
// through LDS - rle0_lds_kernel for any characters, rle0_own_kernel for the kernels' own M - X R).
#include "device_util.hpp"

#include <algorithm>

namespace kbo {
namespace {

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
            const uint32_t rec[7] = {st.start, st.end, st.matches, st.mismatches, st.jumps, st.gap_bases, st.gap_opens};
            __builtin_memcpy(out + (uint64_t)slot * 7u, rec, 28); // (16 + 12 bytes: two stores instead of seven)
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
                                                  uint32_t capacity, uint32_t min_len)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    const uint64_t b = off[s];
    const uint32_t len = (uint32_t)(off[s + 1] - b);
    // (kbo::find's tail, min_len = 3: a sequence of fewer than 3 bases has no alignment - the reference asserts, derandomize.rs:274-276 -
    // and the kernels in front leave its bytes as they were: no run, whatever stands there)
    if (len < min_len) {
        if (!EMIT) counts[s] = 0;
        return;
    }
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
                                                      uint32_t capacity, uint32_t lds_bytes, uint32_t min_len)
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
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (one wave a workgroup: nobody else to wait for)
    __builtin_amdgcn_wave_barrier();
    if (s >= n_seqs) return;
    const uint32_t b = (uint32_t)(off[s] - base), len = (uint32_t)(off[s + 1] - off[s]);
    const uint32_t first = EMIT ? sums[s / kScanBlock] + counts[s] : 0u;
    uint32_t n_out = 0, in_run = 0, start = 0, end = 0, matches = 0, mismatches = 0, jumps = 0, gap_bases = 0, prev_r = 0;
    auto close = [&]() {
        if (EMIT) {
            const uint32_t slot = first + n_out;
            if (slot < capacity) {
                // (two stores, 16 + 12 bytes: seven dword stores are seven partial-line requests per record)
                const uint32_t rec[7] = {start, end, matches, mismatches, jumps, gap_bases, 0u};
                __builtin_memcpy(out + (uint64_t)slot * 7u, rec, 28);
            }
        }
        n_out++;
        in_run = 0;
    };
    for (uint32_t i0 = 0; len >= min_len && i0 < len; i0 += 16u) { // (min_len: as in rle_kernel)
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

// The same for characters that are translate_ms_vec's own (translate.rs:180-216: 'M' 4D, '-' 2D, 'X' 58, 'R' 52 and nothing else - kbo::find's
// tail, lib.rs:816-820): bit 5 of a character says '-', bit 1 'R', bit 0 without bit 5 'M'.  Four characters a word: the three bits of each
// gathered by a multiplication (bits 0, 8, 16, 24 of x land in bits 24 .. 27 of x * 0x01020408, nothing else does), a lane's 16 characters from
// five aligned words of its LDS row - a fifth of the instructions of the byte-by-byte classification above, which took as long as
// map_reads_kernel itself at C3.
template <bool EMIT>
__global__ __launch_bounds__(64) void rle0_own_kernel(const uint8_t *__restrict__ chars, const uint64_t *__restrict__ off,
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
    for (uint32_t o = lane * 16u; o < span; o += 1024u) // stage in (reads <= 15 B past the span)
        *reinterpret_cast<uint4 *>(lds + o) = ld16u(chars + base, o);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (one wave a workgroup: nobody else to wait for)
    __builtin_amdgcn_wave_barrier();
    if (s >= n_seqs) return;
    const uint32_t b = (uint32_t)(off[s] - base), len = (uint32_t)(off[s + 1] - off[s]);
    const uint32_t first = EMIT ? sums[s / kScanBlock] + counts[s] : 0u;
    const uint32_t *row = reinterpret_cast<const uint32_t *>(lds) + (b >> 2);
    const uint32_t sh = b & 3u;
    uint32_t n_out = 0, in_run = 0, start = 0, end = 0, matches = 0, mismatches = 0, jumps = 0, prev_r = 0;
    auto close = [&]() {
        if (EMIT) {
            const uint32_t slot = first + n_out;
            if (slot < capacity) {
                const uint32_t rec[7] = {start, end, matches, mismatches, jumps, 0u, 0u};
                __builtin_memcpy(out + (uint64_t)slot * 7u, rec, 28);
            }
        }
        n_out++;
        in_run = 0;
    };
    uint32_t carry = row[0];
    // (a sequence of fewer than 3 bases has no alignment - the reference asserts, derandomize.rs:274-276 - and no run: what the kernels
    // in front counted for it)
    for (uint32_t i0 = 0; len >= 3u && i0 < len; i0 += 16u) {
        const uint32_t n = min(16u, len - i0);
        uint32_t E = 0, M1 = 0, R = 0; // bit j describes character i0 + j: '-' / bit 0 / 'R'
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) {
            const uint32_t nxt = row[(i0 >> 2) + q + 1u]; // (up to 7 bytes behind the row: the staged span is padded)
            const uint32_t d = __builtin_amdgcn_alignbyte(nxt, carry, sh);
            carry = nxt;
            E |= ((((d >> 5) & 0x01010101u) * 0x01020408u) >> 24) << (4u * q);
            R |= ((((d >> 1) & 0x01010101u) * 0x01020408u) >> 24) << (4u * q);
            M1 |= (((d & 0x01010101u) * 0x01020408u) >> 24) << (4u * q);
        }
        const uint32_t valid = n == 16u ? 0xFFFFu : (1u << n) - 1u;
        const uint32_t M = (M1 & ~E) | R;
        const uint32_t RR = R & ((R << 1) | prev_r); // 'R' right after an 'R' (format.rs:175)
        prev_r = (R >> 15) & 1u;                     // only meaningful when n == 16 (no block follows otherwise)
        uint32_t cur = 0;
        while (cur < n) {
            if (in_run) {
                const uint32_t ev = E & valid & (~0u << cur);
                const uint32_t e = ev ? (uint32_t)__ffs((int)ev) - 1u : n;
                const uint32_t seg = ((1u << e) - 1u) & (~0u << cur); // characters [cur, e) of the block
                matches += __popc(M & seg);
                mismatches += __popc(~M & seg);
                jumps += __popc(RR & seg);
                if (seg) end = i0 + (32u - (uint32_t)__clz((int)seg)); // (no 'D' here: end moves past every character, format.rs:172)
                if (e < n) { // a '-' ends the run (its gap is taken back out: net nothing)
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
                end = matches = mismatches = jumps = 0;
            }
        }
    }
    if (in_run) close();
    if (!EMIT) counts[s] = n_out;
}

template <bool EMIT>
static void launch_rle_pass(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                            uint32_t *local, const uint32_t *sums, uint32_t *d_rles, uint32_t capacity, uint32_t max_seq_len,
                            hipStream_t stream, bool own_alphabet)
{
    const uint32_t min_len = own_alphabet ? 3u : 0u; // (the kernels' own characters = an alignment of kbo::find: none for fewer than 3 bases)
    if (max_seq_len > 0 && max_seq_len <= 480 && max_gap_len == 0) {
        const uint32_t lds_bytes = ((64u * max_seq_len + 15u) / 16u) * 16u + 16u;
        if (own_alphabet)
            hipLaunchKernelGGL((rle0_own_kernel<EMIT>), dim3((n_seqs + 63) / 64), dim3(64), lds_bytes + 16u, stream, d_chars, d_offsets, n_seqs, local,
                               sums, d_rles, capacity, lds_bytes);
        else if (max_seq_len % 32u == 0)
            hipLaunchKernelGGL((rle0_lds_kernel<EMIT, true>), dim3((n_seqs + 63) / 64), dim3(64), lds_bytes + lds_bytes / 32u + 16u,
                               stream, d_chars, d_offsets, n_seqs, local, sums, d_rles, capacity, lds_bytes, min_len);
        else
            hipLaunchKernelGGL((rle0_lds_kernel<EMIT, false>), dim3((n_seqs + 63) / 64), dim3(64), lds_bytes, stream, d_chars,
                               d_offsets, n_seqs, local, sums, d_rles, capacity, lds_bytes, min_len);
    } else {
        hipLaunchKernelGGL((rle_kernel<EMIT>), dim3((n_seqs + 255) / 256), dim3(256), 0, stream, d_chars, d_offsets, n_seqs,
                           max_gap_len, local, sums, d_rles, capacity, min_len);
    }
}

// total number of runs (the scan's grand total) -> one word the host can read after the stream
__global__ void rle_total_kernel(const uint32_t *__restrict__ local, const uint32_t *__restrict__ sums, uint32_t n_seqs,
                                 uint32_t *__restrict__ total)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *total = sums[n_seqs / kScanBlock] + local[n_seqs];
}

} // namespace

// format::run_lengths_gapped over a batch: counts -> exclusive scan (counts[n_seqs] = total slot) -> emit.
// d_scratch: chunk_items_scratch_words(n_seqs) u32 (per-sequence first-run index, block sums);
// d_total: one u32.  Records beyond `capacity` are counted but not written (the caller re-emits).
hipError_t launch_rle_count(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                            uint32_t *d_scratch, uint32_t *d_total, hipStream_t stream, uint32_t max_seq_len, bool own_alphabet)
{
    if (n_seqs == 0) return hipSuccess;
    const uint32_t n = n_seqs + 1;
    uint32_t *local = d_scratch, *sums = d_scratch + n;
    const hipError_t e = hipMemsetAsync(local + n_seqs, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    launch_rle_pass<false>(d_chars, d_offsets, n_seqs, max_gap_len, local, nullptr, nullptr, 0u, max_seq_len, stream, own_alphabet);
    const hipError_t es = launch_scan(local, n, sums, stream);
    if (es != hipSuccess) return es;
    hipLaunchKernelGGL(rle_total_kernel, dim3(1), dim3(64), 0, stream, local, sums, n_seqs, d_total);
    return hipGetLastError();
}

// the same with the counts given (d_scratch[0 .. n_seqs): map_reads_kernel's and, for the reads of its second pass, launch_derand_flagged's): scan + total only
hipError_t launch_rle_scan_counts(uint32_t n_seqs, uint32_t *d_scratch, uint32_t *d_total, hipStream_t stream)
{
    if (n_seqs == 0) return hipSuccess;
    const uint32_t n = n_seqs + 1;
    uint32_t *local = d_scratch, *sums = d_scratch + n;
    const hipError_t e = hipMemsetAsync(local + n_seqs, 0, sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const hipError_t es = launch_scan(local, n, sums, stream);
    if (es != hipSuccess) return es;
    hipLaunchKernelGGL(rle_total_kernel, dim3(1), dim3(64), 0, stream, local, sums, n_seqs, d_total);
    return hipGetLastError();
}

hipError_t launch_rle_emit(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                           uint32_t *d_scratch, uint32_t *d_rles, uint32_t capacity, hipStream_t stream,
                           uint32_t max_seq_len, bool own_alphabet)
{
    if (n_seqs == 0) return hipSuccess;
    uint32_t *local = d_scratch, *sums = d_scratch + n_seqs + 1;
    launch_rle_pass<true>(d_chars, d_offsets, n_seqs, max_gap_len, local, sums, d_rles, capacity, max_seq_len, stream, own_alphabet);
    return hipGetLastError();
}

} // namespace kbo
