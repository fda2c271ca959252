// derand_kernels.hip — gfx950 (MI355X, CDNA4): A5+A6, derandomize_ms_vec (derandomize.rs:269-288) fused with
// translate_ms_vec (translate.rs:263-293) and, optionally, format::relative_to_ref (format.rs:266-287):
//   derand_translate_lds_kernel        reads: one wave per 64 sequences, staged through LDS
//   derand_translate_piece_lds_kernel  long reads / contigs: one lane per piece, staged through LDS
//   derand_translate_kernel            one lane per sequence (fallback, and i32 derandomised output)
//   dl_* kernels                       one very long sequence: three-level scan over chunk tables
//   translate_kernel                   A6 alone (stencil form)
// Integer / byte work only.  Wavefront = 64 lanes.
#include "device_util.hpp"

#include <cstdlib>

#include <algorithm>

namespace kbo {
namespace {

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
__device__ __forceinline__ uint32_t translate_char(int xm, int xc, int xn, uint32_t rel, uint32_t len, int K, int T)
{
    // branch-free; 0 < v < T is written (unsigned)(v - 1) < (unsigned)(T - 1), 2 <= rel <= len-2 as
    // (rel - 2) < (len - 3) (len >= 3)
    const uint32_t Tm1 = (uint32_t)(T - 1);
    const int prev = rel > 1u ? xm : K;
    const int next = rel < len - 1u ? xn : xc;
    const bool inherits = (rel - 2u) < (len - 3u) && xm > T && (uint32_t)(xc - 1) < Tm1;
    const bool own = xc > T && (uint32_t)(next - 1) < Tm1;
    const uint32_t plain = xc <= 0 ? ((next == 1 && prev > 0) ? (uint32_t)'X' : (uint32_t)'-') : (uint32_t)'M';
    return (inherits || own) ? (uint32_t)'R' : plain;
}

struct DtState {
    int x_cur, x_next, x_prev;
};

// one position of the right-to-left pass; J = byte inside the current 16-byte block
template <int J>
__device__ __forceinline__ void dt_step(DtState &st, const uint4 &cur, const uint4 &below, const uint4 &rcur,
                                        uint4 &oblk, uint32_t p, uint32_t len, int K, int T, bool fmt,
                                        int32_t *derand_out_p)
{
    if (p >= len) return; // only in the topmost block
    if (p == len - 1) {   // derandomize.rs:282
        const int a = (int)blk_byte<J>(cur);
        st.x_cur = a > T ? a : 0;
        st.x_next = st.x_cur;
    }
    if (p > 0) { // x[p-1] from noisy[p-1] and x[p] (derandomize.rs:233-246)
        const int a = (int)(J > 0 ? blk_byte<(J + 15) & 15>(cur) : blk_byte<15>(below));
        st.x_prev = (a == K) ? K : ((a > T && st.x_cur < a) ? a : st.x_cur - 1);
    }
    uint32_t ch = translate_char(st.x_prev, st.x_cur, st.x_next, p, len, K, T);
    if (fmt) // format::relative_to_ref: M,R keep the reference base, X and '-' become '-'
        ch = (ch == 'M' || ch == 'R') ? blk_byte<J>(rcur) : (uint32_t)'-';
    blk_or_byte<J>(oblk, ch);
    if (derand_out_p) *derand_out_p = st.x_cur;
    st.x_next = st.x_cur;
    st.x_cur = st.x_prev;
}

// the same for a position with 2 <= p <= len-2 whose block lies wholly inside the sequence and is
// not its first: no position tests at all (most blocks of a long sequence)
template <int J>
__device__ __forceinline__ void dt_step_mid(DtState &st, const uint4 &cur, const uint4 &below, const uint4 &rcur,
                                            uint4 &oblk, int K, int T, bool fmt)
{
    const int a = (int)(J > 0 ? blk_byte<(J + 15) & 15>(cur) : blk_byte<15>(below));
    st.x_prev = (a == K) ? K : ((a > T && st.x_cur < a) ? a : st.x_cur - 1);
    const uint32_t Tm1 = (uint32_t)(T - 1);
    const bool is_r = (st.x_prev > T && (uint32_t)(st.x_cur - 1) < Tm1) || (st.x_cur > T && (uint32_t)(st.x_next - 1) < Tm1);
    const uint32_t plain = st.x_cur <= 0 ? ((st.x_next == 1 && st.x_prev > 0) ? (uint32_t)'X' : (uint32_t)'-') : (uint32_t)'M';
    uint32_t ch = is_r ? (uint32_t)'R' : plain;
    if (fmt) ch = (ch == 'M' || ch == 'R') ? blk_byte<J>(rcur) : (uint32_t)'-';
    blk_or_byte<J>(oblk, ch);
    st.x_next = st.x_cur;
    st.x_cur = st.x_prev;
}

#define KBO_DT_BLOCK_GUARDED(DOUT)                                                                                    \
    {                                                                                                                \
        KBO_DT(15, DOUT) KBO_DT(14, DOUT) KBO_DT(13, DOUT) KBO_DT(12, DOUT) KBO_DT(11, DOUT) KBO_DT(10, DOUT)        \
        KBO_DT(9, DOUT) KBO_DT(8, DOUT) KBO_DT(7, DOUT) KBO_DT(6, DOUT) KBO_DT(5, DOUT) KBO_DT(4, DOUT)              \
        KBO_DT(3, DOUT) KBO_DT(2, DOUT) KBO_DT(1, DOUT) KBO_DT(0, DOUT)                                              \
    }
#define KBO_DT(J, DOUT) dt_step<J>(st, cur, below, rcur, oblk, p0 + J, len, K, T, fmt, (DOUT) ? (DOUT) + p0 + J : nullptr);
#define KBO_DT_BLOCK_MID                                                                                             \
    {                                                                                                                \
        KBO_DM(15) KBO_DM(14) KBO_DM(13) KBO_DM(12) KBO_DM(11) KBO_DM(10) KBO_DM(9) KBO_DM(8)                        \
        KBO_DM(7) KBO_DM(6) KBO_DM(5) KBO_DM(4) KBO_DM(3) KBO_DM(2) KBO_DM(1) KBO_DM(0)                              \
    }
#define KBO_DM(J) dt_step_mid<J>(st, cur, below, rcur, oblk, K, T, fmt);

// One lane per sequence, right to left, one 16-byte block (relative to the sequence start,
// unaligned global accesses) at a time with the block below it already in flight; inside a
// block the 16 positions are unrolled so every byte access is a constant bit-field.
__device__ __forceinline__ void derand_one_sequence(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t s, uint32_t k,
    uint32_t t, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out, int32_t *__restrict__ derand_out, uint32_t max_len)
{
    const uint64_t b = off[s], e = off[s + 1];
    const uint32_t len = (uint32_t)(e - b);
    if (len < 3) return;       // the host side rejects these (derandomize.rs:276)
    if (len > max_len) return; // longer sequences take the chunked path (launch_derand_long)
    const int K = (int)k, T = (int)t;
    const uint8_t *msb = ms + b;
    const bool fmt = ref != nullptr;
    const uint8_t *refb = fmt ? ref + b : msb;
    uint8_t *outb = out + b;
    int32_t *dout = derand_out ? derand_out + b : nullptr;
    const uint32_t nblk = (len + 15u) >> 4;
    uint4 cur = ld16u(msb, 16u * (nblk - 1u));
    uint4 rcur = fmt ? ld16u(refb, 16u * (nblk - 1u)) : make_uint4(0, 0, 0, 0);
    DtState st{0, 0, K};
    for (uint32_t bi = nblk; bi-- > 0;) {
        uint4 below = cur, rbelow = rcur;
        if (bi > 0) {
            below = ld16u(msb, 16u * (bi - 1u));
            if (fmt) rbelow = ld16u(refb, 16u * (bi - 1u));
        }
        uint4 oblk = make_uint4(0, 0, 0, 0);
        const uint32_t p0 = 16u * bi;
        if (bi >= 1u && p0 + 17u <= len && !dout) KBO_DT_BLOCK_MID
        else KBO_DT_BLOCK_GUARDED(dout)
        if (p0 + 16u <= len) st16u(outb, p0, oblk);
        else st_partial(outb + p0, oblk, len - p0); // topmost, partial block of the sequence
        cur = below;
        rcur = rbelow;
    }
}

__global__ __launch_bounds__(256) void derand_translate_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t k,
    uint32_t t, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out, int32_t *__restrict__ derand_out,
    uint32_t max_len, const uint32_t *__restrict__ only)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seqs) return;
    if (only && !only[s]) return; // second pass of the piece-wise path: flagged sequences only
    derand_one_sequence(ms, off, s, k, t, ref, out, derand_out, max_len);
}

// behind map_reads_kernel (map_kernels.hip): the sequences with flags[s] != 0 - a few per cent, one or two per wave of that
// kernel - gathered per block of 4096 into a list in LDS first, so that the lanes that run the pass sit in the same waves (one
// lane per sequence straight off the flags: nearly every wave runs the whole pass for its one flagged lane, 115 us at C2)
constexpr uint32_t kFlaggedPerBlock = 4096;
__global__ __launch_bounds__(256) void derand_flagged_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t k, uint32_t t,
    const uint8_t *__restrict__ ref, uint8_t *__restrict__ out, uint32_t max_len, const uint8_t *__restrict__ flags,
    uint32_t *__restrict__ run_counts)
{
    __shared__ uint32_t list[kFlaggedPerBlock];
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const uint32_t i0 = blockIdx.x * kFlaggedPerBlock + threadIdx.x * 16u;
    if (i0 < n_seqs) {
        const uint4 f = *reinterpret_cast<const uint4 *>(flags + i0); // (the flag array is padded to 16 bytes)
#pragma unroll
        for (uint32_t b = 0; b < 16; b++) {
            const uint32_t w = (b >> 2) == 0 ? f.x : (b >> 2) == 1 ? f.y : (b >> 2) == 2 ? f.z : f.w;
            if (((w >> (8u * (b & 3u))) & 0xFFu) != 0 && i0 + b < n_seqs) list[atomicAdd(&cnt, 1u)] = i0 + b;
        }
    }
    __syncthreads();
    const uint32_t c = cnt;
    for (uint32_t j = threadIdx.x; j < c; j += blockDim.x) {
        const uint32_t s = list[j];
        derand_one_sequence(ms, off, s, k, t, ref, out, nullptr, max_len);
        // kbo::find with max_gap_len = 0 (format.rs:143-193): the runs of this sequence - the maximal stretches without '-' -
        // counted by the lane that has just written its characters (unformatted: M - X R, bit 5 says '-'), 16 at a time; a kernel
        // of its own that looked at every sequence's flag for this took 0.14 ms per 5 M reads
        if (run_counts) {
            const uint64_t b = off[s];
            const uint32_t len = (uint32_t)(off[s + 1] - b);
            uint32_t n = 0, prev_gap = 1;
            for (uint32_t i0 = 0; i0 < len; i0 += 16u) {
                const uint4 v = ld16u(out + b, i0); // (reads <= 15 bytes behind the sequence: the buffer is padded)
                uint32_t E = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4u; q++) {
                    const uint32_t d = q == 0 ? v.x : q == 1 ? v.y : q == 2 ? v.z : v.w;
                    E |= ((((d >> 5) & 0x01010101u) * 0x01020408u) >> 24) << (4u * q);
                }
                const uint32_t nv = min(16u, len - i0), valid = nv == 16u ? 0xFFFFu : (1u << nv) - 1u;
                n += (uint32_t)__popc(~E & ((E << 1) | prev_gap) & valid); // a character that is no gap behind one that is
                prev_gap = (E >> 15) & 1u;
            }
            run_counts[s] = len < 3u ? 0u : n;
        }
    }
}

// ---- LDS-staged variant for batches of short sequences (reads) -------------------------
// One wave per workgroup handles 64 consecutive sequences, whose bytes are contiguous in the
// concatenated buffers: the wave copies that span HBM -> LDS with coalesced 16-byte accesses,
// every lane runs the right-to-left pass over its own sequence inside LDS (bytes in place:
// MS value in, character out), and the wave copies the span back out, applying
// format::relative_to_ref on the way when a reference is given.  Global traffic is fully
// coalesced (the per-lane kernel above issues one 16-byte request per lane instead).
__device__ __forceinline__ uint32_t fmt_word(uint32_t ch, uint32_t rf)
{ // per byte: ch in {'M','R'} ? rf : '-'
    const uint32_t xm = ch ^ 0x4D4D4D4Du, xr = ch ^ 0x52525252u; // zero byte where equal
    const uint32_t zm = ~(((xm & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xm | 0x7F7F7F7Fu);
    const uint32_t zr = ~(((xr & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | xr | 0x7F7F7F7Fu);
    const uint32_t hi = zm | zr;                 // 0x80 in matching bytes
    const uint32_t mask = (hi >> 7) * 0xFFu;     // 0xFF in matching bytes
    return (rf & mask) | (0x2D2D2D2Du & ~mask);
}

// SKEW: the LDS image gets 4 bytes of padding after every 128 bytes.  Lanes touch position p of their own
// sequence in the same step, so with sequences whose common length is a multiple of 32 bytes the flat image
// puts 8..64 lanes on one bank (reads of 128 or 256 bases: 3.2x slower); the padding spreads them.
template <bool SKEW>
__global__ __launch_bounds__(256) void derand_translate_lds_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t k,
    uint32_t t, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out, uint32_t lds_bytes, uint32_t wave_lds)
{
    // (the waves of a workgroup are independent - each has its own slice of the LDS and its own 64 sequences - and wait for
    // nobody: wave barriers around the staged copy instead of workgroup barriers, C2 0.137 -> 0.127 ms; one, two or four
    // waves a workgroup make no difference)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    uint8_t *lds = lds_all + (threadIdx.x >> 6) * wave_lds;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t s0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64u;
    if (s0 >= n_seqs) return;
    const uint32_t s = s0 + lane;
    const uint32_t s_end = min(s0 + 64u, n_seqs);
    const uint64_t base = off[s0];
    const uint32_t span = (uint32_t)(off[s_end] - base);
    if (span > lds_bytes) return; // cannot happen: the host sizes lds_bytes from the longest sequence
    const int K = (int)k, T = (int)t;
    auto at = [&](uint32_t x) -> uint8_t & { return lds[SKEW ? x + ((x >> 7) << 2) : x]; };

    for (uint32_t o = lane * 16u; o < span; o += 1024u) { // stage in (reads <= 15 B past the span)
        const uint4 v = ld16u(ms + base, o);
        if (SKEW) { // a 16-byte chunk never straddles a 128-byte granule, but it is only 4-byte aligned
            uint32_t *d = reinterpret_cast<uint32_t *>(&at(o));
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
            *reinterpret_cast<uint4 *>(lds + o) = v;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    if (s < n_seqs) {
        const uint32_t b = (uint32_t)(off[s] - base), len = (uint32_t)(off[s + 1] - off[s]);
        if (len >= 3) {
            // Branch-free pass with the window (x_prev, x_cur, x_next) = x[p-1], x[p], x[p+1]:
            //   in(v) = 0 < v < T, written (unsigned)(v - 1) < (unsigned)(T - 1);
            //   'R' at p  <=>  (x[p-1] > T && in(x[p]) && 2 <= p < len-1)  ||  (x[p] > T && in(next));
            // the two ends (p = len-1, where next = x[p]; p = 1 and 0, where prev = K and no 'R' is
            // inherited from below) are peeled so that the middle of the sequence carries no
            // position tests.  gt_* / in_* flags move down the window with the values.
            const uint32_t Tm1 = (uint32_t)(T - 1);
            auto step = [&](int a, int x_cur) { return (a == K) ? K : ((a > T && x_cur < a) ? a : x_cur - 1); };
            auto plain = [&](int x_cur, int next, int prev) -> uint32_t { // translate.rs:180-216 without the 'R' cases
                return x_cur <= 0 ? ((next == 1 && prev > 0) ? (uint32_t)'X' : (uint32_t)'-') : (uint32_t)'M';
            };
            int a = at(b + len - 1);
            int x_cur = a > T ? a : 0; // derandomize.rs:282
            int a_below = at(b + len - 2);
            int x_prev = step(a_below, x_cur);
            // p = len-1: next = x_cur itself; inherits 'R' from below (len-1 >= 2 always holds, but
            // the rule needs pos < len-1, so it does not apply here)
            bool in_cur = (uint32_t)(x_cur - 1) < Tm1, gt_cur = x_cur > T;
            at(b + len - 1) = (uint8_t)((gt_cur && in_cur) ? (uint32_t)'R' : plain(x_cur, x_cur, x_prev));
            int x_next = x_cur;
            bool in_next = in_cur;
            x_cur = x_prev;
            gt_cur = x_cur > T;
            in_cur = (uint32_t)(x_cur - 1) < Tm1;
            a_below = at(b + len - 3);
#pragma unroll 4
            for (uint32_t p = len - 2; p >= 2; p--) { // middle: 2 <= p <= len-2
                x_prev = step(a_below, x_cur);
                a_below = at(b + p - 2); // p >= 2
                const bool gt_prev = x_prev > T;
                const bool is_r = (gt_prev && in_cur) || (gt_cur && in_next);
                at(b + p) = (uint8_t)(is_r ? (uint32_t)'R' : plain(x_cur, x_next, x_prev));
                x_next = x_cur;
                in_next = in_cur;
                x_cur = x_prev;
                gt_cur = gt_prev;
                in_cur = (uint32_t)(x_cur - 1) < Tm1;
            }
            // p = 1: prev = K (translate.rs:277 tests pos > 1), no 'R' inherited (needs pos >= 2)
            x_prev = step(a_below, x_cur); // a_below == at(b + 0)
            at(b + 1) = (uint8_t)((gt_cur && in_next) ? (uint32_t)'R' : plain(x_cur, x_next, K));
            // p = 0
            x_next = x_cur;
            in_next = in_cur;
            x_cur = x_prev;
            at(b + 0) = (uint8_t)((x_cur > T && in_next) ? (uint32_t)'R' : plain(x_cur, x_next, K));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    for (uint32_t o = lane * 16u; o < span; o += 1024u) { // stage out
        uint4 c;
        if (SKEW) {
            const uint32_t *d = reinterpret_cast<const uint32_t *>(&at(o));
            c = make_uint4(d[0], d[1], d[2], d[3]);
        } else {
            c = *reinterpret_cast<const uint4 *>(lds + o);
        }
        if (ref) {
            const uint4 rf = ld16u(ref + base, o);
            c.x = fmt_word(c.x, rf.x);
            c.y = fmt_word(c.y, rf.y);
            c.z = fmt_word(c.z, rf.z);
            c.w = fmt_word(c.w, rf.w);
        }
        if (o + 16u <= span) st16u(out + base, o, c);
        else st_partial(out + base + o, c, span - o);
    }
}

// ---- piece-wise, LDS-staged variant for batches of long reads / contigs -----------------
// One lane per piece of kDtPiece positions of a sequence, so that a few thousand sequences of
// 10 kbp still fill the device.  The recurrence runs right to left, so a piece needs x at its
// upper end: the lane looks for the nearest position at or above the piece's end whose value is
// known without context - a hard reset (noisy == k gives x = k whatever follows,
// derandomize.rs:235-238) or the sequence's last position (derandomize.rs:282) - and runs the
// recurrence from there down to the piece.  Exact whenever such a position lies within
// kDtLookahead positions (inside matches every position is a reset); otherwise the sequence is
// flagged and redone by one lane in a second launch (adversarial inputs: long stretches without a
// single full-length match).  One wave takes 64 consecutive pieces (a contiguous span of <= 16 KB
// of the concatenated buffers) and stages it through LDS like the read kernel above: coalesced
// 16-byte traffic (one 16-byte request per lane and block, measured first, is bound by the
// L2-miss request rate at 8x the bytes).  Phase 1: every lane reads what it
// needs from OUTSIDE its piece - the look-ahead up to the nearest hard reset and the value just
// below the piece - and derives its start state; barrier; phase 2: it overwrites its piece in
// place (MS value in, character out); barrier; the span is copied out, relative_to_ref applied
// on the way.  Pieces that give up (no reset in reach) leave their MS bytes in place and flag the
// sequence, which the per-lane kernel redoes afterwards.
// 132 = 33 words: consecutive lanes' pieces start one LDS bank apart (a 256-byte stride puts all 64
// lanes of a step on the same bank: measured 0.67 ms against 0.44 ms for 260 on 200 Mbp); 10 KB
// of LDS per wave keeps 4 waves per SIMD resident
constexpr uint32_t kDtPiece = 132, kDtLookahead = 1024;
constexpr uint32_t kDtBehind = 16; // staged bytes below the span (the value just below the first piece)

__global__ __launch_bounds__(64) void derand_translate_piece_lds_kernel(
    const uint8_t *__restrict__ ms, const uint64_t *__restrict__ off, uint32_t n_seqs, uint64_t total,
    const WalkItem *__restrict__ pieces, uint32_t n_pieces, uint32_t k, uint32_t t, const uint8_t *__restrict__ ref,
    uint8_t *__restrict__ out, uint32_t max_len, uint32_t *__restrict__ redo)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint32_t span_len_sh;
    const uint32_t lane = threadIdx.x;
    const uint32_t first = blockIdx.x * 64u;
    WalkItem it;
    it.start = 0;
    it.len = 0;
    it.warm = 0;
    if (first + lane < n_pieces) it = pieces[first + lane];
    const uint64_t span_base = pieces[first].start; // wave-uniform; slots behind the last piece are empty
    if (lane == 0) span_len_sh = 0;
    __syncthreads();
    if (it.len) atomicMax(&span_len_sh, (uint32_t)(it.start + it.len - span_base));
    __syncthreads();
    const uint32_t span_len = span_len_sh;
    if (span_len == 0) return;
    const uint64_t stage_lo = span_base >= kDtBehind ? span_base - kDtBehind : 0;
    const uint32_t head = (uint32_t)(span_base - stage_lo); // 0 or 16
    const uint32_t stage_len = (uint32_t)(min(span_base + span_len + kDtLookahead, total) - stage_lo);
    for (uint32_t o = lane * 16u; o < stage_len; o += 1024u) // stage in (reads <= 15 B past the batch: buffers are padded)
        *reinterpret_cast<uint4 *>(lds + o) = ld16u(ms + stage_lo, o);
    __syncthreads();

    const int K = (int)k, T = (int)t;
    uint32_t len = 0, c0 = 0, c1 = 0;
    uint8_t *row = lds; // LDS address of position 0 of the lane's sequence (may lie below lds: only [c0-1, ..) is touched)
    int x_cur = 0, x_next = 0, a_under = 0;
    bool active = false;
    // the sequence that holds the span's first byte (wave-uniform search), then, per lane, the one that
    // holds its piece: at most 63 sequences further on
    uint32_t s_first = 0;
    {
        uint32_t hi = n_seqs;
        while (hi - s_first > 1) {
            const uint32_t mid = s_first + (hi - s_first) / 2;
            if (off[mid] <= span_base) s_first = mid;
            else hi = mid;
        }
    }
    if (it.len) {
        uint32_t lo = s_first, hi = min(n_seqs, s_first + 64u);
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (off[mid] <= it.start) lo = mid;
            else hi = mid;
        }
        const uint64_t b = off[lo];
        len = (uint32_t)(off[lo + 1] - b);
        c0 = (uint32_t)(it.start - b);
        c1 = c0 + it.len;
        row = lds + (int64_t)(b - stage_lo);
        active = len >= 3 && len <= max_len;
        if (active) {
            a_under = c0 > 0 ? row[c0 - 1u] : 0;
            if (c1 < len) { // x[c1 - 1], x[c1] from the nearest context-free position at or above c1
                const uint32_t last = len - 1u, stop = min(last, c1 + kDtLookahead - 1u);
                uint32_t p = c1;
                while (p < stop && row[p] != (uint8_t)K) p++;
                const int a = row[p];
                if (p != last && a != K) { // nothing context-free in reach
                    redo[lo] = 1u;
                    active = false;
                } else {
                    int x = p == last ? (a > T ? a : 0) : K, xn = x;
                    for (uint32_t q = p; q-- > c1 - 1u;) {
                        const int aq = row[q];
                        xn = x;
                        x = (aq == K) ? K : ((aq > T && x < aq) ? aq : x - 1);
                    }
                    x_cur = x;
                    x_next = xn;
                }
            } else { // the piece holds the sequence's last position (derandomize.rs:282)
                const int a = row[len - 1u];
                x_cur = a > T ? a : 0;
                x_next = x_cur;
            }
        }
    }
    __syncthreads();
    if (active) {
        int a = c1 - 1u > c0 ? (int)row[c1 - 2u] : a_under; // noisy[p - 1] for p = c1 - 1, fetched one step ahead
#pragma unroll 4
        for (uint32_t p = c1; p-- > c0;) { // (a variant without position tests for interior pieces was slower:
            const int a_here = a;         //  its carried flags cost more mask bookkeeping than the tests)
            a = p > c0 + 1u ? (int)row[p - 2u] : a_under; // for the next step (unused after the last one)
            const int x_prev = p > 0 ? ((a_here == K) ? K : ((a_here > T && x_cur < a_here) ? a_here : x_cur - 1)) : K;
            row[p] = (uint8_t)translate_char(x_prev, x_cur, x_next, p, len, K, T);
            x_next = x_cur;
            x_cur = x_prev;
        }
    }
    __syncthreads();
    if ((head & 15u) == 0) {
        for (uint32_t o = lane * 16u; o < span_len; o += 1024u) { // stage out
            uint4 c = *reinterpret_cast<const uint4 *>(lds + head + o);
            if (ref) {
                const uint4 rf = ld16u(ref + span_base, o);
                c.x = fmt_word(c.x, rf.x);
                c.y = fmt_word(c.y, rf.y);
                c.z = fmt_word(c.z, rf.z);
                c.w = fmt_word(c.w, rf.w);
            }
            if (o + 16u <= span_len) st16u(out + span_base, o, c);
            else st_partial(out + span_base + o, c, span_len - o);
        }
    } else { // a span that starts within the first 16 bytes of the batch but not at byte 0: byte by byte
        for (uint32_t o = lane; o < span_len; o += 64u) {
            uint32_t ch = lds[head + o];
            if (ref) ch = (ch == 'M' || ch == 'R') ? ref[span_base + o] : (uint32_t)'-';
            out[span_base + o] = (uint8_t)ch;
        }
    }
}

// ---- one very long sequence: chunked derandomize ---------------------------------------
// The recurrence x[i] = f(noisy[i], x[i+1]) only ever compares x with values in (t, k], so
// for decisions x matters through the finite state  S(x) = x > t ? x - t : 0  (k - t + 1
// states); a value <= t keeps counting down exactly until some position "fires" (reset
// noisy == k, anchor noisy > t && x < noisy, or the last position's own rule).  Every chunk
// of positions is therefore a function  x_in -> x_out  described by one entry per state:
//     fired ? (exact x_out) : x_in - chunk_len
// Tables compose exactly, which gives a three-level scan: per-chunk tables (parallel),
// per-group tables (parallel over groups x states), one short sequential pass over the
// groups, inputs per chunk (parallel over groups), and the final per-chunk pass that emits
// characters (parallel).  Bit-identical to the sequential loop.
constexpr uint32_t kDlChunk = 128;  // positions per chunk
constexpr uint32_t kDlGroup = 128;  // chunks per group
constexpr int32_t kDlPass = (int32_t)0x80000000; // table value: "not fired, x_out = x_in - len"

__device__ __forceinline__ int dl_step(int a, int x, uint32_t p, uint32_t len, int K, int T, bool &fired)
{
    if (p == len - 1) { fired = true; return a > T ? a : 0; }   // derandomize.rs:282
    if (a == K) { fired = true; return K; }                      // derandomize.rs:235-238
    if (a > T && x < a) { fired = true; return a; }              // derandomize.rs:240-244
    return x - 1;
}
__device__ __forceinline__ uint32_t dl_state(int x, int T) { return x > T ? (uint32_t)(x - T) : 0u; }
__device__ __forceinline__ int dl_apply(int32_t entry, int x_in, uint32_t span) { return entry == kDlPass ? x_in - (int)span : entry; }

// lane = (chunk, state)
__global__ void dl_chunk_tables_kernel(const uint8_t *__restrict__ ms, uint32_t len, uint32_t k, uint32_t t,
                                       uint32_t n_chunks, uint32_t n_states, int32_t *__restrict__ t1)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_chunks * n_states) return;
    const uint32_t c = id / n_states, st = id % n_states;
    const int K = (int)k, T = (int)t;
    const uint32_t p0 = c * kDlChunk, p1 = min(len, p0 + kDlChunk);
    int x = T + (int)st; // representative of the state (st == 0: any value <= t)
    bool fired = false;
    for (uint32_t p = p1; p-- > p0;) x = dl_step(ms[p], x, p, len, K, T, fired);
    t1[id] = fired ? x : kDlPass;
}

// lane = (group, state): compose the chunk tables of the group, right to left
__global__ void dl_group_tables_kernel(const int32_t *__restrict__ t1, uint32_t len, uint32_t t, uint32_t n_chunks,
                                       uint32_t n_groups, uint32_t n_states, int32_t *__restrict__ t2)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_groups * n_states) return;
    const uint32_t g = id / n_states, st = id % n_states;
    const int T = (int)t;
    const uint32_t c0 = g * kDlGroup, c1 = min(n_chunks, c0 + kDlGroup);
    int x = T + (int)st;
    bool fired = false;
    for (uint32_t c = c1; c-- > c0;) {
        const uint32_t span = min(len, (c + 1) * kDlChunk) - c * kDlChunk;
        const int32_t e = t1[c * n_states + dl_state(x, T)];
        fired = fired || e != kDlPass;
        x = dl_apply(e, x, span);
    }
    t2[id] = fired ? x : kDlPass;
}

// one lane: exact x entering every group (from its right)
__global__ void dl_top_kernel(const int32_t *__restrict__ t2, uint32_t len, uint32_t t, uint32_t n_groups,
                              uint32_t n_states, int32_t *__restrict__ g_in)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int T = (int)t;
    int x = 0; // irrelevant: the last position's rule ignores it
    for (uint32_t g = n_groups; g-- > 0;) {
        g_in[g] = x;
        const uint32_t p0 = g * kDlGroup * kDlChunk, p1 = min(len, p0 + kDlGroup * kDlChunk);
        x = dl_apply(t2[g * n_states + dl_state(x, T)], x, p1 - p0);
    }
}

// lane = group: exact x entering every chunk of the group
__global__ void dl_chunk_inputs_kernel(const int32_t *__restrict__ t1, const int32_t *__restrict__ g_in, uint32_t len,
                                       uint32_t t, uint32_t n_chunks, uint32_t n_groups, uint32_t n_states,
                                       int32_t *__restrict__ c_in)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int T = (int)t;
    const uint32_t c0 = g * kDlGroup, c1 = min(n_chunks, c0 + kDlGroup);
    int x = g_in[g];
    for (uint32_t c = c1; c-- > c0;) {
        c_in[c] = x;
        const uint32_t span = min(len, (c + 1) * kDlChunk) - c * kDlChunk;
        x = dl_apply(t1[c * n_states + dl_state(x, T)], x, span);
    }
}

// lane = chunk: final pass with the exact incoming value; emits characters (and values)
__global__ void dl_emit_kernel(const uint8_t *__restrict__ ms, const int32_t *__restrict__ c_in, uint32_t len, uint32_t k,
                               uint32_t t, uint32_t n_chunks, const uint8_t *__restrict__ ref, uint8_t *__restrict__ out,
                               int32_t *__restrict__ derand_out)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const int K = (int)k, T = (int)t;
    const uint32_t p0 = c * kDlChunk, p1 = min(len, p0 + kDlChunk);
    int x_next = c_in[c]; // x[p1] (unused when p1 == len)
    bool fired = false;
    int x_cur = dl_step(ms[p1 - 1], x_next, p1 - 1, len, K, T, fired);
    if (p1 == len) x_next = x_cur;
    for (uint32_t p = p1; p-- > p0;) {
        int x_prev = K;
        if (p > 0) x_prev = dl_step(ms[p - 1], x_cur, p - 1, len, K, T, fired);
        uint32_t ch = translate_char(x_prev, x_cur, x_next, p, len, K, T);
        if (ref) ch = (ch == 'M' || ch == 'R') ? ref[p] : (uint32_t)'-';
        out[p] = (uint8_t)ch;
        if (derand_out) derand_out[p] = x_cur;
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
    out[p] = (uint8_t)translate_char(xm, xc, xn, (uint32_t)p, (uint32_t)len, (int)k, (int)t); // len < 2^32 (checked by the caller)
}

} // namespace

size_t derand_piece_work_bytes(uint32_t n_seqs, uint64_t total_bases)
{
    const uint64_t slots = total_bases / kDtPiece + n_seqs;
    return (size_t)(slots * sizeof(WalkItem) + (chunk_items_scratch_words(n_seqs) + n_seqs) * sizeof(uint32_t) + 64);
}

hipError_t launch_derand_translate(const uint8_t *d_ms, const uint64_t *d_offsets, uint32_t n_seqs,
                                   uint32_t k, uint32_t threshold, const uint8_t *d_ref,
                                   uint8_t *d_chars_out, int32_t *d_derand_out, uint32_t max_seq_len,
                                   uint32_t per_lane_max_len, hipStream_t stream, uint64_t total_bases,
                                   void *d_work, size_t work_bytes)
{
    if (n_seqs == 0) return hipSuccess;
    // short sequences: LDS-staged kernel (64 sequences per wave must fit the LDS budget)
    if (max_seq_len > 0 && max_seq_len <= 480 && d_derand_out == nullptr) {
        const uint32_t lds_bytes = ((64u * max_seq_len + 15u) / 16u) * 16u + 16u;
        static const int env_wpb = std::getenv("KBO_DT_WAVES") ? std::atoi(std::getenv("KBO_DT_WAVES")) : 0; // experiments
        const bool skew = max_seq_len % 32u == 0; // e.g. reads of 128 or 256 bases: padded LDS image (see the kernel)
        const uint32_t wave_lds = ((skew ? lds_bytes + lds_bytes / 32u + 16u : lds_bytes) + 15u) / 16u * 16u;
        uint32_t wpb = env_wpb >= 1 && env_wpb <= 4 ? (uint32_t)env_wpb : 4u;
        while (wpb > 1u && wpb * wave_lds > 65536u) wpb--; // (a workgroup's LDS)
        const uint32_t n_waves = (n_seqs + 63u) / 64u;
        if (skew)
            hipLaunchKernelGGL((derand_translate_lds_kernel<true>), dim3((n_waves + wpb - 1u) / wpb), dim3(64u * wpb), wpb * wave_lds, stream,
                               d_ms, d_offsets, n_seqs, k, threshold, d_ref, d_chars_out, lds_bytes, wave_lds);
        else
            hipLaunchKernelGGL((derand_translate_lds_kernel<false>), dim3((n_waves + wpb - 1u) / wpb), dim3(64u * wpb), wpb * wave_lds, stream,
                               d_ms, d_offsets, n_seqs, k, threshold, d_ref, d_chars_out, lds_bytes, wave_lds);
        return hipGetLastError();
    }
    // long reads / contigs with scratch available: one lane per piece, then the flagged sequences again
    const uint64_t slots = total_bases / kDtPiece + n_seqs;
    if (d_work && d_derand_out == nullptr && total_bases > 0 && slots < (1ull << 31) &&
        work_bytes >= derand_piece_work_bytes(n_seqs, total_bases)) {
        WalkItem *pieces = static_cast<WalkItem *>(d_work);
        uint32_t *scratch = reinterpret_cast<uint32_t *>(pieces + slots);
        uint32_t *redo = scratch + chunk_items_scratch_words(n_seqs);
        hipError_t e = hipMemsetAsync(redo, 0, (size_t)n_seqs * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        e = launch_make_chunk_items(d_offsets, n_seqs, kDtPiece, 1u, (uint32_t)slots, pieces, scratch, stream);
        if (e != hipSuccess) return e;
        const uint32_t lds_bytes = kDtBehind + 64u * kDtPiece + kDtLookahead + 32u;
        hipLaunchKernelGGL(derand_translate_piece_lds_kernel, dim3((unsigned)((slots + 63) / 64)), dim3(64), lds_bytes, stream,
                           d_ms, d_offsets, n_seqs, total_bases, pieces, (uint32_t)slots, k, threshold, d_ref, d_chars_out,
                           per_lane_max_len, redo);
        hipLaunchKernelGGL(derand_translate_kernel, dim3((n_seqs + 255) / 256), dim3(256), 0, stream, d_ms, d_offsets,
                           n_seqs, k, threshold, d_ref, d_chars_out, (int32_t *)nullptr, per_lane_max_len,
                           (const uint32_t *)redo);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(derand_translate_kernel, dim3((n_seqs + 255) / 256), dim3(256), 0, stream, d_ms,
                       d_offsets, n_seqs, k, threshold, d_ref, d_chars_out, d_derand_out, per_lane_max_len,
                       (const uint32_t *)nullptr);
    return hipGetLastError();
}

hipError_t launch_derand_flagged(const uint8_t *d_ms, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t k, uint32_t threshold,
                                 const uint8_t *d_ref, uint8_t *d_chars_out, const uint8_t *d_flags, uint32_t max_seq_len, hipStream_t stream,
                                 uint32_t *d_run_counts)
{
    if (n_seqs == 0) return hipSuccess;
    hipLaunchKernelGGL(derand_flagged_kernel, dim3((n_seqs + kFlaggedPerBlock - 1u) / kFlaggedPerBlock), dim3(256), 0, stream, d_ms, d_offsets,
                       n_seqs, k, threshold, d_ref, d_chars_out, max_seq_len, d_flags, d_ref ? nullptr : d_run_counts);
    return hipGetLastError();
}

size_t derand_long_scratch_bytes(uint64_t len, uint32_t k, uint32_t threshold)
{
    const uint64_t n_chunks = (len + kDlChunk - 1) / kDlChunk, n_groups = (n_chunks + kDlGroup - 1) / kDlGroup;
    const uint64_t n_states = (uint64_t)(k > threshold ? k - threshold : 0) + 1;
    return (size_t)((n_chunks * n_states + n_groups * n_states + n_groups + n_chunks) * sizeof(int32_t) + 64);
}

hipError_t launch_derand_long(const uint8_t *d_ms, uint32_t len, uint32_t k, uint32_t threshold, const uint8_t *d_ref,
                              uint8_t *d_chars_out, int32_t *d_derand_out, void *d_scratch, hipStream_t stream)
{
    if (len < 3) return hipSuccess;
    const uint32_t n_chunks = (len + kDlChunk - 1) / kDlChunk, n_groups = (n_chunks + kDlGroup - 1) / kDlGroup;
    const uint32_t n_states = (k > threshold ? k - threshold : 0) + 1;
    int32_t *t1 = static_cast<int32_t *>(d_scratch);
    int32_t *t2 = t1 + (size_t)n_chunks * n_states;
    int32_t *g_in = t2 + (size_t)n_groups * n_states;
    int32_t *c_in = g_in + n_groups;
    const uint32_t T = 256;
    hipLaunchKernelGGL(dl_chunk_tables_kernel, dim3((n_chunks * n_states + T - 1) / T), dim3(T), 0, stream, d_ms, len, k,
                       threshold, n_chunks, n_states, t1);
    hipLaunchKernelGGL(dl_group_tables_kernel, dim3((n_groups * n_states + T - 1) / T), dim3(T), 0, stream, t1, len,
                       threshold, n_chunks, n_groups, n_states, t2);
    hipLaunchKernelGGL(dl_top_kernel, dim3(1), dim3(64), 0, stream, t2, len, threshold, n_groups, n_states, g_in);
    hipLaunchKernelGGL(dl_chunk_inputs_kernel, dim3((n_groups + T - 1) / T), dim3(T), 0, stream, t1, g_in, len, threshold,
                       n_chunks, n_groups, n_states, c_in);
    hipLaunchKernelGGL(dl_emit_kernel, dim3((n_chunks + T - 1) / T), dim3(T), 0, stream, d_ms, c_in, len, k, threshold,
                       n_chunks, d_ref, d_chars_out, d_derand_out);
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
