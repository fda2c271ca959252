// long_kernels.hip — gfx950 (MI355X, CDNA4): kbo::map / kbo::matches for sequences of ANY length in one launch.
//
// The chain the reference runs per sequence (lib.rs:735-738 for map, lib.rs:624-627 for matches / find):
//     index::query_sbwt (index.rs:243-256)  ->  derandomize_ms_vec (derandomize.rs:269-288)  ->  translate_ms_vec
//     (translate.rs:263-293)  [-> format::relative_to_ref (format.rs:266-287)]
// for contigs, whole reference sequences (kbo::map streams one through the query's index: lib.rs:720-761) and long reads.
// map_kernels.hip does this for reads of up to 160 bases, one LANE per read; here ONE WAVE takes a PIECE of a sequence: `own`
// bases [s, s + n) inside a region [s - k, s + n + k + 1) of at most 1008 bases, lane w holding the 16 positions of word w.
//
//   map_long_kernel  0. the region is loaded once (16 B per lane), turned into 2-bit digits, kept in LDS;
//                    1. STRETCHES: intervals of the region that equal a path of the index's text on one diagonal.  A seed
//                       (DevIndexView::seed_pos, or the anchors of the depth table) gives a diagonal; the 2-bit text around it
//                       (pc_tm) is staged in LDS once and the whole region compared with it in one step (a word per lane).
//                       Where 6 of 16 bases mismatch the diagonal is lost: the 64 lanes try the 64 diagonals beside it (an
//                       insertion or deletion of up to 32 bases) on the 32 bases behind the loss - the diagonal on which the read
//                       goes on soonest wins - and only then the seed table again.  The first loss of a piece tries all its
//                       words at once on the 13 diagonals around the lost one (an insertion or deletion every 80 bases: one pass
//                       instead of a dozen rounds).  Consecutive diagonals alternate between two
//                       BIT PLANES (1 = the base lies in a stretch of that plane), so that two stretches may overlap (an
//                       insertion inside a run of equal bases lies on both diagonals);
//                    2. everything else is a function of the two planes, word-parallel (tools/model/long_form.py has the
//                       derivation and checks it against the oracle):
//                         cov(i)  i lies in a stretch of more than t bases          G(i)  ... at depth > t
//                         chars   'M' where cov; else 'X' if cov(i + 1) and (i <= 1 or cov(i - 1)) else '-';
//                                 'R','R' at i, i + 1 where G(i), not G(i + 1), cov(i + 1)     (translate.rs:195-203, :282-288)
//                         U(e)    the window of `order` bases that ends at e lies in no single stretch;
//                    3. PROOF from the depth table (dtab_kernels.hip): the characters above are derandomize_ms_vec + translate_ms_vec
//                       of the true matching statistics provided that no string of t + 1 bases that lies in no single stretch is in
//                       the index.  Per maximal run of U: its first end, every c-th from there (c = t - order: a window may then
//                       be present as long as nothing deeper than order + 1 ends there), and its last end, are looked up - behind
//                       the filter where the copy has one; a window that is present AND extended to the left by the read's base
//                       (order + 1 bases in the index) is ruled out by any pair of absent strings of order + 1 bases around it - one
//                       ending j bases back, one i bases on, i + j <= c + 1 (j = 1 alone will do); the run's last window - it
//                       starts at the last base in front of the next stretch - needs the window one base on not to be extended
//                       by that base.  The wave's look-ups are dealt to its lanes; a piece whose proof fails is FLAGGED;
//                    4. the characters of the own bases leave in whole lines; x at the piece's first own base goes to xin[]
//                       (what the piece to the left needs should it be flagged: worked out only for a wave's first piece and
//                       behind a piece the wave has just flagged).
//   flagged pieces   long_redo_items_kernel lists them in sub-items of 32 bases (8 / 16 in small batches) for the plain walk (walk_kernels.hip), which gives
//                    their matching statistics; long_derand_kernel runs the literal recurrences over them right to left, from
//                    xin[] of the piece to the right (runs of flagged pieces in one go).
//
// Integer / bit work only: no MFMA.  Four independent waves per workgroup, 2 KB of LDS each.
#include "device_util.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace kbo {
std::atomic<int> g_map_long{1}; // kbo_set_map_long: 0 = never, 1 = where it applies, 2 = ... and every piece flagged (tests of the second pass)
void set_map_long(int mode) { g_map_long = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
namespace {

constexpr uint32_t kLongTextUnits = 76; // staged units of 16 text positions: 64 in front of the region's diagonal, 1024, 128 behind
constexpr uint32_t kLongListCap = 384;  // look-ups a piece may ask for (more: flagged)
// experiments (KBO_LONG_X: phases left out, cycle stamps): compiled in only with -DKBO_LONG_EXPERIMENTS - their state lives across the
// whole piece loop in a kernel that spills scalar registers as it is
#ifdef KBO_LONG_EXPERIMENTS
constexpr bool kLongExp = true;
#else
constexpr bool kLongExp = false;
#endif
constexpr uint32_t kLongTH = 6;         // mismatches in 16 bases that end a diagonal
constexpr uint32_t kLongRun = 10;       // matching bases that start one
constexpr uint32_t kLongLds = 2048;     // bytes of LDS per wave
#ifndef KBO_LONG_WPE
#define KBO_LONG_WPE 4
#endif

__device__ __forceinline__ uint32_t funnel2(uint32_t hi, uint32_t lo, uint32_t r) // 16 digits from digit r of hi on
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (32u - 2u * r));
}
// mismatch bits on the digit grid (bit 2 (15 - j) for base j) -> bit j for base j
__device__ __forceinline__ uint32_t grid_to_mask16(uint32_t m)
{
    m &= 0x55555555u;
    m = (m | (m >> 1)) & 0x33333333u;
    m = (m | (m >> 2)) & 0x0F0F0F0Fu;
    m = (m | (m >> 4)) & 0x00FF00FFu;
    m = (m | (m >> 8)) & 0x0000FFFFu;
    return __builtin_bitreverse32(m) >> 16;
}
// bit j for base j -> bit 2 (15 - j) on the digit grid
__device__ __forceinline__ uint32_t grid_from_mask16(uint32_t m)
{
    m = __builtin_bitreverse32(m) >> 16; // base j at bit 15 - j
    m = (m | (m << 8)) & 0x00FF00FFu;
    m = (m | (m << 4)) & 0x0F0F0F0Fu;
    m = (m | (m << 2)) & 0x33333333u;
    m = (m | (m << 1)) & 0x55555555u;
    return m;
}
// bits t of a word at grid position xa with lo <= xa + t < hi
__device__ __forceinline__ uint32_t range16(int32_t xa, int32_t lo, int32_t hi)
{
    const int32_t a = min(max(lo - xa, 0), 16), b = min(max(hi - xa, 0), 16);
    return b > a ? (((1u << b) - 1u) & ~((1u << a) - 1u)) : 0u;
}
// AND over the L positions that end at each bit (bits in front of bit 0 count as 0), 1 <= L <= 63: by doubling up to the largest
// power of two p <= L, then two windows of p (see ero_at)
__device__ __forceinline__ uint64_t erode_end(uint64_t h, uint32_t L)
{
    const uint32_t lg = 31u - (uint32_t)__builtin_clz(L);
    for (uint32_t w = 0; w < lg; w++) h &= h << (1u << w);
    return h & (h << (L - (1u << lg)));
}
// OR over the L positions that start at each bit (bits behind bit 63 count as 0)
__device__ __forceinline__ uint64_t dilate_fwd(uint64_t f, uint32_t L)
{
    const uint32_t lg = 31u - (uint32_t)__builtin_clz(L);
    for (uint32_t w = 0; w < lg; w++) f |= f >> (1u << w);
    return f | (f >> (L - (1u << lg)));
}
// 4 bits -> 4 bytes of 0xFF / 0x00
__device__ __forceinline__ uint32_t spread4(uint32_t m4) { return (((m4 & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu; }

// the value of the lane d (1 .. 3) below / above, 0 where there is no such lane: wave shifts of the data-parallel primitives (one
// vector instruction each, no trip through the LDS crossbar as ds_bpermute makes)
__device__ __forceinline__ uint32_t shfl_up0(uint32_t v, uint32_t d, uint32_t)
{
    v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true); // wave_shr:1
    if (d >= 2u) v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
    if (d >= 3u) v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);
    return v;
}
__device__ __forceinline__ uint32_t shfl_down0(uint32_t v, uint32_t d, uint32_t)
{
    v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); // wave_shl:1
    if (d >= 2u) v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
    if (d >= 3u) v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);
    return v;
}

// Pieces of a sequence whose first byte is byte b of the buffer: the first one's own bases end where a 16-byte block of the buffer
// ends, the others' own bases are `own` (a multiple of 16) each - so that every piece but a sequence's first and last stores its
// characters in whole aligned blocks (one store instruction a piece; partial blocks only at the two ends of a sequence)
__device__ __forceinline__ uint32_t long_first_own(uint64_t b, uint32_t own) { return own - (uint32_t)((b + own) & 15u); }
// the minimum over the wave's 64 lanes, in every lane: swaps inside quads, mirrors inside rows of 16, two broadcasts across the rows
// - six vector instructions, no trip through the LDS crossbar (tools/ubench/dpp_min.hip checks the encodings on the device)
__device__ __forceinline__ uint32_t wave_min_dpp(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false)); // row_half_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false)); // row_mirror
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false)); // row_bcast:15 into rows 1, 3
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false)); // row_bcast:31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(256) void long_count_kernel(const uint64_t *__restrict__ off, uint32_t n_seqs, uint32_t own, uint32_t *__restrict__ counts)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_seqs) return;
    uint32_t n = 0;
    if (s < n_seqs) {
        const uint64_t b = off[s], len = off[s + 1] - b;
        const uint32_t l1 = long_first_own(b, own);
        n = len == 0 ? 0u : (len <= l1 ? 1u : 1u + (uint32_t)((len - l1 + own - 1u) / own));
    }
    counts[s] = n;
}

// piece t -> { first byte of its region, that byte's position in its sequence, the sequence's length, own0 | own_n << 10 }
__global__ __launch_bounds__(256) void long_items_kernel(const uint64_t *__restrict__ off, const uint32_t *__restrict__ local, const uint32_t *__restrict__ sums,
                                                         uint32_t n_seqs, uint32_t own, uint32_t cb, uint32_t n_slots, uint4 *__restrict__ items,
                                                         uint32_t *__restrict__ n_pieces)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    auto first_item = [&](uint32_t s) { return sums[s / kScanBlock] + local[s]; };
    if (t == 0) *n_pieces = first_item(n_seqs);
    if (t >= n_slots) return;
    uint4 it = make_uint4(0, 0, 0, 0);
    if (t < first_item(n_seqs)) {
        uint32_t lo = 0, hi = n_seqs; // largest s with first_item(s) <= t (empty sequences own no piece)
        while (hi - lo > 1) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if (first_item(mid) <= t) lo = mid;
            else hi = mid;
        }
        const uint64_t b = off[lo], len = off[lo + 1] - b;
        const uint32_t j = t - first_item(lo), l1 = long_first_own(b, own);
        const uint64_t s0 = j == 0 ? 0 : (uint64_t)l1 + (uint64_t)(j - 1u) * own;
        const uint32_t own0 = (uint32_t)min(s0, (uint64_t)cb), own_n = (uint32_t)min((uint64_t)(j == 0 ? l1 : own), len - s0);
        it = make_uint4((uint32_t)(b + s0 - own0), (uint32_t)(s0 - own0), (uint32_t)len, own0 | (own_n << 10));
    }
    items[t] = it;
}

// the doubling chain of erode_end, kept for several lengths over the same bits
struct EroChain {
    uint64_t s1, s2, s4, s8, s16, s32;
};
__device__ __forceinline__ EroChain ero_chain(uint64_t h)
{
    EroChain c;
    c.s1 = h;
    c.s2 = h & (h << 1);
    c.s4 = c.s2 & (c.s2 << 2);
    c.s8 = c.s4 & (c.s4 << 4);
    c.s16 = c.s8 & (c.s8 << 8);
    c.s32 = c.s16 & (c.s16 << 16);
    return c;
}
// AND over the L positions that end at each bit, 1 <= L <= 63: two windows of the largest power of two below L + 1 cover them
// (overlapping windows do not matter to an AND) - one shift and one AND whatever L is, and two small numbers per length instead of
// the six conditions of a bit-by-bit composition (which the compiler kept in scalar registers it did not have: a third of this
// phase's instructions were lane reads of spilled ones)
__device__ __forceinline__ uint64_t ero_at(const EroChain &c, uint32_t L)
{
    const uint32_t lg = 31u - (uint32_t)__builtin_clz(L);
    const uint64_t sp = lg == 0u ? c.s1 : lg == 1u ? c.s2 : lg == 2u ? c.s4 : lg == 3u ? c.s8 : lg == 4u ? c.s16 : c.s32;
    return sp & (sp << (L - (1u << lg)));
}

// One wave takes a.ppw consecutive pieces, one after the other: the next piece's bases are on their way while this one is
// worked on, and a piece that continues the sequence of the one before starts on that one's last diagonal - its text is
// on its way too - instead of asking the seed table (a wrong guess is lost at once and found again like any lost diagonal).
// STATS: the kernel counts its own work (kbo_set_plan_stats) - instrumentation, compiled out of the default instantiation
template <bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KBO_LONG_WPE))) void map_long_kernel(LongArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t long_lds_all[];
    const uint32_t lane = threadIdx.x & 63u;
    // (made uniform for the compiler: the items are then scalar loads, and what is derived from them lives in scalar registers)
    const uint32_t p_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * a.ppw));
    if (p_first >= a.n_items) return; // (wave-uniform; the waves of a workgroup share nothing)
    const uint32_t n_mine = min(a.ppw, a.n_items - p_first);
    uint8_t *lds = long_lds_all + (threadIdx.x >> 6) * kLongLds;
    uint32_t *lin = reinterpret_cast<uint32_t *>(lds) + 2;              // digits of word w, w = -2 .. 67         (280 B)
    uint16_t *invl = reinterpret_cast<uint16_t *>(lds + 288) + 2;       // bytes that are no base, per word       (144 B)
    uint16_t *ufl = reinterpret_cast<uint16_t *>(lds + 432) + 2;        // U of the filter's strings, per word    (144 B)
    uint2 *tx = reinterpret_cast<uint2 *>(lds + 576);                   // staged text units                      (608 B)
    uint16_t *list = reinterpret_cast<uint16_t *>(lds + 576);           // look-ups: position | 0x400 ext | 0x800 last (768 B; where the
                                                                        // text was: it is not needed any more when the list is made)
    const uint32_t k = a.ix.k, thr = a.thr, order = a.ix.dtab_order;
    const int32_t xa = (int32_t)(16u * lane);
    // (the items through the scalar unit - constant address space, uniform index: no vector load whose wait would take the queue of
    // vector loads and stores with it -, each one two pieces ahead of its use)
    typedef uint32_t item_words_t __attribute__((ext_vector_type(4)));
    typedef const item_words_t __attribute__((address_space(4))) *const_items_t;
    const const_items_t items_c = (const_items_t)(uintptr_t)a.items;
    auto item_at = [&](uint32_t i) -> uint4 {
        const item_words_t v = items_c[i];
        return make_uint4(v.x, v.y, v.z, v.w);
    };
    const uint32_t p_last = p_first + n_mine - 1u;
    if (lane < 2u) {
        lin[(int32_t)lane - 2] = 0;
        invl[(int32_t)lane - 2] = 0;
        ufl[(int32_t)lane - 2] = 0;
    }
    if (lane < 4u) {
        lin[64u + lane] = 0;
        invl[64u + lane] = 0;
        ufl[64u + lane] = 0;
    }
    const uint32_t F = a.ix.dfilt ? a.ix.dfilt_bases : 0u;
    const bool by_anchor = a.ix.anchor != nullptr && order >= 12u && order > a.ix.seed_d + 1u; // (as map_reads_kernel seeds)
    const uint32_t D = by_anchor ? order : a.ix.seed_d;
    const uint32_t dmask = D >= 16u ? 0xFFFFFFFFu : ((1u << (2u * D)) - 1u);
    const int32_t n_units = (int32_t)(((uint64_t)a.ix.n + kMapPad + 256u) / 16u + 2u); // (pack_text_units)
    const uint32_t cstep = thr - order, Mrun = cstep * (47u / cstep);
    uint32_t st_seed = 0, st_look = 0, st_filt = 0, st_second = 0, st_band = 0; // (st_band: wave-uniform; tried | taken << 16)

    uint4 it_c = item_at(p_first), it_n = item_at(min(p_first + 1u, p_last));
    // the bases of the wave's first piece
    uint4 vq = make_uint4(0, 0, 0, 0);
    {
        const uint32_t q0 = it_c.x, g00 = it_c.y, sl0 = it_c.z, w0 = it_c.w;
        const uint32_t R0 = min(sl0 - g00, (w0 & 0x3FFu) + ((w0 >> 10) & 0x7FFu) + a.ca), n0 = ((q0 & 15u) + R0 + 15u) >> 4;
        if (((w0 >> 10) & 0x7FFu) != 0u && lane < n0) vq = ld16(a.q, (q0 & ~15u) + 16u * lane);
    }
    // the digits of a piece's 16 bases per lane and which of them are bases: packed as soon as the bytes are there, a piece ahead
    auto pack_piece = [&](const uint4 &v, uint32_t q_off_, uint32_t R_, uint32_t &code_, uint32_t &valid_) {
        bool anyinv;
        pack16_whole(v, code_, anyinv);
        valid_ = 0xFFFFu;
        const uint32_t inr_ = range16(xa, (int32_t)(q_off_ & 15u), (int32_t)((q_off_ & 15u) + R_));
        // (a byte that is no base, or this lane's word is not all the region's: the per-byte mask)
        if (__ballot(anyinv && inr_ != 0u)) pack16(v, code_, valid_);
    };
    uint32_t code_c = 0, valid_c = 0xFFFFu;
    {
        const uint32_t q0 = it_c.x, g00 = it_c.y, sl0 = it_c.z, w0 = it_c.w;
        pack_piece(vq, q0, min(sl0 - g00, (w0 & 0x3FFu) + ((w0 >> 10) & 0x7FFu) + a.ca), code_c, valid_c);
    }
    bool pred_in_lds = false; // ... and that text is in LDS already
    bool pred = false;      // this piece goes on where the last one ended: on diagonal pred_dl, with its text in tn0 / tn1 (units from pred_u0 on)
    int32_t pred_dl = 0, pred_u0 = 0;
    uint2 tn0 = make_uint2(0, 0), tn1 = make_uint2(0, 0);

    const bool stamps = kLongExp && (a.xexp & 128u) != 0; // (experiments: shader cycles per phase, summed into qctl[16 ..] in units of 16)
    uint32_t cyc[5] = {0, 0, 0, 0, 0};
    auto stamp = [&](uint32_t slot, uint64_t &t_last) {
        if (!stamps) return;
        const uint64_t t = __builtin_amdgcn_s_memtime();
        cyc[slot] += (uint32_t)(t - t_last);
        t_last = t;
    };
    uint64_t t_last = stamps ? __builtin_amdgcn_s_memtime() : 0;
    bool need_xin = true; // (x of this piece's first own base: see below)
    for (uint32_t pi = 0; pi < n_mine; pi++) {
        const uint32_t piece = p_first + pi;
        const uint4 itc = it_c, itn = it_n;
        it_c = it_n;
        it_n = item_at(min(piece + 2u, p_last));
        const uint32_t q_off = itc.x, g0 = itc.y, seqlen = itc.z, itw = itc.w;
        const uint32_t own0 = itw & 0x3FFu, own_n = (itw >> 10) & 0x7FFu;
        // (the next piece, whose bases are asked for as soon as this one's are in LDS)
        const bool have_next = pi + 1u < n_mine;
        const uint32_t nq_off = itn.x, ng0 = itn.y, nseqlen = itn.z, nitw = have_next ? itn.w : 0u;
        const uint32_t nown0 = nitw & 0x3FFu, nown_n = (nitw >> 10) & 0x7FFu;
        const uint32_t nR = min(nseqlen - ng0, nown0 + nown_n + a.ca), nnblk = ((nq_off & 15u) + nR + 15u) >> 4;
        const bool skip = own_n == 0 || seqlen < 3u; // (a slot past the batch's last piece; derandomize.rs:274-276 asserts on fewer than 3 values: left unwritten)
        const uint32_t r0 = q_off & 15u, base16 = q_off - r0;
        const uint32_t R = min(seqlen - g0, own0 + own_n + a.ca), xe = r0 + R; // the region on the grid: [r0, xe)

        // ---- 0. the region's 2-bit digits (packed at the end of the last round); the next piece's bases are asked for now and
        // packed at the end of this round, BEFORE this piece's characters are stored: loads and stores leave one queue in order,
        // and a wait for those bases at the top of the next round would be a wait for these stores
        const uint32_t code = code_c, valid = valid_c;
        if (have_next && nown_n != 0u) {
            vq = make_uint4(0, 0, 0, 0);
            if (lane < nnblk) vq = ld16(a.q, (nq_off & ~15u) + 16u * lane); // (reads <= 15 bytes in front of / behind the region: the buffer's own)
        }
        bool tx_next = false; // the next piece's text is in LDS already (finish_round)
        auto finish_round = [&]() {
            if (have_next && nown_n != 0u) pack_piece(vq, nq_off, nR, code_c, valid_c);
            if (pred) { // (the proof's list is done with: the text of the next piece goes where it was)
                __builtin_amdgcn_wave_barrier();
                tx[lane] = tn0;
                if (lane < kLongTextUnits - 64u) tx[64u + lane] = tn1;
                tx_next = true;
            }
        };
        if (skip) {
            if (lane == 0) a.redo[piece] = 0;
            need_xin = false;
            pred = false;
            pred_in_lds = false;
            finish_round();
            continue;
        }
        const uint32_t inr16 = range16(xa, (int32_t)r0, (int32_t)xe);
        const uint32_t inv16 = ~valid & inr16 & 0xFFFFu;
        __builtin_amdgcn_wave_barrier();
        lin[lane] = code;
        invl[lane] = (uint16_t)inv16;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (kLongExp && (a.xexp & 16u)) {
            pred = false;
            pred_in_lds = false;
            finish_round();
            continue;
        }
        stamp(0, t_last); // staging

        auto from_base = [&](uint32_t S) -> uint32_t { // 16 bases from S on, first one most significant
            const uint32_t W = S >> 4, r = S & 15u;
            return funnel2(lin[W], lin[W + 1u], r);
        };
        auto ending_at = [&](uint32_t E) -> uint64_t { // the 17 + E mod 16 bases ending at E, last one least significant
            const uint32_t W = E >> 4, r = E & 15u;
            const uint64_t V = ((uint64_t)lin[(int32_t)W - 1] << 32) | lin[W];
            return V >> (2u * (15u - r));
        };
        auto base_at = [&](uint32_t S) -> uint32_t { return (lin[S >> 4] >> (2u * (15u - (S & 15u)))) & 3u; };
        const bool any_inv = __ballot(inv16 != 0u) != 0; // (wave-uniform: nearly always false)
        auto inv_span = [&](uint32_t E, uint32_t L) -> bool { // any byte that is no base among the L <= 33 positions ending at E
            if (!any_inv) return false;
            const int32_t W = (int32_t)(E >> 4);
            const uint32_t r = E & 15u;
            const uint64_t V = (uint64_t)invl[W - 2] | ((uint64_t)invl[W - 1] << 16) | ((uint64_t)invl[W] << 32);
            return ((V >> (33u + r - L)) & ((1ull << L) - 1ull)) != 0;
        };

        // ---- 1. stretches
        auto seed_at = [&](uint32_t e_) -> uint32_t { // text position of grid position e_ by the window that ends there (bit 31: one of several), or ~0
            const uint64_t win = ending_at(e_);
            if (!by_anchor) return a.ix.seed_pos[(uint32_t)win & dmask];
            const uint64_t key = win & ((1ull << (2u * D)) - 1ull), amask = ((uint64_t)1 << a.ix.anchor_bits) - 1ull;
            uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64u - a.ix.anchor_bits);
            const uint32_t tag = (uint32_t)key + 1u;
            for (uint32_t probe = 0; probe < 16u; probe++) {
                const uint64_t slot = a.ix.anchor[h];
                if (slot == 0) break;
                if ((uint32_t)(slot >> 32) == tag) return (uint32_t)slot & 0x7FFFFFFFu;
                h = (h + 1u) & amask;
            }
            return 0xFFFFFFFFu;
        };
        // windows ending at c + D - 1 + D j, j < nl: the first that ends one row only, else the first that ends any
        auto seed_round = [&](uint32_t c, uint32_t nl, int32_t &dl, uint32_t &A) -> bool {
            const uint32_t e_ = c + D - 1u + D * lane;
            const bool ok = lane < nl && e_ < xe && !inv_span(e_, D);
            uint32_t tp = 0xFFFFFFFFu;
            if (ok) {
                tp = seed_at(e_);
                if (STATS) st_seed++;
            }
            const uint64_t hit_any = __ballot(tp != 0xFFFFFFFFu), hit_one = __ballot(tp != 0xFFFFFFFFu && !(tp >> 31));
            if (!hit_any) return false;
            const int src = (int)__builtin_ctzll(hit_one ? hit_one : hit_any);
            const uint32_t tps = __shfl(tp, src) & 0x7FFFFFFFu, es = c + D - 1u + D * (uint32_t)src;
            dl = (int32_t)tps - (int32_t)es;
            A = es - D + 1u;
            return true;
        };
        int32_t tbase = 0, stage_dl = 0;
        bool staged = false;
        auto stage_text = [&](int32_t dl) {
            const int32_t u0 = (dl + (int32_t)kMapPad - 64) >> 4; // (arithmetic shift: floor)
            for (uint32_t c = lane; c < kLongTextUnits; c += 64u) {
                const int32_t u = u0 + (int32_t)c;
                tx[c] = (u >= 0 && u < n_units) ? a.ix.pc_tm[u] : make_uint2(0u, 0x55555555u);
            }
            tbase = u0 * 16 - (int32_t)kMapPad;
            stage_dl = dl;
            staged = true;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        auto compare = [&](int32_t dl) -> uint32_t { // bit j: position 16 lane + j does not equal the text on diagonal dl (or is no base / outside)
            const uint32_t idx0 = (uint32_t)(dl + xa - tbase), unit = idx0 >> 4, r = idx0 & 15u;
            const uint2 t0 = tx[unit], t1 = tx[unit + 1u];
            const uint32_t x = code ^ funnel2(t0.x, t1.x, r);
            return grid_to_mask16(x | (x >> 1) | funnel2(t0.y, t1.y, r)) | inv16 | (~inr16 & 0xFFFFu);
        };
        // the assignment of a diagonal reaches back from A over sparse mismatches, up to a mismatch with TH of them in the 16 bases ending at it
        auto left_start = [&](uint32_t mm, uint32_t A, int32_t lower) -> uint32_t {
            const uint32_t m = mm & inr16 & range16(xa, lower, (int32_t)A);
            const uint32_t view = shfl_up0(m, 1, lane) | (m << 16);
            if (!__ballot((uint32_t)__popc(view) >= kLongTH)) return (uint32_t)lower; // (no word pair with TH mismatches)
            // (the windows end at every fourth base: the last one with TH mismatches, and the last mismatch inside it)
            uint32_t hits = 0;
#pragma unroll
            for (uint32_t j = 3; j < 16u; j += 4u)
                if ((uint32_t)__popc((view >> (j + 1u)) & 0xFFFFu) >= kLongTH) hits |= 1u << j;
            const uint64_t bal = __ballot(hits != 0);
            if (!bal) return (uint32_t)lower;
            const int L = 63 - (int)__builtin_clzll(bal);
            const uint32_t j = 31u - (uint32_t)__builtin_clz(__shfl(hits, L));
            const uint32_t w = (__shfl(view, L) >> (j + 1u)) & 0xFFFFu; // the window's mismatches: bit 15 = base 16 L + j
            return 16u * (uint32_t)L + j - (uint32_t)(__builtin_clz(w) - 16) + 1u;
        };

        const bool pred_now = pred;
        uint32_t ZA = 0, ZB = 0; // the planes: bit j = position 16 lane + j lies in a stretch
        bool end_on_diag = false; // the region's last bases lie on the diagonal end_dl
        bool band_took = false;
        int32_t end_dl = 0;
        {
            int32_t endz[2] = {(int32_t)r0 - 1, (int32_t)r0 - 1};
            const int32_t J = (int32_t)order - 3;
            uint32_t cur = 0, c = r0, A = 0, start = 0, mm = 0;
            int32_t dl = 0;
            bool have = false, band_tried = false;
            if (pred_now) { // on from the piece before: its last diagonal, the text already here
                if (!pred_in_lds) {
                    tx[lane] = tn0;
                    if (lane < kLongTextUnits - 64u) tx[64u + lane] = tn1;
                }
                tbase = pred_u0 * 16 - (int32_t)kMapPad;
                stage_dl = pred_dl;
                staged = true;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                dl = pred_dl;
                have = true;
                A = start = r0;
                mm = compare(dl);
            }
            for (uint32_t iter = 0; iter < 96u; iter++) {
                if (!have) {
                    int32_t dn = 0;
                    uint32_t An = 0;
                    bool ok = seed_round(c, 4u, dn, An);
                    if (!ok) ok = seed_round(c + 4u * D, 64u, dn, An);
                    if (!ok) break;
                    dl = dn;
                    A = An;
                    have = true;
                    if (!staged || dl - stage_dl > 24 || stage_dl - dl > 24) stage_text(dl);
                    mm = compare(dl);
                    start = left_start(mm, A, max(max((int32_t)r0, endz[cur] + 1), endz[cur ^ 1u] - J));
                }
                // where the diagonal is lost: the first 16 bases from A on with TH mismatches; f = the first of them
                uint32_t f = xe;
                {
                    // (the windows start at every fourth base: where one holds TH mismatches from A on, so does one of these within three
                    // bases or one mismatch of it - whatever is chosen here only decides how many pieces take the second pass)
                    const uint32_t mr = mm & inr16 & range16(xa, (int32_t)A, (int32_t)xe);
                    const uint32_t view = mr | (shfl_down0(mr, 1, lane) << 16);
                    uint32_t loss = 0;
                    if (__ballot((uint32_t)__popc(view) >= kLongTH)) { // (no word pair with TH mismatches: nothing is lost)
#pragma unroll
                        for (uint32_t j = 0; j < 16u; j += 4u)
                            if ((uint32_t)__popc((view >> j) & 0xFFFFu) >= kLongTH) loss |= 1u << j;
                    }
                    const uint64_t bal = __ballot(loss != 0);
                    if (bal) {
                        const int L = (int)__builtin_ctzll(bal);
                        const uint32_t j = (uint32_t)__builtin_ctz(__shfl(loss, L));
                        const uint32_t vw = __shfl(view, L) >> j;
                        f = 16u * (uint32_t)L + j + (uint32_t)__builtin_ctz(vw);
                    }
                }
                const uint32_t add = range16(xa, (int32_t)start, (int32_t)f) & ~mm & 0xFFFFu;
                if (cur == 0) ZA |= add;
                else ZB |= add;
                endz[cur] = (int32_t)f;
                if (f >= xe) {
                    end_on_diag = true;
                    end_dl = dl;
                    break;
                }
                // ---- the first loss of a piece: ALL its words at once, each on the diagonal of the band dl - kBand .. dl + kBand it matches
                // best.  A sequence with an insertion or deletion every 80 bases leaves its diagonal a dozen times a piece, by a
                // base or two each time: one pass of every lane over the band instead of a dozen rounds of this loop (each of which
                // every lane takes part in).  A word that matches none (a junction inside it, a cluster of substitutions) is cut
                // between its neighbours' diagonals where the two together mismatch least.  Any word left without a diagonal: the
                // band was not it (a long insertion, a join) - the planes are dropped and the loop goes on as before.
                if (!band_tried && !(kLongExp && (a.xexp & 8u))) {
                    band_tried = true;
                    constexpr int kBand = 6;
                    const uint32_t idx0 = (uint32_t)(dl - kBand + xa - tbase), unit = idx0 >> 4, r0b = idx0 & 15u;
                    const uint2 t0 = tx[unit], t1 = tx[unit + 1u], t2 = tx[unit + 2u];
                    // 48 digits of text and marks from digit r0b of the first unit on: the word on the band's s-th diagonal is 16 of them
                    auto band_mask = [&](uint32_t sidx) -> uint32_t { // mismatches on the digit grid (bit 2 (15 - j) = base j)
                        const uint32_t rr = r0b + sidx, hi = rr >> 4, r = rr & 15u; // rr <= 15 + 12
                        const uint32_t a0 = hi ? t1.x : t0.x, a1 = hi ? t2.x : t1.x, b0 = hi ? t1.y : t0.y, b1 = hi ? t2.y : t1.y;
                        const uint32_t x = code ^ funnel2(a0, a1, r);
                        return (x | (x >> 1) | funnel2(b0, b1, r)) & 0x55555555u;
                    };
                    const uint32_t gin = grid_from_mask16(inr16 & ~inv16); // the word's bases that count, on the digit grid
                    uint32_t best_c = 99u, best_s = (uint32_t)kBand;
#pragma unroll
                    for (int o = 0; o <= 2 * kBand; o++) { // (from the middle outwards: of equal counts the nearer diagonal)
                        const uint32_t sidx = (uint32_t)(kBand + ((o & 1) ? -((o + 1) / 2) : (o / 2)));
                        const uint32_t c = (uint32_t)__popc(band_mask(sidx) & gin);
                        if (c < best_c) {
                            best_c = c;
                            best_s = sidx;
                        }
                    }
                    const uint32_t n_in = (uint32_t)__popc(inr16);
                    const bool in_lane = n_in != 0u;
                    const bool decided = in_lane && best_c <= 2u && (n_in - (uint32_t)__popc(inv16)) >= 8u; // (at least 8 bases that say so)
                    const uint64_t dec = __ballot(decided), inl = __ballot(in_lane);
                    bool ok = dec != 0;
                    if (ok) {
                        // sR: the diagonal this word hands on to the right (its own when decided, else the next decided word's - the
                        // last one's behind that); sL: what the word to the left hands on (the first word: its own)
                        const uint64_t above = lane < 63u ? dec >> (lane + 1u) : 0ull, below = dec & ((1ull << lane) - 1ull);
                        const uint32_t src = decided ? lane : above ? lane + 1u + (uint32_t)__builtin_ctzll(above) : 63u - (uint32_t)__builtin_clzll(below | 1ull);
                        const uint32_t sR = __shfl(best_s, (int)src);
                        uint32_t sL = (uint32_t)__builtin_amdgcn_update_dpp((int)sR, (int)sR, 0x138, 0xF, 0xF, false); // wave_shr:1, lane 0 keeps its own
                        if (lane == 0) sL = sR;
                        const uint32_t mR = grid_to_mask16(band_mask(sR)) | inv16 | (~inr16 & 0xFFFFu);
                        const bool change = in_lane && sL != sR;
                        uint32_t inbits = 0, outbits = ~mR & 0xFFFFu; // bits of the in-coming / out-going diagonal's stretches in this word
                        uint32_t f1_w = 16u;   // (a word with a change: where its in-coming stretch ends)
                        bool hole = false, soft = false, open_left = false; // open_left: the out-going stretch starts at the word's first base - and may start earlier
                        if (__ballot(change)) {
                            if (change) {
                                const uint32_t mL = grid_to_mask16(band_mask(sL)) | inv16 | (~inr16 & 0xFFFFu);
                                // the cut: bases [0, c) on the in-coming diagonal, [c, 16) on the out-going one
                                uint32_t cost = (uint32_t)__popc(mR & inr16), bestc = cost, cut = 0;
#pragma unroll
                                for (uint32_t c = 1; c <= 16u; c++) {
                                    cost += ((mL >> (c - 1u)) & 1u & (inr16 >> (c - 1u))) - ((mR >> (c - 1u)) & 1u & (inr16 >> (c - 1u)));
                                    if (cost <= bestc) { // (ties: the rightmost - the in-coming diagonal as far as it matches)
                                        bestc = cost;
                                        cut = c;
                                    }
                                }
                                if (bestc > 3u) {
                                    // no cut explains the word: two changes inside it, or a change and a cluster of substitutions.  The in-coming
                                    // stretch then ends at its first mismatch, the out-going one starts behind its last, and what lies between is
                                    // in no stretch (whatever matches there is shorter than a word, hence than the table's windows - unless it
                                    // is not, and then the proof says so).  A few such words a piece; more: the band was not it
                                    soft = true;
                                    const uint32_t f1 = mL ? (uint32_t)__builtin_ctz(mL) : 16u;
                                    uint32_t b2 = mR ? 32u - (uint32_t)__builtin_clz(mR) : 0u;
                                    b2 = max(b2, f1 > (uint32_t)J ? f1 - (uint32_t)J : 0u);
                                    f1_w = f1;
                                    inbits = ~mL & ((1u << f1) - 1u) & 0xFFFFu;
                                    outbits = b2 < 16u ? (~mR & (0xFFFFu << b2) & 0xFFFFu) : 0u;
                                } else {
                                    // the in-coming stretch ends at its first mismatch from the cut on; the out-going one starts behind its last
                                    // mismatch in front of the cut, at most J bases in front of that end - and never at the word's first base (a
                                    // base that is in no stretch of its plane keeps the plane's stretches apart: header)
                                    const uint32_t mLc = mL & (0xFFFFu << cut) & 0xFFFFu;
                                    const uint32_t f1 = mLc ? (uint32_t)__builtin_ctz(mLc) : 16u;
                                    const uint32_t mRc = mR & ((1u << cut) - 1u);
                                    uint32_t b2 = mRc ? 32u - (uint32_t)__builtin_clz(mRc) : 0u;
                                    b2 = max(b2, f1 > (uint32_t)J ? f1 - (uint32_t)J : 0u);
                                    open_left = b2 == 0u;
                                    f1_w = f1;
                                    inbits = ~mL & ((1u << f1) - 1u) & 0xFFFFu;
                                    outbits = b2 < 16u ? (~mR & (0xFFFFu << b2) & 0xFFFFu) : 0u;
                                }
                            }
                        }
                        // (a word that matches no diagonal of the band and stands between two words on one: its bases as they are on that
                        // one - two changes that cancel, a cluster of substitutions -, counted with the words no cut explains)
                        if (in_lane && !decided && !change && (uint32_t)__popc(mR & inr16) > 5u) soft = true;
                        if (hole) inbits = outbits = 0;
                        ok = __ballot(hole) == 0 && __popcll(__ballot(soft)) <= 3;
                        if (ok) {
                            const uint64_t ch = __ballot(change);
                            const uint32_t p_in = (uint32_t)__popcll(ch & ((1ull << lane) - 1ull)) & 1u, p_out = p_in ^ (change ? 1u : 0u);
                            // a stretch that starts at its word's first base goes on to the LEFT, into the last bases of the word in front,
                            // as far as those lie on its diagonal (a junction in a word's last bases: that word chose the other diagonal
                            // and the one behind it carries the change): by at most J bases in all with what lies on both (f1 of the word
                            // with the change), and never up to the end of a stretch of the same plane in that word.  That word looks.
                            uint32_t extbits = 0;
                            {
                                const uint32_t nx = shfl_down0((open_left && change && !hole ? 1u : 0u) | (f1_w << 1) | (sR << 8), 1, lane);
                                if (__ballot(nx & 1u)) {
                                    if (nx & 1u) {
                                        const uint32_t f1n = (nx >> 1) & 0x7Fu, sRn = nx >> 8;
                                        const uint32_t mX = grid_to_mask16(band_mask(sRn)) | inv16 | (~inr16 & 0xFFFFu);
                                        uint32_t ext = mX ? (uint32_t)__builtin_clz(mX << 16) : 16u; // bases at the word's end on that diagonal
                                        ext = min(ext, f1n < (uint32_t)J ? (uint32_t)J - f1n : 0u);
                                        // (this word's own in-coming stretch lies in the same plane when it has a change itself)
                                        ext = min(ext, change ? (f1_w < 15u ? 15u - f1_w : 0u) : 15u);
                                        extbits = ext ? (0xFFFFu << (16u - ext)) & 0xFFFFu : 0u;
                                    }
                                }
                            }
                            // ... and an in-coming stretch that reaches its word's last base goes on to the RIGHT, into the first bases of the word
                            // behind, as far as those lie on its diagonal (a junction in a word's last bases whose first base behind it matches
                            // both diagonals): within J bases of where the out-going stretch started, and short of that word's own
                            // out-going stretch when it lies in the same plane (a change there too).  That word looks.
                            uint32_t rextbits = 0;
                            {
                                const uint32_t b2_w = (outbits & 0xFFFFu) ? (uint32_t)__builtin_ctz(outbits) : 16u; // (where this word's out-going stretch starts)
                                const uint32_t pv = shfl_up0((change && !hole && f1_w == 16u ? 1u : 0u) | (b2_w << 1) | (sL << 8), 1, lane);
                                if (__ballot(pv & 1u)) {
                                    if (pv & 1u) {
                                        const uint32_t b2p = (pv >> 1) & 0x7Fu, sLp = pv >> 8;
                                        const uint32_t mX = grid_to_mask16(band_mask(sLp)) | inv16 | (~inr16 & 0xFFFFu);
                                        uint32_t ext = mX ? (uint32_t)__builtin_ctz(mX) : 16u; // bases at the word's start on that diagonal
                                        const uint32_t over = 16u - min(b2p, 16u); // what the two stretches share in the word in front
                                        ext = min(ext, over < (uint32_t)J ? (uint32_t)J - over : 0u);
                                        ext = min(ext, change ? (b2_w > 0u ? b2_w - 1u : 0u) : 15u);
                                        rextbits = (1u << ext) - 1u;
                                    }
                                }
                            }
                            // (and a stretch that starts at its word's first base right behind a word whose in-coming stretch of the same plane
                            // runs to that word's end: not its first base - a base in neither keeps the two apart)
                            {
                                const uint32_t pv = shfl_up0((change && f1_w == 16u ? 1u : 0u), 1, lane);
                                if (change && open_left && pv && !(shfl_up0(extbits, 1, lane) != 0u)) outbits &= ~1u;
                            }
                            const uint32_t p_ext = p_out ^ 1u; // (the plane the word behind hands on: it has a change)
                            const uint32_t p_rext = p_in ^ 1u; // (the plane the word in front took in: it has a change)
                            ZA = (p_in == 0u ? inbits : 0u) | (p_out == 0u ? outbits : 0u) | (p_ext == 0u ? extbits : 0u) | (p_rext == 0u ? rextbits : 0u);
                            ZB = (p_in == 1u ? inbits : 0u) | (p_out == 1u ? outbits : 0u) | (p_ext == 1u ? extbits : 0u) | (p_rext == 1u ? rextbits : 0u);
                            const uint32_t last_l = 63u - (uint32_t)__builtin_clzll(inl);
                            end_on_diag = true;
                            end_dl = dl - kBand + (int32_t)__shfl(sR, (int)last_l);
                            if (STATS) st_band += 0x10001u;
                            band_took = true;
                            break;
                        }
                    }
                    if (STATS) st_band += 1u; // (tried, not taken)
                }
                // the next diagonal: of the 64 beside this one, the one on which the read goes on soonest - the first run of kLongRun
                // matching bases among the 32 behind f (ties: the longer run, then the nearer diagonal)
                bool found = false;
                int32_t d2 = 0;
                uint32_t A2 = 0;
                if (f + 1u + kLongRun <= xe) {
                    const uint32_t S = f + 1u;
                    const uint32_t w0 = from_base(S), w1 = from_base(S + 16u);
                    const int32_t W = (int32_t)(S >> 4);
                    const uint64_t iv = ((uint64_t)invl[W] | ((uint64_t)invl[W + 1] << 16) | ((uint64_t)invl[W + 2] << 32)) >> (S & 15u);
                    const int32_t sft = (int32_t)lane - 32;
                    const uint32_t idx0 = (uint32_t)(dl + sft + (int32_t)S - tbase), unit = idx0 >> 4, r = idx0 & 15u;
                    const uint2 t0 = tx[unit], t1 = tx[unit + 1u], t2 = tx[unit + 2u];
                    const uint32_t x0 = w0 ^ funnel2(t0.x, t1.x, r), x1 = w1 ^ funnel2(t1.x, t2.x, r);
                    uint32_t m32 = grid_to_mask16(x0 | (x0 >> 1) | funnel2(t0.y, t1.y, r)) | (grid_to_mask16(x1 | (x1 >> 1) | funnel2(t1.y, t2.y, r)) << 16) |
                                   (uint32_t)iv;
                    const uint32_t n_in = xe - S;
                    if (n_in < 32u) m32 |= ~0u << n_in;
                    const uint32_t z = ~m32, e2 = z & (z >> 1), e4 = e2 & (e2 >> 2), e8 = e4 & (e4 >> 4), e10 = e8 & (e2 >> 8);
                    uint32_t key = 0xFFFFFFFFu;
                    if (e10) {
                        const uint32_t at = (uint32_t)__builtin_ctz(e10);
                        const uint32_t rest = ~(z >> at); // (bit `run`: the first mismatch behind the run; beyond bit 31 - at: zeros shifted in read as mismatches)
                        const uint32_t run = rest ? (uint32_t)__builtin_ctz(rest) : 32u;
                        const uint32_t as = (uint32_t)(sft < 0 ? -sft : sft);
                        key = (at << 16) | ((63u - run) << 8) | (as << 1) | (sft > 0 ? 1u : 0u);
                    }
                    const uint32_t best = wave_min_dpp(key);
                    if (best != 0xFFFFFFFFu) {
                        const uint32_t as = (best >> 1) & 0x7Fu;
                        found = true;
                        d2 = dl + ((best & 1u) ? (int32_t)as : -(int32_t)as);
                        A2 = S + (best >> 16);
                    }
                }
                if (!found) found = seed_round(f + 4u, 4u, d2, A2);
                if (!found) {
                    have = false;
                    c = f + 4u + 4u * D;
                    continue;
                }
                if (d2 == dl) { // the same diagonal after all (a cluster of substitutions): on with it
                    start = f;
                    A = A2;
                    continue;
                }
                if (d2 - stage_dl > 24 || stage_dl - d2 > 24) stage_text(d2);
                mm = compare(d2);
                cur ^= 1u;
                start = left_start(mm, A2, max(max((int32_t)r0, (int32_t)f - J), endz[cur] + 1));
                dl = d2;
                A = A2;
            }
        }
        // the piece that goes on with this sequence: its diagonal (in its own grid) and its text, asked for now
        pred = false;
        if (have_next && nown_n != 0u && end_on_diag && nown0 != 0u && nq_off - ng0 == q_off - g0 && ng0 + nown0 == g0 + own0 + own_n) {
            pred = true;
            pred_dl = end_dl + ((int32_t)r0 - (int32_t)g0) - ((int32_t)(nq_off & 15u) - (int32_t)ng0);
            pred_u0 = (pred_dl + (int32_t)kMapPad - 64) >> 4;
            const int32_t u_a = pred_u0 + (int32_t)lane, u_b = u_a + 64;
            tn0 = (u_a >= 0 && u_a < n_units) ? a.ix.pc_tm[u_a] : make_uint2(0u, 0x55555555u);
            tn1 = (lane < kLongTextUnits - 64u && u_b >= 0 && u_b < n_units) ? a.ix.pc_tm[u_b] : make_uint2(0u, 0x55555555u);
        }
        if (kLongExp && (a.xexp & 32u)) {
            finish_round();
            pred_in_lds = tx_next;
            continue;
        }
        stamp(1, t_last); // stretches

        // ---- 2. the planes -> G, cov, characters, U
        auto hist64 = [&](uint32_t z) -> uint64_t { // positions [16 (lane - 3), 16 lane + 16): this lane's at bits 48 .. 63
            const uint32_t z1 = shfl_up0(z, 1, lane), z2 = shfl_up0(z1, 1, lane), z3 = shfl_up0(z2, 1, lane);
            return (uint64_t)z3 | ((uint64_t)z2 << 16) | ((uint64_t)z1 << 32) | ((uint64_t)z << 48);
        };
        auto fwd64 = [&](uint32_t z) -> uint64_t { // positions [16 lane, 16 lane + 64)
            const uint32_t z1 = shfl_down0(z, 1, lane), z2 = shfl_down0(z1, 1, lane), z3 = shfl_down0(z2, 1, lane);
            return (uint64_t)z | ((uint64_t)z1 << 16) | ((uint64_t)z2 << 32) | ((uint64_t)z3 << 48);
        };
        // (plane by plane: most pieces never leave their first diagonal, and the second plane is empty)
        uint32_t G, inO, inF = 0;
        {
            const EroChain CA = ero_chain(hist64(ZA));
            G = (uint32_t)(ero_at(CA, thr + 1u) >> 48);
            inO = (uint32_t)(ero_at(CA, order) >> 48);
            if (F) inF = (uint32_t)(ero_at(CA, F) >> 48);
        }
        if (__ballot(ZB != 0u)) {
            const EroChain CB = ero_chain(hist64(ZB));
            G |= (uint32_t)(ero_at(CB, thr + 1u) >> 48);
            inO |= (uint32_t)(ero_at(CB, order) >> 48);
            if (F) inF |= (uint32_t)(ero_at(CB, F) >> 48);
        }
        const uint32_t cov = (uint32_t)dilate_fwd(fwd64(G), thr + 1u) & 0xFFFFu;
        const uint32_t cov_prev = ((cov << 1) | (shfl_up0(cov, 1, lane) >> 15)) & 0xFFFFu;
        const uint32_t cov_next = ((cov >> 1) | (shfl_down0(cov, 1, lane) << 15)) & 0xFFFFu;
        const uint32_t G_next = ((G >> 1) | (shfl_down0(G, 1, lane) << 15)) & 0xFFFFu;
        // (positions of the sequence: grid x stands for base g0 + x - r0)
        const int32_t gx = (int32_t)r0 - (int32_t)g0; // grid position of the sequence's base 0 (may be negative)
        const uint32_t first_two = range16(xa, gx, gx + 2);
        const uint32_t isX = ~cov & cov_next & (cov_prev | first_two) & inr16;
        const uint32_t R1 = G & ~G_next & cov_next;
        // (the second 'R' at base p needs 2 <= p < len - 1: translate.rs:282-288)
        const uint32_t R2 = ((R1 << 1) | (shfl_up0(R1, 1, lane) >> 15)) & range16(xa, gx + 2, gx + (int32_t)min(seqlen - 1u, 0x3FFFFFFFu));
        const uint32_t isR = (R1 | R2) & 0xFFFFu;
        const uint32_t isM = cov & ~isR;
        const uint32_t U = ~inO & inr16 & range16(xa, (int32_t)(r0 + order - 1u), (int32_t)xe);
        if (F) ufl[lane] = (uint16_t)(~inF & inr16 & range16(xa, (int32_t)(r0 + F - 1u), (int32_t)xe));

        stamp(2, t_last); // analysis
        // ---- 3. the proof
        bool flag = false;
        uint32_t why = 0; // (counted with the work counters: what sent the piece to the second pass)
        if (!(kLongExp && (a.xexp & 1u)) && __ballot(U != 0u)) {
            // the points of every run of U: its first position and every cstep-th from there while the run's start is known (Mrun
            // positions back), every cstep-th position of the grid beyond
            const uint64_t HU = hist64(U);
            const uint64_t P0 = HU & ~(HU << 1);
            const uint64_t Ec1 = erode_end(HU, cstep + 1u);
            uint64_t Ej = HU, pts = P0;
            uint64_t far = 0; // the run started more than Mrun positions back
            for (uint32_t jc = cstep;; jc += cstep) {
                if (jc > Mrun) {
                    far = Ej & (HU << (Mrun + 1u));
                    break;
                }
                Ej &= Ec1 << (jc - cstep);
                if (!__ballot((Ej >> 48) != 0)) break; // (no run of the piece is that long)
                pts |= (P0 << jc) & Ej;
            }
            uint32_t gridm = 0; // positions of this word that are multiples of cstep
            for (uint32_t j = ((uint32_t)xa + cstep - 1u) / cstep * cstep - (uint32_t)xa; j < 16u; j += cstep) gridm |= 1u << j;
            const uint32_t point = ((uint32_t)(pts >> 48) | (gridm & (uint32_t)(far >> 48))) & U;
            const uint32_t U_next = ((U >> 1) | (shfl_down0(U, 1, lane) << 15)) & 0xFFFFu;
            const uint32_t lastU = U & ~U_next;                  // the last window of a run
            const uint32_t lastf = lastU & point;                // ... that is a point: the window one base on is looked up too when it is present
            uint32_t extn = lastU & ~point;                      // ... that is none: the window one base on is looked up instead
            const uint32_t at_end = extn & range16(xa, (int32_t)xe - 1, (int32_t)xe); // (nothing behind the region: the window itself)
            extn &= ~at_end;
            const uint32_t normal = point | at_end;
            const uint32_t ext = ((extn << 1) | (shfl_up0(extn, 1, lane) >> 15)) & 0xFFFFu;
            const uint32_t mine = (uint32_t)__popc(normal) + (uint32_t)__popc(ext);
            uint32_t incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t t = __shfl_up(incl, o);
                if ((int)lane >= o) incl += t;
            }
            const uint32_t total = __shfl(incl, 63);
            __builtin_amdgcn_wave_barrier(); // (the list goes where the text was)
            if (total > kLongListCap) {
                flag = true;
                if (STATS) why = 1u;
            }
            else {
                uint32_t at = incl - mine;
                uint32_t nm = normal;
                while (nm) {
                    const uint32_t j = (uint32_t)__builtin_ctz(nm);
                    nm &= nm - 1u;
                    list[at++] = (uint16_t)(((uint32_t)xa + j) | (((lastf >> j) & 1u) ? 0x800u : 0u));
                }
                uint32_t em = ext;
                while (em) {
                    const uint32_t j = (uint32_t)__builtin_ctz(em);
                    em &= em - 1u;
                    list[at++] = (uint16_t)(((uint32_t)xa + j) | 0x400u);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t omask = (1ull << (2u * order)) - 1ull;
            auto table = [&](uint32_t e_, uint32_t g) -> uint32_t { // the entry of the window ending at e_ (g: its place in the grouped line)
                const uint64_t key = ending_at(e_) & omask;
                if (STATS) st_look++;
                return a.ix.dtab_grouped ? a.ix.dtab[dtab_grouped_addr(key, g, order)] : a.ix.dtab[key];
            };
            // the window ending at e_ is in the index AND the read's base in front of it extends it
            auto present_ext = [&](uint32_t byte, uint32_t e_) -> bool {
                return (byte & 0x80u) && e_ >= r0 + order && !inv_span(e_ - order, 1u) && ((byte >> base_at(e_ - order)) & 1u);
            };
            for (uint32_t i0 = 0; !flag && i0 < total; i0 += 64u) {
                const uint32_t i = i0 + lane;
                bool act = i < total;
                const uint32_t ent = act ? (uint32_t)list[i] : 0u;
                const uint32_t x = ent & 0x3FFu;
                const bool is_ext = (ent & 0x400u) != 0, is_last = (ent & 0x800u) != 0;
                // (a window with a byte that is no base is in no index; the string an ext look-up asks about holds the base in front)
                if (act && (inv_span(x, order) || (is_ext && inv_span(x - order, 1u)))) act = false;
                if (F && !(kLongExp && (a.xexp & 2u))) { // the filter: a string of F bases of the window that lies in no single stretch - absent: so is the window
                    bool fl = false;
                    uint32_t ef = 0;
                    if (act && is_ext) {
                        ef = x - order + F - 1u;
                        fl = true;
                    } else if (act) {
                        const int32_t W = (int32_t)(x >> 4);
                        const uint32_t r = x & 15u;
                        uint32_t V = (uint32_t)ufl[W - 1] | ((uint32_t)ufl[W] << 16); // positions 16 (W - 1) .. 16 W + 15
                        V &= (2u << (16u + r)) - 1u;                                // <= x
                        const uint32_t lowbit = 16u + r - (order - F);              // >= x - order + F
                        V &= ~((1u << lowbit) - 1u);
                        if (V) {
                            ef = 16u * (uint32_t)(W - 1) + (31u - (uint32_t)__builtin_clz(V));
                            fl = true;
                        }
                    }
                    if (fl) {
                        const uint32_t fk = (uint32_t)ending_at(ef) & ((1u << (2u * F)) - 1u);
                        if (STATS) st_filt++;
                        if (!((a.ix.dfilt[fk >> 5] >> (fk & 31u)) & 1u)) act = false; // absent
                    }
                }
                uint32_t byte = 0;
                if (act && !(kLongExp && (a.xexp & 4u))) byte = table(x, 1u);
                bool fail = false;
                bool need_back = false, need_on = false;
                if (act && is_ext) fail = present_ext(byte, x);
                else if (act && (byte & 0x80u)) {
                    need_back = present_ext(byte, x);
                    need_on = is_last && x + 1u < xe;
                }
                if (__ballot(need_back)) {
                    // order + 1 bases of the read end here and are in the index.  The windows of t + 1 bases that hold them end at x ..
                    // x + cstep - 1.  A string of order + 1 bases ending at x - j that is absent rules out those ending at x + cstep - j
                    // or before, one ending at x + i those ending at x + i or later: any pair with i + j <= cstep + 1 (j = 1 alone) will
                    // do.  (A string that starts in front of the region or ends behind it lies in none of the windows that matter; one
                    // with a byte that is no base is absent.)  A chance match of order + 2 bases - one in some thousand windows - is
                    // what most flagged pieces of reads with substitutions only were flagged for; each further base divides that by four.
                    if (need_back) {
                        auto both = [&](uint32_t e_, uint32_t g) -> bool {
                            if (e_ < r0 + order || e_ >= xe || inv_span(e_, order)) return false;
                            if (STATS) st_second++;
                            return present_ext(table(e_, g), e_);
                        };
                        uint32_t j = 1;
                        while (j <= cstep && both(x - j, j == 1u ? 0u : 1u)) j++;
                        fail = j > cstep;
                        if (!fail && j > 1u) {
                            uint32_t i = 1;
                            while (i <= cstep + 1u - j && both(x + i, i == 1u ? 2u : 1u)) i++;
                            fail = i > cstep + 1u - j;
                        }
                    }
                }
                uint32_t y_fail = x; // where the string of order + 1 bases that is in the index ends
                if (__ballot(need_on)) { // the run's last window is present: the one inside the next stretch must not be extended by the base in front
                    if (need_on && !fail) {
                        if (STATS) st_second++;
                        if (!inv_span(x + 1u, order + 1u)) fail = present_ext(table(x + 1u, 2u), x + 1u);
                        if (fail) y_fail = x + 1u;
                    }
                }
                uint64_t fm = __ballot(fail);
                if (fm) {
                    // (round 6) what fails is SEARCHED, as map_reads_kernel does: order + 1 bases ending at y are in the index; the strings
                    // of thr + 1 bases that hold them end at y .. y + cstep, one a lane - first seed_d bases by the interval table, the
                    // rest by extend-right steps over the rank blocks - and none of them in the index is the proof, exactly (one with a byte
                    // that is no base is absent).  Two searches a round of look-ups; more, one that finds its string, or one whose strings do
                    // not all lie inside the region: the second pass as before
                    const uint32_t D_ = a.ix.seed_d;
                    const bool can_search = !(kLongExp && (a.xexp & 256u)) && a.ix.seed_tab != nullptr && D_ >= 4u && D_ <= 16u && thr + 1u >= D_ && thr + 1u <= 33u;
                    bool unresolved = !can_search;
                    uint32_t n_search = 0;
                    while (fm && !unresolved) {
                        const int src = __builtin_ctzll(fm);
                        fm &= fm - 1ull;
                        if (++n_search > 2u) {
                            unresolved = true;
                            break;
                        }
                        const uint32_t y = __shfl(y_fail, src);
                        // (order + 1 bases ending at y lie inside thr + 1 bases ending at y .. y + cstep: cstep + 1 candidates; one that does not
                        // lie inside the region cannot be looked at here: the second pass)
                        const uint32_t e_c = y + lane; // this lane's candidate ends here
                        const bool mine_c = lane <= cstep;
                        const bool outside = mine_c && (e_c >= xe || e_c < r0 + thr);
                        if (__ballot(outside)) {
                            unresolved = true;
                            break;
                        }
                        bool alive = mine_c && !inv_span(e_c, thr + 1u);
                        const uint32_t first = e_c - thr;
                        uint32_t l_ = 0, r_ = 0;
                        if (alive) {
                            const uint32_t key = (uint32_t)ending_at(first + D_ - 1u) & (D_ >= 16u ? 0xFFFFFFFFu : (1u << (2u * D_)) - 1u);
                            const uint2 iv = a.ix.seed_tab[key];
                            if (STATS) st_second++;
                            l_ = iv.x;
                            r_ = iv.y;
                            alive = l_ < r_;
                        }
                        const uint8_t *arena_ = reinterpret_cast<const uint8_t *>(a.ix.arena);
                        for (uint32_t j_ = D_; j_ <= thr; j_++) {
                            if (__ballot(alive) == 0) break;
                            if (alive) {
                                const uint32_t cb_ = base_at(first + j_) * a.ix.n_blocks, bl_ = div96(l_), br_ = div96(r_);
                                const uint4 xA = ld16(arena_, (cb_ + bl_) << 4), xB = ld16(arena_, (cb_ + br_) << 4);
                                l_ = rank_eval(xA, l_ - bl_ * 96u);
                                r_ = rank_eval(xB, r_ - br_ * 96u);
                                alive = l_ < r_;
                            }
                        }
                        if (__ballot(alive)) unresolved = true; // a string of thr + 1 bases that lies in no single stretch IS in the index
                    }
                    if (unresolved) {
                        flag = true;
                        if (STATS) why = __ballot(fail && is_ext) ? 2u : __ballot(fail && need_back) ? 3u : 4u;
                    }
                }
            }
        }

        stamp(3, t_last); // proof
        finish_round();
        pred_in_lds = tx_next;
        // ---- 4. the characters of the own bases, in whole lines; format::relative_to_ref (format.rs:270-286) on the way
        {
            uint32_t w[4];
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) {
                if (a.fmt) { // ('M' and 'R' keep the read's base, everything else is '-': one mask)
                    const uint32_t d8 = (code >> (24u - 8u * q)) & 0xFFu;
                    const uint32_t sel = ((d8 >> 6) & 3u) | (((d8 >> 4) & 3u) << 8) | (((d8 >> 2) & 3u) << 16) | ((d8 & 3u) << 24);
                    const uint32_t letters = __builtin_amdgcn_perm(0u, 0x54474341u, sel), keep = spread4((isM | isR) >> (4u * q));
                    w[q] = (letters & keep) | (0x2D2D2D2Du & ~keep);
                } else {
                    const uint32_t eM = spread4(isM >> (4u * q)), eX = spread4(isX >> (4u * q)), eR = spread4(isR >> (4u * q));
                    w[q] = (eM & 0x4D4D4D4Du) | (eX & 0x58585858u) | (eR & 0x52525252u) | (~(eM | eX | eR) & 0x2D2D2D2Du);
                }
            }
            const uint4 out = make_uint4(w[0], w[1], w[2], w[3]);
            const int32_t o_lo = (int32_t)(r0 + own0), o_hi = o_lo + (int32_t)own_n;
            const uint32_t lo_t = (uint32_t)min(max(o_lo - xa, 0), 16), hi_t = (uint32_t)min(max(o_hi - xa, 0), 16);
            uint8_t *dst = a.chars_out + base16 + 16u * lane;
            const bool whole = lo_t == 0u && hi_t == 16u, part = !whole && hi_t > lo_t;
            if (whole) __builtin_memcpy(dst, &out, 16);
            // (pieces start and end where 16-byte blocks of the buffer do, but for a sequence's first and last: rarely any partial block)
            if (__ballot(part) != 0 && part) { // (the first and the last word of the own bases: whole 4-byte words, then bytes)
#pragma unroll
                for (uint32_t q = 0; q < 4u; q++) {
                    if (4u * q >= lo_t && 4u * q + 4u <= hi_t) __builtin_memcpy(dst + 4u * q, &w[q], 4);
                    else if (4u * q + 4u > lo_t && 4u * q < hi_t) {
#pragma unroll
                        for (uint32_t t = 4u * q; t < 4u * q + 4u; t++)
                            if (t >= lo_t && t < hi_t) dst[t] = (uint8_t)(w[q] >> ((t & 3u) * 8u));
                    }
                }
            }
        }
        // x of the first own base, 0 .. k (what derandomize_ms_vec gives there: the depth in the stretch of more than t bases that
        // covers it, 0 when none does) - for the piece to the left, should that one be flagged.  Per plane: the ones that end at that
        // base (Lp) and those behind it (Rp), through the words that are all ones by one ballot
        // (only the piece behind a flagged one is ever asked for it - long_derand_kernel starts a run of flagged pieces from the xin of
        // the piece to its right: worked out for the wave's first piece, whose left neighbour is another wave's, and behind a piece
        // this wave has just flagged)
        const bool fl = flag || (a.xexp & 64u) != 0; // (64: every piece to the second pass - tests)
        if (need_xin) {
            const uint32_t xs = r0 + own0, Lo = xs >> 4, b = xs & 15u;
            uint32_t best = 0;
#pragma unroll
            for (uint32_t p = 0; p < 2u; p++) {
                const uint32_t w = p == 0 ? ZA : ZB;
                const uint64_t full = __ballot(w == 0xFFFFu);
                const uint32_t wo = __shfl(w, (int)Lo);
                if (!((wo >> b) & 1u)) continue; // (wave-uniform)
                // down from bit b of word Lo
                uint32_t Lp;
                const uint32_t zd = ~wo & ((2u << b) - 1u);
                if (zd) Lp = b - (31u - (uint32_t)__builtin_clz(zd));
                else {
                    const uint64_t nf = ~full & ((1ull << Lo) - 1ull); // words below Lo that are not all ones
                    const uint32_t hi = nf ? 63u - (uint32_t)__builtin_clzll(nf) : 0u, n_full = nf ? Lo - 1u - hi : Lo;
                    const uint32_t wp = nf ? (uint32_t)__shfl(w, (int)hi) : 0u;
                    const uint32_t part = nf ? (uint32_t)__builtin_clz(~(wp << 16)) : 0u; // ones from bit 15 of that word down
                    Lp = b + 1u + 16u * n_full + min(part, 16u);
                }
                // up from bit b + 1
                uint32_t Rp;
                const uint32_t zu = ~wo & 0xFFFFu & ~((2u << b) - 1u);
                if (zu) Rp = (uint32_t)__builtin_ctz(zu) - b - 1u;
                else {
                    const uint64_t nf = Lo < 63u ? (~full & (~0ull << (Lo + 1u))) : 0ull;
                    const uint32_t lo_l = nf ? (uint32_t)__builtin_ctzll(nf) : 64u, n_full = nf ? lo_l - Lo - 1u : 63u - Lo;
                    const uint32_t wn = nf ? (uint32_t)__shfl(w, (int)lo_l) : 0u;
                    const uint32_t part = nf ? (uint32_t)__builtin_ctz(~wn) : 0u; // ones from bit 0 of that word up
                    Rp = 15u - b + 16u * n_full + min(part, 16u);
                }
                if (Lp + Rp > thr) best = max(best, min(Lp, k));
            }
            if (lane == 0) a.xin[piece] = (uint8_t)best;
        }
        if (lane == 0) {
            a.redo[piece] = fl ? 1 : 0;
            if (fl) atomicAdd(a.qctl + 4, 1u);
            if (STATS && fl && a.pstats) atomicAdd(a.qctl + 8u + why + (band_took ? 4u : 0u), 1u);
        }
        need_xin = fl;
        stamp(4, t_last); // output
    }
    if (stamps && lane == 0)
        for (uint32_t i = 0; i < 5u; i++) atomicAdd(a.qctl + 16u + i, cyc[i] >> 4);
    // the wave's work counters, when the launch counts (kbo_set_plan_stats): every wave adding to a handful of words was a third of
    // the kernel's time - atomics on one address take about 10 ns each, whoever sends them
    if (!STATS || !a.pstats) return;
    // (two sums instead of four: seed and second look-ups stay below 2^16 per wave, filter and table look-ups as well)
    const uint32_t s0 = wave_sum(st_seed | (st_second << 16)), s1 = wave_sum(st_filt | (st_look << 16));
    if (lane == 0) {
        uint32_t *st = a.pstats + ((p_first / a.ppw) % kPlanStatSlots) * kPlanStatWords;
        atomicAdd(st + kPlanStatSeedLookups, s0 & 0xFFFFu);
        atomicAdd(st + kPlanStatTabAnchored, s0 >> 16);
        atomicAdd(st + kPlanStatSeedExtensions, s1 & 0xFFFFu);
        atomicAdd(st + kPlanStatTabLookups, s1 >> 16);
        atomicAdd(st + kPlanStatUnits, st_band & 0xFFFFu);    // pieces that tried all their words on a band of diagonals at once
        atomicAdd(st + kPlanStatAccepted, st_band >> 16);     // ... and kept the result
    }
}

// ---- flagged pieces: sub-items of 8 / 16 / 32 bases (+ k - 1 warm-up bases) for the plain walk: the MS values of [s - 1, s + n).
// The walk is a chain of dependent steps, as long as a sub-item with its warm-up: while the flagged pieces are few, short sub-items
// (more lanes, each done sooner) - as long as there are not more of them than the device holds lanes at once.  One pair of atomics
// per workgroup: every flagged piece adding to the same two words took 50 us for three thousand of them.
constexpr uint32_t kLongSub = 32;
constexpr uint32_t kLongSubLanes = 100000;
__global__ __launch_bounds__(1024) void long_redo_items_kernel(LongArgs a, WalkItem *__restrict__ out, uint32_t cap, uint32_t *__restrict__ count,
                                                               uint32_t *__restrict__ flist)
{
    __shared__ uint32_t s_sub[16], s_fl[16], s_base[2];
    const uint32_t piece = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t n_fl = a.qctl[4], own_max = (kLongRegion - 2u * a.ix.k - 1u) & ~15u;
    uint32_t sub = kLongSub;
    for (uint32_t s_ = 8u; s_ < kLongSub; s_ <<= 1) {
        const uint64_t n = (uint64_t)n_fl * ((own_max + s_) / s_);
        if (n <= kLongSubLanes && n + 64u <= cap) {
            sub = s_;
            break;
        }
    }
    bool fl = piece < a.n_items && a.redo[piece] != 0;
    uint32_t s = 0, own_n = 0;
    uint64_t seq0 = 0;
    if (fl) {
        const uint4 it = reinterpret_cast<const uint4 *>(a.items)[piece];
        const uint32_t own0 = it.w & 0x3FFu;
        own_n = (it.w >> 10) & 0x7FFu;
        s = it.y + own0;                  // the piece's first own base in its sequence
        seq0 = (uint64_t)it.x - it.y;     // the sequence's first byte
        fl = own_n != 0u;
    }
    const uint32_t lo = s > 0 ? s - 1u : 0u, hi = s + own_n, n_sub = fl ? (hi - lo + sub - 1u) / sub : 0u;
    // this piece's place among the workgroup's: within the wave, then across the waves
    uint32_t incl = n_sub;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if ((int)lane >= o) incl += t;
    }
    const uint64_t bal = __ballot(fl);
    if (lane == 63u) {
        s_sub[wv] = incl;
        s_fl[wv] = (uint32_t)__popcll(bal);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ts = 0, tf = 0;
        for (uint32_t w = 0; w < (blockDim.x >> 6); w++) {
            const uint32_t a_ = s_sub[w], b_ = s_fl[w];
            s_sub[w] = ts;
            s_fl[w] = tf;
            ts += a_;
            tf += b_;
        }
        s_base[0] = ts ? atomicAdd(count, ts) : 0u;
        s_base[1] = tf ? atomicAdd(count + 1, tf) : 0u; // (count + 1 = qctl[2]: the flagged pieces, listed for long_derand_kernel)
    }
    __syncthreads();
    if (!fl) return;
    const uint32_t base = s_base[0] + s_sub[wv] + incl - n_sub;
    flist[s_base[1] + s_fl[wv] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = piece;
    const uint32_t warm_max = a.ix.k > 0 ? a.ix.k - 1u : 0u;
    for (uint32_t p = 0; p < n_sub; p++) {
        const uint32_t o0 = lo + p * sub, o1 = min(o0 + sub, hi), warm = min(o0, warm_max);
        WalkItem w;
        w.start = seq0 + o0 - warm;
        w.len = o1 - o0 + warm;
        w.warm = warm;
        if (base + p < cap) out[base + p] = w;
    }
}

// derandomize_ms_vec (derandomize.rs:269-288) + translate_ms_vec (translate.rs:263-293) over the flagged pieces, from their true MS
// values: one WAVE per run of flagged pieces - the wave of its rightmost piece, which starts from x of the base behind it (xin of
// the unflagged piece to the right; the sequence's end: derandomize.rs:282) and goes on through the flagged pieces to its left - a
// lane per 16 bases.  The recurrence x[p] = F(a[p], x[p + 1]) (derandomize.rs:233-246) in parallel:
//   * x <= a wherever a < k, and a rises by at most one per base.  So at a base with a > t ("good") x is a or a - 1: with
//     d = a - x, d[p] = 0 when a[p] = k or the base to the right is not good or has a smaller a; d[p] = not d[p + 1] when it has the
//     same a; d[p] = d[p + 1] when it has a + 1: a one-bit recurrence of resets, copies and negations - composed per lane, then
//     across the lanes by a scan of (mask, constant) pairs;
//   * everywhere else x[p] = x[q] - (q - p) with q the next good base to the right (or the base behind the piece): x - p is
//     copied from there - a scan of "take mine if I have one";
//   * translate_ms_vec's window (x[p - 1], x[p], x[p + 1]) from the neighbouring lanes.
__global__ __launch_bounds__(64) void long_derand_kernel(LongArgs a, const uint8_t *__restrict__ ms, const uint32_t *__restrict__ flist)
{
    const uint32_t lane = threadIdx.x;
    const uint4 *items = reinterpret_cast<const uint4 *>(a.items);
    const uint32_t n_flagged = min(a.qctl[2], a.n_items);
    const int K = (int)a.ix.k, T = (int)a.thr;
    auto step = [&](int av, int x) { return av == K ? K : ((av > T && x < av) ? av : x - 1); };
    for (uint32_t fi = blockIdx.x; fi < n_flagged; fi += gridDim.x) {
        uint32_t piece = flist[fi];
        const uint4 it = items[piece];
        uint32_t own_n = (it.w >> 10) & 0x7FFu;
        const uint32_t seqlen = it.z;
        if (own_n == 0) continue;
        uint32_t s = it.y + (it.w & 0x3FFu), e = s + own_n; // own bases [s, e) of the sequence
        const bool at_end = e >= seqlen;
        if (!at_end && piece + 1u < a.n_items && a.redo[piece + 1u]) continue; // (the wave of a piece further right takes this one)
        const uint64_t seq0 = (uint64_t)it.x - it.y;
        const uint64_t room = a.q_bytes - seq0; // bytes of the two buffers from the sequence's first on
        const uint8_t *m = ms + seq0;
        const uint8_t *qs = a.q + seq0;
        uint8_t *out = a.chars_out + seq0;
        // x of base e; at the sequence's end a value that makes the last base's x what derandomize.rs:282 says: a > t ? a : 0
        int x_right = at_end ? 1 : (int)a.xin[piece + 1u];
        for (;;) {
            const uint32_t n = own_n, p0 = s + 16u * lane; // this lane's bases: p0 .. p0 + 15, those below e
            const uint32_t cnt = p0 < e ? min(16u, e - p0) : 0u;
            // their MS values (a byte that would lie behind the buffers is not read)
            uint32_t av[16];
            {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (cnt && (uint64_t)p0 + 16u <= room) v = ld16u(m, p0);
                else if (cnt) {
                    uint8_t tmp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    for (uint32_t t = 0; t < 16u && (uint64_t)p0 + t < room; t++) tmp[t] = m[p0 + t];
                    __builtin_memcpy(&v, tmp, 16);
                }
#pragma unroll
                for (uint32_t t = 0; t < 16u; t++) {
                    const uint32_t w = (t >> 2) == 0 ? v.x : (t >> 2) == 1 ? v.y : (t >> 2) == 2 ? v.z : v.w;
                    av[t] = t < cnt ? (w >> ((t & 3u) * 8u)) & 0xFFu : 0u;
                }
            }
            uint4 qv = make_uint4(0, 0, 0, 0);
            if (a.fmt && cnt) {
                if ((uint64_t)p0 + 16u <= room) qv = ld16u(qs, p0);
                else {
                    uint8_t tmp[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    for (uint32_t t = 0; t < 16u && (uint64_t)p0 + t < room; t++) tmp[t] = qs[p0 + t];
                    __builtin_memcpy(&qv, tmp, 16);
                }
            }
            const int a_left = s > 0 ? (int)m[s - 1u] : K; // MS of the base in front of the piece (base -1: anything, its x is not used)
            // the value to the right of this lane's last base: the next lane's first, or (the piece's last lane) none: the boundary
            const uint32_t a_first_next = __shfl_down(av[0], 1);
            const bool last_lane = cnt != 0 && p0 + cnt == e;
            // ---- d within good runs: per base (mask, constant), d = constant ^ (mask & d of the base to the right)
            uint32_t mk[16], cs[16];
            uint32_t M = 1, C = 0; // the lane's composition, from its last base down to its first: d_first = C ^ (M & d_in)
#pragma unroll
            for (int t = 15; t >= 0; t--) {
                const uint32_t an = t == 15 ? a_first_next : av[t == 15 ? 15 : t + 1];
                const bool in = (uint32_t)t < cnt, is_last = last_lane && (uint32_t)t + 1u == cnt;
                const bool good = in && (int)av[t] > T;
                uint32_t mm = 0, cc = 0;
                if (is_last) cc = (good && (int)av[t] != K && x_right == (int)av[t]) ? 1u : 0u; // (x_right = a + 1: the decrement gives a as well)
                else if (good && (int)av[t] != K && (int)an > T) {
                    mm = (an == av[t] || an == av[t] + 1u) ? 1u : 0u;
                    cc = an == av[t] ? 1u : 0u;
                }
                if (!in) { // (bases behind the piece in its last lane: pass d through - it is not used)
                    mm = 1;
                    cc = 0;
                }
                mk[t] = mm;
                cs[t] = cc;
                // compose: this base applied after what is to its right
                C = cc ^ (mm & C);
                M = mm & M;
            }
            // suffix scan over the lanes (lane L needs the composition of the lanes to its right applied to d = 0 behind the piece)
            uint32_t SM = M, SC = C; // composition of lanes L .. L + 2^i - 1
            uint32_t din = 0;        // d of the first base of lane L + 1
            {
                // exclusive: start from the right neighbour's inclusive composition
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t m2 = __shfl_down(SM, o), c2 = __shfl_down(SC, o);
                    if (lane + (uint32_t)o < 64u) { // lanes [L, L + o) then [L + o, L + 2 o): f_L(f_{L+o}(d))
                        SC = SC ^ (SM & c2);
                        SM = SM & m2;
                    }
                }
                // SC now = d of this lane's first base when d = 0 behind everything (the masks see to it that the boundary is a reset)
                din = __shfl_down(SC, 1);
                if (lane == 63u) din = 0;
            }
            int xv[16];
            // d of every base, then x of the good ones; Y = x - position for the others
            int Yin; // x - position of the next good base to the right of this lane (or of the base behind the piece)
            {
                uint32_t d = din;
                uint32_t has = 0;
                int firstY = 0;
#pragma unroll
                for (int t = 15; t >= 0; t--) {
                    d = cs[t] ^ (mk[t] & d);
                    const bool good = (uint32_t)t < cnt && (int)av[t] > T;
                    xv[t] = (int)av[t] - (int)d;
                    if (good) {
                        has = 1;
                        firstY = xv[t] - (int)(p0 + (uint32_t)t);
                    }
                }
                // suffix scan: the leftmost good base of the nearest lane to the right that has one
                uint32_t H = has;
                int Y = firstY;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t h2 = __shfl_down(H, o);
                    const int y2 = __shfl_down(Y, o);
                    if (lane + (uint32_t)o < 64u && !H) {
                        H = h2;
                        Y = y2;
                    }
                }
                const uint32_t Hn = __shfl_down(H, 1);
                const int Yn = __shfl_down(Y, 1);
                const int Yb = x_right - (int)e;
                Yin = (lane < 63u && Hn) ? Yn : Yb;
            }
            {
                int carry = Yin;
#pragma unroll
                for (int t = 15; t >= 0; t--) {
                    const bool in = (uint32_t)t < cnt;
                    const bool good = in && (int)av[t] > T;
                    if (good) carry = xv[t] - (int)(p0 + (uint32_t)t);
                    else if (in) xv[t] = carry + (int)(p0 + (uint32_t)t);
                }
            }
            // ---- translate_ms_vec: (x[p - 1], x[p], x[p + 1])
            const int x_first = __shfl(xv[0], 0); // x of base s
            const int x_front = s > 0 ? step(a_left, x_first) : K; // x of base s - 1
            int x_prev_in = __shfl_up(xv[15], 1);
            if (lane == 0) x_prev_in = x_front;
            int x_next_in = __shfl_down(xv[0], 1); // (the next lane's first base; the piece's last base: the base behind it)
            uint32_t ow[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t t = 0; t < 16u; t++) {
                const uint32_t p = p0 + t;
                const bool lastb = last_lane && t + 1u == cnt;
                const int cur = xv[t];
                const int prv = t == 0 ? x_prev_in : xv[t == 0 ? 0 : t - 1];
                int nxt = t == 15 ? x_next_in : xv[t == 15 ? 15 : t + 1];
                if (lastb) nxt = x_right;
                if (p + 1u >= seqlen) nxt = cur; // translate.rs:279
                const int prev = p > 1u ? prv : K; // translate.rs:277
                // res[p] = 'R' when 2 <= p < len - 1 and the base in front starts an ('R','R') (translate.rs:282-288), else translate_ms_val's first
                const bool r2 = p >= 2u && p + 1u < seqlen && prv > T && cur > 0 && cur < T;
                const bool r1 = cur > T && nxt > 0 && nxt < T;
                uint32_t ch = cur <= 0 ? ((nxt == 1 && prev > 0) ? (uint32_t)'X' : (uint32_t)'-') : (uint32_t)'M';
                ch = (r1 || r2) ? (uint32_t)'R' : ch;
                if (a.fmt) {
                    const uint32_t qw = (t >> 2) == 0 ? qv.x : (t >> 2) == 1 ? qv.y : (t >> 2) == 2 ? qv.z : qv.w;
                    ch = (ch == (uint32_t)'M' || ch == (uint32_t)'R') ? ((qw >> ((t & 3u) * 8u)) & 0xFFu) : (uint32_t)'-';
                }
                ow[t >> 2] |= ch << ((t & 3u) * 8u);
            }
            if (cnt == 16u) {
                const uint4 v = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                __builtin_memcpy(out + p0, &v, 16);
            } else if (cnt) st_partial(out + p0, make_uint4(ow[0], ow[1], ow[2], ow[3]), cnt);
            x_right = x_first; // x of base s: what the piece to the left starts from
            (void)n;
            // the flagged piece to the left, if it is this sequence's
            if (s == 0 || piece == 0 || !a.redo[piece - 1u]) break;
            const uint4 pit = items[piece - 1u];
            const uint32_t p_n = (pit.w >> 10) & 0x7FFu, p_s = pit.y + (pit.w & 0x3FFu);
            if (p_n == 0 || (uint64_t)pit.x - pit.y != seq0 || p_s + p_n != s) break;
            piece--;
            s = p_s;
            e = p_s + p_n;
            own_n = p_n;
        }
    }
}

} // namespace

namespace {
struct LongLayout {
    size_t items, scan, redo, xin, flist, ctl, pstats, subs, end;
    uint32_t own, n_slots, sub_cap;
};
LongLayout long_layout(size_t n_seqs, uint64_t total_bases, uint32_t k)
{
    LongLayout L{};
    L.own = (kLongRegion - 2u * k - 1u) & ~15u; // (a multiple of 16: long_first_own)
    const uint64_t slots = total_bases / L.own + 2 * n_seqs + 1; // (a sequence's first piece may be up to 15 bases short)
    L.n_slots = (uint32_t)std::min<uint64_t>(slots, 0x7FFFFF00ull);
    size_t w = 0;
    L.items = w;
    w += slots * 16;
    L.scan = w;
    w += (chunk_items_scratch_words((uint32_t)n_seqs) * 4 + 15) / 16 * 16;
    L.redo = w;
    w += (slots + 15) / 16 * 16;
    L.xin = w;
    w += (slots + 15) / 16 * 16;
    L.flist = w;
    w += (slots * 4 + 15) / 16 * 16;
    L.ctl = w; // [0] pieces, [1] sub-items of the flagged pieces, [2] flagged pieces listed, [4] flagged pieces
    w += 256;
    L.pstats = w;
    w += kPlanStatSlots * kPlanStatWords * 4;
    L.subs = w; // sub-items of the flagged pieces: all of them, at worst
    const uint64_t cap = total_bases / kLongSub + 3 * slots + 64;
    L.sub_cap = (uint32_t)std::min<uint64_t>(cap, 0x7FFFFF00ull);
    w += cap * sizeof(WalkItem);
    L.end = w + 64;
    return L;
}
} // namespace

size_t long_work_bytes(size_t n_seqs, uint64_t total_bases, uint32_t k)
{
    if (2u * k + 1u + kLongOwnMin > kLongRegion) return 0;
    return long_layout(n_seqs, total_bases, k).end;
}

bool map_long_applies(const DevIndexView &ix, uint32_t thr)
{
    static const int env_on = std::getenv("KBO_MAP_LONG") ? std::atoi(std::getenv("KBO_MAP_LONG")) : 1; // experiments
    if (!(env_on != 0 && g_map_long.load() != 0 && ix.dtab && ix.pc_tm && ix.seed_pos && ix.seed_d >= 4u && ix.seed_d <= 14u && ix.dtab_order >= 4u && ix.dtab_order <= 17u &&
          ix.dtab_order < thr && thr < ix.k && thr <= 47u && 2u * ix.k + 1u + kLongOwnMin <= kLongRegion))
        return false;
    if (ix.dfilt && (ix.dfilt_bases >= ix.dtab_order || ix.dtab_order - ix.dfilt_bases > 5u || ix.dfilt_bases > 16u)) return false;
    // (the proof takes a window every thr - order bases: at most six per mismatch)
    const uint32_t c = thr - ix.dtab_order;
    return (ix.dtab_order + c - 1u) / c + 1u <= 6u;
}

// the pieces of the batch and the kernel; `a` as launch_map_long_redo needs it
hipError_t launch_map_long(const DevIndexView &ix, const uint8_t *d_q, const uint64_t *d_off, uint32_t n_seqs, uint64_t total_bases, uint32_t thr,
                           bool fmt, uint8_t *d_chars, void *d_work, hipStream_t stream, LongArgs &a, bool count)
{
    const LongLayout L = long_layout(n_seqs, total_bases, ix.k);
    uint8_t *w = static_cast<uint8_t *>(d_work);
    uint32_t *local = reinterpret_cast<uint32_t *>(w + L.scan), *sums = local + n_seqs + 1u;
    uint32_t *ctl = reinterpret_cast<uint32_t *>(w + L.ctl);
    hipError_t e = hipMemsetAsync(ctl, 0, 256 + kPlanStatSlots * kPlanStatWords * 4, stream);
    if (e != hipSuccess) return e;
    // pieces per sequence, their scan, the items (the grid covers the most pieces the batch can have; slots past the last are empty)
    hipLaunchKernelGGL(long_count_kernel, dim3((n_seqs + 1u + 255u) / 256u), dim3(256), 0, stream, d_off, n_seqs, L.own, local);
    e = launch_scan(local, n_seqs + 1u, sums, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(long_items_kernel, dim3((L.n_slots + 255u) / 256u), dim3(256), 0, stream, d_off, local, sums, n_seqs, L.own, ix.k, L.n_slots,
                       reinterpret_cast<uint4 *>(w + L.items), ctl);
    a = LongArgs{};
    a.ix = ix;
    a.q = d_q;
    a.items = w + L.items;
    a.n_items = L.n_slots;
    a.chars_out = d_chars;
    a.redo = w + L.redo;
    a.xin = w + L.xin;
    a.qctl = ctl;
    a.pstats = count ? reinterpret_cast<uint32_t *>(w + L.pstats) : nullptr; // (instrumentation: kbo_set_plan_stats)
    a.thr = thr;
    a.fmt = fmt ? 1u : 0u;
    static const int env_x = std::getenv("KBO_LONG_X") ? std::atoi(std::getenv("KBO_LONG_X")) : 0; // experiments: phases left out (timing only)
    a.xexp = (uint32_t)env_x | (g_map_long.load() == 2 ? 64u : 0u);
    // pieces per wave: consecutive pieces share a wave's set-up and hand their diagonal and text on (8 / 12 / 16 / 32 per wave on 165 000
    // pieces: 401 / 403 / 406 / 409 Gbp/s) - as many as leave the device two rounds of waves, sixteen at most
    static const int env_ppw = std::getenv("KBO_LONG_PPW") ? std::atoi(std::getenv("KBO_LONG_PPW")) : 0; // experiments
    a.ppw = env_ppw > 0 ? (uint32_t)std::min(64, env_ppw) : std::min(16u, std::max(1u, L.n_slots / 8192u));
    a.ca = ix.k + 1u;
    a.subs = w + L.subs;
    a.flist = reinterpret_cast<uint32_t *>(w + L.flist);
    a.sub_cap = L.sub_cap;
    a.q_bytes = total_bases;
    static const int env_wpb = std::getenv("KBO_LONG_WPB") ? std::atoi(std::getenv("KBO_LONG_WPB")) : 4; // experiments: waves per workgroup
    const uint32_t wpb = (uint32_t)std::min(4, std::max(1, env_wpb));
    const uint32_t n_waves = (L.n_slots + a.ppw - 1u) / a.ppw;
    if (a.pstats) hipLaunchKernelGGL(map_long_kernel<true>, dim3((n_waves + wpb - 1u) / wpb), dim3(64u * wpb), kLongLds * wpb, stream, a);
    else hipLaunchKernelGGL(map_long_kernel<false>, dim3((n_waves + wpb - 1u) / wpb), dim3(64u * wpb), kLongLds * wpb, stream, a);
    return hipGetLastError();
}

// the flagged pieces: their matching statistics by the plain walk (into d_ms), then the literal recurrences
hipError_t launch_map_long_redo(const LongArgs &a, uint8_t *d_ms, hipStream_t stream)
{
    hipLaunchKernelGGL(long_redo_items_kernel, dim3((a.n_items + 1023u) / 1024u), dim3(1024), 0, stream, a, static_cast<WalkItem *>(a.subs), a.sub_cap, a.qctl + 1,
                       a.flist);
    WalkArgs wa{};
    wa.ix = a.ix;
    wa.q = a.q;
    wa.q_bytes = a.q_bytes;
    wa.d_out = d_ms;
    // (a few per cent of the pieces are flagged: a lane per sub-item while there are kLongSubLanes or two per piece)
    const uint32_t lanes = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)a.n_items * 2u + 4096u, kLongSubLanes + 4096u), a.sub_cap);
    hipError_t e = launch_walk_list(wa, static_cast<const WalkItem *>(a.subs), a.sub_cap, a.qctl + 1, lanes, stream);
    if (e != hipSuccess) return e;
    // (a wave per run of flagged pieces; the list's length is on the device: the waves share it)
    hipLaunchKernelGGL(long_derand_kernel, dim3(std::min<uint32_t>(a.n_items, 4096u)), dim3(64), 0, stream, a, d_ms, a.flist);
    return hipGetLastError();
}

hipError_t long_read_stats(const void *d_work, size_t n_seqs, uint64_t total_bases, uint32_t k, uint32_t ctl[32], uint32_t *stats, hipStream_t stream)
{
    const LongLayout L = long_layout(n_seqs, total_bases, k);
    const uint8_t *w = static_cast<const uint8_t *>(d_work);
    hipError_t e = hipMemcpyAsync(ctl, w + L.ctl, 128, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(stats, w + L.pstats, kPlanStatSlots * kPlanStatWords * 4, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(stream);
}

} // namespace kbo
