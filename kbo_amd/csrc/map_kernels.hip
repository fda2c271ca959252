// map_kernels.hip — gfx950 (MI355X, CDNA4): kbo::map / kbo::matches for a batch of READS in one launch.
//
// The chain the reference runs per sequence (lib.rs:735-738 for map, lib.rs:624-627 for matches / find):
//     index::query_sbwt (index.rs:243-256)  ->  derandomize_ms_vec (derandomize.rs:269-288)  ->  translate_ms_vec
//     (translate.rs:263-293)  [-> format::relative_to_ref (format.rs:266-287)]
// as ONE kernel over 64 reads per wave, everything 2-bit packed except the bytes that leave:
//
//   map_reads_kernel   0. the wave's stretch of the query buffer (its 64 reads lie back to back) is loaded once, in whole
//                         lines, and turned into 2-bit digits on the way into LDS (a quarter of the bytes; a byte that is no
//                         base sends its read to the plain walk);
//                      1. seed: the first D bases of a read - D = what a seed must be deep, log4(rows) + 3 - are looked up in
//                         a table of text positions (DevIndexView::seed_pos): one load gives the read's diagonal of the
//                         path-cover text (plan_kernels.hip has the idea; there it took a table of intervals and a second,
//                         dependent load of the row's position);
//                      2. compare: 16 bases per XOR against the 2-bit text (DevIndexView::pc_tm: text and path-start marks
//                         interleaved, 0.5 B per row: 2.5 MB on a 5 Mbp index, L2-resident where the byte text was not);
//                      2b. (direct form) a read that leaves its diagonal - an insertion, a deletion, a chimera, a first seed that
//                         sat elsewhere - gets a SECOND diagonal seeded from its last bases and is cut where the two together
//                         mismatch least, behind the stretch both match (the cut is a break like a mismatch, with the second
//                         diagonal's ramp starting in front of it by the bases both match; no break of its own when it stands
//                         right behind a mismatch);
//                      3. the stretches behind the mismatches from the depth table (dtab_kernels.hip has the rule), the wave's
//                         mismatches dealt out to its lanes as in plan_kernel<FUSE>; the MS values - k where nothing
//                         happened, the ramp behind a mismatch, the table's values right behind it - are put together in LDS
//                         bytes by the lane that resolved the mismatch;
//                      4. derandomize + translate, right to left, one lane per read, in place over those bytes;
//                      5. the characters leave in whole lines, relative_to_ref applied on the way out (the bases come from the
//                         2-bit copy).
//                      (direct form, round 5: a window that is in the index and that neither the anchors nor its place settle is
//                      asked of the windows AROUND it - six entries, judged at the kernel's end; a diagonal on which every third
//                      base mismatches is a chance seed, and its read is judged like one without a seed.)
//   Reads it cannot finish - a base deeper than the table knows, more mismatches than a list holds, no seed and a deep match,
//   a byte that is no base - are flagged exactly as plan_kernel<FUSE> flags them; redo_collect_kernel + the plain walk give
//   their MS values and launch_derand_flagged their characters.  Nothing depends on a diagonal being right.
//
//   (IO = 1, 2)        packed-native: the reads come as 2-bit words (kbo_matches_batch_packed's layout) and go into the digit
//                      string as they are; with IO = 2 the characters leave as 2-bit words as well.
//
//   pack_text_kernel / seed_pos_kernel   build pc_tm and seed_pos on the device from the byte text and the interval table.
//
// Integer / byte work only: no MFMA.  One 64-lane wave per workgroup (the LDS of a CU then holds twelve of them).
#include "device_util.hpp"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

#ifndef KBO_STAGE_UNROLL
#define KBO_STAGE_UNROLL 6
#endif
#ifndef KBO_MAP_LB
#define KBO_MAP_LB 256 // (threads per workgroup the kernel is compiled for: KBO_MAP_WPB waves of 64)
#endif
namespace kbo {
extern std::atomic<int> g_plan_cap; // plan_kernels.hip: bases of a read in which a seed may start (kbo_set_plan)
namespace {

// stores bytes [lo, hi) of a 16-byte block to o + lo .. o + hi (0 <= lo <= hi <= 16)
__device__ __forceinline__ void st_range16(uint8_t *o, const uint4 &v, uint32_t lo, uint32_t hi)
{
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) {
        const uint32_t w = (t >> 2) == 0 ? v.x : (t >> 2) == 1 ? v.y : (t >> 2) == 2 ? v.z : v.w;
        if (t >= lo && t < hi) o[t] = (uint8_t)(w >> ((t & 3u) * 8u));
    }
}

// bit 2 (15 - j) set for every bit j of the low 16 bits of x that is set (the marks of 16 positions, spread to the digit grid)
__device__ __forceinline__ uint32_t spread_marks(uint32_t x)
{
    x = __builtin_bitreverse32(x) >> 16; // bit j -> bit 15 - j
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// pc_tm[u] = { 2-bit digits of text positions [16 u - kMapPad, 16 u - kMapPad + 16), first one most significant;
//              01 at every position that matches nothing: a path start, the padding, in front of / beyond the text }
// (text_padded holds positions -kPlanPad .. : kMapPad - kPlanPad is a multiple of 16)
__global__ __launch_bounds__(256) void pack_text_kernel(const uint8_t *__restrict__ text_padded, uint64_t n_bytes, uint2 *__restrict__ out,
                                                        uint64_t n_units)
{
    static_assert(kMapPad >= kPlanPad && (kMapPad - kPlanPad) % 16u == 0, "padding of the 2-bit text");
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_units) return;
    uint4 v = make_uint4(0, 0, 0, 0);
    const uint64_t b = 16ull * u - (kMapPad - kPlanPad); // (wraps for the units in front of the byte text: they stay all marks)
    if (16ull * u < kMapPad - kPlanPad) {
    } else if (b + 16u <= n_bytes) v = *reinterpret_cast<const uint4 *>(text_padded + b);
    else if (b < n_bytes) {
        uint8_t tmp[16];
        for (uint32_t t = 0; t < 16; t++) tmp[t] = b + t < n_bytes ? text_padded[b + t] : (uint8_t)0;
        __builtin_memcpy(&v, tmp, 16);
    }
    uint32_t code, valid;
    pack16(v, code, valid);
    out[u] = make_uint2(code, spread_marks(~valid & 0xFFFFu));
}

__global__ __launch_bounds__(256) void seed_pos_kernel(const uint2 *__restrict__ seed_tab, const uint32_t *__restrict__ pc_pos,
                                                       uint32_t *__restrict__ out, uint64_t n)
{
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const uint2 iv = seed_tab[w];
    // (bit 31: more than one row ends with this string - the position is the first one's; text positions are below 2^31 wherever
    // this table is made: copies with a depth table have at most 1.2 * 10^9 rows)
    out[w] = iv.x < iv.y ? (pc_pos[iv.x] | (iv.y - iv.x > 1u ? 0x80000000u : 0u)) : 0xFFFFFFFFu;
}

__device__ __forceinline__ bool thr_gt(uint32_t thr, uint32_t order) { return thr > order; } // (windows order .. thr apart exist)
constexpr uint32_t kMapWords = 10;       // 16-base words per read: reads of up to 160 bases
// experiments of earlier rounds (KBO_MAP_X bits 0 - 4: ambiguous seeds as they come, a second filter stretch, half-window seed
// retries, no search beside the diagonal, no look at the window one base on): compiled in only with -DKBO_MAP_EXPERIMENTS - the second
// filter stretch alone is two more words per window live through the proof's loop.  Bits 5 and 6 (this round's rules off) stay: they sit
// in rare paths (tools/dbg_witness.py)
#ifdef KBO_MAP_EXPERIMENTS
constexpr bool kMapExp = true;
#else
constexpr bool kMapExp = false;
#endif
constexpr uint32_t kMapSlack = 48;       // bytes of the byte region behind the staged stretch
// windows that are in the index and wait for the anchors / the windows around them, per wave (two or three at 1 % substitutions, a
// dozen at 5 %; more: their reads take the plain walk).  20: a wave of 64 reads of 150 bases then takes 13 312 bytes of LDS - 26
// allocation units of 512 - and a CU holds TWELVE of them; with 64 entries it held eleven
constexpr uint32_t kMapPend = 20, kMapPendBytes = kMapPend * 8u + 16u;
// the wave's LDS: [characters] [digits: lin_words + 4 words] [64 x 16 bytes of mismatch lists] [pending windows].  The characters of
// the direct form are 2-BIT CODES (round 6: M - R X = 0 1 2 3, sixteen a word, expanded at the whole-line store) - a quarter of the
// bytes they were: 6.1 instead of 13.3 KB a wave of 150-base reads, so that the kernel's registers (74: six waves a SIMD) and no longer
// its LDS (twelve waves a CU) bound the resident waves - what large indexes, whose every look-up is a trip to HBM, run on (C3: time =
// 0.96 + 15.7 / waves ms).  The other instantiation keeps a byte a base: its region holds the MS values.
__host__ __device__ __forceinline__ uint32_t map_char_bytes(bool direct, uint32_t stage_bytes) { return direct ? ((stage_bytes >> 2) + 15u) & ~15u : stage_bytes; }
__host__ __device__ __forceinline__ uint32_t map_wave_lds(bool direct, uint32_t stage_bytes, uint32_t lin_words)
{
    return (map_char_bytes(direct, stage_bytes) + 4u * (lin_words + 4u) + 1024u + kMapPendBytes + 15u) & ~15u;
}
// a character's code: OR-able in the order the closed form writes them ('-' first, then perhaps 'X' over it; 'R' only over 'M')
constexpr uint32_t kCodeDash = 1u, kCodeR = 2u, kCodeX = 3u; // ('M' = 0: what a word of codes starts as)

// NP = bases a stretch can take: 16 (tables of up to 15 bases: 32-bit keys) or 18 (16 / 17 bases).
// DIRECT: the characters straight from the mismatch positions, no MS bytes at all.  Where the table's order is at most the
// derandomisation threshold t (and t < k: always, with the default error probability) every value the table certifies - the
// stretch behind a mismatch - is <= t, and derandomize_ms_val (derandomize.rs:233-246) then never anchors inside a stretch:
// with x the derandomised value, j the distance to the last mismatch m, L the distance from m to the next one (or the end),
//     x[m + j] = k for j >= k,  j + d otherwise, one d per segment:  d = 0 when L > k;  at the read's end d = 0 when L - 1 > t,
//     else -(L - 1);  else with d' the next segment's:  d = 0 when d' - L <= -2 and L - 1 > t (the ramp's top value anchors),
//     else d' - L
// (derivation: DESIGN.md section 4.9; the in-place pass of the other instantiation is the literal recurrence, and
// tests/test_gpu_map_reads.py runs both against the oracle).  x rises by one per base inside a segment and is <= 0 behind
// a break, so translate_ms_val (translate.rs:180-216) never sees its 'R' case: every base is 'M' except the -d + 1 bases
// from a mismatch on - '-', or 'X' for a lone mismatch whose left neighbour is positive - and the two first bases of the
// read, whose `prev` is k (translate.rs:277).
// What the table has to PROVE for that is only that no string of t + 1 bases that contains a mismatch is a substring of the index
// (the matching statistic is then at most t wherever it is not the distance to the last mismatch).  A present string of t + 1
// bases has every window of `order` bases inside it present, and the windows inside it that contain the mismatch m end at
// cov = t - order + 2 consecutive bases: so the windows ending at m, m + cov, m + 2 cov, .., m + order - 1 - three of them at
// C2 (15 bases, t = 22) and C3 (17, 24) - each not a suffix of any row (bit 7 of their entries clear), rule every such string
// out: three byte look-ups per mismatch instead of the fourteen of the value-by-value rule, no rule to evaluate.  A read
// without a seed has every string of t + 1 bases ruled out the same way (a window every cov bases): all its bases are '-'.
// IO: 0 = the reads as bytes, the characters as bytes;  1 = the reads as 2-bit words (kbo_matches_batch_packed's layout: every read
// starts a word, first base in the lowest bits), the characters as bytes;  2 = both as 2-bit words (DIRECT without relative_to_ref
// only: the alphabet is M - X R).  The words go straight into the digit string (every read then starts a word of it: 16 bases of
// padding at most between two reads), no byte of a read is ever staged, and with IO = 2 a base costs a quarter of a byte each way.
// Reads that hold a byte that is no base come with a flag (launch_flag_exceptions) and take the plain walk, like the reads this
// kernel cannot finish: launch_unpack_flagged gives those their bytes, launch_pack_flagged packs their characters.
__device__ __forceinline__ uint32_t reverse_digits(uint32_t w)
{
    const uint32_t r = __builtin_bitreverse32(w);
    return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
// derandomize_ms_vec + translate_ms_vec of one read, right to left, in place: `at` holds its len >= 3 matching statistics (bytes, LDS)
// and then its characters M - X R (derand_kernels.hip has the derivation of the closed form of the 'R','R' look-ahead): x = derandomised
// value, window (x_prev, x_cur, x_next) = x[p-1], x[p], x[p+1]
__device__ __forceinline__ void literal_chars(uint8_t *at, uint32_t len, int K, int T)
{
    const uint32_t Tm1 = (uint32_t)(T - 1);
    auto step = [&](int av, int x_cur) { return (av == K) ? K : ((av > T && x_cur < av) ? av : x_cur - 1); };
    auto plain = [&](int x_cur, int next, int prev) -> uint32_t {
        return x_cur <= 0 ? ((next == 1 && prev > 0) ? (uint32_t)'X' : (uint32_t)'-') : (uint32_t)'M';
    };
    int av = at[len - 1];
    int x_cur = av > T ? av : 0; // derandomize.rs:282
    int a_below = at[len - 2];
    int x_prev = step(a_below, x_cur);
    bool in_cur = (uint32_t)(x_cur - 1) < Tm1, gt_cur = x_cur > T;
    at[len - 1] = (uint8_t)((gt_cur && in_cur) ? (uint32_t)'R' : plain(x_cur, x_cur, x_prev));
    int x_next = x_cur;
    bool in_next = in_cur;
    x_cur = x_prev;
    gt_cur = x_cur > T;
    in_cur = (uint32_t)(x_cur - 1) < Tm1;
    a_below = at[len - 3];
#pragma unroll 4
    for (uint32_t p = len - 2; p >= 2; p--) {
        x_prev = step(a_below, x_cur);
        a_below = at[p - 2];
        const bool gt_prev = x_prev > T;
        const bool is_r = (gt_prev && in_cur) || (gt_cur && in_next);
        at[p] = (uint8_t)(is_r ? (uint32_t)'R' : plain(x_cur, x_next, x_prev));
        x_next = x_cur;
        in_next = in_cur;
        x_cur = x_prev;
        gt_cur = gt_prev;
        in_cur = (uint32_t)(x_cur - 1) < Tm1;
    }
    x_prev = step(a_below, x_cur); // a_below == at[0]
    at[1] = (uint8_t)((gt_cur && in_next) ? (uint32_t)'R' : plain(x_cur, x_next, K));
    x_next = x_cur;
    in_next = in_cur;
    x_cur = x_prev;
    at[0] = (uint8_t)((x_cur > T && in_next) ? (uint32_t)'R' : plain(x_cur, x_next, K));
}
// STATS: the kernel counts its own work (kbo_set_plan_stats) - instrumentation, compiled out of the default instantiations
template <int NP, bool DIRECT, int IO = 0, bool STATS = false>
__global__ __launch_bounds__(KBO_MAP_LB) void map_reads_kernel(WalkArgs a, uint32_t stage_bytes, uint32_t lin_words)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t map_lds_all[];
    const uint32_t lane = threadIdx.x & 63u;
    // (the waves of a workgroup share nothing: each has its own part of the LDS and never waits for another)
    const uint32_t wave_lds = map_wave_lds(DIRECT, stage_bytes, lin_words), char_bytes = map_char_bytes(DIRECT, stage_bytes);
    uint8_t *map_lds = map_lds_all + (threadIdx.x >> 6) * wave_lds;
    const uint32_t idx = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64u + lane;
    uint8_t *so = map_lds;                                                       // (not DIRECT) MS bytes, then characters: the wave's stretch
    uint32_t *cw = reinterpret_cast<uint32_t *>(map_lds);                        // (DIRECT) the characters as 2-bit codes: position p of the stretch in bits 2 (p mod 16) of word p / 16
    uint32_t *lin = reinterpret_cast<uint32_t *>(map_lds + char_bytes) + 4;      // the stretch as 2-bit digits (lin[-1] = 0)
    uint8_t *spw = map_lds + char_bytes + 4u * (lin_words + 4u);                 // 64 x 16 bytes: mismatch positions 0 .. 12, flag, prefix
    uint8_t *sp = spw + lane * 16u;
    uint2 *pend = reinterpret_cast<uint2 *>(spw + 1024u); // DIRECT: windows that are present and wait for their exact depth (kMapPend entries)
    uint32_t *pend_n = reinterpret_cast<uint32_t *>(spw + 1024u + kMapPend * 8u);
    const uint32_t k = a.ix.k;
    const uint8_t *qb = a.q;
    uint32_t start = 0, len = 0, warm = 0, tail = 0;
    const bool have_item = idx < a.n_items;
    if (have_item && a.seq_off && a.uniform_len) { // (reads of one length: no offset but the batch's first is looked at)
        start = (uint32_t)a.seq_off[0] + idx * a.uniform_len;
        len = a.uniform_len;
    } else if (have_item && a.seq_off) { // (8 bytes per read instead of a 16-byte record somebody had to write first)
        const uint64_t o0 = a.seq_off[idx], o1 = a.seq_off[idx + 1u];
        start = (uint32_t)o0; // launches cover < 4 GiB of query
        len = (uint32_t)(o1 - o0);
    } else if (have_item) {
        const uint4 it = ld16(reinterpret_cast<const uint8_t *>(a.items), idx * 16u);
        start = it.x;
        len = it.z;
        warm = it.w & 0xFFFFu;
        tail = it.w >> 16;
    }
    const bool plannable = have_item && len > 0;
    const uint32_t cap = a.plan_cap;

    // ---- 0. the wave's stretch: reads back to back, whole (no warm-up bases: these are reads, not chunks)
    const uint64_t have = __ballot(have_item); // (item lanes are the wave's first lanes)
    if (have == 0) return;
    const uint32_t last = (uint32_t)__popcll(have) - 1u;
    const uint32_t nxt_start = __shfl_down(start, 1);
    const bool bad_item = have_item && ((lane < last && nxt_start != start + len) || warm != 0 || tail != 0 || len > 16u * kMapWords);
    const uint32_t lo = __shfl(start, 0), wave_hi = __shfl(start + len, (int)last);
    // ooff: where the read's characters (MS bytes) stand in the byte region, soff: where its bases stand in the digit string
    const uint32_t base16 = lo & ~15u, ooff_b = start - base16, span_b = wave_hi - base16;
    uint32_t wst = 0; // IO != 0: the read's first word
    if (IO != 0 && have_item) wst = a.qp_wps ? idx * a.qp_wps : a.qp_sums[idx / kScanBlock] + a.qp_data[idx];
    const uint32_t nw_mine = (len + 15u) >> 4;
    const uint32_t w_lo = __shfl(wst, 0), w_hi = __shfl(wst + nw_mine, (int)last), nwords = w_hi - w_lo;
    const uint32_t soff = IO != 0 ? 16u * (wst - w_lo) : ooff_b, ooff = IO == 2 ? soff : ooff_b;
    const uint32_t span = IO == 2 ? 16u * nwords : span_b;
    const uint32_t nxt_wst = __shfl_down(wst, 1);
    const bool bad_words = IO != 0 && have_item && lane < last && nxt_wst != wst + nw_mine;
    const bool staged = __ballot(bad_item || bad_words) == 0 && wave_hi > lo && (uint64_t)span + kMapSlack <= stage_bytes &&
                        (IO == 0 || (uint64_t)16u * nwords + kMapSlack <= stage_bytes);
    if (!staged) { // (cannot happen for a batch of reads the host sent here; if it does, every item takes the plain walk)
        if (have_item) a.redo[idx] = len != 0 ? 1 : 0;
        const uint64_t fm = __ballot(have_item && len != 0);
        if (lane == 0 && fm) atomicAdd(a.qctl + 4, (uint32_t)__popcll(fm));
        if (have_item && len != 0) reinterpret_cast<uint32_t *>(a.units)[atomicAdd(a.qctl + 6, 1u)] = idx;
        return;
    }
    const uint32_t nblk = IO != 0 ? nwords : (span + 15u) >> 4;
    bool has_invalid = false;
    if (lane == 0) lin[-1] = 0;
    if (IO != 0) {
        for (uint32_t c0 = 0; c0 < nblk + 2u; c0 += 64u) {
            const uint32_t c = c0 + lane;
            if (c < lin_words) lin[c] = c < nblk ? reverse_digits(a.qp[w_lo + c]) : 0u;
        }
        for (uint32_t c = lane; 16u * c < span + 16u; c += 64u) cw[c] = 0u; // (the packed forms are DIRECT: every character starts as 'M')
        has_invalid = plannable && a.qp_exc != nullptr && a.qp_exc[idx] != 0;
    } else {
    // (six blocks per lane in flight: the loads of a stretch go out in two rounds instead of one per 1 KB (eight or eleven at a time:
    // the same, 217 - 219 us) - a wave's life is the
    // sum of its dependent memory rounds, 19 of them before this, and the kernel's time follows it: neither fewer fills, nor fewer
    // instructions, nor more resident waves had changed it)
    constexpr uint32_t kStageUnroll = KBO_STAGE_UNROLL;
    for (uint32_t c00 = 0; c00 < nblk + 2u; c00 += 64u * kStageUnroll) {
        uint4 vv[kStageUnroll];
#pragma unroll
        for (uint32_t u = 0; u < kStageUnroll; u++) {
            const uint32_t c = c00 + 64u * u + lane;
            vv[u] = make_uint4(0, 0, 0, 0);
            if (c < nblk) vv[u] = ld16u(qb, base16 + 16u * c); // (reads <= 15 bytes past the last read)
        }
#pragma unroll
        for (uint32_t u = 0; u < kStageUnroll; u++) {
            const uint32_t c0 = c00 + 64u * u;
            if (c0 < nblk + 2u) {
            const uint32_t c = c0 + lane;
            uint32_t code = 0, valid = 0xFFFFu;
            if (c < nblk) {
                const uint4 v = vv[u];
                // only the bytes of this wave's reads count: [lo, wave_hi)
                const uint32_t b0 = base16 + 16u * c;
                const uint32_t from = lo > b0 ? lo - b0 : 0u, to = min(16u, wave_hi - b0);
                if (from == 0u && to == 16u) { // (all but the wave's first and last block)
                    bool bad;
                    pack16_whole(v, code, bad);
                    valid = bad ? 0u : 0xFFFFu;
                } else {
                    pack16(v, code, valid);
                    const uint32_t inr = ((1u << to) - 1u) & ~((1u << from) - 1u);
                    valid |= ~inr;
                }
            }
            if (c < lin_words) lin[c] = code;
            // DIRECT: the characters start out as what nearly all of them are - 'M' (with relative_to_ref: the read's own base, which the
            // digits keep)
            if (DIRECT && 4u * c < char_bytes) cw[c] = 0u;
            uint64_t bm = __ballot((valid & 0xFFFFu) != 0xFFFFu);
            while (bm) { // (rare: a byte that is no base - its read, and a neighbour that shares the block, take the plain walk)
                const uint32_t L = (uint32_t)__ffsll((long long)bm) - 1u;
                bm &= bm - 1ull;
                const uint32_t blo = 16u * (c0 + L);
                has_invalid = has_invalid || (plannable && soff < blo + 16u && soff + len > blo);
            }
            }
        }
    }
    }
    if (!DIRECT)
        for (uint32_t c = lane * 16u; c < span + 16u; c += 1024u) // MS bytes: k wherever nothing says otherwise
            *reinterpret_cast<uint4 *>(so + c) = make_uint4(k * 0x01010101u, k * 0x01010101u, k * 0x01010101u, k * 0x01010101u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // 16 bases of the stretch from base S on (first one most significant) / the 16 bases ending at base E (last one least)
    auto from_base = [&](uint32_t S) -> uint32_t {
        const uint32_t W = S >> 4, r = S & 15u;
        const uint64_t V = ((uint64_t)lin[W] << 32) | lin[W + 1u];
        return (uint32_t)(V >> (32u - 2u * r));
    };
    auto ending_at = [&](uint32_t E) -> uint64_t { // (17 + E mod 16 bases are there: enough for the windows of every table)
        const uint32_t W = E >> 4, r = E & 15u;
        const uint64_t V = ((uint64_t)lin[(int32_t)W - 1] << 32) | lin[W];
        return V >> (2u * (15u - r));
    };
    auto base_at = [&](uint32_t S) -> uint32_t { return (lin[S >> 4] >> (2u * (15u - (S & 15u)))) & 3u; };

    // ---- 1. seed: one table look-up gives the diagonal
    uint32_t qw[kMapWords];
#pragma unroll
    for (uint32_t g = 0; g < kMapWords; g++) qw[g] = (plannable && 16u * g < len) ? from_base(soff + 16u * g) : 0u;
    // A seed is a window of the read whose text position a table knows: the D = log4(rows) + 3 <= 14 bases of seed_pos (any row that
    // ends with them: good while such a string is rare in the index - 2 % of all strings at 5 * 10^6 rows - but at 2.5 * 10^8 rows
    // a string of 14 bases occurs 0.9 times by chance and half of those seeds sat on another occurrence), or, where the copy has
    // them, the ANCHORS of its depth table: the strings of `order` (16, 17) bases that are the suffix of exactly one row, hashed to
    // that row's text position.  A wrong seed costs nothing but the second attempt: the compare step shows it.
    // (where the seed table's strings are nearly as long as the anchors' - small indexes: 14 against 15 bases - the table's one
    // load is the better seed: kernel 0.241 against 0.270 ms at C2; the anchors then only price windows that are present)
    const bool by_anchor = a.ix.anchor != nullptr && a.ix.dtab_order >= 12u && a.ix.dtab_order > a.ix.seed_d + 1u;
    const uint32_t D = by_anchor ? a.ix.dtab_order : a.ix.seed_d;
    const uint32_t dmask = D >= 16u ? 0xFFFFFFFFu : ((1u << (2u * D)) - 1u);
    auto seed_at = [&](uint32_t e_) -> uint32_t { // text position of the base e_ of the read by the window that ends there, or ~0
        const uint64_t win = ending_at(soff + e_);
        if (!by_anchor) return a.ix.seed_pos[(uint32_t)win & dmask];
        const uint64_t key = win & ((1ull << (2u * D)) - 1ull), amask = ((uint64_t)1 << a.ix.anchor_bits) - 1ull;
        uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> (64u - a.ix.anchor_bits);
        const uint32_t tag = (uint32_t)key + 1u;
        for (uint32_t probe = 0; probe < 16u; probe++) {
            const uint64_t slot = a.ix.anchor[h];
            if (slot == 0) break;
            if ((uint32_t)(slot >> 32) == tag) return (uint32_t)slot;
            h = (h + 1u) & amask;
        }
        return 0xFFFFFFFFu;
    };
    const uint32_t jmax = min(len, cap);
    uint32_t e = D - 1u, p0 = 0, st_lookups = 0, p_amb = 0;
    bool seeded = false, have_amb = false;
    for (;;) {
        const bool act = plannable && !has_invalid && !seeded && e < jmax;
        if (__ballot(act) == 0) break;
        if (act) {
            const uint32_t tp = seed_at(e);
            if (STATS) st_lookups++;
            if (tp != 0xFFFFFFFFu && (by_anchor || !(tp & 0x80000000u) || (kMapExp && (a.rounds & 1u)))) { // (a.rounds: experiment switches)
                seeded = true;
                p0 = (by_anchor ? tp : (tp & 0x7FFFFFFFu)) - e;
            } else {
                // a string that ends several rows (2 % of the seeds at 5 * 10^6 rows, the wrong row half of the time - and a wrong
                // diagonal sends the whole wave through the second-diagonal search): kept for later, the next window first
                if (tp != 0xFFFFFFFFu && !have_amb) {
                    have_amb = true;
                    p_amb = (tp & 0x7FFFFFFFu) - e;
                }
                e += (kMapExp && (a.rounds & 4u)) ? (D + 1u) / 2u : D; // (the next window shares no base with this one: whatever broke this one - a
                                                          // substitution in 13 reads of 100 - does not break that one too; half a window on: 225 against 220 us)
            }
        }
    }
    if (plannable && !has_invalid && !seeded && have_amb) { // (no window that ends one row only: a repeat - its first copy will do)
        seeded = true;
        p0 = p_amb;
    }

    // ---- 2. compare with the text on the diagonal: mm[g] has bit 2 (15 - j) set where base 16 g + j differs from the text (or the
    // text has a mark there: a path start, the padding)
    auto compare = [&](uint32_t pd, uint32_t (&mm)[kMapWords]) -> uint32_t {
        const uint32_t q0 = pd + kMapPad, r = q0 & 15u;
        const uint8_t *tp = reinterpret_cast<const uint8_t *>(a.ix.pc_tm) + (size_t)(q0 >> 4) * 8u;
        uint4 L[6];
#pragma unroll
        for (int i = 0; i < 6; i++) __builtin_memcpy(&L[i], tp + 16 * i, 16);
        uint32_t T[12], M[12];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            T[2 * i] = L[i].x;
            M[2 * i] = L[i].y;
            T[2 * i + 1] = L[i].z;
            M[2 * i + 1] = L[i].w;
        }
        uint32_t c = 0;
#pragma unroll
        for (uint32_t g = 0; g < kMapWords; g++) {
            mm[g] = 0;
            if (16u * g < len) {
                const uint32_t tw = (uint32_t)((((uint64_t)T[g] << 32) | T[g + 1]) >> (32u - 2u * r));
                const uint32_t mk = (uint32_t)((((uint64_t)M[g] << 32) | M[g + 1]) >> (32u - 2u * r));
                const uint32_t x = qw[g] ^ tw;
                uint32_t m1 = (x | (x >> 1) | mk) & 0x55555555u;
                const uint32_t nb = len - 16u * g;
                if (nb < 16u) m1 &= ~0u << (32u - 2u * nb);
                mm[g] = m1;
                c += (uint32_t)__popc(m1);
            }
        }
        return c;
    };
    uint32_t cnt = 0;
    uint32_t junction = 0xFFu; // DIRECT: index in the read's list of the entry that is no mismatch but the last base of its first diagonal
    uint32_t jov = 0;          // ... and the bases in front of the cut that lie on both diagonals
    uint32_t mmw[kMapWords];
#pragma unroll
    for (uint32_t g = 0; g < kMapWords; g++) mmw[g] = 0;
    if (__ballot(seeded)) {
        if (seeded) cnt = compare(p0, mmw);
    }
    // ---- 2b. (DIRECT) a SECOND diagonal for a read that leaves its first - an insertion or a deletion, a chimera, or a first seed
    // that sat on a substitution and matched elsewhere (3 reads in 1000; all their bases mismatch from there on) -, and for a read
    // whose first bases gave no seed: seeded from its LAST bases backwards; the read is cut where the two diagonals together
    // mismatch least - [0, g) on the first, [g, len) on the second.  The cut is a break like a mismatch (nothing deeper than the
    // threshold may run through it: the table proves that too), except that its base g - 1 keeps the value of the first
    // diagonal: list entry `junction`.  Whatever is chosen here only decides how many reads take the plain walk.
    if (DIRECT) {
        // (also a read whose list would hold its mismatches but has six of them among its first or its last 16 bases: a break
        // within 16 + 8 bases of an end - windows behind it lie on the other diagonal, are in the index, and fail the proof)
        const uint32_t listmax = a.plan_list + 1u;
        uint32_t tail_c = 0;
#pragma unroll
        for (uint32_t g = 0; g < kMapWords; g++) {
            const uint32_t lo_b = 16u * g, t0 = len - 16u;
            if (lo_b + 16u > t0 && lo_b < len) tail_c += (uint32_t)__popc(mmw[g] & (t0 > lo_b ? (1u << (2u * (16u - (t0 - lo_b)))) - 1u : ~0u));
        }
        const bool dense = seeded && len >= 32u && (tail_c >= 6u || (uint32_t)__popc(mmw[0]) >= 6u) && !(kMapExp && (a.rounds & 8u));
        const bool need2 = plannable && !has_invalid && len >= 2u * D && (!seeded || cnt > listmax || dense);
        if (__ballot(need2)) {
            uint32_t pB = 0, eb = len - 1u, tries = 0;
            bool seedB = false;
            const uint32_t lowest = len > cap ? len - cap : 0u;
            // A second diagonal one to three bases beside the first - a short insertion or deletion, most of what this phase sees -
            // needs no table: the read's last 16 bases against the text at p0 - 3 .. p0 + 3, out of three units of the text that
            // the compare step has just had in cache, instead of one to three dependent look-ups in a table that is not.  The
            // same for its first 16 bases: the seed then came from behind the break (a substitution in the read's first window),
            // and the diagonal found beside it becomes the first one.  (a first diagonal that explains less than a third of the
            // read is a wrong seed: the table.)
            const bool near = need2 && seeded && 10u * cnt <= 7u * len && !(kMapExp && (a.rounds & 8u));
            bool front = false;
            uint32_t pA = 0;
            if (__ballot(near)) {
                if (near) {
                    // qb: the text base three in front of the 16 bases on p0; -> fewest mismatches over the six shifts, j = shift + 3
                    const uint32_t qt = p0 + kMapPad + len - 19u, qh = p0 + kMapPad - 3u;
                    const uint8_t *tt = reinterpret_cast<const uint8_t *>(a.ix.pc_tm) + (size_t)(qt >> 4) * 8u;
                    const uint8_t *th = reinterpret_cast<const uint8_t *>(a.ix.pc_tm) + (size_t)(qh >> 4) * 8u;
                    uint2 U[3], H[3];
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        __builtin_memcpy(&U[i], tt + 8 * i, 8);
                        __builtin_memcpy(&H[i], th + 8 * i, 8);
                    }
                    // (a break within 16 bases of that end: the 8 bases at the end, all of them - 6 / 4^8 by chance)
                    auto beside = [&](const uint2 (&X)[3], uint32_t qb, uint32_t word, uint32_t end8, uint32_t &best_j) -> uint32_t {
                        uint32_t best = 99u, j8 = 3u;
                        best_j = 3u;
#pragma unroll
                        for (uint32_t j = 0; j < 7u; j++) {
                            if (j == 3u) continue;
                            const uint32_t rel = (qb & 15u) + j, r = rel & 15u;
                            const uint2 hi = rel < 16u ? X[0] : X[1], lo = rel < 16u ? X[1] : X[2];
                            const uint32_t tw = (uint32_t)((((uint64_t)hi.x << 32) | lo.x) >> (32u - 2u * r));
                            const uint32_t mk = (uint32_t)((((uint64_t)hi.y << 32) | lo.y) >> (32u - 2u * r));
                            const uint32_t x = word ^ tw;
                            const uint32_t mb = (x | (x >> 1) | mk) & 0x55555555u, c = (uint32_t)__popc(mb);
                            if (c < best) {
                                best = c;
                                best_j = j;
                            }
                            if ((mb & end8) == 0u) j8 = j;
                        }
                        if (best > 2u && j8 != 3u) {
                            best = 0u;
                            best_j = j8;
                        }
                        return best;
                    };
                    uint32_t jt, jh;
                    const uint32_t ct = beside(U, qt, from_base(soff + len - 16u), 0x0000FFFFu, jt), ch = beside(H, qh, qw[0], 0xFFFF0000u, jh);
                    // (by chance: 6 * 1129 / 4^16 = 1.6e-6 - and then it costs a flag, not a wrong character)
                    if (ct <= 2u) {
                        seedB = true;
                        pB = p0 + jt - 3u;
                    } else if (ch <= 2u) {
                        front = true;
                        pA = p0 + jh - 3u;
                    }
                }
            }
            if (__ballot(front)) {
                if (front) { // the seed's diagonal is the second one
                    pB = p0;
                    p0 = pA;
                    cnt = compare(p0, mmw);
                    seedB = true;
                }
            }
            // (three windows: a read that follows a second diagonal to its end has a seed there unless substitutions sit in all of
            // them; every further window is another dependent load for the whole wave - reads that match nothing pay them all)
            for (;; tries++) {
                const bool act = need2 && !seedB && (!seeded || cnt > listmax) && tries < 3u && eb + 1u >= D + lowest && eb + 1u >= D;
                if (__ballot(act) == 0) break;
                if (act) {
                    const uint32_t tp = seed_at(eb);
                    if (STATS) st_lookups++;
                    if (tp != 0xFFFFFFFFu) {
                        seedB = true;
                        pB = (by_anchor ? tp : (tp & 0x7FFFFFFFu)) - eb;
                    } else eb -= min(eb, D);
                }
            }
            const bool two = seedB && (!seeded || pB != p0);
            if (__ballot(two)) {
                uint32_t mmB[kMapWords];
#pragma unroll
                for (uint32_t g = 0; g < kMapWords; g++) mmB[g] = 0;
                uint32_t cB = 0;
                if (two) cB = compare(pB, mmB);
                if (two && !seeded) { // no first diagonal: the whole read on the second
#pragma unroll
                    for (uint32_t g = 0; g < kMapWords; g++) mmw[g] = mmB[g];
                    cnt = cB;
                    p0 = pB;
                    seeded = true;
                } else if (two) {
                    // the cut: first by words (mismatches of the first diagonal in front of word w + of the second from it on) ...
                    uint32_t best_w = 0, best_c = cB, run = 0, suf = cB;
#pragma unroll
                    for (uint32_t w = 0; w < kMapWords; w++) {
                        run += (uint32_t)__popc(mmw[w]);
                        suf -= (uint32_t)__popc(mmB[w]);
                        if (16u * (w + 1u) <= len + 15u && run + suf <= best_c) { // (ties: the rightmost)
                            best_c = run + suf;
                            best_w = w + 1u;
                        }
                    }
                    // ... then base by base through the word in front of that boundary and the one behind it
                    const uint32_t w0 = best_w > 0u ? best_w - 1u : 0u;
                    uint32_t a0 = 0, a1 = 0, b0 = 0, b1 = 0, c0 = cB; // the two words of either mask, the cost of cutting at 16 w0
#pragma unroll
                    for (uint32_t w = 0; w < kMapWords; w++) {
                        a0 = w == w0 ? mmw[w] : a0;
                        a1 = w == w0 + 1u ? mmw[w] : a1;
                        b0 = w == w0 ? mmB[w] : b0;
                        b1 = w == w0 + 1u ? mmB[w] : b1;
                        if (w < w0) c0 = c0 + (uint32_t)__popc(mmw[w]) - (uint32_t)__popc(mmB[w]);
                    }
                    uint32_t gcut = 16u * w0, cmin = c0, cc = c0;
                    for (uint32_t x = 0; x < 32u; x++) { // cutting behind base 16 w0 + x instead of in front of it
                        const uint32_t wa = x < 16u ? a0 : a1, wb = x < 16u ? b0 : b1, sh = 2u * (15u - (x & 15u));
                        cc = cc + ((wa >> sh) & 1u) - ((wb >> sh) & 1u);
                        if (16u * w0 + x + 1u <= len && cc <= cmin) { // (ties: the rightmost - the first diagonal as far as it matches)
                            cmin = cc;
                            gcut = 16u * w0 + x + 1u;
                        }
                    }
                    if (cB <= cmin) gcut = 0u; // (the second diagonal alone is as good: no cut)
                    if (gcut == 0u) { // all of it on the second diagonal
#pragma unroll
                        for (uint32_t g = 0; g < kMapWords; g++) mmw[g] = mmB[g];
                        cnt = cB;
                        p0 = pB;
                    } else if (gcut < len) {
                        // bases [0, gcut) keep the first diagonal's marks, [gcut, len) get the second's; base gcut - 1 gets a mark as
                        // the junction unless it mismatches already (a mismatch there is the same break)
                        uint32_t jm = 0xFFu, before = 0;
                        int32_t lastA = -1; // the first diagonal's last mismatch in front of the cut
                        cnt = 0;
#pragma unroll
                        for (uint32_t g = 0; g < kMapWords; g++) {
                            const uint32_t lo_b = 16u * g; // bases of this word: [lo_b, lo_b + 16)
                            uint32_t keepA = gcut <= lo_b ? 0u : (gcut >= lo_b + 16u ? ~0u : ~0u << (32u - 2u * (gcut - lo_b)));
                            if (mmw[g] & keepA) lastA = (int32_t)(lo_b + 15u - ((uint32_t)__builtin_ctz(mmw[g] & keepA) >> 1));
                            uint32_t m1 = (mmw[g] & keepA) | (mmB[g] & ~keepA);
                            if (gcut - 1u >= lo_b && gcut - 1u < lo_b + 16u) {
                                const uint32_t bit = 0x40000000u >> (2u * (gcut - 1u - lo_b));
                                if (!(m1 & bit)) {
                                    jm = before + (uint32_t)__popc(m1 & ~(bit | (bit - 1u))); // entries in front of it
                                    m1 |= bit;
                                }
                            }
                            before += (uint32_t)__popc(m1);
                            mmw[g] = m1;
                            cnt += (uint32_t)__popc(m1);
                        }
                        junction = jm;
                        // the bases right in front of the cut that lie on the second diagonal as well (an insertion or deletion inside
                        // a run of equal bases, a base that matches both by chance): its ramp starts that much earlier, and the
                        // windows of its proof have to reach back to the base in front of them
                        if (jm != 0xFFu) {
                            const uint32_t E = gcut - 1u, W = E >> 4;
                            uint32_t wlo = 0, whi = 0;
#pragma unroll
                            for (uint32_t g = 0; g < kMapWords; g++) {
                                wlo = g == W ? mmB[g] : wlo;
                                whi = g + 1u == W ? mmB[g] : whi;
                            }
                            const uint64_t V = ((((uint64_t)whi << 32) | wlo) >> (2u * (15u - (E & 15u)))) & 0x5555555555555555ull; // flag of base E - x at bit 2 x
                            const uint32_t run0 = V ? (uint32_t)__builtin_ctzll(V) >> 1 : 17u;
                            if (lastA >= 0 && run0 >= E - (uint32_t)lastA) {
                                // every base between the first diagonal's last mismatch and the cut lies on the second diagonal as
                                // well (the last inserted base, then a base or two that match both by chance): that mismatch is the
                                // break, the second diagonal's stretch starts behind it - no entry of its own for the cut
                                const uint32_t bit = 0x40000000u >> (2u * (E & 15u));
#pragma unroll
                                for (uint32_t g = 0; g < kMapWords; g++) mmw[g] = g == W ? mmw[g] & ~bit : mmw[g];
                                cnt--;
                                junction = 0xFFu;
                            } else
                                jov = min(min(run0, a.ix.dtab_order >= 3u ? a.ix.dtab_order - 3u : 0u), min(E - (uint32_t)(lastA + 1), gcut - 1u));
                        }
                    }
                }
            }
        }
    }
    if (__ballot(seeded)) {
        uint32_t filled = 0;
#pragma unroll
        for (uint32_t g = 0; g < kMapWords; g++) {
            uint32_t mm = mmw[g];
            while (__ballot(mm != 0)) {
                if (mm) {
                    const uint32_t j = (uint32_t)__clz((int)mm) >> 1;
                    if (filled < 13u) sp[filled] = (uint8_t)(16u * g + j);
                    filled++;
                    mm &= ~(0x40000000u >> (2u * j));
                }
            }
        }
    }
    // (DIRECT) a diagonal on which every third base mismatches is no diagonal of this read: a seed of D bases that some other place of
    // the index shares by chance - one window in fifty at 5 * 10^6 rows, so one read in fourteen of another genome or the other
    // strand.  Such a read is judged like one without a seed (every window of the proof: absent, or settled by the windows
    // around it) instead of being left to the plain walk for its list's length.
    if (DIRECT && seeded && cnt > a.plan_list + 1u && 3u * cnt > len && !(a.rounds & 64u)) { // (a.rounds bit 6, experiment: off)
        seeded = false;
        junction = 0xFFu;
        jov = 0;
        cnt = 0;
    }
    // the ramp of a read's first bases (depth i + 1 while nothing mismatches: the bases equal a path of the text)
    bool flag = plannable && seeded && cnt > a.plan_list + 1u; // more mismatches than the list holds: the plain walk
    const bool no_plan = plannable && !seeded && !has_invalid; // no seed at all: every base from the table, 16 at a time
    flag = flag || (plannable && has_invalid);
    if (!DIRECT) {
        const uint32_t first_mm = (seeded && cnt > 0) ? (uint32_t)sp[0] : len;
        const uint32_t lim = (seeded && !flag) ? min(min(k - 1u, first_mm), len) : 0u;
        const uint32_t most = min(k - 1u, 16u * kMapWords);
        for (uint32_t i = 0; i < most; i++) {
            if (__ballot(i < lim) == 0) break;
            if (i < lim) so[soff + i] = (uint8_t)(i + 1u);
        }
    }

    // (DIRECT) a window that is in the index and is left to the windows around it: asked for in stage 3, judged at the kernel's end
    bool w_around = false, w_unsettled = false;
    uint32_t w_e = 0, w_owner = 0, w_at = 0, w_len = 0, w_pre[6] = {0, 0, 0, 0, 0, 0};
    // ---- 3. the stretches behind the mismatches from the depth table (rule: dtab_kernels.hip), dealt out to the lanes
    uint32_t st_look = 0, st_written = 0, st_anch = 0, st_filt = 0; // (direct form: st_written = windows the filter settled)
    {
        const uint32_t order = a.ix.dtab_order;
        // DIRECT: cov = bases between two windows of the proof (header), n_e = windows per mismatch (<= 4: launch_map_reads);
        // a read without a seed: windows ending at order - 1, + cov, .. and at its last base, four per unit of work
        // (with anchors - copies whose table has a thin margin over log4(rows): C3, C4 - the windows of a mismatch stand two bases
        // closer, so that one of them may be present as long as its exact depth, read off the path-cover text, is at most
        // order + 1: header)
        const bool have_anch = DIRECT && a.ix.anchor != nullptr && thr_gt(a.map_thr, order);
        const uint32_t thr = a.map_thr, covb = DIRECT ? thr - order + 2u : 1u, cov = have_anch ? thr - order : covb;
        const uint32_t n_blockwin = (DIRECT && len > thr) ? (len - order + covb - 1u) / covb + 1u : 0u;
        const uint32_t my_n = !plannable || flag ? 0u : DIRECT ? (no_plan ? (n_blockwin + 3u) / 4u : (len > thr ? cnt : 0u))
                                                               : (no_plan ? (len + 15u) / 16u : cnt);
        uint32_t incl = my_n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if ((int)lane >= off) incl += t;
        }
        const uint32_t total = __shfl(incl, 63);
        sp[13] = 0;
        if (lane == 0) *pend_n = 0;
        *reinterpret_cast<uint16_t *>(sp + 14) = (uint16_t)incl;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        using code_t = typename std::conditional<NP == 16, uint32_t, uint64_t>::type;
        const code_t omask = (code_t)((1ull << (2u * order)) - 1ull);
        for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
            const uint32_t w = w0 + lane;
            const bool work = w < total;
            uint32_t lo_l = 0, hi_l = 63; // owner: the first lane whose inclusive prefix exceeds w
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const uint32_t mid = (lo_l + hi_l) >> 1;
                const uint32_t pm = *reinterpret_cast<const uint16_t *>(spw + mid * 16u + 14u);
                if (pm > w) hi_l = mid;
                else lo_l = mid + 1u;
            }
            const uint32_t owner = work ? lo_l : lane;
            const uint32_t o_incl = __shfl(incl, (int)owner), o_n = __shfl(my_n, (int)owner), o_soff = __shfl(soff, (int)owner),
                           o_len = __shfl(len, (int)owner), o_np = __shfl(no_plan ? 1u : 0u, (int)owner), o_junc = __shfl(junction, (int)owner), o_jov = __shfl(jov, (int)owner);
            const bool blockmode = o_np != 0;
            if (DIRECT) {
                if (work) {
                    const uint32_t t = w - (o_incl - o_n);
                    const uint32_t m = blockmode ? 0u : (uint32_t)spw[owner * 16u + t];
                    uint32_t bytes[4], ee[4];
                    bool use[4], lastw[4], look[4];
#pragma unroll
                    for (uint32_t i = 0; i < 4u; i++) {
                        // the base the window ends at, and whether the window counts: inside the read, `order` bases long, and not the
                        // same window again (the last one of a mismatch is clamped to m + order - 1, of a read to its last base)
                        if (blockmode) {
                            const uint32_t u = 4u * t + i;
                            ee[i] = min(order - 1u + u * covb, o_len - 1u);
                            use[i] = o_len >= order && (u == 0u || order - 1u + (u - 1u) * covb < o_len - 1u);
                            lastw[i] = true;
                        } else {
                            // (the junction of two diagonals: the windows that hold both its bases, m and m + 1)
                            // (and the bases in front of m that lie on the second diagonal too: the windows reach back over them)
                            const uint32_t jn = t == o_junc ? 1u + o_jov : 0u, j1 = t == o_junc ? 1u : 0u;
                            ee[i] = m + j1 + min(i * cov, order - 1u - jn);
                            use[i] = ee[i] < o_len && ee[i] + 1u >= order && (i == 0u || (i - 1u) * cov < order - 1u - jn);
                            lastw[i] = i * cov >= order - 1u - jn; // the window that starts at the break: nothing stands behind it
                        }
                        bytes[i] = 0;
                        look[i] = use[i];
                    }
                    // the filter in front of the table (small indexes: DevIndexView::dfilt, 2 MB the L2 keeps): a window is absent as
                    // soon as F of its bases are - F bases that hold the whole break (the mismatch; the junction's bases and what
                    // lies on both diagonals in front of it), or the window would only repeat what the text says.  Two of them
                    // where two differ: the rightmost and the leftmost such stretch inside the window.  Three windows in four
                    // never reach the table (a line fill each) at C2
                    if (a.ix.dfilt) {
                        const uint32_t F = a.ix.dfilt_bases, fmask = (1u << (2u * F)) - 1u;
                        uint32_t zlo, zhi; // the break inside the read (a read without a seed: any stretch will do)
                        if (blockmode) zlo = zhi = 0xFFFFFFFFu;
                        else if (t == o_junc) { zlo = m - o_jov; zhi = m + 1u; }
                        else zlo = zhi = m;
                        if (blockmode || zhi - zlo < F) {
                            uint32_t fw[4][2];
                            bool two[4];
#pragma unroll
                            for (uint32_t i = 0; i < 4u; i++) {
                                fw[i][0] = fw[i][1] = 0xFFFFFFFFu;
                                two[i] = false;
                                if (use[i]) {
                                    const uint32_t e = ee[i], left = e - (order - F); // (ends of the rightmost / leftmost stretch of F bases)
                                    const uint32_t e1 = blockmode ? e : min(e, zlo + F - 1u), e2 = blockmode ? left : max(zhi, left);
                                    const uint32_t k1 = (uint32_t)ending_at(o_soff + e1) & fmask, k2 = (uint32_t)ending_at(o_soff + e2) & fmask;
                                    fw[i][0] = a.ix.dfilt[k1 >> 5] >> (k1 & 31u);
                                    two[i] = kMapExp && e2 != e1 && (a.rounds & 2u); // (a.rounds bit 1, experiment: a second stretch - fewer table look-ups, 1.4 % slower)
                                    if (two[i]) fw[i][1] = a.ix.dfilt[k2 >> 5] >> (k2 & 31u);
                                    if (STATS) st_filt += two[i] ? 2u : 1u;
                                }
                            }
#pragma unroll
                            for (uint32_t i = 0; i < 4u; i++)
                                if (use[i] && !((fw[i][0] & 1u) && (!two[i] || (fw[i][1] & 1u)))) {
                                    look[i] = false; // absent: nothing to look up (bytes[i] stays 0: an absent window)
                                    if (STATS) st_written++;
                                }
                        }
                    }
#pragma unroll
                    for (uint32_t i = 0; i < 4u; i++) {
                        if (look[i]) {
                            const code_t key = (code_t)ending_at(o_soff + ee[i]) & omask;
                            bytes[i] = !a.ix.dtab_grouped ? a.ix.dtab[key]
                                                          : a.ix.dtab[NP == 16 ? dtab_grouped_addr32((uint32_t)key, ee[i] % 3u, order) : dtab_grouped_addr((uint64_t)key, ee[i] % 3u, order)];
                            if (STATS) st_look++;
                        }
                    }
                    bool fail = false;
                    // a window that IS in the index and that neither the anchors nor its place settle: the windows around it may (behind
                    // the loop, with the anchors' windows: `pend`) - unless it belongs to a junction
                    auto ask_around = [&](uint32_t e_) {
                        if (order < k && (blockmode || t != o_junc) && !(a.rounds & 32u)) { // (a.rounds bit 5, experiment: off)
                            const uint32_t slot = atomicAdd(pend_n, 1u);
                            if (slot < kMapPend) pend[slot] = make_uint2(o_soff + e_, owner | (e_ << 8) | (0x7FFFu << 16));
                            else fail = true;
                        } else if (order < k && a.ix.seed_tab != nullptr && !(a.rounds & (32u | 128u))) {
                            // a junction's window: nothing to ask the windows around it (their entries speak of one diagonal), but the
                            // strings that hold it can be searched like any other's (kernel's end)
                            const uint32_t slot = atomicAdd(pend_n, 1u);
                            if (slot < kMapPend) pend[slot] = make_uint2(o_soff + e_, owner | (e_ << 8) | (0x7FFEu << 16));
                            else fail = true;
                        } else
                            fail = true;
                    };
#pragma unroll
                    for (uint32_t i = 0; i < 4u; i++) {
                        if (use[i] && (bytes[i] & 0x80u)) { // the window is a suffix of a row
                            if (have_anch && lastw[i] && t != o_junc && order < k && ee[i] >= order && !(kMapExp && (a.rounds & 16u)) &&
                                !((bytes[i] >> base_at(o_soff + ee[i] - order)) & 1u)) {
                                // the window that STARTS at the mismatch m (one substitution in 170 has it: its 14 bases behind m
                                // occur elsewhere behind the read's base).  Not deeper: no string through m that ends with it has
                                // more than `order` bases.  And one that ends further right holds the window's bases + the next one:
                                // the entry of the window one base on says whether the read's base at m extends THAT to the left
                                // (behind the loop too).  Without this the read's only way was the plain walk: 0.6 % per mismatch
                                const uint32_t e1 = ee[i] + 1u;
                                if (e1 < o_len) {
                                    const uint32_t slot = atomicAdd(pend_n, 1u);
                                    if (slot < kMapPend) pend[slot] = make_uint2(o_soff + e1, owner | (e1 << 8) | 0x80000000u);
                                    else fail = true;
                                }
                            } else if (!have_anch || lastw[i]) ask_around(ee[i]);
                            else {
                                // its exact depth off the path-cover text decides - behind the loop, all such windows of the wave at once
                                // (one in 200 windows: looked up here, a hash probe and two loads of text for ONE lane held every
                                // round of the loop up: the proof was 95 us of the kernel's 240 with it, 75 without anchors)
                                const uint32_t e_i = ee[i];
                                const uint32_t nxt_e = (i + 1u < 4u && use[i + 1u < 4u ? i + 1u : i]) ? ee[i + 1u < 4u ? i + 1u : i] : o_len;
                                // (the entry's low bits say which bases in front of the window extend it: three times in four the read's
                                // own does not - the depth is the window's length, nothing to look up)
                                const bool deeper = e_i >= order && order < k && ((bytes[i] >> base_at(o_soff + e_i - order)) & 1u);
                                if (!deeper) {
                                    if (order + (nxt_e - e_i) - 1u > thr) ask_around(e_i);
                                } else {
                                    if (STATS) st_anch++;
                                    const uint32_t slot = atomicAdd(pend_n, 1u);
                                    if (slot < kMapPend) pend[slot] = make_uint2(o_soff + e_i, owner | (e_i << 8) | ((nxt_e - e_i) << 16));
                                    else fail = true; // (no room: the plain walk decides)
                                }
                            }
                        }
                    }
                    if (fail) spw[owner * 16u + 13u] = 1; // the proof fails: the read takes the plain walk
                }
            } else if (work) {
                const uint32_t t = w - (o_incl - o_n);
                const uint8_t *osp = spw + owner * 16u;
                const uint32_t m = blockmode ? 16u * t : (uint32_t)osp[t];
                const uint32_t nxt = (blockmode || t + 1u >= o_n) ? o_len : (uint32_t)osp[t + 1u];
                const uint32_t P = min(min(blockmode ? 16u : order + 1u, (uint32_t)NP), min(nxt, o_len) - m); // bases looked up: m .. m + P - 1
                uint32_t tv[NP];
                uint32_t evalmask = 0, unkmask = 0, satmask = 0;
                uint32_t outv[5] = {0, 0, 0, 0, 0};
                // window ending at base i of the owner's read: its `order` newest bases are the key; v = bases it has inside the read
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)NP; j++) {
                    tv[j] = 0;
                    if (j < P) {
                        const uint32_t i = m + j;
                        const code_t key = (code_t)ending_at(o_soff + i) & omask;
                        // (bases in front of the read inside the window belong to its neighbour, or are zeros: the rule cuts the value
                        // to the bases that count, min(byte, v), and suffixes of a present string are present)
                        tv[j] = !a.ix.dtab_grouped ? a.ix.dtab[key]
                                                   : a.ix.dtab[NP == 16 ? dtab_grouped_addr32((uint32_t)key, i % 3u, order) : dtab_grouped_addr((uint64_t)key, i % 3u, order)];
                        if (STATS) st_look++;
                    }
                }
                bool done = false;
                uint32_t n_eval = 0; // bases the stretch decided (a block: all of them)
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)NP; j++) {
                    if (j < P && !done) {
                        const uint32_t i = m + j;
                        const uint32_t vv = min(i + 1u, 31u); // bases of the read up to and including i
                        const uint32_t byte = tv[j];
                        uint32_t Lv = k + 1u;
                        if (!(byte & 0x80u)) Lv = min(byte, vv);
                        else if (vv <= order || order >= k) Lv = min(vv, order);
                        else {
                            // all `order` bases are a suffix of a row: deeper when the base in front of them extends it
                            const uint32_t eb = base_at(o_soff + i - order); // (vv > order: that base is the read's own)
                            if ((byte >> eb) & 1u) satmask |= 1u << j;
                            else Lv = order;
                        }
                        outv[j >> 2] |= min(Lv, k) << (8u * (j & 3u));
                        evalmask |= 1u << j;
                        n_eval = j + 1u;
                        done = !blockmode && Lv <= j;
                    }
                }
                if (STATS) st_anch += (uint32_t)__popc(satmask);
                unkmask |= satmask;
                if (unkmask) spw[owner * 16u + 13u] = 1; // the owner's read goes to the plain walk
                // (a stretch with an unknown base writes nothing; a block writes the bases it knows)
                const uint32_t wmask = blockmode ? evalmask & ~unkmask : (unkmask ? 0u : evalmask);
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)NP; j++)
                    if ((wmask >> j) & 1u) {
                        so[o_soff + m + j] = (uint8_t)(outv[j >> 2] >> (8u * (j & 3u)));
                        if (STATS) st_written++;
                    }
                // behind the stretch: the ramp up to the next mismatch (depth = bases since this one), k from k bases on (in place)
                if (!blockmode && !unkmask) {
                    const uint32_t rend = min(nxt - m, k);
                    for (uint32_t j = n_eval; j < rend; j++) so[o_soff + m + j] = (uint8_t)j;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (DIRECT) {
            // a present window's exact depth (dtab_anchor_depth: the window is the suffix of ONE row, whose characters stand in the text in
            // front of its position).  Strings through the break that end between this window and the next start no further left than
            // this depth says: with the next window (or the read's end) g bases on, none is longer than depth + g - 1
            const uint32_t n_pend = min(*pend_n, kMapPend);
            bool around = false; // this lane's window is left to the windows around it
            if (lane < n_pend) {
                const uint2 pe = pend[lane];
                const uint32_t p_owner = pe.y & 0xFFu, e_i = (pe.y >> 8) & 0xFFu, gap = (pe.y >> 16) & 0x7FFFu, at_e = pe.x;
                w_owner = p_owner;
                w_e = e_i;
                w_at = at_e;
                if (pe.y >> 31) { // the window one base behind the one that starts at a mismatch: extended to the left by the read's base there?
                    const code_t key = (code_t)ending_at(at_e) & omask;
                    const uint32_t byte = !a.ix.dtab_grouped ? a.ix.dtab[key]
                                                             : a.ix.dtab[NP == 16 ? dtab_grouped_addr32((uint32_t)key, e_i % 3u, order) : dtab_grouped_addr((uint64_t)key, e_i % 3u, order)];
                    if ((byte & 0x80u) && ((byte >> base_at(at_e - order)) & 1u)) { // (the window in question: the one in front of it)
                        around = true;
                        w_e = e_i - 1u;
                        w_at = at_e - 1u;
                    }
                } else if (gap == 0x7FFFu) around = true;
                else if (gap == 0x7FFEu) w_unsettled = true; // (searched at the kernel's end)
                else {
                    const uint32_t V = dtab_anchor_depth(a.ix, e_i + 1u, [&](uint32_t tt) -> uint32_t {
                        return (0x54474341u >> (8u * base_at(at_e - tt))) & 0xFFu;
                    });
                    if (V == kDtabUnknown || V + gap - 1u > thr) around = true;
                }
                if (around && (a.rounds & 32u)) { // (experiment: as before the rule below)
                    spw[p_owner * 16u + 13u] = 1;
                    around = false;
                }
            }
            // ... and what neither settles, by the windows AROUND it (rule: at the kernel's end, where it is evaluated).  Here only the
            // entries of e - 2 .. e + 3 are asked for - nearly always all it takes -, so that no wave waits for them: the characters are
            // made in the meantime, and a read that turns out flagged after all has them rewritten by the second pass like any other
            w_len = __shfl(len, (int)w_owner);
            w_around = around;
            if (__ballot(around)) {
                if (around) {
                    const uint32_t soff_ = w_at - w_e;
                    const uint32_t e_lo = max(w_e, thr), e_hi = min(w_e + thr + 1u - order, w_len - 1u);
#pragma unroll
                    for (uint32_t q = 0; q < 6u; q++) {
                        const uint32_t w_ = w_e + q - 2u; // (e >= order - 1 >= 2)
                        w_pre[q] = 0;
                        if (e_lo <= e_hi && w_ + 1u >= order && w_ < w_len) {
                            const code_t key = (code_t)ending_at(soff_ + w_) & omask;
                            if (STATS) st_look++;
                            w_pre[q] = !a.ix.dtab_grouped ? a.ix.dtab[key]
                                                          : a.ix.dtab[NP == 16 ? dtab_grouped_addr32((uint32_t)key, w_ % 3u, order) : dtab_grouped_addr((uint64_t)key, w_ % 3u, order)];
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        flag = flag || sp[13] != 0;
    }
    // ---- the MS values themselves, when the caller wants them too (whole lines)
    if (!DIRECT && a.map_want_ms) {
        for (uint32_t c = lane * 16u; c < span; c += 1024u) {
            const uint4 v = *reinterpret_cast<const uint4 *>(so + c);
            const uint32_t g0 = base16 + c;
            if (g0 >= lo && g0 + 16u <= wave_hi) __builtin_memcpy(a.d_out + g0, &v, 16);
            else st_range16(a.d_out + g0, v, lo > g0 ? min(lo - g0, 16u) : 0u, min(wave_hi - g0, 16u));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- 4. derandomize_ms_vec + translate_ms_vec, right to left, in place (derand_kernels.hip has the derivation of the
    // closed form of the 'R','R' look-ahead): x = derandomised value, window (x_prev, x_cur, x_next) = x[p-1], x[p], x[p+1]
    if (DIRECT) {
        // the characters that are not 'M', segment by segment from the right (header): e = the break to the right (the read's
        // end first), dn = its segment's d, cand = the base at e waits for its left neighbour to decide between 'X' and '-'
        const int K = (int)k, T = (int)a.map_thr;
        const uint32_t cX = a.map_fmt ? kCodeDash : kCodeX; // (relative_to_ref: 'X' becomes '-' as well)
        // (a word of codes may hold the end of one read and the head of the next - two lanes: LDS atomics, which cost what a store does)
        auto put = [&](int i, uint32_t code) {
            const uint32_t p_ = ooff + (uint32_t)i;
            atomicOr(cw + (p_ >> 4), code << (2u * (p_ & 15u)));
        };
        auto put_dashes = [&](int i0, int i1) { // bases [i0, i1) of the read
            const uint32_t p0_ = ooff + (uint32_t)i0, p1_ = ooff + (uint32_t)i1;
            for (uint32_t w_ = p0_ >> 4; 16u * w_ < p1_; w_++) {
                const uint32_t lo_ = max(p0_, 16u * w_) - 16u * w_, hi_ = min(p1_, 16u * w_ + 16u) - 16u * w_;
                const uint32_t m_ = (hi_ >= 16u ? 0xFFFFFFFFu : (1u << (2u * hi_)) - 1u) & ~((1u << (2u * lo_)) - 1u);
                atomicOr(cw + w_, (0x55555555u * kCodeDash) & m_);
            }
        };
        const bool live = plannable && !flag && len >= 3u;
        if (__ballot(live && no_plan)) { // no seed, and the table vouches for every base (<= order <= t): x <= 0 throughout
            if (live && no_plan) put_dashes(0, (int)len);
        }
        int e = (int)len, dn = 0;
        bool at_end = true, cand = false;
        for (int q = (int)cnt - 1;; q--) {
            const bool act = live && seeded && q >= -1;
            if (__ballot(act) == 0) break;
            if (act) {
                // break q: a mismatch at m (its own x is the segment's j = 0), or - the read's head (m = -1) and the junction of two
                // diagonals (m = the last base of the first) - a break BEHIND base m: the segment starts at j = 1
                const bool isj = q >= 0 && (uint32_t)q == junction;
                const int m = q >= 0 ? (int)sp[q] - (isj ? (int)jov : 0) : -1;
                const int j0 = (q < 0 ? 1 : 0) + (isj ? 1 + (int)jov : 0);
                const int L = e - m, jl = L - 1;
                if (jl >= j0) { // (else an empty segment - a head in front of a mismatch at base 0, a junction right in front of the next break)
                    int d;
                    if (L > K) d = 0;
                    else if (at_end) d = jl > T ? 0 : -jl;
                    else {
                        const int din = dn - L;
                        d = (din <= -2 && jl > T) ? 0 : din;
                    }
                    const int xt = L > K ? K : jl + d; // x of the segment's last base (the left neighbour of e)
                    if (cand) put(e, ((e <= 1 ? K : xt) > 0) ? cX : kCodeDash);
                    cand = false;
                    // the junction to the right starts with x = 1 and this segment ends above the threshold: translate_ms_val's
                    // ('R', 'R') (translate.rs:195-203; the second 'R' under translate_ms_vec's position rules, :282-288);
                    // relative_to_ref keeps the read's bases for both
                    if (!at_end && dn > 0 && dn < T && xt > T && !a.map_fmt) {
                        put(e - 1, kCodeR);
                        if (e >= 2 && e < (int)len - 1) put(e, kCodeR);
                    }
                    if (-d == j0) { // the segment's first base has x = 0: 'X' when next == 1 and prev > 0 (translate.rs:204-210)
                        const int next = -d + 1 <= jl ? 1 : (at_end ? 0 : dn);
                        if (next == 1) cand = true;
                        else put(m - d, kCodeDash);
                    } else if (-d > j0) {
                        const int j1 = min(-d, jl); // bases with x <= 0
                        put_dashes(m + j0, m + j1 + 1);
                        // the base with x = 0 whose successor has x = 1: an 'X' when it is one of the read's two first bases, whose
                        // prev is k (elsewhere its prev is the x = -1 in front of it)
                        const int pz = m - d;
                        const int nz = -d + 1 <= jl ? 1 : (-d == jl ? (at_end ? 0 : dn) : 0);
                        if (nz == 1 && pz >= 0 && pz <= 1) put(pz, cX);
                    }
                    dn = d + j0; // x of the base the next segment to the left ends in front of
                    e = m + j0;
                    at_end = false;
                }
            }
        }
        if (cand) put(e, cX); // (the read's first base: its prev is k)
        (void)K;
    } else if (a.chars_out != nullptr && plannable && !flag && len >= 3u) { // (chars_out == nullptr: kbo_ms_batch_dev - the MS values were all that was asked for)
        literal_chars(so + soff, len, (int)k, (int)a.map_thr);
    }
    // kbo::find: the runs of the read (format::run_lengths_gapped with max_gap_len = 0 closes a run at every '-': rle_kernels.hip),
    // counted off the characters while they are in LDS - four at a time, a start wherever a character that is not '-' follows one
    // that is (or the read's head)
    if (DIRECT && a.run_counts && have_item) {
        uint32_t n_runs = 0;
        if (plannable && !flag && len >= 3u) { // (fewer than 3 bases: no alignment - the reference asserts, derandomize.rs:274-276 - and no run)
            const uint32_t b0 = ooff, b1 = ooff + len;
            uint32_t carry = 0; // bit 0: the character in front of the word was not a '-'
            for (uint32_t w_ = b0 >> 4; 16u * w_ < b1; w_++) { // sixteen characters a word
                const uint32_t v = cw[w_];
                uint32_t nd = ~(v & ~(v >> 1)) & 0x55555555u; // bit 2 p of every character that is not '-' (code 01)
                // characters outside [b0, b1) count as '-'
                const uint32_t lo_ = 16u * w_ < b0 ? b0 - 16u * w_ : 0u, hi_ = 16u * w_ + 16u > b1 ? b1 - 16u * w_ : 16u;
                nd &= (hi_ >= 16u ? 0xFFFFFFFFu : (1u << (2u * hi_)) - 1u) & ~((1u << (2u * lo_)) - 1u);
                n_runs += (uint32_t)__popc(nd & ~((nd << 2) | carry));
                carry = nd >> 30;
            }
        }
        a.run_counts[idx] = n_runs;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- 5. the characters leave in whole lines; format::relative_to_ref (format.rs:270-286) on the way: 'M' and 'R' keep the
    // read's base, everything else becomes '-' (flagged reads' bytes are rewritten by launch_derand_flagged)
    if (IO == 2) { // as 2-bit words, where the read's own words stand in the batch (the packed alphabet is M - X R: codes 2 and 3 change places)
        for (uint32_t c = lane; c < nwords; c += 64u) {
            const uint32_t w_ = cw[c];
            a.packed_out[w_lo + c] = w_ ^ ((w_ >> 1) & 0x55555555u);
        }
    } else if (DIRECT) {
        // sixteen codes a lane -> sixteen characters: the codes of four characters spread to a byte each select out of "M-RX" (one byte
        // permute); with relative_to_ref the read's own bases, from the digits, where the code says 'M'
        for (uint32_t c = lane; 16u * c < span; c += 64u) {
            const uint32_t w_ = cw[c], dg = a.map_fmt ? lin[c] : 0u; // (relative_to_ref: bytes in, IO == 0 - the digits of block c are word c)
            auto four = [&](uint32_t c8, uint32_t d8) -> uint32_t {
                const uint32_t cs = (c8 | (c8 << 6) | (c8 << 12) | (c8 << 18)) & 0x03030303u;
                if (!a.map_fmt) return __builtin_amdgcn_perm(0u, 0x58522D4Du, cs);
                const uint32_t sel = ((d8 >> 6) & 3u) | (((d8 >> 4) & 3u) << 8) | (((d8 >> 2) & 3u) << 16) | ((d8 & 3u) << 24);
                const uint32_t letters = __builtin_amdgcn_perm(0u, 0x54474341u, sel);
                const uint32_t keep = (((cs | (cs >> 1)) & 0x01010101u) ^ 0x01010101u) * 0xFFu; // bytes whose code is 'M'
                return (letters & keep) | (0x2D2D2D2Du & ~keep);
            };
            uint4 v;
            v.x = four(w_ & 0xFFu, dg >> 24);
            v.y = four((w_ >> 8) & 0xFFu, (dg >> 16) & 0xFFu);
            v.z = four((w_ >> 16) & 0xFFu, (dg >> 8) & 0xFFu);
            v.w = four(w_ >> 24, dg & 0xFFu);
            const uint32_t g0 = base16 + 16u * c;
            if (g0 >= lo && g0 + 16u <= wave_hi) __builtin_memcpy(a.chars_out + g0, &v, 16);
            else st_range16(a.chars_out + g0, v, lo > g0 ? min(lo - g0, 16u) : 0u, min(wave_hi - g0, 16u));
        }
    } else if (a.chars_out != nullptr)
    for (uint32_t c = lane * 16u; c < span; c += 1024u) {
        uint4 v = *reinterpret_cast<const uint4 *>(so + c);
        if (a.map_fmt) {
            const uint32_t dg = lin[c >> 4];
            auto fmt4 = [&](uint32_t ch, uint32_t d8) -> uint32_t { // four characters and the four bases behind them
                const uint32_t sel = ((d8 >> 6) & 3u) | (((d8 >> 4) & 3u) << 8) | (((d8 >> 2) & 3u) << 16) | ((d8 & 3u) << 24);
                const uint32_t letters = __builtin_amdgcn_perm(0u, 0x54474341u, sel);
                const uint32_t keep = ((ch >> 6) & (ch | ~(ch >> 3)) & 0x01010101u) * 0xFFu; // bytes that are 'M' or 'R' (of M - X R)
                return (letters & keep) | (0x2D2D2D2Du & ~keep);
            };
            v.x = fmt4(v.x, dg >> 24);
            v.y = fmt4(v.y, (dg >> 16) & 0xFFu);
            v.z = fmt4(v.z, (dg >> 8) & 0xFFu);
            v.w = fmt4(v.w, dg & 0xFFu);
        }
        const uint32_t g0 = base16 + c;
        if (g0 >= lo && g0 + 16u <= wave_hi) __builtin_memcpy(a.chars_out + g0, &v, 16);
        else st_range16(a.chars_out + g0, v, lo > g0 ? min(lo - g0, 16u) : 0u, min(wave_hi - g0, 16u));
    }
    // ---- the windows that were left to the windows around them (stage 3).  The strings of thr + 1 bases that hold the window ending at
    // e end at e .. e + c, c = thr + 1 - order (all of them through the break the window was looked up for), inside the read from thr
    // on.  A window ending at e + i that is absent - or whose string with the read's base in front of it is: its entry's
    // left-extension bit - rules out those ending at e + i or later; one ending at e - j those ending at e - j + c or before (a base
    // fewer when only its extended string is absent; j = 0: the window's own bit).  Without it the read took the plain walk: every
    // seventh read of another genome, every twentieth with 5 % substitutions.  Brute-force check of the rule:
    // tests/test_proof_rule_model.py
    if (DIRECT && __ballot(w_around || w_unsettled)) {
        const uint32_t order = a.ix.dtab_order, thr = a.map_thr;
        if (w_around) {
            const uint32_t soff_ = w_at - w_e, e = w_e;
            auto ext_of = [&](uint32_t byte_, uint32_t w_) -> bool { return ((byte_ >> base_at(soff_ + w_ - order)) & 1u) != 0; }; // (w_ >= order)
            auto entry = [&](uint32_t w_) -> uint32_t {
                const uint32_t q = w_ + 2u - e; // (unsigned: far to the left wraps to a large value)
                // (further out: "present, and extended by every base" - no more trips to the table, each of which the whole wave would wait
                // for.  A window that the six entries do not settle mostly stands in a true repeat, where nine more would not either)
                return q == 0u ? w_pre[0] : q == 1u ? w_pre[1] : q == 2u ? w_pre[2] : q == 3u ? w_pre[3] : q == 4u ? w_pre[4] : q == 5u ? w_pre[5] : 0xFFu;
            };
            const uint32_t c = thr + 1u - order;
            const uint32_t e_lo = max(e, thr), e_hi = min(e + c, w_len - 1u);
            bool ok = e_lo > e_hi;
            if (!ok) {
                uint32_t R = 0xFFFFFFFFu; // strings ending at R or later are ruled out
                for (uint32_t i2 = 1; i2 <= c + 1u; i2++) {
                    const uint32_t w_ = e + i2;
                    if (w_ > e_hi) {
                        R = e_hi + 1u;
                        break;
                    }
                    const uint32_t b_ = entry(w_);
                    if (!(b_ & 0x80u) || !ext_of(b_, w_)) {
                        R = w_;
                        break;
                    }
                }
                if (R != 0xFFFFFFFFu) {
                    if (R - 1u < e_lo) ok = true;
                    else if (e >= order && !ext_of(entry(e), e) && e + c >= R) ok = true;
                    else
                        for (uint32_t j = 1; j <= c; j++) {
                            if (e < j + order - 1u) break; // (no such window inside the read)
                            const uint32_t w_ = e - j;
                            if (w_ + c + 1u < R) break;
                            const uint32_t b_ = entry(w_);
                            if (!(b_ & 0x80u) || (w_ >= order && !ext_of(b_, w_) && w_ + c >= R)) {
                                ok = true;
                                break;
                            }
                        }
                }
            }
            w_unsettled = !ok;
        }
        // (w_len of a junction's window: set with the others')
        // ---- what the windows around it do not settle is SEARCHED (round 6): the strings of thr + 1 bases that hold the window end at
        // e_lo .. e_hi - nine of them at most - and each is in the index or not: its first seed_d bases by the interval table (one load),
        // the others by extend-right steps over the rank blocks, thr + 1 - seed_d (nine at C2) dependent rounds of two 16-byte loads, one
        // candidate a lane, nearly all of them dead after the first round or two (a string of 14 random bases is in a 5 Mbp index one time
        // in fifty).  None present: no string longer than thr runs through the break - the proof, exactly.  Before, such a read went to the
        // second pass: a chain of ~ 80 dependent walk steps for two reads in a thousand, 0.12 ms whatever their number - and it still does
        // when a candidate IS present (a true repeat of more than thr bases through a mismatch).
        const uint32_t D_ = a.ix.seed_d;
        const bool can_search = a.ix.seed_tab != nullptr && D_ >= 4u && D_ <= 16u && thr + 1u >= D_ && !(a.rounds & 128u); // (a.rounds bit 7, experiment: off)
        uint64_t um = __ballot(w_unsettled);
        if (!can_search) {
            if (w_unsettled) spw[w_owner * 16u + 13u] = 1;
            um = 0;
        }
        // (a read without a seed whose windows are in the index - a read of the genome whose head gave no seed: every one of its windows,
        // a pending list full of them - is the second pass's as before, and so is whatever a wave has beyond two searches: a search is ten
        // dependent rounds for the whole wave, and forty-four such reads in a million were a tenth of the kernel's time as its last waves)
        uint32_t n_search = 0;
        while (um) {
            const int src = __builtin_ctzll(um);
            um &= um - 1ull;
            const uint32_t e_ = __shfl(w_e, src), at_ = __shfl(w_at, src), wl_ = __shfl(w_len, src), own_ = __shfl(w_owner, src);
            const bool own_np = __shfl(no_plan ? 1u : 0u, (int)own_) != 0u;
            if (own_np || n_search >= 2u || spw[own_ * 16u + 13u] != 0) { // (wave-uniform)
                if (lane == 0) spw[own_ * 16u + 13u] = 1;
                continue;
            }
            n_search++;
            const uint32_t soff_ = at_ - e_, c_ = thr + 1u - order;
            const uint32_t e_lo = max(e_, thr), e_hi = min(e_ + c_, wl_ - 1u);
            bool alive = e_lo <= e_hi && lane <= e_hi - e_lo;
            const uint32_t first = soff_ + e_lo + lane - thr; // the candidate's first base in the stretch
            uint32_t l_ = 0, r_ = 0;
            if (alive) {
                const uint32_t key = (uint32_t)ending_at(first + D_ - 1u) & (D_ >= 16u ? 0xFFFFFFFFu : (1u << (2u * D_)) - 1u);
                const uint2 iv = a.ix.seed_tab[key];
                if (STATS) st_look++;
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
            if (__ballot(alive) != 0 && lane == 0) spw[own_ * 16u + 13u] = 1; // a string of thr + 1 bases through the break IS in the index
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        flag = flag || sp[13] != 0;
    }
    if (STATS) plan_stats_add(a.pstats, kPlanStatSeedLookups, st_lookups, kPlanStatSeedExtensions, st_filt /* (this kernel: filter look-ups) */, kPlanStatMismatches, seeded ? cnt : 0u, 0, 0);
    if (STATS) plan_stats_add(a.pstats, kPlanStatTabLookups, st_look, kPlanStatTabWritten, st_written, kPlanStatTabFlagged, flag ? 1u : 0u,
                   kPlanStatTabAnchored, st_anch);
    const uint64_t fm = __ballot(flag), nm = __ballot(no_plan);
    if (lane == 0) {
        if (fm) atomicAdd(a.qctl + 4, (uint32_t)__popcll(fm));
        if (nm) atomicAdd(a.qctl + 5, (uint32_t)__popcll(nm));
    }
    if (have_item) a.redo[idx] = flag ? 1 : 0;
    // ... and listed for finish_reads_kernel (qctl[6] counts them; the list - where the guided walk's units would be - holds every read)
    if (flag && have_item) reinterpret_cast<uint32_t *>(a.units)[atomicAdd(a.qctl + 6, 1u)] = idx;
}

// ---- the reads map_reads_kernel could not finish, in ONE kernel behind it (round 6; before: redo_collect_kernel's list of pieces, the
// plain walk over it, derand_flagged_kernel - three launches whose empty run alone is 0.09 ms behind a batch of a million reads, and a chain
// of 16 + k - 1 walk steps and a lane's pass over a whole read: 0.12 ms for the two reads in ten thousand that need it at C2, two thirds
// of the kernel's own time for a caller with one batch at a time).  R reads a wave, 64 / R lanes each: a lane walks len / (64 / R) bases of
// its read behind k - 1 bases that only bring the state up (walk_kernels.hip's walk: extend by two rank blocks, or climb one level of the
// LCS interval tree by two contraction entries and try again), the values go to LDS, the read's first lane runs derandomize_ms_vec +
// translate_ms_vec over them (literal_chars), and the lanes write what was asked for: the matching statistics, the characters
// (format::relative_to_ref on the way), the read's number of runs for kbo::find.  R by the number of such reads - one while they are fewer
// than the launch has waves (a chain of k + 1 + len / 64 steps), four or sixteen beyond (a batch with 5 % substitutions leaves 3 % of its
// reads) -, so that a batch's second pass is as long as its longest chain, and that is short.
constexpr uint32_t kFinishStride = 176, kFinishMaxR = 16, kFinishWaves = 8192, kFinishLds = 2u * kFinishMaxR * kFinishStride;
__global__ __launch_bounds__(256) void finish_reads_kernel(WalkArgs a, uint32_t r1, uint32_t r4)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fin_lds_all[];
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *lds = fin_lds_all + (threadIdx.x >> 6) * kFinishLds;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
    const uint32_t n_f = min(a.qctl[6], a.n_items);
    if (wave == 0 && lane == 0 && a.qctl[4] > a.unit_bail) { // most of the batch: the copy's plan is held off for the batches to come (as redo_collect_kernel)
        a.qctl[2] = 1;
        if (a.host_bailed) *a.host_bailed = 1u;
    }
    const uint32_t R = n_f <= r1 ? 1u : (n_f <= r4 ? 4u : kFinishMaxR), L = 64u / R;
    const uint32_t sub = lane / L, s = lane % L;
    const uint32_t *list = reinterpret_cast<const uint32_t *>(a.units);
    const uint32_t n = a.ix.n, k = a.ix.k, nblk = a.ix.n_blocks, null_blk = 4u * nblk;
    const uint8_t *arena = reinterpret_cast<const uint8_t *>(a.ix.arena);
    const uint8_t *ent = a.ix.big ? a.ix.ent : arena + ((uint64_t)a.ix.lcs_off << 4);
    uint8_t *qL = lds + sub * kFinishStride, *mL = lds + (kFinishMaxR + sub) * kFinishStride;
    for (uint32_t g = wave; (uint64_t)g * R < n_f; g += n_waves) {
        const uint32_t slot = g * R + sub;
        uint32_t ridx = 0, len = 0;
        uint64_t o0 = 0;
        if (slot < n_f) {
            ridx = list[slot];
            o0 = a.seq_off[ridx];
            len = (uint32_t)(a.seq_off[ridx + 1u] - o0);
            if (len + 16u > kFinishStride) len = 0; // (cannot happen: the kernel in front of this one takes reads of up to 160 bases)
        }
        for (uint32_t o = 16u * s; o < len; o += 16u * L) { // (reads <= 15 bytes past the read: the batch is padded)
            uint4 v;
            __builtin_memcpy(&v, a.q + o0 + o, 16);
            *reinterpret_cast<uint4 *>(qL + o) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t per = (len + L - 1u) / L;
        const uint32_t own0 = min(s * per, len), own1 = min(own0 + per, len);
        uint32_t i = own0 < own1 ? (own0 > k - 1u ? own0 - (k - 1u) : 0u) : own1;
        uint32_t l = 0, r = n, d = 0;
        while (__ballot(i < own1)) {
            if (i < own1) {
                const uint32_t c = decode_base(qL[i]);
                const uint32_t cb = c < 4u ? c * nblk : null_blk, bl = c < 4u ? div96(l) : 0u, br = c < 4u ? div96(r) : 0u;
                const uint4 xA = ld16(arena, (cb + bl) << 4), xB = ld16(arena, (cb + br) << 4);
                const uint32_t l2 = rank_eval(xA, l - div96(l) * 96u), r2 = rank_eval(xB, r - div96(r) * 96u);
                if (l2 < r2 || d == 0u) { // extended - or nothing left to give up: the root stays, the value is 0
                    if (l2 < r2) {
                        l = l2;
                        r = r2;
                        d = min(d + 1u, k);
                    }
                    if (i >= own0) mL[i] = (uint8_t)d;
                    i++;
                } else { // contract_left down to the level where the interval changes: max(lcs[l], lcs[r]); the side(s) that hold it move
                    uint4 eA, eB;
                    __builtin_memcpy(&eA, ent + (uint64_t)l * 12u, 16);
                    __builtin_memcpy(&eB, ent + (uint64_t)r * 12u, 16);
                    const uint32_t lv = max(eA.x, eB.x);
                    l = lv == 0u ? 0u : (eA.x == lv ? eA.y : l);
                    r = lv == 0u ? n : (eB.x == lv ? eB.z : r);
                    d = lv;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (a.d_out && (a.map_want_ms || !a.chars_out))
            for (uint32_t p = s; p < len; p += L) a.d_out[o0 + p] = mL[p];
        if (a.chars_out) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (s == 0u && slot < n_f) {
                uint32_t n_runs = 0;
                if (len >= 3u) { // (fewer: no alignment - the reference asserts, derandomize.rs:274-276 -, no character, no run)
                    literal_chars(mL, len, (int)k, (int)a.map_thr);
                    uint32_t gap = 1; // kbo::find: a character that is no '-' behind one that is, or at the read's head
                    for (uint32_t p = 0; p < len; p++) {
                        const uint32_t gp = mL[p] == (uint8_t)'-' ? 1u : 0u;
                        n_runs += gap & (gp ^ 1u);
                        gap = gp;
                    }
                }
                if (a.run_counts) a.run_counts[ridx] = n_runs;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (len >= 3u)
                for (uint32_t p = s; p < len; p += L) {
                    uint32_t ch = mL[p];
                    if (a.map_fmt) ch = (ch == 'M' || ch == 'R') ? (uint32_t)qL[p] : (uint32_t)'-'; // format::relative_to_ref (format.rs:270-286)
                    a.chars_out[o0 + p] = (uint8_t)ch;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

} // namespace

hipError_t launch_pack_text(const uint8_t *d_text_padded, uint64_t n_bytes, uint2 *d_out, uint64_t n_units, hipStream_t stream)
{
    hipLaunchKernelGGL(pack_text_kernel, dim3((unsigned)((n_units + 255) / 256)), dim3(256), 0, stream, d_text_padded, n_bytes, d_out, n_units);
    return hipGetLastError();
}

hipError_t launch_seed_pos(const uint2 *d_seed_tab, const uint32_t *d_pc_pos, uint32_t *d_out, uint32_t seed_d, hipStream_t stream)
{
    const uint64_t n = 1ull << (2u * seed_d);
    hipLaunchKernelGGL(seed_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_seed_tab, d_pc_pos, d_out, n);
    return hipGetLastError();
}

bool map_reads_applies(const WalkArgs &a)
{
    return a.ix.dtab && a.ix.pc_tm && a.ix.seed_pos && a.ix.seed_d >= 4u && a.ix.seed_d <= 14u && a.ix.dtab_order >= 4u && a.ix.dtab_order <= 17u &&
           a.ix.dtab_order <= a.ix.k && !a.call_sites && !a.lo_out && a.max_item_len != 0 && a.max_item_len <= 16u * kMapWords && a.redo && a.qctl &&
           a.units;
}

// the characters straight from the mismatch positions where no table value can anchor (order <= t < k) and the MS values
// are not asked for; else the MS bytes in LDS and the literal pass over them
bool map_reads_direct(const WalkArgs &a)
{
    static const int env_direct = std::getenv("KBO_MAP_DIRECT") ? std::atoi(std::getenv("KBO_MAP_DIRECT")) : 1; // experiments
    // (and the proof takes at most four windows per mismatch: cov = t - order + 2 bases apart over order bases)
    const uint32_t sp_w = (a.ix.anchor && a.map_thr > a.ix.dtab_order) ? a.map_thr - a.ix.dtab_order : a.map_thr - a.ix.dtab_order + 2u;
    return env_direct != 0 && !a.map_want_ms && a.ix.dtab_order <= a.map_thr && a.map_thr < a.ix.k && (a.ix.dtab_order - 1u + sp_w - 1u) / sp_w + 1u <= 4u;
}

// the packed-native instantiations: the direct form only, and the characters packed only without relative_to_ref
bool map_reads_packed_applies(const WalkArgs &a, bool packed_out) { (void)packed_out; return map_reads_applies(a) && map_reads_direct(a) && !a.map_fmt; }

// the kernel + the list of the reads it could not finish (redo_collect_kernel, for the plain walk: launch_map_reads_redo)
hipError_t launch_map_reads(WalkArgs &a, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    a.table_mode = 1;
    a.table_fused = 1;
    a.unit_bail = a.n_items / 2u + 64u;
    a.plan_list = kPlanList;
    a.plan_cap = (uint32_t)g_plan_cap.load();
    static const int env_x = std::getenv("KBO_MAP_X") ? std::atoi(std::getenv("KBO_MAP_X")) : 0; // experiments: bit 0 = ambiguous seeds as they come
    a.rounds = (uint32_t)env_x;
    static const int env_piece = std::getenv("KBO_REDO_PIECE") ? std::atoi(std::getenv("KBO_REDO_PIECE")) : 0; // experiments
    // (pieces of 16 bases + k - 1 warm-up bases: the pass is as long as its longest chain, and the kernel leaves it under 2 % of the reads:
    // 32 / 16 / 8 bases at C2 0.422 / 0.414 / 0.472 ms per step)
    a.redo_piece = env_piece >= 4 ? (uint32_t)env_piece : 16u;
    hipError_t e = hipMemsetAsync(a.qctl, 0, 64 + kPlanStatSlots * kPlanStatWords * 4, stream);
    if (e != hipSuccess) return e;
    const bool direct = map_reads_direct(a);
    static const bool env_nouni = std::getenv("KBO_MAP_NO_UNIFORM") != nullptr; // experiments
    a.uniform_len = (a.seq_off && !env_nouni && a.max_item_len != 0 && (uint64_t)a.n_items * a.max_item_len == a.q_bytes) ? a.max_item_len : 0u;
    const int io = a.qp ? (a.packed_out ? 2 : 1) : 0;
    if (io != 0 && (!direct || a.map_fmt)) return hipErrorInvalidValue; // (callers ask map_reads_packed_applies first; the packed entry points are kbo::matches: no relative_to_ref)
    // (packed: every read starts a word of the digit string and, with the characters packed as well, of the byte region)
    const uint32_t stage_bytes = io ? (64u * 16u * ((a.max_item_len + 15u) / 16u) + 32u + kMapSlack) : (64u * a.max_item_len + 16u + kMapSlack + 15u) / 16u * 16u;
    const uint32_t lin_words = stage_bytes / 16u + 4u;
    const uint32_t lds_wave = map_wave_lds(direct, stage_bytes, lin_words);
    static const int env_wpb = std::getenv("KBO_MAP_WPB") ? std::atoi(std::getenv("KBO_MAP_WPB")) : 1; // experiments: waves per workgroup
    const uint32_t wpb = (uint32_t)std::min(4, std::max(1, env_wpb));
    const uint32_t n_waves = (a.n_items + 63u) / 64u;
    const dim3 grid((n_waves + wpb - 1u) / wpb), block(64u * wpb);
    static const int env_pad = std::getenv("KBO_MAP_LDS_PAD") ? std::atoi(std::getenv("KBO_MAP_LDS_PAD")) : 0; // experiments: fewer resident waves
    const uint32_t lds = lds_wave * wpb + (uint32_t)env_pad;
    // (the counting instantiations only while kbo_set_plan_stats is on)
#define KBO_MAP_LAUNCH(NP_, DIRECT_, IO_)                                                                                                         \
    do {                                                                                                                                          \
        if (a.pstats) hipLaunchKernelGGL((map_reads_kernel<NP_, DIRECT_, IO_, true>), grid, block, lds, stream, a, stage_bytes, lin_words);       \
        else hipLaunchKernelGGL((map_reads_kernel<NP_, DIRECT_, IO_, false>), grid, block, lds, stream, a, stage_bytes, lin_words);               \
    } while (0)
    if (a.ix.dtab_order <= 15u) {
        if (io == 2) KBO_MAP_LAUNCH(16, true, 2);
        else if (io == 1) KBO_MAP_LAUNCH(16, true, 1);
        else if (direct) KBO_MAP_LAUNCH(16, true, 0);
        else KBO_MAP_LAUNCH(16, false, 0);
    } else {
        if (io == 2) KBO_MAP_LAUNCH(18, true, 2);
        else if (io == 1) KBO_MAP_LAUNCH(18, true, 1);
        else if (direct) KBO_MAP_LAUNCH(18, true, 0);
        else KBO_MAP_LAUNCH(18, false, 0);
    }
#undef KBO_MAP_LAUNCH
    return hipGetLastError();
}

// the reads launch_map_reads' kernel left (its list: qctl[6] entries where the units would be), finished by one kernel: their matching
// statistics where a.d_out wants them (a.map_want_ms, or no characters asked for), their characters, their runs (a.run_counts)
bool map_reads_finish_applies(const WalkArgs &a)
{
    static const int env_finish = std::getenv("KBO_MAP_FINISH") ? std::atoi(std::getenv("KBO_MAP_FINISH")) : 1; // experiments: 0 = the three launches
    return env_finish != 0 && a.seq_off != nullptr && a.qp == nullptr && a.units != nullptr && a.max_item_len != 0 && a.max_item_len <= 16u * kMapWords;
}

hipError_t launch_map_reads_finish(const WalkArgs &a, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    const uint32_t waves = std::min(kFinishWaves, ((a.n_items + 63u) / 64u + 3u) & ~3u);
    // (reads a wave by their number: one up to r1, four up to r4, sixteen beyond - a read takes 64 x 34, 16 x 42 or 4 x 70 lane-steps, and
    // beside other batches' kernels a long list costs by those, not by its chain)
    static const int env_r1 = std::getenv("KBO_FINISH_R1") ? std::atoi(std::getenv("KBO_FINISH_R1")) : 1024; // experiments
    static const int env_r4 = std::getenv("KBO_FINISH_R4") ? std::atoi(std::getenv("KBO_FINISH_R4")) : 4096;
    hipLaunchKernelGGL(finish_reads_kernel, dim3(waves / 4u), dim3(256), 4u * kFinishLds, stream, a, (uint32_t)env_r1, (uint32_t)env_r4);
    return hipGetLastError();
}

} // namespace kbo
