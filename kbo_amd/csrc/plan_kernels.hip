// plan_kernels.hip — gfx950 (MI355X, CDNA4): the plan-guided form of A1 (sbwt::StreamingIndex::matching_statistics,
// called at reference index.rs:251-252) for batches that ask for the MS values only.
//
//   plan_kernel              per work item: a diagonal of the path-cover text, the mismatches of the item against it,
//                            the MS values this predicts (queries and predictions staged through LDS in whole lines)
//   plan_count / plan_emit   mismatch lists -> units (the stretches that have to be walked), two-level scan between them
//   ms_walk_guided_kernel    the extend / contract walk of walk_kernels.hip over the units: from the first mismatch of
//                            a group until the walk itself proves it is back on the diagonal; everything else keeps
//                            the predicted value.  8 resident waves per CU: the lines of the lanes in flight stay in L2
//   ms_walk_recovery_kernel  the same over the recovery lines (sbwt_index.hpp; indexes far beyond L2): extension,
//                            contraction levels and the retry of a failing base from ONE 128-byte line
//   (CALL instantiations)    the breakpoint scan of call_variants carried by the units (kbo_call_walk_dev)
//   redo_collect_kernel      items a unit could not vouch for -> list for the plain kernel, long ones in pieces
//   call_fix_sites_kernel    call mode: rows of matches placed on the diagonal, sites of redone items voided
//
// Why the skipped values are exact (path_cover.cpp has the graph side): if the walk's interval is the single row
// node_at[p] at depth d and the next base equals text[p+1], the reference's step gives the single row node_at[p+1]
// at depth min(d+1, k).  A walk state (d, [l, l+1)) with d == min(k, distance to the last mismatch against the
// diagonal) has a suffix of d bases that equals the text, so l IS the diagonal's node; from there to the next
// mismatch nothing but such steps happens.  Whatever plan_kernel decides (which diagonal, or none) therefore only
// changes how much is walked, never a value.
//
// Integer / byte work only: no MFMA.  Wavefront = 64 lanes; one lane per work item (plan, emit) or per unit (walk).
#include "device_util.hpp"

#include <algorithm>
#include <type_traits>
#include <atomic>
#include <cmath>
#include <cstdlib>

namespace kbo {
namespace {

// bit t (0..3) set where byte t of x is non-zero
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t x)
{
    const uint32_t m = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; // bit 7 of every non-zero byte
    return ((m >> 7) * 0x01020408u) >> 24;                                     // gathered into bits 0..3
}

__device__ __forceinline__ uint32_t sel4(const uint4 &v, uint32_t w)
{
    const uint32_t lo = (w & 1u) ? v.y : v.x, hi = (w & 1u) ? v.w : v.z;
    return (w & 2u) ? hi : lo;
}

// stores bytes [lo, hi) of a 16-byte block to o + lo .. o + hi (0 <= lo <= hi <= 16)
__device__ __forceinline__ void st_range(uint8_t *o, const uint4 &v, uint32_t lo, uint32_t hi)
{
#pragma unroll
    for (uint32_t t = 0; t < 16; t++) { // (compile-time byte positions: no indexed copy of v in scratch)
        const uint32_t w = (t >> 2) == 0 ? v.x : (t >> 2) == 1 ? v.y : (t >> 2) == 2 ? v.z : v.w;
        if (t >= lo && t < hi) o[t] = (uint8_t)(w >> ((t & 3u) * 8u));
    }
}

// -------------------------------------------------------------------------------------------------------------
// plan_kernel: one lane per item, the wave in lock step.
//  0. staging: the items of a wave are neighbours in the query buffer (reads back to back; chunks of a long sequence
//     overlap by their warm-up bases only), so the wave copies its stretch of the buffer into LDS with full-line loads,
//     every lane reads its item from there, writes the predicted values over it (in place: a lane only writes its own
//     output bytes, and within a step all reads come before all writes), and the wave writes the stretch out with
//     full-line stores.  Per-lane bursts would fetch every query / output line once per step instead (a line does not
//     survive in L2 between two steps of a lane: thousands of waves are in flight).  Waves whose items are not
//     neighbours, or do not fit the wave's LDS, go to the buffers directly.
//  1. seed: extend from the root over the item's first bases; when an extension fails, start again from the root
//     with the failing base (no contraction: only a diagonal is wanted, not the MS of these bases).  Done when the
//     interval is a single row at depth >= dmin; given up after `cap` bases.
//  2. compare the whole item with text[p0 ..], p0 = pos[row] - j, kPlanStep blocks of text at a time (all loads of a
//     step go out together: the lines they touch are fetched once); a text byte of 0 (path start, padding) matches
//     nothing.  Predicted MS of base t = min(k, t - last mismatch at or before t) (0 at a mismatch: the guided walk
//     always walks those itself); mismatch positions go to the item's list.
//  3. j_conv: when the seed never restarted and nothing mismatches up to its end, the seed WAS the exact walk of
//     those bases and ended on the diagonal's node, so the guided walk may start at the first mismatch.
constexpr int kPlanStep = 10;           // 16-byte blocks per compare step (reads of up to 160 bases: one step)
constexpr uint32_t kPlanLdsSlack = 48;  // bytes of a wave's LDS behind the staged stretch (block reads run past an item)
constexpr uint32_t kPlanPackBytes = 10u * 64u * 4u; // FUSE: the wave's queries as 2-bit digits
// FUSE (table mode, reads of at most 160 bases): the wave keeps a 2-bit copy of its queries next to the staged stretch (the
// predicted values overwrite the bytes), and every lane then resolves the stretches behind mismatches from the depth table
// itself (dtab_kernels.hip has the rule and the stand-alone kernel for items that cannot be staged): up to 16 / 18
// independent look-ups per mismatch go out together, their values patch the predictions in LDS, and the item's record and
// mismatch list never leave the kernel.  wave_lds = staged stretch + 3840 bytes of digits + 64 x 16 bytes of mismatch
// positions: 14.5 KB for reads of 150 bases, eleven one-wave workgroups a CU (a second full region for the predictions
// instead - 20.4 KB, eight waves a CU - was 0.625 ms at C2: the kernel waits on memory half of its time, and with half of
// those waves it takes 0.93).
// NP = bases a stretch can take: 16 (tables of up to 15 bases: 32-bit keys) or 18 (16 / 17 bases).
template <bool FUSE, int NP = 16>
__global__ __launch_bounds__(256) void plan_kernel(WalkArgs a, uint32_t wave_lds, uint32_t stage_ok, uint32_t stage_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t plan_lds[];
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint8_t *sm = plan_lds + (threadIdx.x >> 6) * wave_lds;
    uint8_t *so = sm;                                            // the predicted values go over the staged queries
    // FUSE: the queries survive as 2-bit digits next to the staged stretch - word w of lane L at pk[w * 64 + L], ten words of 16
    // bases each (the first base most significant; an item with a byte that is no base takes the plain walk: no validity
    // bits) - followed by the lanes' mismatch positions (16 bytes each)
    uint32_t *pk = reinterpret_cast<uint32_t *>(sm + stage_bytes);
    uint8_t *sp = sm + stage_bytes + kPlanPackBytes + lane * 16u;
    const uint32_t n = a.ix.n, k = a.ix.k, nblk = a.ix.n_blocks, null_blk = 4u * nblk;
    const uint8_t *arena = reinterpret_cast<const uint8_t *>(a.ix.arena);
    const uint8_t *qb = a.q;
    uint32_t start = 0, len = 0, warm = 0, olen = 0;
    const bool have_item = idx < a.n_items;
    if (have_item) {
        const uint4 it = ld16(reinterpret_cast<const uint8_t *>(a.items), idx * 16u);
        start = it.x; // launches cover < 4 GiB of query: the high word of WalkItem::start is 0
        len = it.z;
        warm = it.w & 0xFFFFu; // (call mode keeps the bases borrowed from the next chunk in the high half: WalkItem)
        olen = len - min(len, it.w >> 16);
    }
    const bool plannable = have_item && len > 0;
    const uint32_t dmin = max(1u, min(k, a.plan_dmin)), cap = a.plan_cap;

    // ---- 0. staging (everything here is wave-uniform)
    bool staged = false;
    uint32_t base16 = 0, wave_hi = 0, out_lo = 0, soff = 0;
    {
        const uint64_t have = __ballot(have_item); // (item lanes are the wave's first lanes)
        if (stage_ok != 0 && have != 0) {
            const uint32_t last = (uint32_t)__popcll(have) - 1u;
            const uint32_t nxt_start = __shfl_down(start, 1), nxt_warm = __shfl_down(warm, 1);
            const bool bad = (lane < last && (nxt_start < start || nxt_start + nxt_warm != start + len)) || olen != len;
            const uint32_t lo = __shfl(start, 0);
            wave_hi = __shfl(start + len, (int)last);
            out_lo = lo + __shfl(warm, 0);
            base16 = lo & ~15u;
            staged = __ballot(bad) == 0 && wave_hi > lo && (uint64_t)(wave_hi - base16) + kPlanLdsSlack <= (FUSE ? stage_bytes : wave_lds);
            soff = start - base16;
        }
    }
    const bool xpose = !FUSE && !staged && wave_lds >= 64u * 16u * (uint32_t)kPlanStep; // unstaged: outputs transposed through LDS
    if (staged) {
        for (uint32_t c = lane * 16u; c < wave_hi - base16; c += 1024u) // (reads <= 15 bytes past the last item)
            *reinterpret_cast<uint4 *>(sm + c) = ld16u(qb, base16 + c);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (FUSE && staged) { // the 2-bit copy of the item (its bases are overwritten by the predictions further down)
#pragma unroll
        for (uint32_t g = 0; g < (uint32_t)kPlanStep; g++) {
            uint32_t code = 0;
            if (plannable && 16u * g < len) {
                uint4 qv;
                __builtin_memcpy(&qv, sm + soff + 16u * g, 16);
                code = digits16(qv);
            }
            pk[g * 64u + lane] = code;
        }
    }
    auto qld = [&](uint32_t x) -> uint4 { // 16 bytes of the item from base x on
        uint4 v;
        if (staged) __builtin_memcpy(&v, sm + soff + x, 16);
        else __builtin_memcpy(&v, qb + start + x, 16);
        return v;
    };

    // ---- 1. seed
    // state: the interval [l, r) at depth d of the bases [.., j); tab = the next step is a look-up of the seed_d bases
    // from j on in the seed table (the start of a seed: the item's first bases, or the bases behind a failure)
    uint32_t l = 0, r = n, d = 0, j = 0, j0 = 0;
    bool clean = true, seeded = false;
    const uint32_t D = a.ix.seed_tab ? a.ix.seed_d : 0u;
    bool tab = D != 0;
    uint4 qblk = make_uint4(0, 0, 0, 0);
    if (plannable) qblk = qld(0);
    uint32_t qbase = 0; // item position of qblk's first byte
    uint32_t st_lookups = 0, st_ext = 0; // work counters (kPlanStat*)
    for (;;) {
        const bool act = plannable && !seeded && j < len && j < cap;
        if (__ballot(act) == 0) break;
        if (act) {
            if (tab && j + D <= len) {
                // ---- D bases at once (they may straddle two query blocks: bytes are fetched one by one from the
                // current block and the one after it)
                const uint4 qn = (j + D > qbase + 16u) ? qld(qbase + 16u) : qblk;
                uint32_t w = 0, okc = 1;
                for (uint32_t t = 0; t < D; t++) {
                    const uint32_t o = j + t - qbase; // 0 .. 31
                    const uint32_t word = o < 16u ? sel4(qblk, (o >> 2) & 3u) : sel4(qn, (o >> 2) & 3u);
                    const uint32_t c = decode_base((word >> ((o & 3u) * 8u)) & 0xFFu);
                    okc &= c < 4u ? 1u : 0u;
                    w = (w << 2) | (c & 3u);
                }
                uint2 iv = make_uint2(0, 0);
                if (okc) iv = a.ix.seed_tab[w];
                st_lookups += okc;
                if (iv.x < iv.y) {
                    l = iv.x;
                    r = iv.y;
                    d = D;
                    j += D;
                    tab = false;
                    if (r == l + 1u && d >= dmin) {
                        seeded = true;
                        j0 = j - 1u;
                    }
                } else { // no such string in the index: a seed that starts a few bases further on
                    clean = false;
                    j += (D + 1u) / 2u;
                }
                if (j >= qbase + 16u && !seeded) {
                    qbase = j & ~15u;
                    qblk = qld(qbase);
                }
            } else {
                const uint32_t o = j - qbase;
                const uint32_t ch = (sel4(qblk, (o >> 2) & 3u) >> ((o & 3u) * 8u)) & 0xFFu;
                const uint32_t c = decode_base(ch);
                const uint32_t cbk = c < 4u ? c * nblk : null_blk, bmask = c < 4u ? ~0u : 0u;
                const uint32_t bl = div96(l), br = div96(r);
                const uint4 xA = ld16(arena, (cbk + (bl & bmask)) << 4), xB = ld16(arena, (cbk + (br & bmask)) << 4);
                uint32_t l2 = rank_eval(xA, l - bl * kRankRows), r2 = rank_eval(xB, r - br * kRankRows);
                uint32_t dbase = d;
                bool again = false;
                st_ext++;
                if (l2 >= r2) { // the seed ends here: start again behind this base (table) or with it (no table)
                    clean = false;
                    dbase = 0;
                    again = D != 0;
                    l2 = c == 0 ? a.ix.C[0] : c == 1 ? a.ix.C[1] : c == 2 ? a.ix.C[2] : c == 3 ? a.ix.C[3] : 0u;
                    r2 = c == 0 ? a.ix.C[1] : c == 1 ? a.ix.C[2] : c == 2 ? a.ix.C[3] : c == 3 ? a.ix.C[4] : 0u;
                }
                const bool ok = l2 < r2 && !again;
                l = ok ? l2 : 0u;
                r = ok ? r2 : n;
                d = ok ? min(dbase + 1u, k) : 0u;
                tab = again;
                if (r == l + 1u && d >= dmin) {
                    seeded = true;
                    j0 = j;
                }
                j++;
                if (j >= qbase + 16u && !seeded) {
                    qbase = j & ~15u;
                    qblk = qld(qbase); // (direct: stays within the 16-byte slack behind the queries)
                }
            }
        }
    }
    uint32_t p0 = 0;
    if (seeded) p0 = a.ix.pc_pos[l] - j0;

    // ---- 2. compare + predict
    const uint8_t *tb = a.ix.pc_text - kPlanPad; // start of the padded text buffer
    uint16_t *list = a.glist + (size_t)idx * a.plan_list;
    int32_t i_last = -1;
    uint32_t cnt = 0, mm0 = kPlanInf;
    bool has_invalid = false; // FUSE: the item has a byte that is not A, C, G or T
    for (uint32_t base0 = 0;; base0 += 16u * kPlanStep) {
        const bool act = seeded && base0 < len;
        if (__ballot(act) == 0) break;
        uint32_t mmw[kPlanStep / 2]; // mismatch masks of the step's blocks, two per word
#pragma unroll
        for (int g = 0; g < kPlanStep / 2; g++) mmw[g] = 0;
        if (act) {
            uint4 tv[kPlanStep];
#pragma unroll
            for (int g = 0; g < kPlanStep; g++) {
                const uint32_t base = base0 + 16u * g;
                tv[g] = make_uint4(0, 0, 0, 0);
                const uint32_t u = p0 + base + kPlanPad; // offset into the padded buffer (mod 2^32)
                if (base < len && u <= n + 2u * kPlanPad - 16u) __builtin_memcpy(&tv[g], tb + u, 16);
            }
#pragma unroll
            for (int g = 0; g < kPlanStep; g++) {
                const uint32_t base = base0 + 16u * g;
                if (base < len) {
                    const uint4 qv = qld(base);
                    const uint32_t nb = min(16u, len - base);
                    uint32_t mm = (nonzero_bytes(qv.x ^ tv[g].x) | (nonzero_bytes(tv[g].x) ^ 0xFu)) |
                                  ((nonzero_bytes(qv.y ^ tv[g].y) | (nonzero_bytes(tv[g].y) ^ 0xFu)) << 4) |
                                  ((nonzero_bytes(qv.z ^ tv[g].z) | (nonzero_bytes(tv[g].z) ^ 0xFu)) << 8) |
                                  ((nonzero_bytes(qv.w ^ tv[g].w) | (nonzero_bytes(tv[g].w) ^ 0xFu)) << 12);
                    mm &= (1u << nb) - 1u;
                    mmw[g >> 1] |= mm << (16 * (g & 1));
                }
            }
        }
        // (staged: every lane has read its query bytes of this step before any lane overwrites them)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (act) {
#pragma unroll
            for (int g = 0; g < kPlanStep; g++) {
                const uint32_t base = base0 + 16u * g;
                if (base < len) {
                    uint32_t mm = (mmw[g >> 1] >> (16 * (g & 1))) & 0xFFFFu;
                    uint4 o4 = make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        i_last = ((mm >> t) & 1u) ? (int32_t)(base + t) : i_last;
                        const uint32_t val = min((uint32_t)((int32_t)(base + t) - i_last), k);
                        const uint32_t sh = val << ((t & 3) * 8);
                        if ((t >> 2) == 0) o4.x |= sh;
                        else if ((t >> 2) == 1) o4.y |= sh;
                        else if ((t >> 2) == 2) o4.z |= sh;
                        else o4.w |= sh;
                    }
                    // (a constant / SWAR-ramp fast path for blocks without a mismatch - nine of ten - was tried: the wave runs
                    // all three paths anyway, plan_kernel 227 -> 262 us, VALU 79.9 -> 87.6 M instructions.  Removed.)
                    while (mm) { // the item's mismatch list: entry 0 in the item record, entries 1..12 in the list
                        const uint32_t pos = base + (uint32_t)__ffs((int)mm) - 1u;
                        if (FUSE) {
                            if (cnt < 16u) sp[cnt] = (uint8_t)pos;
                            // (a byte that is no base is a mismatch against any text: the only places to look for one.  The
                            // item's bytes of this block are still there: its predictions are written below)
                            if (staged) has_invalid = has_invalid || decode_base(sm[soff + pos]) >= 4u;
                        } else if (cnt == 0) mm0 = pos;
                        else if (cnt <= a.plan_list) list[cnt - 1u] = (uint16_t)pos;
                        cnt++;
                        mm &= mm - 1u;
                    }
                    // (olen: call mode's items end with bases of the next chunk, whose MS bytes are that chunk's to write)
                    const uint32_t nb = olen > base ? min(16u, olen - base) : 0u;
                    const uint32_t lo = min(warm > base ? min(warm - base, 16u) : 0u, nb);
                    if (staged) {
                        uint8_t *o = so + soff + base;
                        if (lo == 0 && nb == 16u) __builtin_memcpy(o, &o4, 16);
                        else st_range(o, o4, lo, nb);
                    } else if (xpose) { // (goes out below, with the other lanes' blocks)
                        *reinterpret_cast<uint4 *>(sm + (lane * (uint32_t)kPlanStep + (uint32_t)g) * 16u) = o4;
                    } else {
                        uint8_t *o = a.d_out + (start + base);
                        if (lo == 0 && nb == 16u) __builtin_memcpy(o, &o4, 16); // (plain store: the blocks of a step merge in L2)
                        else st_range(o, o4, lo, nb);
                    }
                }
            }
        }
        if (xpose) {
            // the step's blocks of all lanes, lane-major in LDS, go out in that order: ten consecutive lanes write one
            // item's 160 bytes (whole lines) - a lane storing its own ten blocks one by one puts 64 partial lines into
            // every store instruction (unstaged plan kernel on 830-base chunks: 0.35 ms, 0.14 of it these stores)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t actw = act ? 1u : 0u;
#pragma unroll 1
            for (uint32_t i = 0; i < (uint32_t)kPlanStep; i++) {
                const uint32_t t = i * 64u + lane, c = t / (uint32_t)kPlanStep, g = t - c * (uint32_t)kPlanStep;
                const uint32_t c_start = __shfl(start, (int)c), c_len = __shfl(len, (int)c), c_olen = __shfl(olen, (int)c),
                               c_warm = __shfl(warm, (int)c), c_act = __shfl(actw, (int)c);
                const uint32_t base = base0 + 16u * g;
                if (c_act && base < c_len) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(sm + t * 16u);
                    const uint32_t nb = c_olen > base ? min(16u, c_olen - base) : 0u;
                    const uint32_t lo = min(c_warm > base ? min(c_warm - base, 16u) : 0u, nb);
                    uint8_t *o = a.d_out + (c_start + base);
                    if (lo == 0 && nb == 16u) __builtin_memcpy(o, &v, 16);
                    else st_range(o, v, lo, nb);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // ---- FUSE: the stretches behind the mismatches from the depth table (rule and counters: dtab_resolve_kernel).
    // The wave's mismatches are dealt out to its lanes whatever item they belong to (a lane with five mismatches would
    // otherwise keep 63 others waiting): work w belongs to the lane whose prefix of mismatch counts covers it.
    uint32_t st_look = 0, st_written = 0, st_anch = 0;
    bool flag = false, no_plan = false;
    if (FUSE) {
        const uint32_t order = a.ix.dtab_order;
        flag = have_item && len != 0 && !staged; // (an unstaged wave cannot resolve here: its items take the plain walk)
        // more mismatches than the list holds (a seed on a wrong diagonal, an error-dense read): the plain walk
        flag = flag || (plannable && staged && seeded && cnt > a.plan_list + 1u);
        // no seed at all (unrelated reads, the other strand): EVERY base of the item from the table, 16 at a time - reads that
        // match nothing deeper than the table knows (most reads that match nothing) are done with 150 independent look-ups
        // instead of a walk; one base the table cannot tell sends the item to the plain walk.  (An item that did seed has a
        // match of log4(rows) + 3 bases and more: looking all its bases up would mostly find what the table cannot tell.)
        no_plan = plannable && staged && !seeded;
        if (__ballot(no_plan)) { // (their bytes were never compared with a text - nor overwritten: look for one that is no base)
            if (no_plan)
                for (uint32_t x = 0; x < len; x++) has_invalid = has_invalid || decode_base(sm[soff + x]) >= 4u;
        }
        // an item with a byte that is no base: the plain walk (the 2-bit copy cannot say where a window ends)
        flag = flag || (plannable && has_invalid);
        const uint32_t my_n = !plannable || flag ? 0u : (no_plan ? (len + 15u) / 16u : cnt);
        uint32_t incl = my_n; // inclusive scan of the counts over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if ((int)lane >= off) incl += t;
        }
        const uint32_t total = __shfl(incl, 63);
        uint8_t *spw = sm + stage_bytes + kPlanPackBytes; // the wave's 64 x 16 bytes: positions 0 .. 12, flag at 13, prefix (u16) at 14
        sp[13] = 0;
        *reinterpret_cast<uint16_t *>(sp + 14) = (uint16_t)incl;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        using code_t = typename std::conditional<NP == 16, uint32_t, uint64_t>::type; // `order` + 1 bases as 2-bit digits
        const code_t omask = (code_t)((1ull << (2u * order)) - 1ull);
        // stretches with a base deeper than the table knows are left to the anchors - when the copy has them - in the wave's
        // list for dtab_stretch_kernel (kernels.hpp DtabStretch*): an anchor is two more dependent loads, and all 64 lanes
        // would wait for the one that needs them
        const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        uint8_t *sblk = reinterpret_cast<uint8_t *>(a.units) + (size_t)wave_id * kDtabStretchBlockBytes;
        const bool have_anchors = a.ix.anchor != nullptr;
        uint32_t n_def = 0; // (wave-uniform)
        for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
            const uint32_t w = w0 + lane;
            const bool work = w < total;
            // owner: the first lane whose inclusive prefix exceeds w
            uint32_t lo_l = 0, hi_l = 63;
#pragma unroll
            for (int it = 0; it < 6; it++) {
                const uint32_t mid = (lo_l + hi_l) >> 1;
                const uint32_t pm = *reinterpret_cast<const uint16_t *>(spw + mid * 16u + 14u);
                if (pm > w) hi_l = mid;
                else lo_l = mid + 1u;
            }
            const uint32_t owner = work ? lo_l : lane;
            const uint32_t o_incl = __shfl(incl, (int)owner), o_n = __shfl(my_n, (int)owner), o_soff = __shfl(soff, (int)owner),
                           o_len = __shfl(len, (int)owner), o_warm = __shfl(warm, (int)owner), o_start = __shfl(start, (int)owner),
                           o_np = __shfl(no_plan ? 1u : 0u, (int)owner), o_idx = __shfl(idx, (int)owner);
            uint32_t m = 0, nxt = 0, evalmask = 0, unkmask = 0, satmask = 0;
            uint32_t outv[5] = {0, 0, 0, 0, 0};
            const bool blockmode = o_np != 0; // 16 bases of an item without a plan: every value on its own, no stretch logic
            if (work) {
                const uint32_t t = w - (o_incl - o_n);
                const uint8_t *osp = spw + owner * 16u;
                m = blockmode ? 16u * t : (uint32_t)osp[t];
                nxt = (blockmode || t + 1u >= o_n) ? o_len : (uint32_t)osp[t + 1u];
                const uint32_t P = min(min(blockmode ? 16u : order + 1u, (uint32_t)NP), min(nxt, o_len) - m); // bases looked up: m .. m + P - 1
                // the owner's bases, from its 2-bit copy: base x = digit 15 - x mod 16 of word x / 16 (all of them are bases)
                uint32_t cw_i = ~0u, cw = 0;
                auto base_at = [&](uint32_t x) -> uint32_t {
                    if ((x >> 4) != cw_i) {
                        cw_i = x >> 4;
                        cw = pk[cw_i * 64u + owner];
                    }
                    return (cw >> (2u * (15u - (x & 15u)))) & 3u;
                };
                // the bases in front of m: `order` of them are enough (a run that reaches further back counts as "> order")
                code_t code = 0;
                uint32_t v = m > order ? order : m; // bases in front of m, inside the item
                for (uint32_t x = m - v; x < m; x++) code = (code << 2) | base_at(x);
                uint32_t tv[NP];               // the table's bytes
                uint64_t meta0 = 0, meta1 = 0, meta2 = 0; // per base: min(v, 31) | extension base << 5 | no window << 7
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)NP; j++) {
                    tv[j] = 0;
                    if (j < P) {
                        const uint32_t i = m + j;
                        code = (code << 2) | base_at(i);
                        v++;
                        const code_t key = code & omask;
                        const bool nowin = (uint64_t)o_start + i + 1u < 32u; // (as the stand-alone kernel: the buffer's first bytes)
                        const uint64_t me = (uint64_t)(min(v, 31u) | (((uint32_t)(code >> (2u * order)) & 3u) << 5) | (nowin ? 128u : 0u));
                        if (j < 8) meta0 |= me << (8u * j);
                        else if (j < 16) meta1 |= me << (8u * (j - 8u));
                        else meta2 |= me << (8u * (j - 16u));
                        if (!nowin) {
                            tv[j] = !a.ix.dtab_grouped ? a.ix.dtab[key]
                                                         : a.ix.dtab[NP == 16 ? dtab_grouped_addr32((uint32_t)key, i % 3u, order) : dtab_grouped_addr((uint64_t)key, i % 3u, order)];
                            st_look++;
                        }
                    }
                }
                // a stretch: the bases in front of the first one where the longest present suffix is the j bases behind the
                // mismatch (a block: all of them), as far as the table tells
                bool done = false;
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)NP; j++) {
                    if (j < P && !done) {
                        const uint32_t me = (uint32_t)((j < 8 ? meta0 >> (8u * j) : j < 16 ? meta1 >> (8u * (j - 8u)) : meta2 >> (8u * (j - 16u))) & 0xFFu);
                        const uint32_t vv = me & 31u, eb = (me >> 5) & 3u, byte = tv[j];
                        uint32_t L = k + 1u;
                        if (me & 128u) unkmask |= 1u << j; // (no window: unknown, and no end of the stretch here)
                        else if (!(byte & 0x80u)) L = min(byte, vv);
                        else if (vv <= order || order >= k) L = min(vv, order);
                        else if ((byte >> eb) & 1u) satmask |= 1u << j; // deeper than the table knows (no end of the stretch either)
                        else L = order;
                        outv[j >> 2] |= min(L, k) << (8u * (j & 3u));
                        evalmask |= 1u << j;
                        done = !blockmode && L <= j;
                    }
                }
            }
            // bases deeper than the table knows: a stretch goes to the wave's list when the copy has anchors (and nothing else
            // is unknown in it); otherwise - and for the blocks of an item without a plan - they are unknown
            const bool defer = work && have_anchors && !blockmode && satmask != 0 && unkmask == 0;
            const uint64_t dm = __ballot(defer);
            if (dm) {
                const uint32_t slot = n_def + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull));
                if (defer) {
                    if (slot < kDtabStretchCap)
                        *reinterpret_cast<uint4 *>(sblk + 16u + (size_t)slot * 16u) = make_uint4(o_idx, o_start, m | (nxt << 16), o_len | (o_warm << 16));
                    else unkmask |= satmask; // (no room in the list: the item takes the plain walk)
                }
                n_def += (uint32_t)__popcll(dm);
            }
            if (work) {
                const bool deferred = defer && !(unkmask & satmask);
                if (!deferred) {
                    st_anch += (uint32_t)__popc(satmask); // (counted as tried, like the stand-alone kernel without anchors)
                    unkmask |= satmask;
                }
                if (unkmask) spw[owner * 16u + 13u] = 1; // the owner's item goes to the plain walk
                // (a stretch with an unknown base - or one left to the anchors - writes nothing; a block writes the bases it knows)
                const uint32_t wmask = deferred ? 0u : blockmode ? evalmask & ~unkmask : (unkmask ? 0u : evalmask);
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)NP; j++)
                    if (((wmask >> j) & 1u) && m + j >= o_warm) {
                        so[o_soff + m + j] = (uint8_t)(outv[j >> 2] >> (8u * (j & 3u)));
                        st_written++;
                    }
            }
        }
        if (have_anchors && lane == 0) *reinterpret_cast<uint32_t *>(sblk) = min(n_def, kDtabStretchCap);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        flag = flag || sp[13] != 0;
    }
    if (staged) { // the wave's output bytes [out_lo, wave_hi), in blocks aligned like the staged copy
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t c = lane * 16u; c < wave_hi - base16; c += 1024u) {
            const uint4 v = *reinterpret_cast<const uint4 *>(so + c);
            const uint32_t g0 = base16 + c;
            if (g0 >= out_lo && g0 + 16u <= wave_hi) __builtin_memcpy(a.d_out + g0, &v, 16);
            else st_range(a.d_out + g0, v, out_lo > g0 ? min(out_lo - g0, 16u) : 0u, min(wave_hi - g0, 16u));
        }
    }
    plan_stats_add(a.pstats, kPlanStatSeedLookups, st_lookups, kPlanStatSeedExtensions, st_ext, kPlanStatMismatches, cnt, 0, 0);
    if (FUSE) {
        plan_stats_add(a.pstats, kPlanStatTabLookups, st_look, kPlanStatTabWritten, st_written, kPlanStatTabFlagged, flag ? 1u : 0u,
                       kPlanStatTabAnchored, st_anch);
        const uint64_t fm = __ballot(flag), nm = __ballot(no_plan);
        if (lane == 0) {
            if (fm) atomicAdd(a.qctl + 4, (uint32_t)__popcll(fm));
            if (nm) atomicAdd(a.qctl + 5, (uint32_t)__popcll(nm));
        }
        if (have_item) a.redo[idx] = flag ? 1 : 0;
        return; // (no record, no list: nothing reads them in this form)
    }
    if (!have_item) return;
    const uint32_t j_conv = (seeded && clean && (cnt == 0 || mm0 > j0)) ? j0 + 1u : 0u;
    const uint32_t n_mm = seeded ? min(cnt, 254u) : kPlanNone;
    uint4 g;
    g.x = start;
    g.y = p0;
    g.z = (len & 0xFFFFu) | (j_conv << 16);
    g.w = (mm0 & 0xFFFFu) | ((warm & 0xFFu) << 16) | (n_mm << 24);
    *reinterpret_cast<uint4 *>(reinterpret_cast<uint8_t *>(a.gitems) + (size_t)idx * 16u) = g;
    a.redo[idx] = 0;
}

// -------------------------------------------------------------------------------------------------------------
// Units of one item, from its record and mismatch list; EMIT = false only counts them (same code, so that the
// counts the scan is built from are the counts the second pass writes).
//  * item without a plan (no diagonal, or more mismatches than the list holds): chunks of `chunk` output bases, each
//    walked from the root k-1 bases upstream (MS depends on the last k bases only), no convergence test;
//  * otherwise one unit per group of mismatches closer than `gap` (>= 2, so that the base in front of a group's first
//    mismatch is a match and the diagonal's node there is the walk's state once the previous group has converged);
//    the first group of an item whose first bases were not walked exactly by plan_kernel starts at base 0 from the root.
// Units come in two weights, heavy ones (chunks, the head of an item, groups that span more than a few bases) first
// in the queue: the longest units then start early instead of stretching the end of the launch.
// Returns heavy | light << 16.
// lim_len != 0: call mode - bases of the item that are its own | the item's length << 16, copied into every unit (the walk's
// breakpoint scan needs them); the chunks of an item without a plan then own `chunk` bases each (their breakpoints and
// MS bytes), start k bases upstream and may walk up to k bases past their end to resolve the breakpoints still waiting.
template <bool EMIT>
__device__ __forceinline__ uint32_t make_units(const uint4 &g, const uint16_t *list, uint32_t k, uint32_t gap, uint32_t chunk,
                                               uint32_t item, WalkUnit *out_heavy, WalkUnit *out_light, uint32_t lim_len = 0,
                                               uint32_t plist = kPlanList /* list entries per item */)
{
    const uint32_t len = g.z & 0xFFFFu, j_conv = g.z >> 16, mm0 = g.w & 0xFFFFu, warm = (g.w >> 16) & 0xFFu, n_mm = g.w >> 24;
    if (len == 0) return 0;
    uint32_t nh = 0, nl = 0;
    auto put = [&](uint32_t pos, uint32_t out_from, int32_t last_mm, uint32_t bound, uint32_t d_start, uint32_t flags, uint32_t ll) {
        const bool heavy = (flags & (kUnitPlain | kUnitHead)) || last_mm - (int32_t)pos > 8;
        WalkUnit *out = heavy ? out_heavy : out_light;
        uint32_t &nu = heavy ? nh : nl;
        if (EMIT) {
            uint4 w0, w1;
            w0.x = g.x;
            w0.y = g.y;
            w0.z = pos | (out_from << 16);
            w0.w = ((uint32_t)last_mm & 0xFFFFu) | (bound << 16);
            w1.x = d_start | (flags << 8) | (warm << 16);
            w1.y = item;
            w1.z = ll;
            w1.w = 0;
            uint4 *o = reinterpret_cast<uint4 *>(out + nu);
            o[0] = w0;
            o[1] = w1;
        }
        nu++;
    };
    if (n_mm == kPlanNone || n_mm > plist + 1u) {
        const bool call = lim_len != 0;
        const uint32_t own = call ? (lim_len & 0xFFFFu) : len, wk = call ? k : k - 1u;
        for (uint32_t c0 = warm; c0 < own; c0 += chunk) {
            const uint32_t own_end = min(own, c0 + chunk), bound = call ? min(len, own_end + k) : own_end;
            put(c0 > wk ? c0 - wk : 0u, c0, -1, bound, 0u, kUnitHead | kUnitPlain | (bound == len ? kUnitToEnd : 0u),
                call ? (own_end | (len << 16)) : 0u);
        }
        return nh | (nl << 16);
    }
    auto mm_at = [&](uint32_t t) -> uint32_t { return t == 0 ? mm0 : (uint32_t)list[t - 1u]; };
    uint32_t t = 0;
    int32_t prev = -1;
    bool head = j_conv == 0;
    while (head || t < n_mm) {
        uint32_t pos = 0, d_start = 0, flags = head ? kUnitHead : 0u;
        int32_t last = -1;
        if (!head) {
            pos = mm_at(t);
            d_start = min((uint32_t)((int32_t)pos - 1 - prev), k);
            last = prev = (int32_t)pos;
            t++;
        }
        head = false;
        while (t < n_mm && (int32_t)mm_at(t) - last < (int32_t)gap) {
            last = prev = (int32_t)mm_at(t);
            t++;
        }
        const uint32_t bound = t < n_mm ? mm_at(t) : len;
        put(pos, max(warm, pos), last, bound, d_start, flags | (t < n_mm ? 0u : kUnitToEnd), lim_len);
    }
    return nh | (nl << 16);
}

__global__ __launch_bounds__(256) void plan_count_kernel(WalkArgs a)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_items) return;
    const uint4 g = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.gitems) + (size_t)idx * 16u);
    uint32_t lim_len = 0;
    if (a.call_sites) { // (call mode: see plan_emit_kernel)
        const uint4 it = ld16(reinterpret_cast<const uint8_t *>(a.items), idx * 16u);
        lim_len = (it.z - (it.w >> 16)) | (it.z << 16);
    }
    const uint32_t c = make_units<false>(g, a.glist + (size_t)idx * a.plan_list, a.ix.k, a.plan_gap, a.plan_chunk, idx, nullptr, nullptr, lim_len, a.plan_list);
    a.ucount[idx] = c & 0xFFFFu;              // class-major: all heavy counts, then all light counts, then one 0
    a.ucount[a.n_items + idx] = c >> 16;
    if (idx == 0) a.ucount[2u * a.n_items] = 0; // the prefix of this extra entry is the number of units
}

__global__ __launch_bounds__(256) void plan_emit_kernel(WalkArgs a)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.n_items) return;
    const uint4 g = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint8_t *>(a.gitems) + (size_t)idx * 16u);
    auto prefix = [&](uint32_t e) -> uint32_t { return a.usums[e / kScanBlock] + a.ucount[e]; };
    if (prefix(2u * a.n_items) > a.unit_bail) return; // too much to walk for the plan to pay: every item takes the
                                                      // plain walk (redo_collect_kernel lists them all)
    uint32_t lim_len = 0;
    if (a.call_sites) { // call mode: the walk of a unit needs the item's own length and whole length for its breakpoint scan
        const uint4 it = ld16(reinterpret_cast<const uint8_t *>(a.items), idx * 16u);
        lim_len = (it.z - (it.w >> 16)) | (it.z << 16);
    }
    const uint32_t hs = prefix(idx), he = prefix(idx + 1u), ls = prefix(a.n_items + idx), le = prefix(a.n_items + idx + 1u);
    if (le > a.unit_cap || he > a.unit_cap) { // no room for this item's units: it takes the full walk instead
        if (he > hs || le > ls) a.redo[idx] = 1;
        for (uint32_t part = 0; part < 2; part++) // (slots below the capacity are still consumed: empty units)
            for (uint32_t sl = part ? ls : hs; sl < min(part ? le : he, a.unit_cap); sl++) {
                uint4 *o = reinterpret_cast<uint4 *>(a.units + sl);
                o[0] = make_uint4(0, 0, 0, 0);
                o[1] = make_uint4((kUnitHead | kUnitPlain) << 8, idx, 0, 0);
            }
        return;
    }
    make_units<true>(g, a.glist + (size_t)idx * a.plan_list, a.ix.k, a.plan_gap, a.plan_chunk, idx, a.units + hs, a.units + ls, lim_len,
                     a.plan_list);
}

// Collects the items the guided walk flagged into a list for the plain kernel; qctl[1] counts its entries.  The list is
// built in the unit array (all units are walked by now; 2 * unit_cap item records fit: more than every item cut into
// pieces).  A flagged item longer than 96 output bases is cut into pieces of 64 with their own warm-up (and, in call mode,
// borrowed) bases, like the chunks of a long sequence: the redo pass is a handful of items, and one lane walking 800 bases
// on its own would be most of the stage's time.  A plan that was given up: all items as they are, in order.
constexpr uint32_t kRedoPiece = 64, kRedoPieceTable = 32; // (table mode: a few per cent of the reads come here, see below)
constexpr uint32_t kRedoBlock = 1024;
constexpr uint32_t kRedoWholeFrom = 50000; // flagged items from which the redo pass walks them whole (table mode): their
                                           // pieces of 32 would be more than half of the lanes the device holds
__global__ __launch_bounds__(1024) void redo_collect_kernel(WalkArgs a)
{
    __shared__ uint32_t wave_tot[kRedoBlock / 64], block_base;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = idx < a.n_items;
    WalkItem *list = reinterpret_cast<WalkItem *>(a.units);
    uint4 it = make_uint4(0, 0, 0, 0);
    if (have && a.seq_off) { // (map_reads_kernel's batches: item s is sequence s, whole - no item list was ever written)
        const uint64_t o0 = a.seq_off[idx], o1 = a.seq_off[idx + 1u];
        it = make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), (uint32_t)(o1 - o0), 0u);
    } else if (have) it = ld16(reinterpret_cast<const uint8_t *>(a.items), idx * 16u);
    // plan given up: all items, in order.  The same when a wave of the guided walk left through its no-progress guard
    // (qctl[3]; it cannot, but then units are unwalked): every item is walked again in full, so the batch stays exact.
    // (table mode: the plan is given up when more than unit_bail items had no plan or were left unresolved by the table)
    const bool gave_up = a.table_mode ? a.qctl[4] > a.unit_bail
                                      : a.usums[(2u * a.n_items) / kScanBlock] + a.ucount[2u * a.n_items] > a.unit_bail;
    if (gave_up || a.qctl[3]) { // (block-uniform)
        if (have) reinterpret_cast<uint4 *>(list)[idx] = it;
        if (idx == 0) {
            a.qctl[1] = a.n_items;
            if (gave_up) a.qctl[2] = 1; // tells the host (plan_after_launch) that planning did not pay for this batch
            if (gave_up && a.host_bailed) *a.host_bailed = 1u; // (... straight into its pinned word where the launch gave one)
        }
        return;
    }
    // one counter update per block of 1024 items (returning atomics on one address take about 10 ns each: with a few per cent
    // of the items flagged - the table form - one per wave was 0.13 ms)
    const bool f = have && a.redo[idx] != 0;
    const uint32_t len = it.z, warm = it.w & 0xFFFFu, tail = it.w >> 16, body = len - warm - tail;
    const uint32_t marg = a.call_sites ? a.ix.k : (a.ix.k > 0 ? a.ix.k - 1u : 0u); // warm-up of a chunk (make_chunk_items_kernel)
    // a flagged item longer than a piece and a half is cut into pieces with their own warm-up: the pass is bound by the
    // longest chain of dependent look-ups, not by their number (pieces of 64 bases when a handful of items come here, of 32
    // in table mode)
    // (table mode: pieces of 32 while the pass is bound by its longest chain - C2: 20 k flagged reads, 16 / 32 / 64 bases
    // 0.692 / 0.679 / 0.706 ms -, whole reads once there are enough of them to fill the device (50 k: 250 k pieces), when the warm-up bases of the
    // pieces are what costs - C3: 233 k flagged reads per slab, 16 / 32 / 64 / whole 7.99 / 7.34 / 7.05 / 6.76 ms)
    const uint32_t piece = a.table_mode ? (a.qctl[4] > kRedoWholeFrom ? 0xFFFFu : a.redo_piece) : kRedoPiece;
    const uint32_t np = !f ? 0u : (body > piece + piece / 2u ? (body + piece - 1u) / piece : 1u);
    uint32_t incl = np; // inclusive scan over the wave
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)lane >= off) incl += t;
    }
    if (lane == 63u) wave_tot[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < blockDim.x / 64u; w++) tot += wave_tot[w];
        block_base = tot ? atomicAdd(a.qctl + 1, tot) : 0u;
    }
    __syncthreads();
    uint32_t base = block_base + incl - np;
    for (uint32_t w = 0; w < wv; w++) base += wave_tot[w];
    if (np == 1) {
        if (base < a.redo_cap) reinterpret_cast<uint4 *>(list)[base] = it;
    } else {
        for (uint32_t p = 0; p < np; p++) {
            const uint32_t out_lo = warm + p * piece, out_hi = min(out_lo + piece, warm + body);
            const uint32_t w2 = p == 0 ? warm : min(marg, out_lo);
            const uint32_t t2 = a.call_sites ? min(a.ix.k, len - out_hi) : 0u;
            if (base + p < a.redo_cap)
                reinterpret_cast<uint4 *>(list)[base + p] = make_uint4(it.x + out_lo - w2, it.y, (out_hi - out_lo) + w2 + t2, w2 | (t2 << 16));
        }
    }
}

// The same behind map_reads_kernel (items = the batch's sequences, whole; table mode): sixteen flags a lane, so that a batch of 5 M
// reads is 1 200 workgroups instead of 4 900 - each of which waits for its returning atomic - and nearly every lane is done when it
// has seen sixteen zero bytes (C3: 0.26 -> ms per 5 M reads beside another batch's kernel).
constexpr uint32_t kRedoReadsBlock = 256;
__global__ __launch_bounds__(kRedoReadsBlock) void redo_collect_reads_kernel(WalkArgs a)
{
    __shared__ uint32_t wave_tot[kRedoReadsBlock / 64], block_base;
    const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    WalkItem *list = reinterpret_cast<WalkItem *>(a.units);
    auto item_of = [&](uint32_t idx) -> uint4 {
        const uint64_t o0 = a.seq_off[idx], o1 = a.seq_off[idx + 1u];
        return make_uint4((uint32_t)o0, (uint32_t)(o0 >> 32), (uint32_t)(o1 - o0), 0u);
    };
    const bool gave_up = a.qctl[4] > a.unit_bail;
    if (gave_up || a.qctl[3]) { // (block-uniform; see redo_collect_kernel)
        for (uint32_t j = 0; j < 16u; j++)
            if (i0 + j < a.n_items) reinterpret_cast<uint4 *>(list)[i0 + j] = item_of(i0 + j);
        if (i0 == 0) {
            a.qctl[1] = a.n_items;
            if (gave_up) a.qctl[2] = 1;
            if (gave_up && a.host_bailed) *a.host_bailed = 1u;
        }
        return;
    }
    uint32_t fl = 0; // bit j: item i0 + j is flagged
    if (i0 < a.n_items) {
        const uint4 f = *reinterpret_cast<const uint4 *>(a.redo + i0); // (the flag array is padded to 16 bytes)
        fl = nonzero_bytes(f.x) | (nonzero_bytes(f.y) << 4) | (nonzero_bytes(f.z) << 8) | (nonzero_bytes(f.w) << 12);
        if (a.n_items - i0 < 16u) fl &= (1u << (a.n_items - i0)) - 1u;
    }
    const uint32_t marg = a.ix.k > 0 ? a.ix.k - 1u : 0u;
    const uint32_t piece = a.qctl[4] > kRedoWholeFrom ? 0xFFFFu : a.redo_piece; // (as redo_collect_kernel's table mode)
    auto pieces_of = [&](uint32_t body) -> uint32_t { return body > piece + piece / 2u ? (body + piece - 1u) / piece : 1u; };
    uint32_t np = 0;
    for (uint32_t m = fl; m; m &= m - 1u) {
        const uint32_t idx = i0 + (uint32_t)__builtin_ctz(m);
        np += pieces_of((uint32_t)(a.seq_off[idx + 1u] - a.seq_off[idx]));
    }
    uint32_t incl = np; // inclusive scan over the wave
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if ((int)lane >= off) incl += t;
    }
    if (lane == 63u) wave_tot[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (uint32_t w = 0; w < blockDim.x / 64u; w++) tot += wave_tot[w];
        block_base = tot ? atomicAdd(a.qctl + 1, tot) : 0u;
    }
    __syncthreads();
    uint32_t base = block_base + incl - np;
    for (uint32_t w = 0; w < wv; w++) base += wave_tot[w];
    for (uint32_t m = fl; m; m &= m - 1u) {
        const uint32_t idx = i0 + (uint32_t)__builtin_ctz(m);
        const uint4 it = item_of(idx);
        const uint32_t body = it.z, n = pieces_of(body);
        if (n == 1u) {
            if (base < a.redo_cap) reinterpret_cast<uint4 *>(list)[base] = it;
        } else {
            for (uint32_t p = 0; p < n; p++) {
                const uint32_t out_lo = p * piece, out_hi = min(out_lo + piece, body);
                const uint32_t w2 = p == 0 ? 0u : min(marg, out_lo);
                if (base + p < a.redo_cap) reinterpret_cast<uint4 *>(list)[base + p] = make_uint4(it.x + out_lo - w2, it.y, (out_hi - out_lo) + w2, w2);
            }
        }
        base += n;
    }
}

// Call mode, after the guided walk: a site written by a unit carries its item (w = item + 1); the sites of items that are
// flagged for the redo pass are made void (x = ~0: that pass scans the item again), the others get the row of a match
// that was placed on the diagonal (w's top bit: z holds the text position) and w = 0.
__global__ __launch_bounds__(256) void call_fix_sites_kernel(WalkArgs a)
{
    const uint32_t seg = blockIdx.x;
    const uint32_t n = min(a.call_counts[seg * 16u], a.call_cap);
    uint4 *list = a.call_sites + (size_t)seg * a.call_cap;
    for (uint32_t sl = threadIdx.x; sl < n; sl += blockDim.x) {
        uint4 v = list[sl];
        if (v.w == 0) continue;
        const uint32_t item = (v.w & 0x7FFFFFFFu) - 1u;
        if (a.redo[item] || a.qctl[3]) v.x = ~0u; // (qctl[3]: every item is scanned again, see redo_collect_kernel)
        else if (v.w >> 31) v.z = a.ix.pc_node[v.z];
        v.w = 0;
        list[sl] = v;
    }
}

// per-lane flag bits of the guided walk
enum : uint32_t {
    G_QF = 1u,    // fetch the query block after the current one
    G_CON = 2u,   // contracting: loads contraction entries instead of rank blocks
    G_HAVE = 4u,  // next unit (record, start row, first query block) is prefetched
    G_PF = 8u,    // next unit's record is in flight, its row and query block not yet requested
    G_ENT = 64u,  // (recovery lines) the next contraction level is read from the entries
    // the bits below take the lane out of the hot path until the bookkeeping block has run
    G_DONE = 16u, // finished its unit, wants the next one
    G_FIN = 32u,  // no units left
    G_FLUSH = 128u, // its window of output bytes is full: written out by the bookkeeping block
    G_BLOCKED = G_DONE | G_FIN | G_FLUSH
};

// -------------------------------------------------------------------------------------------------------------
// ms_walk_guided_kernel.  Same walk as ms_walk_kernel (walk_kernels.hip: extend by rank blocks, contraction by
// {lcs, psv, nsv} entries, one loop with a hot path and a bookkeeping block) over units instead of items:
//  * a unit starts from the diagonal's row in front of its first base (prefetched with the unit's record and query
//    block by the bookkeeping block) or from the root;
//  * after every accepted base, once the group's last mismatch is behind:
//    converged = single-row interval && d == min(k, bases since that mismatch) -> the unit is done;
//  * a unit that reaches its bound first (and the bound is not the item's end) flags the item for the full walk;
//  * output in words (4 bases), bytes at the two ends of the walked stretch, so that a unit patches the predicted
//    values without touching its neighbours;
//  * units come off one queue in chunks of 64 per wave; a lane that finishes takes the next one.
// STATS: the kernel counts its own work (kPlanStat*; kbo_set_plan_stats) - instrumentation, compiled out of the default
// instantiation (about 1 % of the kernel's time)
template <bool BIG, bool CALL, bool STATS>
__global__ __launch_bounds__(256) void ms_walk_guided_kernel(WalkArgs a)
{
    const uint32_t n = a.ix.n, k = a.ix.k;
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t *arena = reinterpret_cast<const uint8_t *>(a.ix.arena);
    const uint8_t *qb = a.q;
    const uint8_t *utb = reinterpret_cast<const uint8_t *>(a.units);
    const uint8_t *nodeb = reinterpret_cast<const uint8_t *>(a.ix.pc_node);
    const uint32_t q_end = (uint32_t)a.q_bytes;
    const uint32_t nblk = a.ix.n_blocks;
    const uint32_t null_blk = 4u * nblk;
    const uint32_t ent_byte0 = a.ix.lcs_off << 4;

    const uint32_t u_total = a.usums[(2u * a.n_items) / kScanBlock] + a.ucount[2u * a.n_items];
    const uint32_t q_total = u_total > a.unit_bail ? 0u : min(u_total, a.unit_cap); // (see plan_emit_kernel)
    uint32_t pool_next = 0, pool_end = 0;
    bool drained = false; // (wave-uniform) the queue has nothing left

    uint32_t flags = G_DONE;
    uint32_t l = 0, r = n, d = 0, m = 0, cb = 0, tgt_l = 0, tgt_r = 0;
    uint32_t i = 0, start = 0, warm = 0, bound = 0, out_from = 0, uflags = 0, item = 0;
    int32_t last_mm = -1;
    uint4 qblk = make_uint4(0, 0, 0, 0), qnxt = make_uint4(0, 0, 0, 0);
    uint32_t qcur = 0, ocur = 0;
    // output window: the unit's MS bytes wait in registers (32 bytes from output index wbase on, wfirst .. wend - 1 of them
    // the unit's) and go out together when the unit ends: a word stored on its own every fourth base is a line fill of its own
    // (the window's words live in LDS, word w of lane L at [w * 64 + L] of the wave's 2 KB: one ds_write when a word is
    // complete instead of a compare-and-select per window register)
    __shared__ uint32_t ow_lds[4 * 512];
    uint32_t *ow = ow_lds + (threadIdx.x >> 6) * 512u + lane;
    uint32_t wbase = 0, wfirst = 0, wend = 0;
    uint4 nu0 = make_uint4(0, 0, 0, 0), nq0 = make_uint4(0, 0, 0, 0), nq1 = make_uint4(0, 0, 0, 0);
    uint4 nu1 = make_uint4(0, 0, 0, 0);
    // CALL: the breakpoint scan of call_variants over what the unit walks (ms_walk_kernel's, walk_kernels.hip); when the
    // unit converges with breakpoints still waiting, the match that resolves them is the base where the depth on the
    // diagonal reaches the threshold - known without walking there (units are at least threshold + 1 bases apart)
    uint32_t lim = 0, ilen = 0, up0 = 0, dprev = 0, np = 0, pend0 = 0, pend1 = 0, pend2 = 0, pend3 = 0;
    uint32_t nrow = 0;
    bool want = true; // wants to claim a unit
    uint32_t st_units = 0, st_acc = 0, st_fail = 0, st_con = 0; // work counters (kPlanStat*)
#ifdef KBO_WALK_DEBUG
    uint32_t dbg_iter = 0, dbg_rare = 0, dbg_acc = 0, dbg_fail = 0, dbg_con = 0, dbg_wdone = 0, dbg_wfin = 0, dbg_units = 0,
             dbg_flagged = 0;
#endif

    for (;;) {
        // ============================== bookkeeping block ==============================
        {
#ifdef KBO_WALK_DEBUG
            dbg_rare++;
#endif
            // ---- the next unit's record is here: request its start row and first query block
            if (flags & G_PF) {
                const uint32_t pos = nu0.z & 0xFFFFu;
                nq0 = ld16u(qb, nu0.x + (pos & ~15u));
                nq1 = ld16u(qb, min(nu0.x + (pos & ~15u) + 16u, q_end)); // (with the first: one fill for the line they share)
                if (!((nu1.x >> 8) & kUnitHead)) nrow = *reinterpret_cast<const uint32_t *>(nodeb + (uint64_t)(nu0.y + pos - 1u) * 4u);
                flags = (flags & ~G_PF) | G_HAVE;
            }
            // ---- finished units (and full windows): the output bytes, whole words where the unit owns them
            {
                const bool fl = (flags & (G_DONE | G_FLUSH)) && wend != 0;
                if (__ballot(fl)) {
                    if (fl) {
                        uint8_t *o = a.d_out + (start + warm + wbase);
#pragma unroll
                        for (uint32_t w = 0; w < 8; w++) {
                            const uint32_t lo = max(wfirst, 4u * w), hi = min(wend, 4u * w + 4u); // bytes [lo, hi) of word w
                            if (lo < hi) {
                                const uint32_t v = ow[w * 64u];
                                if (lo == 4u * w && hi == 4u * w + 4u) st4u(o + 4u * w, v);
                                else {
#pragma unroll
                                    for (uint32_t t = 0; t < 4; t++)
                                        if (4u * w + t >= lo && 4u * w + t < hi) o[4u * w + t] = (uint8_t)(v >> (8u * t));
                                }
                            }
                        }
                        wbase += 32u;
                        wfirst = 0;
                        wend = 0;
                        flags &= ~G_FLUSH;
                    }
                }
            }
            // ---- switch finished lanes to their prefetched unit
            if (flags & G_DONE) {
                if (flags & G_HAVE) {
                    start = nu0.x;
                    i = nu0.z & 0xFFFFu;
                    out_from = nu0.z >> 16;
                    last_mm = (int32_t)(int16_t)(nu0.w & 0xFFFFu);
                    bound = nu0.w >> 16;
                    uflags = (nu1.x >> 8) & 0xFFu;
                    warm = (nu1.x >> 16) & 0xFFu;
                    item = nu1.y;
                    const bool head = (uflags & kUnitHead) != 0;
                    l = head ? 0u : nrow;
                    r = head ? n : nrow + 1u;
                    d = head ? 0u : (nu1.x & 0xFFu);
                    m = 0;
                    qblk = nq0;
                    qnxt = nq1;
                    qcur = sel4(qblk, (i >> 2) & 3u);
                    const uint32_t c = decode_base((qcur >> ((i & 3u) * 8u)) & 0xFFu);
                    cb = c < 4u ? c * nblk : null_blk;
                    if (CALL) {
                        lim = nu1.z & 0xFFFFu;
                        ilen = nu1.z >> 16;
                        up0 = nu0.y;
                        dprev = d;
                        np = 0;
                    }
                    ocur = 0;
                    wbase = (out_from - warm) & ~3u;
                    wfirst = (out_from - warm) & 3u; // first byte of the first output word that is this unit's
                    flags = i < bound ? 0u : G_DONE; // (empty units: see plan_emit_kernel)
                    if (STATS) st_units += i < bound ? 1u : 0u;
                    want = true;
#ifdef KBO_WALK_DEBUG
                    dbg_units++;
#endif
                } else {
                    flags = ((flags & G_PF) || want) ? flags : G_FIN;
                }
            }
            // ---- claim units for the lanes that have none in the pipeline
            {
                const uint64_t wmask = __ballot(want);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
                const uint32_t n_want = (uint32_t)__popcll(wmask);
                const uint32_t avail = pool_end - pool_next;
                uint32_t mine = pool_next + rank, lim = pool_end;
                if (n_want > avail && !drained) { // (wave-uniform) the chunk runs out: take the next one off the queue
                    uint32_t nb = 0;
                    if (lane == 0) nb = atomicAdd(a.qctl, 64u);
                    nb = min(__shfl(nb, 0), q_total);
                    drained = nb >= q_total;
                    const uint32_t ne = min(nb + 64u, q_total);
                    if (rank >= avail) {
                        mine = nb + (rank - avail);
                        lim = ne;
                    }
                    pool_next = min(nb + (n_want - avail), ne);
                    pool_end = ne;
                } else {
                    pool_next = min(pool_end, pool_next + n_want);
                }
                if (want && mine < lim) {
                    nu0 = ld16(utb, mine * 32u);
                    __builtin_memcpy(&nu1, utb + (size_t)mine * 32u + 16u, 16);
                    flags |= G_PF;
                }
                want = false;
            }
            if (__ballot(flags != G_FIN) == 0) break;
        }

#pragma unroll 1
        for (uint32_t it = 0; it < a.rare_period; it++) {
#ifdef KBO_WALK_DEBUG
            dbg_iter++;
            dbg_wdone += (flags & G_DONE) ? 1u : 0u;
            dbg_wfin += (flags & G_FIN) ? 1u : 0u;
#endif
            if (!(flags & G_BLOCKED)) {
                const bool con = (flags & G_CON) != 0;
                const uint32_t bl = div96(l), br = div96(r);
                const uint32_t bmask = cb == null_blk ? 0u : ~0u;
                const uint32_t rkA = (cb + (bl & bmask)) << 4, rkB = (cb + (br & bmask)) << 4;
                uint4 xA, xB;
                if (BIG) { // entries live in their own region, 64-bit offsets
                    const uint8_t *pA = con ? a.ix.ent + (uint64_t)l * 12u : arena + rkA;
                    const uint8_t *pB = con ? a.ix.ent + (uint64_t)r * 12u : arena + rkB;
                    __builtin_memcpy(&xA, pA, 16);
                    __builtin_memcpy(&xB, pB, 16);
                } else {
                    const uint32_t enA = ent_byte0 + ((l + (l << 1)) << 2), enB = ent_byte0 + ((r + (r << 1)) << 2);
                    xA = ld16u(arena, con ? enA : rkA);
                    xB = ld16u(arena, con ? enB : rkB);
                }
                if (flags & G_QF) { // the query block after the current one (reads <= 16 bytes past the item)
                    qnxt = ld16u(qb, min(start + (i & ~15u) + 16u, q_end));
                    flags &= ~G_QF;
                }
                // ---- contracting lanes: one level up the LCS interval tree
                const uint32_t lv = max(xA.x, xB.x);
                const bool root = lv == 0;
                const uint32_t cl = root ? 0u : (xA.x == lv ? xA.y : l);
                const uint32_t cr = root ? n : (xB.x == lv ? xB.z : r);
                const bool cstop = root || !m || cl <= tgt_l || cr >= tgt_r;
                // ---- extending lanes
                const uint32_t ol = l - bl * kRankRows, orr = r - br * kRankRows;
                const uint32_t l2 = rank_eval(xA, ol), r2 = rank_eval(xB, orr);
                const bool ok = !con && l2 < r2;
                const bool accept = !con && (l2 < r2 || d == 0);
                const bool fail = !con && !accept;
                if (STATS) {
                    st_con += con ? 1u : 0u;
                    st_acc += accept ? 1u : 0u;
                    st_fail += fail ? 1u : 0u;
                }
#ifdef KBO_WALK_DEBUG
                dbg_con += con ? 1u : 0u;
                dbg_acc += accept ? 1u : 0u;
                dbg_fail += fail ? 1u : 0u;
#endif
                uint32_t dl = 0, dr = 0;
                {
                    const uint32_t wsel = ol >> 5, pb = ol & 31u;
                    const uint32_t W = wsel == 0 ? xA.y : (wsel == 1 ? xA.z : xA.w);
                    const uint32_t below = W & ((1u << pb) - 1u);
                    dl = below ? pb - (31u - (uint32_t)__clz((int)below)) : 0u;
                }
                {
                    const uint32_t wsel = orr >> 5, pb = orr & 31u;
                    const uint32_t W = wsel == 0 ? xB.y : (wsel == 1 ? xB.z : xB.w);
                    const uint32_t above = W & (~0u << pb);
                    dr = above ? (uint32_t)__ffs((int)above) - pb : 0u;
                }
                m = fail ? ((dl && dr) ? 1u : 0u) : m;
                tgt_l = fail ? l - dl : tgt_l;
                tgt_r = fail ? r + dr : tgt_r;
                l = con ? cl : (ok ? l2 : l);
                r = con ? cr : (ok ? r2 : r);
                d = con ? lv : (ok ? min(d + 1u, k) : d);
                flags = (con && cstop) ? (flags & ~G_CON) : (fail ? (flags | G_CON) : flags);
                if (accept) {
                    bool fin = i + 1u == bound;
                    const uint32_t e = i - warm; // output index (wraps below warm; only its low bits are used then)
                    // converged: the walk is provably back on the diagonal (see the header); the unit ends here
                    const bool conv = !(uflags & kUnitPlain) && (int32_t)i >= last_mm && r == l + 1u &&
                                      d == min((uint32_t)((int32_t)i - last_mm), k);
                    if (CALL) {
                        const uint32_t thr = a.call_thr, seg = (((blockIdx.x * blockDim.x + threadIdx.x) >> 6)) % kCallSegs;
                        if (i >= out_from && i > 0 && i < lim && d < dprev && dprev >= thr && d < thr) { // a breakpoint this unit owns
                            while (np && pend0 + k < i) { pend0 = pend1; pend1 = pend2; pend2 = pend3; np--; }
                            if (np == 4u) atomicAdd(a.call_counts + 16u * kCallSegs, 1u);
                            else {
                                if (np == 0) pend0 = i; else if (np == 1) pend1 = i; else if (np == 2) pend2 = i; else pend3 = i;
                                np++;
                            }
                        } else if (np && d >= thr && r == l + 1u) { // the first unique match to the right of the waiting ones
                            for (uint32_t x = 0; x < np; x++) {
                                const uint32_t bp = x == 0 ? pend0 : x == 1 ? pend1 : x == 2 ? pend2 : pend3;
                                if (i <= bp + k) {
                                    const uint32_t slot = atomicAdd(a.call_counts + seg * 16u, 1u);
                                    if (slot < a.call_cap) a.call_sites[(size_t)seg * a.call_cap + slot] = make_uint4(start + bp, start + i, l, item + 1u);
                                }
                            }
                            np = 0;
                        }
                        dprev = d;
                        if (conv && np) { // back on the diagonal: the depth reaches thr at js, on the diagonal's node there
                            const uint32_t js = (uint32_t)(last_mm + (int32_t)thr);
                            for (uint32_t x = 0; x < np; x++) {
                                const uint32_t bp = x == 0 ? pend0 : x == 1 ? pend1 : x == 2 ? pend2 : pend3;
                                if (js < ilen && js <= bp + k) {
                                    const uint32_t slot = atomicAdd(a.call_counts + seg * 16u, 1u);
                                    if (slot < a.call_cap) // (row = node_at[text position]: filled in by call_fix_sites_kernel)
                                        a.call_sites[(size_t)seg * a.call_cap + slot] = make_uint4(start + bp, start + js, up0 + js, (item + 1u) | 0x80000000u);
                                }
                            }
                            np = 0;
                        }
                        // a chunk of an item without a plan: done with its own bases once no breakpoint waits any more
                        fin = fin || ((uflags & kUnitPlain) && i + 1u >= lim && np == 0u);
                    }
                    const bool word_done = (e & 3u) == 3u || fin || conv;
                    if (i >= out_from && (!CALL || i < lim)) { // (call mode: the bases borrowed from the next chunk are that chunk's to write)
                        const uint32_t wi = e - wbase; // 0 .. 31
                        ocur |= d << ((e & 3u) * 8u);
                        wend = wi + 1u;
                        if (word_done) {
                            ow[(wi >> 2) * 64u] = ocur;
                            ocur = 0;
                        }
                        flags |= (wi == 31u && !(fin || conv)) ? G_FLUSH : 0u; // (long units: the window goes out in between)
                    }
                    if (fin && !conv && !(uflags & (kUnitPlain | kUnitToEnd))) { // reached the next group unconverged
                        a.redo[item] = 1;
#ifdef KBO_WALK_DEBUG
                        dbg_flagged++;
#endif
                    }
                    i++;
                    const bool newblk = (i & 15u) == 0;
                    qblk.x = newblk ? qnxt.x : qblk.x;
                    qblk.y = newblk ? qnxt.y : qblk.y;
                    qblk.z = newblk ? qnxt.z : qblk.z;
                    qblk.w = newblk ? qnxt.w : qblk.w;
                    flags |= (fin || conv) ? G_DONE : (newblk ? G_QF : 0u);
                    qcur = sel4(qblk, (i >> 2) & 3u);
                    const uint32_t c = decode_base((qcur >> ((i & 3u) * 8u)) & 0xFFu);
                    cb = c < 4u ? c * nblk : null_blk;
                }
            }
        } // hot loop
    }
    if (STATS) plan_stats_add(a.pstats, kPlanStatUnits, st_units, kPlanStatAccepted, st_acc, kPlanStatFailed, st_fail, kPlanStatLevels, st_con);
#ifdef KBO_WALK_DEBUG
    if (a.lo_out == nullptr && a.hi_out != nullptr) { // debug build: hi_out doubles as the counter sink
        if (lane == 0) {
            atomicAdd(a.hi_out + 0, dbg_iter);
            atomicAdd(a.hi_out + 1, dbg_rare);
            atomicAdd(a.hi_out + 3, 1u);
            atomicMax(a.hi_out + 13, dbg_iter);
        }
        atomicAdd(a.hi_out + 4, dbg_acc);
        atomicAdd(a.hi_out + 5, dbg_fail);
        atomicAdd(a.hi_out + 6, dbg_con);
        atomicAdd(a.hi_out + 7, dbg_flagged);
        atomicAdd(a.hi_out + 8, dbg_wdone);
        atomicAdd(a.hi_out + 9, dbg_wfin);
        atomicAdd(a.hi_out + 12, dbg_units);
    }
#endif
}

// x < y in every byte (yb = y in all four bytes), as bits 0..3
__device__ __forceinline__ uint32_t lt4(uint32_t x, uint32_t yb)
{
    const uint32_t H = 0x80808080u;
    const uint32_t t = (x | H) - (yb & ~H);               // bit 7 of a byte: its low seven bits are >= y's
    const uint32_t lt = ((~x & yb) | (~(x ^ yb) & ~t)) & H; // top bit decides, else the low bits
    return ((lt >> 7) * 0x01020408u) >> 24;
}

// the same for values below 128 on both sides (k <= 127: every LCS value and level)
__device__ __forceinline__ uint32_t lt4_7(uint32_t x, uint32_t yb)
{
    const uint32_t H = 0x80808080u;
    const uint32_t lt = ~((x | H) - yb) & H;
    return ((lt >> 7) * 0x01020408u) >> 24;
}

// -------------------------------------------------------------------------------------------------------------
// ms_walk_recovery_kernel: ms_walk_guided_kernel over the recovery lines (sbwt_index.hpp) instead of the rank blocks
// and contraction entries.  A unit is the stretch behind a mismatch: nearly every base lands on a row that is random
// with respect to the last one, fails to extend about every other time and then needs the LCS values around its
// rows - with rank blocks and entries two line fills per failing base, here the line of the failed extension holds
// them.  Contraction: lv = max(LCS[l], LCS[r]); the level's ends are the previous value < lv to the left of l and
// the next one to the right of r, searched in the 16 values [.., l] and [r, ..] of the line(s); when a window ends
// first (end of the line, long run of equal suffixes) the level is taken from the {lcs, psv, nsv} entries in the
// next iteration instead.  Everything else is ms_walk_guided_kernel.
template <bool BIG, bool CALL, bool K7, bool STATS>
__global__ __launch_bounds__(256) void ms_walk_recovery_kernel(WalkArgs a)
{
    const uint32_t n = a.ix.n, k = a.ix.k;
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t *arena = reinterpret_cast<const uint8_t *>(a.ix.arena);
    const uint8_t *qb = a.q;
    const uint8_t *utb = reinterpret_cast<const uint8_t *>(a.units);
    const uint8_t *nodeb = reinterpret_cast<const uint8_t *>(a.ix.pc_node);
    const uint32_t q_end = (uint32_t)a.q_bytes;
    const uint8_t *fat = a.ix.fat;
    const uint32_t null_line = a.ix.fat_null;
    const uint32_t ent_byte0 = a.ix.lcs_off << 4;

    const uint32_t u_total = a.usums[(2u * a.n_items) / kScanBlock] + a.ucount[2u * a.n_items];
    const uint32_t q_total = u_total > a.unit_bail ? 0u : min(u_total, a.unit_cap); // (see plan_emit_kernel)
    uint32_t pool_next = 0, pool_end = 0;
    bool drained = false; // (wave-uniform) the queue has nothing left

    uint32_t flags = G_DONE;
    uint32_t l = 0, r = n, d = 0, cb = 0;
    uint32_t i = 0, start = 0, warm = 0, bound = 0, out_from = 0, uflags = 0, item = 0;
    int32_t last_mm = -1;
    uint4 qblk = make_uint4(0, 0, 0, 0), qnxt = make_uint4(0, 0, 0, 0);
    uint32_t qcur = 0, ocur = 0;
    // output window: the unit's MS bytes wait in registers (32 bytes from output index wbase on, wfirst .. wend - 1 of them
    // the unit's) and go out together when the unit ends: a word stored on its own every fourth base is a line fill of its own
    // (the window's words live in LDS, word w of lane L at [w * 64 + L] of the wave's 2 KB: one ds_write when a word is
    // complete instead of a compare-and-select per window register)
    __shared__ uint32_t ow_lds[4 * 512];
    uint32_t *ow = ow_lds + (threadIdx.x >> 6) * 512u + lane;
    uint32_t wbase = 0, wfirst = 0, wend = 0;
    uint4 nu0 = make_uint4(0, 0, 0, 0), nq0 = make_uint4(0, 0, 0, 0), nq1 = make_uint4(0, 0, 0, 0);
    uint4 nu1 = make_uint4(0, 0, 0, 0);
    // CALL: the breakpoint scan of call_variants over what the unit walks (ms_walk_kernel's, walk_kernels.hip); when the
    // unit converges with breakpoints still waiting, the match that resolves them is the base where the depth on the
    // diagonal reaches the threshold - known without walking there (units are at least threshold + 1 bases apart)
    uint32_t lim = 0, ilen = 0, up0 = 0, dprev = 0, np = 0, pend0 = 0, pend1 = 0, pend2 = 0, pend3 = 0;
    uint32_t nrow = 0;
    bool want = true; // wants to claim a unit
    uint32_t visits = 0;
    uint32_t st_units = 0, st_acc = 0, st_fail = 0, st_con = 0, st_ent = 0; // work counters (kPlanStat*)
#ifdef KBO_WALK_DEBUG
    uint32_t dbg_iter = 0, dbg_rare = 0, dbg_acc = 0, dbg_fail = 0, dbg_con = 0, dbg_wdone = 0, dbg_wfin = 0, dbg_units = 0,
             dbg_flagged = 0, dbg_short = 0;
#endif

    for (;;) {
        // ============================== bookkeeping block ==============================
        {
#ifdef KBO_WALK_DEBUG
            dbg_rare++;
#endif
            // ---- the next unit's record is here: request its start row and first query block
            if (flags & G_PF) {
                const uint32_t pos = nu0.z & 0xFFFFu;
                nq0 = ld16u(qb, nu0.x + (pos & ~15u));
                nq1 = ld16u(qb, min(nu0.x + (pos & ~15u) + 16u, q_end)); // (with the first: one fill for the line they share)
                if (!((nu1.x >> 8) & kUnitHead)) nrow = *reinterpret_cast<const uint32_t *>(nodeb + (uint64_t)(nu0.y + pos - 1u) * 4u);
                flags = (flags & ~G_PF) | G_HAVE;
            }
            // ---- finished units (and full windows): the output bytes, whole words where the unit owns them
            {
                const bool fl = (flags & (G_DONE | G_FLUSH)) && wend != 0;
                if (__ballot(fl)) {
                    if (fl) {
                        uint8_t *o = a.d_out + (start + warm + wbase);
#pragma unroll
                        for (uint32_t w = 0; w < 8; w++) {
                            const uint32_t lo = max(wfirst, 4u * w), hi = min(wend, 4u * w + 4u); // bytes [lo, hi) of word w
                            if (lo < hi) {
                                const uint32_t v = ow[w * 64u];
                                if (lo == 4u * w && hi == 4u * w + 4u) st4u(o + 4u * w, v);
                                else {
#pragma unroll
                                    for (uint32_t t = 0; t < 4; t++)
                                        if (4u * w + t >= lo && 4u * w + t < hi) o[4u * w + t] = (uint8_t)(v >> (8u * t));
                                }
                            }
                        }
                        wbase += 32u;
                        wfirst = 0;
                        wend = 0;
                        flags &= ~G_FLUSH;
                    }
                }
            }
            // ---- switch finished lanes to their prefetched unit
            if (flags & G_DONE) {
                if (flags & G_HAVE) {
                    start = nu0.x;
                    i = nu0.z & 0xFFFFu;
                    out_from = nu0.z >> 16;
                    last_mm = (int32_t)(int16_t)(nu0.w & 0xFFFFu);
                    bound = nu0.w >> 16;
                    uflags = (nu1.x >> 8) & 0xFFu;
                    warm = (nu1.x >> 16) & 0xFFu;
                    item = nu1.y;
                    const bool head = (uflags & kUnitHead) != 0;
                    l = head ? 0u : nrow;
                    r = head ? n : nrow + 1u;
                    d = head ? 0u : (nu1.x & 0xFFu);
                    qblk = nq0;
                    qnxt = nq1;
                    qcur = sel4(qblk, (i >> 2) & 3u);
                    const uint32_t c = decode_base((qcur >> ((i & 3u) * 8u)) & 0xFFu);
                    cb = c < 4u ? c << 4 : ~0u; // offset of the base's rank block inside a line (~0: no such base)
                    if (CALL) {
                        lim = nu1.z & 0xFFFFu;
                        ilen = nu1.z >> 16;
                        up0 = nu0.y;
                        dprev = d;
                        np = 0;
                    }
                    ocur = 0;
                    wbase = (out_from - warm) & ~3u;
                    wfirst = (out_from - warm) & 3u; // first byte of the first output word that is this unit's
                    flags = i < bound ? 0u : G_DONE; // (empty units: see plan_emit_kernel)
                    if (STATS) st_units += i < bound ? 1u : 0u;
                    want = true;
#ifdef KBO_WALK_DEBUG
                    dbg_units++;
#endif
                } else {
                    flags = ((flags & G_PF) || want) ? flags : G_FIN;
                }
            }
            // ---- claim units for the lanes that have none in the pipeline
            {
                const uint64_t wmask = __ballot(want);
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(wmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wmask, 0u));
                const uint32_t n_want = (uint32_t)__popcll(wmask);
                const uint32_t avail = pool_end - pool_next;
                uint32_t mine = pool_next + rank, lim = pool_end;
                if (n_want > avail && !drained) { // (wave-uniform) the chunk runs out: take the next one off the queue
                    uint32_t nb = 0;
                    if (lane == 0) nb = atomicAdd(a.qctl, 64u);
                    nb = min(__shfl(nb, 0), q_total);
                    drained = nb >= q_total;
                    const uint32_t ne = min(nb + 64u, q_total);
                    if (rank >= avail) {
                        mine = nb + (rank - avail);
                        lim = ne;
                    }
                    pool_next = min(nb + (n_want - avail), ne);
                    pool_end = ne;
                } else {
                    pool_next = min(pool_end, pool_next + n_want);
                }
                if (want && mine < lim) {
                    nu0 = ld16(utb, mine * 32u);
                    __builtin_memcpy(&nu1, utb + (size_t)mine * 32u + 16u, 16);
                    flags |= G_PF;
                }
                want = false;
            }
            if (__ballot(flags != G_FIN) == 0) break;
            if (++visits > (1u << 21)) { // (cannot happen: a wave sees a few thousand visits in the largest launch; ends a
                if (lane == 0) a.qctl[3] = 1; // walk that does not make progress instead of hanging the device)
                break;
            }
        }

#pragma unroll 1
        for (uint32_t it = 0; it < a.rare_period; it++) {
#ifdef KBO_WALK_DEBUG
            dbg_iter++;
            dbg_wdone += (flags & G_DONE) ? 1u : 0u;
            dbg_wfin += (flags & G_FIN) ? 1u : 0u;
#endif
            if (!(flags & G_BLOCKED)) {
                const bool ent = (flags & G_ENT) != 0; // this iteration takes one contraction level from the entries
                const uint32_t bl = l >> 6, br = r >> 6, ol = l & 63u, orr = r & 63u;
                // 16-row windows of LCS values, [.., l] and [r, ..], cut at the line's ends
                const uint32_t wl = ol > 15u ? ol - 15u : 0u, wr = min(orr, 48u);
                const bool cnull = cb == ~0u;
                uint4 xA, xB, wA, wB;
                if (BIG) {
                    const uint64_t oA = (uint64_t)bl << 7, oB = (uint64_t)br << 7, oN = (uint64_t)null_line << 7;
                    __builtin_memcpy(&xA, fat + (cnull ? oN : oA + cb), 16);
                    __builtin_memcpy(&xB, fat + (cnull ? oN : oB + cb), 16);
                    __builtin_memcpy(&wA, fat + oA + 64u + wl, 16);
                    __builtin_memcpy(&wB, fat + oB + 64u + wr, 16);
                } else {
                    const uint32_t oA = bl << 7, oB = br << 7, oN = null_line << 7;
                    xA = ld16(fat, cnull ? oN : oA + cb);
                    xB = ld16(fat, cnull ? oN : oB + cb);
                    wA = ld16u(fat, oA + 64u + wl);
                    wB = ld16u(fat, oB + 64u + wr);
                }
                if (__ballot(ent)) { // (rare: the windows did not hold a level's ends)
                    if (ent) {
                        if (BIG) {
                            __builtin_memcpy(&wA, a.ix.ent + (uint64_t)l * 12u, 16);
                            __builtin_memcpy(&wB, a.ix.ent + (uint64_t)r * 12u, 16);
                        } else {
                            wA = ld16u(arena, ent_byte0 + ((l + (l << 1)) << 2));
                            wB = ld16u(arena, ent_byte0 + ((r + (r << 1)) << 2));
                        }
                    }
                }
                if (flags & G_QF) { // the query block after the current one (reads <= 16 bytes past the item)
                    qnxt = ld16u(qb, min(start + (i & ~15u) + 16u, q_end));
                    flags &= ~G_QF;
                }
                // ---- the base's extension from [l, r); when it is empty: contraction levels out of the windows, the
                // extension tried again after each from the two rank blocks that are already here (a level found in
                // the windows ends inside the two lines)
                const uint64_t WA = ((uint64_t)xA.z << 32) | xA.y, WB = ((uint64_t)xB.z << 32) | xB.y;
                uint32_t l2 = xA.x + (uint32_t)__popcll(WA & ((1ull << ol) - 1ull));
                uint32_t r2 = xB.x + (uint32_t)__popcll(WB & ((1ull << orr) - 1ull));
                bool ok = !ent && l2 < r2;
                bool short_win = false; // the windows end before the level does
                if (STATS) {
                    st_fail += (!ent && !ok && d != 0) ? 1u : 0u;
                    st_ent += ent ? 1u : 0u;
                }
#ifdef KBO_WALK_DEBUG
                dbg_fail += (!ent && !ok && d != 0) ? 1u : 0u;
#endif
                if (ent) { // one level from the entries; the extension comes with the next iteration (other lines)
                    const uint32_t lve = max(wA.x, wB.x);
                    const bool roote = lve == 0;
                    const uint32_t cle = roote ? 0u : (wA.x == lve ? wA.y : l), cre = roote ? n : (wB.x == lve ? wB.z : r);
                    l = cle;
                    r = cre;
                    d = lve;
                    flags &= ~G_ENT;
                }
#pragma unroll 1
                for (uint32_t lev = 0; lev < 4u; lev++) {
                    const bool need = !ent && !ok && d != 0 && !short_win;
                    if (__ballot(need) == 0) break;
                    if (need) {
                        const uint32_t pl = l - (bl << 6) - wl, pr = r - (br << 6) - wr; // 0..15: l and r stay in the windows
                        const uint32_t lcs_l = (sel4(wA, pl >> 2) >> ((pl & 3u) * 8u)) & 0xFFu;
                        const uint32_t lcs_r = (sel4(wB, pr >> 2) >> ((pr & 3u) * 8u)) & 0xFFu;
                        const uint32_t lvw = max(lcs_l, lcs_r), yb = lvw * 0x01010101u;
                        const uint32_t mA = K7 ? lt4_7(wA.x, yb) | (lt4_7(wA.y, yb) << 4) | (lt4_7(wA.z, yb) << 8) | (lt4_7(wA.w, yb) << 12)
                                               : lt4(wA.x, yb) | (lt4(wA.y, yb) << 4) | (lt4(wA.z, yb) << 8) | (lt4(wA.w, yb) << 12);
                        const uint32_t mB = K7 ? lt4_7(wB.x, yb) | (lt4_7(wB.y, yb) << 4) | (lt4_7(wB.z, yb) << 8) | (lt4_7(wB.w, yb) << 12)
                                               : lt4(wB.x, yb) | (lt4(wB.y, yb) << 4) | (lt4(wB.z, yb) << 8) | (lt4(wB.w, yb) << 12);
                        const uint32_t below = mA & ((1u << pl) - 1u), above = mB & ~((2u << pr) - 1u);
                        const bool need_l = lcs_l == lvw, need_r = lcs_r == lvw;
                        if (lvw == 0) { // the root: its extension is [C[c], C[c+1])
                            const uint32_t c = cb >> 4;
                            l = 0;
                            r = n;
                            d = 0;
                            l2 = c == 0 ? a.ix.C[0] : c == 1 ? a.ix.C[1] : c == 2 ? a.ix.C[2] : c == 3 ? a.ix.C[3] : 0u;
                            r2 = c == 0 ? a.ix.C[1] : c == 1 ? a.ix.C[2] : c == 2 ? a.ix.C[3] : c == 3 ? a.ix.C[4] : 0u;
                            ok = l2 < r2;
                        } else if ((need_l && !below) || (need_r && !above)) {
                            short_win = true;
                        } else {
                            l = need_l ? (bl << 6) + wl + (31u - (uint32_t)__clz((int)below)) : l;
                            r = need_r ? (br << 6) + wr + (uint32_t)__ffs((int)above) - 1u : r;
                            d = lvw;
                            l2 = xA.x + (uint32_t)__popcll(WA & ((1ull << (l - (bl << 6))) - 1ull));
                            r2 = xB.x + (uint32_t)__popcll(WB & ((1ull << (r - (br << 6))) - 1ull));
                            ok = l2 < r2;
                        }
                        if (STATS) st_con++;
#ifdef KBO_WALK_DEBUG
                        dbg_con++;
                        dbg_short += short_win ? 1u : 0u;
#endif
                    }
                }
                // accepted: the extension exists, or the walk is at the root (a base without an edge there keeps d = 0)
                const bool accept = !ent && (ok || d == 0);
                flags |= short_win ? G_ENT : 0u;
                l = ok ? l2 : l;
                r = ok ? r2 : r;
                d = ok ? min(d + 1u, k) : d;
                if (STATS) st_acc += accept ? 1u : 0u;
#ifdef KBO_WALK_DEBUG
                dbg_acc += accept ? 1u : 0u;
#endif
                if (accept) {
                    bool fin = i + 1u == bound;
                    const uint32_t e = i - warm; // output index (wraps below warm; only its low bits are used then)
                    // converged: the walk is provably back on the diagonal (see the header); the unit ends here
                    const bool conv = !(uflags & kUnitPlain) && (int32_t)i >= last_mm && r == l + 1u &&
                                      d == min((uint32_t)((int32_t)i - last_mm), k);
                    if (CALL) {
                        const uint32_t thr = a.call_thr, seg = (((blockIdx.x * blockDim.x + threadIdx.x) >> 6)) % kCallSegs;
                        if (i >= out_from && i > 0 && i < lim && d < dprev && dprev >= thr && d < thr) { // a breakpoint this unit owns
                            while (np && pend0 + k < i) { pend0 = pend1; pend1 = pend2; pend2 = pend3; np--; }
                            if (np == 4u) atomicAdd(a.call_counts + 16u * kCallSegs, 1u);
                            else {
                                if (np == 0) pend0 = i; else if (np == 1) pend1 = i; else if (np == 2) pend2 = i; else pend3 = i;
                                np++;
                            }
                        } else if (np && d >= thr && r == l + 1u) { // the first unique match to the right of the waiting ones
                            for (uint32_t x = 0; x < np; x++) {
                                const uint32_t bp = x == 0 ? pend0 : x == 1 ? pend1 : x == 2 ? pend2 : pend3;
                                if (i <= bp + k) {
                                    const uint32_t slot = atomicAdd(a.call_counts + seg * 16u, 1u);
                                    if (slot < a.call_cap) a.call_sites[(size_t)seg * a.call_cap + slot] = make_uint4(start + bp, start + i, l, item + 1u);
                                }
                            }
                            np = 0;
                        }
                        dprev = d;
                        if (conv && np) { // back on the diagonal: the depth reaches thr at js, on the diagonal's node there
                            const uint32_t js = (uint32_t)(last_mm + (int32_t)thr);
                            for (uint32_t x = 0; x < np; x++) {
                                const uint32_t bp = x == 0 ? pend0 : x == 1 ? pend1 : x == 2 ? pend2 : pend3;
                                if (js < ilen && js <= bp + k) {
                                    const uint32_t slot = atomicAdd(a.call_counts + seg * 16u, 1u);
                                    if (slot < a.call_cap) // (row = node_at[text position]: filled in by call_fix_sites_kernel)
                                        a.call_sites[(size_t)seg * a.call_cap + slot] = make_uint4(start + bp, start + js, up0 + js, (item + 1u) | 0x80000000u);
                                }
                            }
                            np = 0;
                        }
                        // a chunk of an item without a plan: done with its own bases once no breakpoint waits any more
                        fin = fin || ((uflags & kUnitPlain) && i + 1u >= lim && np == 0u);
                    }
                    const bool word_done = (e & 3u) == 3u || fin || conv;
                    if (i >= out_from && (!CALL || i < lim)) { // (call mode: the bases borrowed from the next chunk are that chunk's to write)
                        const uint32_t wi = e - wbase; // 0 .. 31
                        ocur |= d << ((e & 3u) * 8u);
                        wend = wi + 1u;
                        if (word_done) {
                            ow[(wi >> 2) * 64u] = ocur;
                            ocur = 0;
                        }
                        flags |= (wi == 31u && !(fin || conv)) ? G_FLUSH : 0u; // (long units: the window goes out in between)
                    }
                    if (fin && !conv && !(uflags & (kUnitPlain | kUnitToEnd))) { // reached the next group unconverged
                        a.redo[item] = 1;
#ifdef KBO_WALK_DEBUG
                        dbg_flagged++;
#endif
                    }
                    i++;
                    const bool newblk = (i & 15u) == 0;
                    qblk.x = newblk ? qnxt.x : qblk.x;
                    qblk.y = newblk ? qnxt.y : qblk.y;
                    qblk.z = newblk ? qnxt.z : qblk.z;
                    qblk.w = newblk ? qnxt.w : qblk.w;
                    flags |= (fin || conv) ? G_DONE : (newblk ? G_QF : 0u);
                    qcur = sel4(qblk, (i >> 2) & 3u);
                    const uint32_t c = decode_base((qcur >> ((i & 3u) * 8u)) & 0xFFu);
                    cb = c < 4u ? c << 4 : ~0u;
                }
            }
        } // hot loop
    }
    if (STATS) {
        plan_stats_add(a.pstats, kPlanStatUnits, st_units, kPlanStatAccepted, st_acc, kPlanStatFailed, st_fail, kPlanStatLevels, st_con);
        plan_stats_add(a.pstats, kPlanStatEntryLevels, st_ent, 0, 0, 0, 0, 0, 0);
    }
#ifdef KBO_WALK_DEBUG
    if (a.lo_out == nullptr && a.hi_out != nullptr) { // debug build: hi_out doubles as the counter sink
        if (lane == 0) {
            atomicAdd(a.hi_out + 0, dbg_iter);
            atomicAdd(a.hi_out + 1, dbg_rare);
            atomicAdd(a.hi_out + 3, 1u);
            atomicMax(a.hi_out + 13, dbg_iter);
        }
        atomicAdd(a.hi_out + 4, dbg_acc);
        atomicAdd(a.hi_out + 5, dbg_fail);
        atomicAdd(a.hi_out + 6, dbg_con);
        atomicAdd(a.hi_out + 7, dbg_flagged);
        atomicAdd(a.hi_out + 8, dbg_wdone);
        atomicAdd(a.hi_out + 9, dbg_wfin);
        atomicAdd(a.hi_out + 12, dbg_units);
        atomicAdd(a.hi_out + 14, dbg_short);
    }
#endif
}

} // namespace

std::atomic<int> g_plan_dmin{0} /* 0: by index size */, g_plan_cap{64}, g_plan_gap{0} /* 0: by index size */, g_plan_chunk{32};
std::atomic<int> g_plan_stage{1};     // plan_kernel stages queries and predictions through LDS (0: experiments)
std::atomic<int> g_plan_bail_x16{50}; // give the plan up when there are more than this many units per 16 items
void set_plan_stage(int on) { g_plan_stage = on != 0; }
void set_plan_bail(int units_per_16_items) { g_plan_bail_x16 = std::max(0, units_per_16_items); }
void set_plan_params(int dmin, int cap, int gap, int chunk)
{
    if (dmin != 0) g_plan_dmin = std::max(0, dmin); // (< 0: back to the automatic choice)
    if (cap > 0) g_plan_cap = std::min((int)kPlanPad, cap); // the text is padded by kPlanPad >= cap bytes in front
    if (gap != 0) g_plan_gap = gap < 0 ? 0 : std::max(2, gap); // (< 0: back to the automatic choice)
    if (chunk > 0) g_plan_chunk = std::max(16, chunk);
}

// plan_kernel with the parameters of the launch filled into `a` (the later launches need them)
static hipError_t launch_plan_kernel(WalkArgs &a, hipStream_t stream)
{
    // seed depth: a single-row interval is trusted as the item's diagonal from log4(rows) + 3 bases on (measured: 14 on the
    // 5 Mbp index, 16 on the 100 Mbp one - 12 / 13 / 14 / 16 bases: 1.19 / 1.16 / 1.14 / 1.18 ms, 13 / 14 / 15 / 16 / 18:
    // 4.10 / 3.80 / 3.58 / 3.56 / 3.69 ms; shallower seeds put items on wrong diagonals, deeper ones cost extensions)
    const int dmin_set = g_plan_dmin.load();
    a.plan_dmin = dmin_set > 0 ? (uint32_t)dmin_set : (uint32_t)(std::lround(std::log2((double)std::max<uint32_t>(a.ix.n, 4u)) / 2.0) + 3);
    a.plan_cap = (uint32_t)g_plan_cap.load();
    // mismatches closer than this share a unit: a unit needs about log4(rows) + 2 bases behind its last mismatch to converge,
    // and one that has not when the next begins sends its read to the redo pass (5 Mbp index: 20; 100 Mbp, A1 per 3 M reads:
    // 18 / 20 / 22 / 24 -> 3.54 / 3.51 / 3.36 / 3.36 ms)
    const int gap_set = g_plan_gap.load();
    a.plan_gap = gap_set > 0 ? (uint32_t)gap_set : (uint32_t)(std::lround(std::log2((double)std::max<uint32_t>(a.ix.n, 4u)) / 2.0) + 9);
    if (a.call_sites) a.plan_gap = std::max(a.plan_gap, a.call_thr + 1u); // (a unit resolves its breakpoints before the next one starts)
    a.plan_chunk = (uint32_t)g_plan_chunk.load();
    // mismatches an item's list holds: 13 for reads (more than that on 150 bases is a wrong diagonal), 29 for the chunks of
    // long sequences (800 bases at 1 % substitutions exceed 13 every twentieth time)
    a.plan_list = (a.max_item_len != 0 && a.max_item_len <= 255u) ? kPlanList : kPlanListMax;
    // (queue head, redo count, flags and - behind them - the launch's work counters)
    const hipError_t e = hipMemsetAsync(a.qctl, 0, 64 + kPlanStatSlots * kPlanStatWords * 4, stream);
    if (e != hipSuccess) return e;
    // LDS for the staged stretch of every wave: 64 items of at most max_item_len bases (not known, or too long for
    // four waves to share 64 KiB: no staging)
    // items that cannot be staged: 10 KB per wave for the transposed write-out of a step's predicted values
    uint32_t wave_lds = 64u * 16u * (uint32_t)kPlanStep, stage_ok = 0, stage_bytes = 0;
    bool fuse = false;
    static const int env_stage = std::getenv("KBO_PLAN_STAGE") ? std::atoi(std::getenv("KBO_PLAN_STAGE")) : -1; // experiments
    static const int env_fuse = std::getenv("KBO_PLAN_FUSE") ? std::atoi(std::getenv("KBO_PLAN_FUSE")) : 1;      // experiments
    if (a.max_item_len != 0 && (env_stage >= 0 ? env_stage != 0 : g_plan_stage.load() != 0)) {
        const uint64_t need = (64ull * a.max_item_len + 16u + kPlanLdsSlack + 15u) / 16u * 16u;
        if (need <= 16384u) {
            wave_lds = std::max<uint32_t>(wave_lds, (uint32_t)need);
            stage_ok = 1;
            // table mode, reads: queries and predictions side by side, the look-ups in this kernel (two workgroups of four
            // waves still share a CU's 160 KB)
            if (a.table_mode && !a.call_sites && env_fuse != 0 && a.max_item_len <= 16u * (uint32_t)kPlanStep && a.ix.dtab_order <= 17u) {
                fuse = true;
                stage_bytes = (uint32_t)need;
                wave_lds = stage_bytes + kPlanPackBytes + 1024u;
            }
        }
    }
    if (env_stage == 0) wave_lds = 0; // (experiments: neither staging nor the transposed write-out)
    static const int env_blk = std::getenv("KBO_PLAN_BLOCK") ? std::atoi(std::getenv("KBO_PLAN_BLOCK")) : 0; // experiments
    // (fused: one wave a workgroup - the LDS of a CU then holds eleven of them instead of two workgroups of four)
    const uint32_t bt = env_blk == 64 || env_blk == 128 || env_blk == 256 ? (uint32_t)env_blk : (fuse ? 64u : 256u);
    a.table_fused = fuse ? 1u : 0u;
    if (fuse)
        if (a.ix.dtab_order <= 15u)
            hipLaunchKernelGGL((plan_kernel<true, 16>), dim3((a.n_items + bt - 1u) / bt), dim3(bt), (bt / 64u) * wave_lds, stream, a, wave_lds, stage_ok, stage_bytes);
        else
            hipLaunchKernelGGL((plan_kernel<true, 18>), dim3((a.n_items + bt - 1u) / bt), dim3(bt), (bt / 64u) * wave_lds, stream, a, wave_lds, stage_ok, stage_bytes);
    else
        hipLaunchKernelGGL((plan_kernel<false, 16>), dim3((a.n_items + bt - 1u) / bt), dim3(bt), (bt / 64u) * wave_lds, stream, a, wave_lds, stage_ok, 0u);
    return hipGetLastError();
}

// plan -> unit counts -> scan -> units (the guided walk and the redo pass are launched by launch_ms_walk)
hipError_t launch_plan(WalkArgs &a, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    a.table_mode = 0;
    // (per read of 150 bases: chunks of long sequences hold several reads' worth of units)
    a.unit_bail = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(a.n_items, a.q_bytes / 150u) * (uint64_t)g_plan_bail_x16.load() / 16u + 64u, 0xFFFFFFFFu);
    const hipError_t e = launch_plan_kernel(a, stream);
    if (e != hipSuccess) return e;
    const uint32_t nb = (a.n_items + 255u) / 256u;
    hipLaunchKernelGGL(plan_count_kernel, dim3(nb), dim3(256), 0, stream, a);
    const hipError_t es = launch_scan(a.ucount, 2u * a.n_items + 1u, a.usums, stream);
    if (es != hipSuccess) return es;
    hipLaunchKernelGGL(plan_emit_kernel, dim3(nb), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// table mode (dtab_kernels.hip): plan -> the stretches behind the mismatches from the depth table -> the items it could not
// resolve listed for the plain kernel.  A launch that leaves more than half of its items unresolved gives the plan up (every
// item is walked plainly, and the host skips planning for the next launches: plan_after_launch).
hipError_t launch_plan_table(WalkArgs &a, hipStream_t stream)
{
    if (a.n_items == 0) return hipSuccess;
    a.table_mode = 1;
    a.unit_bail = a.n_items / 2u + 64u;
    static const int env_piece = std::getenv("KBO_REDO_PIECE") ? std::atoi(std::getenv("KBO_REDO_PIECE")) : 0; // experiments
    a.redo_piece = env_piece >= 4 ? (uint32_t)env_piece : kRedoPieceTable;
    hipError_t e = launch_plan_kernel(a, stream);
    if (e != hipSuccess) return e;
    if (!a.table_fused) { // (reads: plan_kernel has done the look-ups itself)
        e = launch_dtab_resolve(a, stream);
        if (e != hipSuccess) return e;
    } else if (a.ix.anchor) { // ... and left the stretches with bases deeper than the table knows to the anchors
        e = launch_dtab_stretches(a, (a.n_items + 63u) / 64u, stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(redo_collect_kernel, dim3((a.n_items + kRedoBlock - 1u) / kRedoBlock), dim3(kRedoBlock), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_redo_collect(const WalkArgs &a, hipStream_t stream)
{
    if (a.seq_off && a.table_mode && !a.call_sites) { // behind map_reads_kernel: the batch's sequences, sixteen flags a lane
        const uint32_t per = kRedoReadsBlock * 16u;
        hipLaunchKernelGGL(redo_collect_reads_kernel, dim3((a.n_items + per - 1u) / per), dim3(kRedoReadsBlock), 0, stream, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(redo_collect_kernel, dim3((a.n_items + kRedoBlock - 1u) / kRedoBlock), dim3(kRedoBlock), 0, stream, a);
    return hipGetLastError();
}

template <bool CALL, bool STATS> static void launch_guided_variant(const WalkArgs &a, uint32_t grid, uint32_t threads, hipStream_t stream)
{
    if (guided_uses_recovery_lines(a)) {
        if (a.ix.k <= 127u) {
            if (a.ix.big) hipLaunchKernelGGL((ms_walk_recovery_kernel<true, CALL, true, STATS>), dim3(grid), dim3(threads), 0, stream, a);
            else hipLaunchKernelGGL((ms_walk_recovery_kernel<false, CALL, true, STATS>), dim3(grid), dim3(threads), 0, stream, a);
        } else if (a.ix.big) hipLaunchKernelGGL((ms_walk_recovery_kernel<true, CALL, false, STATS>), dim3(grid), dim3(threads), 0, stream, a);
        else hipLaunchKernelGGL((ms_walk_recovery_kernel<false, CALL, false, STATS>), dim3(grid), dim3(threads), 0, stream, a);
    } else if (a.ix.big) hipLaunchKernelGGL((ms_walk_guided_kernel<true, CALL, STATS>), dim3(grid), dim3(threads), 0, stream, a);
    else hipLaunchKernelGGL((ms_walk_guided_kernel<false, CALL, STATS>), dim3(grid), dim3(threads), 0, stream, a);
}

hipError_t launch_ms_walk_guided(WalkArgs a, uint32_t grid, uint32_t threads, hipStream_t stream)
{
    // (the counting instantiations only when the launch was given a place for its counters: kbo_set_plan_stats)
    if (a.call_sites) {
        if (a.pstats) launch_guided_variant<true, true>(a, grid, threads, stream);
        else launch_guided_variant<true, false>(a, grid, threads, stream);
    } else if (a.pstats) launch_guided_variant<false, true>(a, grid, threads, stream);
    else launch_guided_variant<false, false>(a, grid, threads, stream);
    hipLaunchKernelGGL(redo_collect_kernel, dim3((a.n_items + kRedoBlock - 1u) / kRedoBlock), dim3(kRedoBlock), 0, stream, a);
    // call mode: sites of items that go to the redo pass are void (that pass finds them again), the others get their rows
    if (a.call_sites) hipLaunchKernelGGL(call_fix_sites_kernel, dim3(kCallSegs), dim3(256), 0, stream, a);
    return hipGetLastError();
}

} // namespace kbo
