// kernels.hpp — launch interface of the gfx950 kernels (walk_kernels.hip, derand_kernels.hip, rle_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace kbo {

// Device-resident index (32-bit positions).  Layout: sbwt_index.hpp.
struct DevIndexView {
    const uint4 *arena;   // one allocation: rank blocks of A,C,G,T, then the LCS windows
    uint32_t n_blocks;    // rank blocks per character (character c starts at c * n_blocks)
    uint32_t lcs_off;     // arena index (16-byte units) of contraction entry 0 (32-bit build)
    uint32_t pair_off;    // arena index (16-byte units) of the two-base extension blocks, 0 = none
    const uint8_t *ent;   // contraction entries as their own region (used when `big`)
    uint32_t big;         // 1: entries are addressed with 64-bit offsets (n_sets * 12 B >= 4 GiB)
    uint32_t n;           // n_sets
    uint32_t k;
    // path cover of the de Bruijn graph (sbwt_index.hpp PathCover), nullptr when the device copy has none:
    const uint8_t *pc_text;    // points at text position 0; kPlanPad zero bytes on either side
    const uint32_t *pc_pos;    // row -> text position
    const uint32_t *pc_node;   // text position -> row
    uint32_t C[5];             // C[c] of the index, C[4] = n_sets: extend(root, c) = [C[c], C[c+1])
    // plan_kernel's seed table: interval {l, r} of every string of seed_d bases (code = bases as 2-bit digits, first
    // base most significant), l >= r when it is no suffix of a row; nullptr / 0 when the index has none
    const uint2 *seed_tab;
    uint32_t seed_d;
    // recovery lines of the guided walk (sbwt_index.hpp make_recovery_lines), nullptr when the device copy has none
    const uint8_t *fat;
    uint32_t fat_null; // line index of an all-zero line (extensions by a non-ACGT byte)
    // depth table (dtab_kernels.hip), nullptr / 0 when the device copy has none: for every string of dtab_order bases (2-bit
    // digits, the newest base least significant) the longest suffix of it that is a suffix of a row; 0x80 | e when all of it
    // is, bit c of e = so is the string with base c in front of it
    const uint8_t *dtab;
    uint32_t dtab_order;
    uint32_t dtab_grouped; // 1: the entries of three consecutive bases share a 64-byte line (dtab_kernels.hip), 4^(order+1) bytes
    // anchors of the depth table, nullptr / 0 when there are none: open-addressing hash (2^anchor_bits slots) of the strings of
    // dtab_order bases that are the suffix of exactly ONE row: slot = (low 32 bits of key + 1) << 32 | text position of that
    // row in the path cover.  A base whose value the table cannot tell (deeper than dtab_order) is read off the path-cover
    // text in front of that position
    const uint64_t *anchor;
    uint32_t anchor_bits;
    // what map_reads_kernel (map_kernels.hip) reads instead of the byte text and the interval table, nullptr when the copy has
    // none: pc_tm[u] = { 2-bit digits of text positions [16 u - kMapPad, + 16), first one most significant; 01 at every
    // position that matches nothing (path start, padding) }; seed_pos[key of seed_d bases] = text position of the first row
    // whose k-mer ends with them, 0xFFFFFFFF when there is none
    const uint2 *pc_tm;
    const uint32_t *seed_pos;
    // ... and its filter in front of the depth table (small indexes): bit [key of dfilt_bases bases] set where that string is a
    // suffix of a row - 4^dfilt_bases / 8 bytes (2 MB for 12 bases: stays in L2), nullptr when the copy has none
    const uint32_t *dfilt;
    uint32_t dfilt_bases;
};

// One unit of walk work: `len` bases starting at absolute offset `start` of the
// concatenated query buffer; the first `warm` of them only warm the state up (no
// output) — chunks of long sequences restart k-1 bases upstream.
struct alignas(16) WalkItem {
    uint64_t start;
    uint32_t len;
    uint32_t warm; // low 16 bits: warm-up bases; high 16 bits (call mode): bases at the end that are walked only to finish
                   // the search to the right of the breakpoints in front of them (they belong to the next chunk)
};

// ---- plan-guided walk (plan_kernels.hip) -------------------------------------------------------------------------
// plan_kernel finds, for every work item, a diagonal of the path-cover text (a short exact walk over the item's first
// bases to a single row u: diagonal p0 = pos[u] - j), compares the whole item with the text on that diagonal, writes the
// MS values this predicts - min(k, distance to the last mismatch) - and the list of mismatch positions.
// plan_emit_kernel turns the lists into UNITS: stretches that have to be walked, one per group of mismatches closer
// than `plan_gap`, starting on the diagonal in front of the group's first mismatch.  ms_walk_guided_kernel walks every
// unit until the walk itself proves it is back on the diagonal (a single-row interval whose depth equals the distance
// to the group's last mismatch); a unit that reaches the next group first flags its item, and flagged items are
// walked again in full by the plain kernel.  Every value that is not walked is exact: see path_cover.cpp.
constexpr uint32_t kPlanPad = 64;        // zero bytes in front of / behind the device text (== PathCover::kPad)
constexpr uint32_t kPlanList = 12;       // u16 list entries per item after the first (13 mismatch positions in all): reads
constexpr uint32_t kPlanListMax = 28;    // the same for long items (WalkArgs::plan_list); the list array is sized for it
constexpr uint32_t kPlanNone = 0xFF;     // GuidedItem::n_mm: no diagonal found, walk the item plainly
constexpr uint32_t kPlanInf = 0xFFFE;    // GuidedItem::mm0: no mismatch
struct alignas(16) GuidedItem {
    uint32_t start;   // absolute offset of the item's first base in the query buffer (< 4 GiB per launch)
    uint32_t p0;      // text position of item base 0 on the diagonal (mod 2^32)
    uint16_t len;     // bases, warm-up included
    uint16_t j_conv;  // > 0: the item's first j_conv bases match the diagonal and were walked exactly by plan_kernel
    uint16_t mm0;     // first mismatch position (kPlanInf: none)
    uint8_t warm;     // leading bases without output
    uint8_t n_mm;     // kPlanNone, or min(mismatches, 254); more than plan_list + 1: the list is incomplete
};
enum : uint32_t { kUnitHead = 1u, kUnitPlain = 2u, kUnitToEnd = 4u };
struct alignas(16) WalkUnit { // 32 bytes
    uint32_t start;    // as GuidedItem
    uint32_t p0;
    uint16_t pos;      // first base the unit walks (item-relative)
    uint16_t out_from; // first base it writes
    int16_t last_mm;   // last mismatch of its group (-1: none, head of an item without early mismatches)
    uint16_t bound;    // one past the last base it may walk: the next group's first mismatch, or the item's length
    uint8_t d_start;   // depth of the walk in front of `pos` (units that start on the diagonal)
    uint8_t flags;     // kUnitHead: starts at the root; kUnitPlain: no convergence test (chunk of an item without a
                       // plan: k-1 warm-up bases from the root); kUnitToEnd: bound is the item's end
    uint8_t warm;      // as GuidedItem (positions of the output words)
    uint8_t pad0;
    uint32_t item;     // work item it belongs to (for the redo flag)
    uint32_t lim_len;  // (call mode) bases of the item that are its own (the rest belong to the next chunk) | length << 16
    uint32_t pad1;
};

struct WalkArgs {
    DevIndexView ix;
    const uint8_t *q;      // concatenated queries (ASCII), 4-byte aligned
    uint64_t q_bytes;      // total bytes in q
    const WalkItem *items; // n_items
    uint32_t n_items;
    uint32_t rounds;       // items per lane (set by launch_ms_walk)
    uint32_t rare_period;  // hot-loop iterations between two visits of the rare block (set by launch_ms_walk)
    uint32_t pair_min_d;   // two-base steps only from matches at least this deep (set by launch_ms_walk)
    uint32_t lane_limit;   // plain kernel: lanes from this one on get no items (64 = all work; experiments)
    uint8_t *d_out;        // 1 byte per base, same indexing as q
    uint32_t *lo_out;      // optional (nullptr): interval start per base
    uint32_t *hi_out;      // optional: interval end per base
    // plan-guided walk (all or none): work buffers sized by plan_work_bytes(), see attach_plan()
    GuidedItem *gitems;    // nullptr: plain walk
    uint16_t *glist;       // plan_list entries per item
    uint32_t *ucount;      // units per item, heavy ones then light ones (2 n_items + 1 entries), scanned in place
    uint32_t *usums;       // block sums of that scan
    uint8_t *redo;         // per item: 1 = a unit could not vouch for its successor, walk the item again in full
    WalkUnit *units;       // unit_cap records (items whose units do not fit are flagged for the full walk instead)
    uint32_t unit_cap;
    uint32_t redo_cap;     // WalkItem records the unit array holds (the redo pass's list is built there)
    uint32_t unit_bail;    // more units than this in a launch: the plan is given up, every item takes the plain walk
    uint32_t *qctl;        // [0] queue head of the guided walk, [1] entries of the redo list, [2] plan given up, [3] a walk left through its guard,
                           // [4] (table mode) items the table could not resolve, [5] items without a plan, [6] entries of map_reads_kernel's list of the reads it left (finish_reads_kernel)
    uint32_t *pstats;      // work counters of the launch (kPlanStat*): kPlanStatSlots slots of 8 u32, summed by the host
    uint32_t plan_dmin;    // plan_kernel: a seed must be this deep (capped at k) before its row is trusted
    uint32_t plan_cap;     // plan_kernel: seed iterations before an item is given up as unplanned
    uint32_t plan_gap;     // plan_emit_kernel: mismatches closer than this share a unit (>= 2)
    uint32_t plan_chunk;   // plan_emit_kernel: bases per unit of an item without a plan
    uint32_t plan_list;    // mismatch-list entries per item (kPlanList or kPlanListMax, set by launch_plan)
    // call mode of the plain kernel (all or none; call_kernels.hip has the stand-alone scan): the breakpoint scan of
    // call_variants (variant_calling.rs:268-273) done by the walking lane itself, sites {first base of the item + i, .. + j,
    // row, 0} appended to kCallSegs lists of call_cap records each (counters 64 bytes apart)
    uint4 *call_sites;
    uint32_t *call_counts;
    uint32_t call_cap;     // records per list
    uint32_t call_thr;     // derandomisation threshold t of the predicate
    uint32_t table_mode;   // 1: the stretches behind mismatches come from the depth table (set by launch_ms_walk)
    uint32_t table_fused;  // 1: plan_kernel did the table look-ups itself (reads; set by launch_plan_table)
    uint32_t redo_piece;   // table mode: output bases per piece of a flagged item in the redo pass
    uint32_t max_item_len; // 0 = not known, else no item is longer than this (plan_kernel sizes its LDS staging from it)
    const uint32_t *n_items_dev; // plain kernel: nullptr, or where the number of items is (the redo pass: qctl + 1)
    // map_reads_kernel (map_kernels.hip): where the characters go, the derandomisation threshold, 1 = format::relative_to_ref
    // on the way out, 1 = the MS values go to d_out as well
    uint8_t *chars_out;
    uint32_t map_thr, map_fmt, map_want_ms;
    // ... its packed-native instantiations (pack_kernels.hip has the layout): the reads as 2-bit words - qp_wps words per read, or
    // 0 and the scanned words-per-read (qp_data / qp_sums) -, one byte per read that is non-zero where the read holds a byte that
    // is no base (nullptr: none does), and, when the characters leave packed as well, where their words go
    uint32_t *host_bailed;   // (map_reads_kernel's route: pinned host word redo_collect_kernel sets when the plan is given up - the copy's
                             // hold-off, DevCopy::PlanState::bailed - instead of an 8-byte copy behind every launch)
    uint32_t *run_counts;    // (kbo::find with max_gap_len == 0: the number of runs - maximal stretches without '-' - of every read the
                             // kernel finishes itself, or nullptr: format::run_lengths_gapped then needs no counting pass of its own)
    const uint64_t *seq_off; // (reads: the batch's offsets instead of the item list - item s is sequence s, whole: the list is then
                             // only made for the second pass)
    uint32_t uniform_len;    // (... and when every sequence has max_item_len bases - n_items * max_item_len == q_bytes says so without
                             // looking at an offset - read s starts at seq_off[0] + s * uniform_len: one dependent load less per wave)
    const uint32_t *qp;
    uint32_t qp_wps;
    const uint32_t *qp_data, *qp_sums;
    const uint8_t *qp_exc;
    uint32_t *packed_out;
};
// Work counters the plan-guided stage keeps about itself (one wave-level atomic per counter and wave, spread over slots):
// what the CPU model of the stage (oracle/plan_model.c) is pinned to, tests/test_gpu_model.py
enum : uint32_t { kPlanStatUnits = 0, kPlanStatAccepted, kPlanStatFailed, kPlanStatLevels, kPlanStatEntryLevels,
                  kPlanStatSeedLookups, kPlanStatSeedExtensions, kPlanStatMismatches,
                  kPlanStatTabLookups, kPlanStatTabWritten, kPlanStatTabFlagged, kPlanStatTabAnchored, kPlanStatWords };
constexpr uint32_t kPlanStatSlots = 8;
// where the pieces of a launch's plan work live inside its work buffer (attach_plan)
struct PlanLayout {
    size_t gitems, units, glist, ucount, usums, qctl, pstats, redo, end;
    uint32_t unit_cap;
};
// capacity of the unit array and bytes of plan work for a launch of n_items items over total_bases bases
inline size_t plan_unit_cap(size_t n_items, uint64_t total_bases) { return 3 * n_items + total_bases / 64 + 64; }
inline PlanLayout plan_layout(size_t n_items, uint64_t total_bases)
{
    PlanLayout L;
    size_t w = 0;
    L.gitems = w;
    w += n_items * sizeof(GuidedItem);
    L.unit_cap = (uint32_t)(plan_unit_cap(n_items, total_bases) < 0x7FFFFF00ull ? plan_unit_cap(n_items, total_bases) : 0x7FFFFF00ull);
    L.units = w;
    w += (size_t)L.unit_cap * sizeof(WalkUnit);
    L.glist = w;
    w += (n_items * kPlanListMax * 2 + 15) / 16 * 16;
    L.ucount = w;
    w += (2 * n_items + 1) * 4;
    L.usums = w;
    w += (n_items / 512 + 4) * 4;
    w = (w + 15) / 16 * 16;
    L.qctl = w;
    w += 64;
    L.pstats = w;
    w += kPlanStatSlots * kPlanStatWords * 4;
    L.redo = w;
    w += (n_items + 15) / 16 * 16;
    L.end = w;
    return L;
}
inline size_t plan_work_bytes(size_t n_items, uint64_t total_bases) { return plan_layout(n_items, total_bases).end + 64; }
hipError_t launch_plan(WalkArgs &a, hipStream_t stream); // fills in the plan parameters of `a` (the later launches need them)
hipError_t launch_ms_walk_guided(WalkArgs a, uint32_t grid, uint32_t threads, hipStream_t stream);
// table mode: plan_kernel, then the stretches behind the mismatches from the depth table, then the list of the items the table
// could not resolve (for the plain kernel, like the guided walk's redo pass)
hipError_t launch_plan_table(WalkArgs &a, hipStream_t stream);
hipError_t launch_dtab_resolve(const WalkArgs &a, hipStream_t stream);
// ---- kbo::map / matches for a batch of reads in one launch (map_kernels.hip)
// the 2-bit text with its path-start marks from the padded byte text (n_bytes = kPlanPad + n_sets + kPlanPad), n_units of
// 16 positions; the text positions of the seed table's intervals
hipError_t launch_pack_text(const uint8_t *d_text_padded, uint64_t n_bytes, uint2 *d_out, uint64_t n_units, hipStream_t stream);
// (a read's diagonal may start up to 160 bases in front of the text - it is seeded from its last bases too - and end 176 behind it)
constexpr uint32_t kMapPad = 192;
inline uint64_t pack_text_units(uint64_t n_sets) { return (n_sets + kMapPad + 256u) / 16u + 2u; }
hipError_t launch_seed_pos(const uint2 *d_seed_tab, const uint32_t *d_pc_pos, uint32_t *d_out, uint32_t seed_d, hipStream_t stream);
// true when the launch described by `a` (plan work attached, chars_out set) can take map_reads_kernel: reads of at most 160
// bases, MS values / characters only, an index copy with a depth table, the 2-bit text and the position table
bool map_reads_applies(const WalkArgs &a);
// the kernel; the reads it could not finish are flagged in a.redo (launch_redo_pass walks them, launch_derand_flagged
// translates them)
hipError_t launch_map_reads(WalkArgs &a, hipStream_t stream);
bool map_reads_finish_applies(const WalkArgs &a);
hipError_t launch_map_reads_finish(const WalkArgs &a, hipStream_t stream);
bool map_reads_direct(const WalkArgs &a);
bool map_reads_packed_applies(const WalkArgs &a, bool packed_out); // (a.qp set: the reads as 2-bit words; a.packed_out: the characters too)
// packed-native batches (pack_kernels.hip): exc[s] = 1 for every read that holds a listed byte (d_exc zeroed first); the bytes of the
// flagged reads from their words (for the plain walk; the listed bytes then go over them: launch_exceptions); the characters of
// the flagged reads into their words
hipError_t launch_flag_exceptions(const uint64_t *d_pos, uint32_t n, uint64_t base, const uint64_t *d_off, uint32_t n_seqs, uint8_t *d_exc, hipStream_t stream);
hipError_t launch_unpack_flagged(const uint32_t *d_packed, const uint64_t *d_off, uint32_t n_seqs, uint32_t uniform_wps, const uint32_t *d_scratch,
                                 const uint8_t *d_flags, uint8_t *d_q, hipStream_t stream);
hipError_t launch_pack_flagged(const uint8_t *d_chars, const uint64_t *d_off, uint32_t n_seqs, uint32_t uniform_wps, const uint32_t *d_scratch,
                               const uint8_t *d_flags, uint32_t *d_packed, hipStream_t stream);
// the list of the flagged items (redo_collect_kernel) and their plain walk; `a` as the plan launch left it
hipError_t launch_redo_pass(WalkArgs a, hipStream_t stream);
// A5 + A6 (+ relative_to_ref) for the sequences with flags[s] != 0 only, one lane each; d_run_counts (unformatted characters only):
// the number of runs of each of those sequences for format::run_lengths_gapped with max_gap_len = 0, next to map_reads_kernel's own
hipError_t launch_derand_flagged(const uint8_t *d_ms, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t k, uint32_t threshold,
                                 const uint8_t *d_ref, uint8_t *d_chars_out, const uint8_t *d_flags, uint32_t max_seq_len, hipStream_t stream,
                                 uint32_t *d_run_counts = nullptr);
// stretches the fused plan_kernel leaves to the anchors: one block per plan_kernel wave in the unit array (free in table mode
// until redo_collect_kernel builds its list there): {count, pad[3]} + kDtabStretchCap entries {item, start, m | next << 16,
// len | warm << 16}; a wave with more of them flags the items of the rest
constexpr uint32_t kDtabStretchCap = 96, kDtabStretchBlockBytes = 16u + kDtabStretchCap * 16u;
hipError_t launch_dtab_stretches(const WalkArgs &a, uint32_t n_waves, hipStream_t stream);
// the depth table of `order` bases (<= 17, <= k) of the index behind `ix`: 4^order bytes at d_tab; d_tmp: dtab_tmp_bytes(cap)
// with cap >= n rows + 1.  Synchronous.
size_t dtab_tmp_bytes(uint64_t frontier_cap);
inline size_t dtab_bytes(uint32_t order, bool grouped) { return grouped ? (size_t)64 << (2u * (order - 2u)) : (size_t)1 << (2u * order); }
hipError_t regroup_depth_table(const uint8_t *d_plain, uint32_t order, uint8_t *d_grouped, hipStream_t stream);
hipError_t build_depth_table(const DevIndexView &ix, uint32_t order, uint8_t *d_tab, void *d_tmp, uint64_t frontier_cap, hipStream_t stream,
                             uint64_t *d_anchor = nullptr, uint32_t anchor_bits = 0 /* ix.pc_pos must be set when d_anchor is */,
                             uint2 *d_seed = nullptr, uint32_t seed_d = 0 /* <= order: plan_kernel's seed table ({l, r} per string) */,
                             uint32_t *d_filter = nullptr, uint32_t filter_bases = 0 /* < order: one bit per string of that many bases,
                             set where it is a suffix of a row (4^filter_bases / 8 bytes) */);
// slots (log2) of the anchor hash of an index of n rows and a table of `order` bases: twice the strings it can hold
inline uint32_t dtab_anchor_bits(uint64_t n_rows, uint32_t order)
{
    const uint64_t most = order >= 16u ? n_rows : (n_rows < ((uint64_t)1 << (2u * order)) ? n_rows : (uint64_t)1 << (2u * order));
    uint32_t b = 4;
    while (((uint64_t)1 << b) < 2 * most + 16) b++;
    return b;
}
void set_guided_walk(int waves_per_cu, int recovery_lines); // tuning: see kbo_set_guided_walk
bool guided_uses_recovery_lines(const WalkArgs &a);
void set_plan_stage(int on); // experiments: plan_kernel with (default) / without its LDS staging
void set_plan_bail(int units_per_16_items); // tuning: launches with more units than this per 16 items give the plan up
void set_plan_params(int dmin, int cap, int gap = 0, int chunk = 0); // tuning (<= 0 keeps): seed depth / seed iterations, unit gap / chunk

// offsets (n_seqs+1) -> one item per sequence
hipError_t launch_make_items(const uint64_t *d_offsets, uint32_t n_seqs, WalkItem *d_items,
                             hipStream_t stream);
// offsets -> items of at most `chunk` emitted bases (+ k-1 warm-up bases; call mode: `chunk` a multiple of 4, else
// hipErrorInvalidValue: see the launcher), n_slots >= total/chunk + n_seqs item
// slots are written (unused ones as empty items); d_scratch: chunk_items_scratch_words(n_seqs) u32
size_t chunk_items_scratch_words(uint32_t n_seqs);
hipError_t launch_make_chunk_items(const uint64_t *d_offsets, uint32_t n_seqs, uint32_t chunk, uint32_t k,
                                   uint32_t n_slots, WalkItem *d_items, uint32_t *d_scratch, hipStream_t stream,
                                   bool call = false /* items for the call mode of the walk */);
// format::run_lengths_gapped over a batch of translated sequences (see rle_kernels.hip); records are 7 u32
// {start, end, matches, mismatches, jumps, gap_bases, gap_opens}; after launch_rle_count the first-run index of
// sequence s is d_scratch[n_seqs + 1 + s / 1024] + d_scratch[s] and *d_total the number of runs
hipError_t launch_rle_count(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                            uint32_t *d_scratch, uint32_t *d_total, hipStream_t stream, uint32_t max_seq_len = 0,
                            bool own_alphabet = false /* the characters are translate_ms_vec's own, unformatted: M - X R and nothing else */);
hipError_t launch_rle_scan_counts(uint32_t n_seqs, uint32_t *d_scratch, uint32_t *d_total, hipStream_t stream);
hipError_t launch_rle_emit(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                           uint32_t *d_scratch, uint32_t *d_rles, uint32_t capacity, hipStream_t stream,
                           uint32_t max_seq_len = 0 /* longest sequence if known: reads take the LDS-staged kernels */,
                           bool own_alphabet = false);
// A1: k-bounded matching statistics over all items
hipError_t launch_ms_walk(WalkArgs a, int max_waves, hipStream_t stream);
// A5+A6 (+ optional relative_to_ref when ref != nullptr, + optional i32 derandomised
// values when derand_out != nullptr): one lane per sequence.  max_seq_len = length of the
// longest sequence if known (0 = unknown); short reads take the LDS-staged kernel.
hipError_t launch_derand_translate(const uint8_t *d_ms, const uint64_t *d_offsets, uint32_t n_seqs,
                                   uint32_t k, uint32_t threshold, const uint8_t *d_ref,
                                   uint8_t *d_chars_out, int32_t *d_derand_out, uint32_t max_seq_len,
                                   uint32_t per_lane_max_len, hipStream_t stream, uint64_t total_bases = 0,
                                   void *d_work = nullptr, size_t work_bytes = 0);
// scratch that lets launch_derand_translate split long reads / contigs into pieces (one lane each)
size_t derand_piece_work_bytes(uint32_t n_seqs, uint64_t total_bases);
// A5+A6 for ONE very long sequence (pointers already offset to its first byte): chunked
// three-level scan over per-chunk transition tables; d_scratch >= derand_long_scratch_bytes().
size_t derand_long_scratch_bytes(uint64_t len, uint32_t k, uint32_t threshold);
hipError_t launch_derand_long(const uint8_t *d_ms, uint32_t len, uint32_t k, uint32_t threshold, const uint8_t *d_ref,
                              uint8_t *d_chars_out, int32_t *d_derand_out, void *d_scratch, hipStream_t stream);
constexpr uint32_t kLongSeq = 1u << 16; // sequences longer than this take the chunked path
// A6 alone on clamped i32 derandomised values: one lane per position.
hipError_t launch_translate(const int32_t *d_derand, uint64_t len, uint32_t k, uint32_t threshold,
                            uint8_t *d_chars_out, hipStream_t stream);

// the breakpoint scan of call_variants over a batch (call_kernels.hip): 16-byte records {sequence, i, j, row of ms[j]} in
// kCallSegs lists (see launch_call_sites)
constexpr uint32_t kCallSegs = 256;
hipError_t launch_call_sites(const uint8_t *d_ms, const uint32_t *d_lo, const uint32_t *d_hi, const uint64_t *d_off,
                             uint32_t n_seqs, uint64_t total, uint32_t k, uint32_t threshold, void *d_sites, uint32_t cap,
                             uint32_t *d_count, hipStream_t stream);

// behind the first pass of a batch's call: every site of the kCallSegs lists becomes {sequence, i, j, row} (void records stay
// void) at index d_prefix[list] + slot of d_recs, and its window - the k MS bytes ending at j, the k characters of the row -
// goes to d_win (call_kernels.hip call_finalize_kernel); record stride = call_gather_stride(k) bytes: MS bytes at 0,
// characters at kpad, flag byte at 2 kpad
inline uint32_t call_gather_stride(uint32_t k) { return 2u * ((k + 15u) / 16u * 16u) + 16u; }
// the second pass of kbo::call on the device (call_second_kernels.hip): per-sequence tables of q-mer start positions, then per site
// { rpeak | qpeak << 8 | csl << 16 | flags << 24 } (0xFF = no peak; flags bit 0 = left to the host; ~0 for a void record)
hipError_t launch_call_qmer_index(const uint8_t *d_q, const uint64_t *d_off, uint32_t n_seqs, uint32_t qlen, const uint64_t *d_tab_off,
                                  uint32_t *d_tab, uint8_t *d_seq_flag, uint64_t total_slots, uint32_t max_slots, hipStream_t stream);
hipError_t launch_call_depths(const void *d_recs, const uint8_t *d_win, uint32_t stride, uint32_t n_sites, const uint8_t *d_q,
                              const uint64_t *d_off, uint32_t k, uint32_t thr, uint32_t qlen, bool revcomp, const uint64_t *d_tab_off,
                              const uint32_t *d_tab, const uint8_t *d_seq_flag, uint32_t *d_out, hipStream_t stream,
                              const uint32_t *d_n_sites = nullptr /* the number of sites on the device (n_sites then bounds it) */,
                              bool per_lane = false /* k <= 64 too through the kernel in which every lane extends its own matches (tests) */);
hipError_t launch_call_finalize(const void *d_lists, const uint32_t *d_counts, const uint32_t *d_prefix, uint32_t seg_cap, uint32_t max_count,
                                bool by_walk, const uint64_t *d_off, uint32_t n_seqs, uint32_t k, const uint8_t *d_ms,
                                const DevIndexView &ix, void *d_recs, uint8_t *d_win, uint32_t stride, hipStream_t stream);

// ---- a device copy's rank blocks / contraction entries / two-base blocks made on the device (layout_kernels.hip; formats: sbwt_index.hpp)
size_t device_layout_scratch_bytes(uint64_t n_sets);
hipError_t build_device_layout(const uint64_t *const d_rows[4], uint64_t n_words, const uint8_t *d_lcs, uint64_t n, const uint64_t C[4],
                               uint32_t n_blocks, uint4 *d_rank, uint32_t *d_ent, uint4 *d_pair, void *d_scratch, hipStream_t stream);

// ---- the path cover laid out on the device (cover_kernels.hip; what it is: path_cover.cpp).  *ok = false: rows on cycles - the host's decides
hipError_t build_path_cover_device(const uint4 *d_rank, const uint32_t *d_ent, uint64_t n, uint32_t n_blocks, uint32_t k, const uint64_t C[4],
                                   uint8_t *d_text, uint32_t *d_pos, uint32_t *d_node, hipStream_t stream, bool *ok);

// ---- the tail of kbo::call on the device (call_emit_kernels.hip): the variants of a slab's sites in the order of (sequence, query
// position), as flat arrays.  d_meta: kCallMetaWords words
constexpr uint32_t kCallMetaSites = 0, kCallMetaValid = 1, kCallMetaVariants = 2, kCallMetaChars = 3, kCallMetaHost = 4, kCallMetaFlags = 5,
                   kCallMetaWorst = 6, kCallMetaWords = 16;
constexpr uint32_t kCallMaxRank = 4096;          // sites of one sequence the device puts in order itself (more: the host's)
constexpr uint32_t kCallNoVariant = 0xFFFFFFFFu; // per-site word: Err(ResolveVariantErr)
constexpr uint32_t kCallHostSite = 0xFFFFFFFEu;  // ... left to the host
struct CallEmitArgs {
    const uint4 *recs;       // call_finalize_kernel's records {sequence, i, j, row}; first word ~0 = void
    const uint32_t *codes;   // call_depths_kernel's word per site
    const uint8_t *win;      // ... and windows (row characters at kpad)
    uint32_t stride, kpad, k;
    const uint8_t *q;        // the slab's bases
    const uint64_t *off;     // ... and offsets
    uint32_t n_seqs;
    const uint32_t *n_sites; // on the device: d_prefix[kCallSegs]
    uint32_t cap;            // what bounds it
    uint32_t *seq_cnt, *seq_sums, *seq_fill; // n_seqs + 1 (+ scan sums), n_seqs
    uint32_t *bucket, *bkey, *sorted, *vrec; // cap each
    uint32_t *vcnt, *vsums, *ccnt, *csums;   // cap + 1 (+ scan sums) each
    uint32_t *out_pos, *out_lens;            // cap: query_pos; query_len | ref_len << 16
    uint8_t *out_chars;
    uint32_t chars_cap;
    uint32_t *seq_vfirst;    // n_seqs + 1: variants in front of every sequence
    uint32_t *host_list;     // sites left to the host (indices into recs), host_cap of them at most
    uint32_t host_cap;
    uint32_t *meta;
};
inline size_t call_scan_sums_words(size_t n) { return (n + 1023) / 1024 + 1; }
hipError_t launch_call_prefix(const uint32_t *d_counts, uint32_t seg_cap, uint32_t *d_prefix /* kCallSegs + 1 */, uint32_t *d_meta, hipStream_t stream);
hipError_t launch_call_emit(const CallEmitArgs &a, void *d_host_recs, uint8_t *d_host_win, hipStream_t stream);

// 2-bit packed reads in / packed alignments out (pack_kernels.hip): sequence s occupies ceil(len / 16) u32 words, base i in
// bits 2 (i mod 16) of word i / 16.  uniform_wps != 0: all sequences have that many words (no prefix needed); otherwise
// d_scratch holds the scanned words-per-sequence (launch_packed_prefix; chunk_items_scratch_words(n_seqs) u32)
hipError_t launch_uniform_offsets(uint64_t *d_off, uint32_t n_seqs, uint32_t len, hipStream_t stream);
hipError_t launch_packed_prefix(const uint64_t *d_off, uint32_t n_seqs, uint32_t *d_scratch, hipStream_t stream);
hipError_t launch_unpack2(const uint32_t *d_packed, uint32_t n_words, const uint64_t *d_off, uint32_t n_seqs, uint32_t uniform_wps,
                          const uint32_t *d_scratch, uint8_t *d_q, hipStream_t stream);
hipError_t launch_exceptions(const uint64_t *d_pos, const uint8_t *d_byte, uint32_t n, uint64_t base, uint8_t *d_q, hipStream_t stream);
hipError_t launch_pack2(const uint8_t *d_chars, uint32_t n_words, const uint64_t *d_off, uint32_t n_seqs, uint32_t uniform_wps,
                        const uint32_t *d_scratch, uint32_t *d_packed, hipStream_t stream);

// a[i] = max(a[i], b[i]) over n bytes (n rounded up to 16: both buffers have that slack): the MS values of a further shard of a
// sharded index folded into the batch's (pack_kernels.hip)
hipError_t launch_max_bytes(uint8_t *d_a, const uint8_t *d_b, uint64_t n, hipStream_t stream);

// ---- kbo::map / matches for sequences of any length in one launch (long_kernels.hip): one wave per PIECE of a sequence - `own`
// bases inside a region of at most kLongRegion bases (k bases of the sequence in front of them, k + 1 behind)
constexpr uint32_t kLongRegion = 1008; // (+ up to 15 bases of alignment: 64 words of 16 positions)
constexpr uint32_t kLongOwnMin = 256;  // own bases a piece must have for the kernel to apply (k <= 375)
struct LongArgs {
    DevIndexView ix;
    const uint8_t *q;     // concatenated queries (ASCII), 16-byte aligned
    uint64_t q_bytes;
    const void *items;    // per piece { first byte of its region, that byte's place in its sequence, the sequence's length, own0 | own_n << 10 }
    uint32_t n_items;     // slots (those past the batch's last piece are empty)
    uint8_t *chars_out;
    uint8_t *redo;        // per piece: 1 = its proof failed (the plain walk + the literal recurrences decide: launch_map_long_redo)
    uint8_t *xin;         // per piece: the derandomised value of its first own base, 0 .. k (<= 0 as 0)
    uint32_t *qctl;       // [0] pieces, [1] sub-items of the flagged pieces, [2] flagged pieces listed, [4] flagged pieces
    uint32_t *pstats;     // work counters (kPlanStat*)
    uint32_t ppw;         // consecutive pieces a wave takes
    uint32_t xexp;        // experiment switches (KBO_LONG_X): timing only, results are wrong with any of them
    uint32_t thr, fmt, ca; // derandomisation threshold, 1 = format::relative_to_ref on the way out, bases of a region behind the own ones
    void *subs;           // WalkItem records of the flagged pieces' sub-items
    uint32_t *flist;      // the flagged pieces, listed (qctl[2] of them)
    uint32_t sub_cap;
};
size_t long_work_bytes(size_t n_seqs, uint64_t total_bases, uint32_t k); // bytes of work memory of a launch (0: k too large)
// true when the copy has what the kernel needs (depth table of fewer bases than the threshold, 2-bit text, seed positions)
bool map_long_applies(const DevIndexView &ix, uint32_t thr);
void set_map_long(int mode); // tuning / tests: see kbo_set_map_long
hipError_t launch_map_long(const DevIndexView &ix, const uint8_t *d_q, const uint64_t *d_off, uint32_t n_seqs, uint64_t total_bases, uint32_t thr,
                           bool fmt, uint8_t *d_chars, void *d_work, hipStream_t stream, LongArgs &a, bool count /* work counters: kbo_set_plan_stats */);
hipError_t launch_map_long_redo(const LongArgs &a, uint8_t *d_ms, hipStream_t stream);
// the control words (32) and work counters of the last launch over d_work; synchronises the stream
hipError_t long_read_stats(const void *d_work, size_t n_seqs, uint64_t total_bases, uint32_t k, uint32_t ctl[32], uint32_t *stats, hipStream_t stream);
// the plain walk over a list of items whose number is counted on the device (walk_kernels.hip): `lanes` lanes share them
hipError_t launch_walk_list(WalkArgs a, const WalkItem *d_list, uint32_t cap, const uint32_t *d_count, uint32_t lanes, hipStream_t stream);

constexpr int kWalkThreads = 64; // default workgroup size (waves are independent: no LDS, no barriers)
void set_walk_threads(int threads); // tuning: 64, 128 or 256
void set_walk_experiment(int lane_limit, int dummy_lds_bytes); // experiments behind DESIGN.md section 6
void set_walk_rare(int period);            // tuning: hot-loop iterations between rare-block visits
void set_pair_min_depth(int d);            // tuning: depth from which two-base steps are tried
constexpr uint32_t kRankRows = 96; // rows per 16-byte rank block (== kRankRowsPerBlock)

} // namespace kbo
