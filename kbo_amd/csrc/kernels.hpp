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
};

// One unit of walk work: `len` bases starting at absolute offset `start` of the
// concatenated query buffer; the first `warm` of them only warm the state up (no
// output) — chunks of long sequences restart k-1 bases upstream.
struct alignas(16) WalkItem {
    uint64_t start;
    uint32_t len;
    uint32_t warm;
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
    uint8_t *d_out;        // 1 byte per base, same indexing as q
    uint32_t *lo_out;      // optional (nullptr): interval start per base
    uint32_t *hi_out;      // optional: interval end per base
};

// offsets (n_seqs+1) -> one item per sequence
hipError_t launch_make_items(const uint64_t *d_offsets, uint32_t n_seqs, WalkItem *d_items,
                             hipStream_t stream);
// offsets -> items of at most `chunk` emitted bases (+ k-1 warm-up bases), n_slots >= total/chunk + n_seqs item
// slots are written (unused ones as empty items); d_scratch: chunk_items_scratch_words(n_seqs) u32
size_t chunk_items_scratch_words(uint32_t n_seqs);
hipError_t launch_make_chunk_items(const uint64_t *d_offsets, uint32_t n_seqs, uint32_t chunk, uint32_t k,
                                   uint32_t n_slots, WalkItem *d_items, uint32_t *d_scratch, hipStream_t stream);
// format::run_lengths_gapped over a batch of translated sequences (see rle_kernels.hip); records are 7 u32
// {start, end, matches, mismatches, jumps, gap_bases, gap_opens}; after launch_rle_count the first-run index of
// sequence s is d_scratch[n_seqs + 1 + s / 1024] + d_scratch[s] and *d_total the number of runs
hipError_t launch_rle_count(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                            uint32_t *d_scratch, uint32_t *d_total, hipStream_t stream, uint32_t max_seq_len = 0);
hipError_t launch_rle_emit(const uint8_t *d_chars, const uint64_t *d_offsets, uint32_t n_seqs, uint32_t max_gap_len,
                           uint32_t *d_scratch, uint32_t *d_rles, uint32_t capacity, hipStream_t stream,
                           uint32_t max_seq_len = 0 /* longest sequence if known: reads take the LDS-staged kernels */);
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

constexpr int kWalkThreads = 64; // default workgroup size (waves are independent: no LDS, no barriers)
void set_walk_threads(int threads); // tuning: 64, 128 or 256
void set_walk_rare(int period);            // tuning: hot-loop iterations between rare-block visits
void set_pair_min_depth(int d);            // tuning: depth from which two-base steps are tried
constexpr uint32_t kRankRows = 96; // rows per 16-byte rank block (== kRankRowsPerBlock)

} // namespace kbo
