// kernels.hpp — launch interface of the gfx950 kernels (ms_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace kbo {

// Device-resident index (32-bit positions).  Layout: sbwt_index.hpp.
struct DevIndexView {
    const uint4 *rank;    // 16-byte rank blocks, character c at rank + c * n_blocks
    uint32_t n_blocks;    // blocks per character
    const uint4 *lcs16;   // LCS bytes viewed as aligned 16-byte windows
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
    uint8_t *d_out;        // 1 byte per base, same indexing as q
    uint32_t *lo_out;      // optional (nullptr): interval start per base
    uint32_t *hi_out;      // optional: interval end per base
};

// offsets (n_seqs+1) -> one item per sequence
hipError_t launch_make_items(const uint64_t *d_offsets, uint32_t n_seqs, WalkItem *d_items,
                             hipStream_t stream);
// A1: k-bounded matching statistics over all items
hipError_t launch_ms_walk(const WalkArgs &a, int blocks, hipStream_t stream);
// A5+A6 (+ optional relative_to_ref when ref != nullptr, + optional i32 derandomised
// values when derand_out != nullptr): one lane per sequence.
hipError_t launch_derand_translate(const uint8_t *d_ms, const uint64_t *d_offsets, uint32_t n_seqs,
                                   uint32_t k, uint32_t threshold, const uint8_t *d_ref,
                                   uint8_t *d_chars_out, int32_t *d_derand_out, hipStream_t stream);
// A6 alone on clamped i32 derandomised values: one lane per position.
hipError_t launch_translate(const int32_t *d_derand, uint64_t len, uint32_t k, uint32_t threshold,
                            uint8_t *d_chars_out, hipStream_t stream);

constexpr int kWalkThreads = 256;
constexpr uint32_t kRankRows = 96; // rows per 16-byte rank block (== kRankRowsPerBlock)

} // namespace kbo
