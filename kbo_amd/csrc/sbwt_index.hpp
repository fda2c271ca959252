// sbwt_index.hpp — host-side abstract SBWT index (the content of
// sbwt::SbwtIndex<SubsetMatrix> + sbwt::LcsArray that kbo::build returns,
// reference lib.rs:501-506 / index.rs:56-99) and its MI355X device layout.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace kbo {

// Abstract index content.  rows[c] is the subset-matrix row of character c
// ("ACGT"[c]) as 64-bit words, bit (i & 63) of word (i >> 6) = row i.
struct HostIndex {
    uint32_t k = 0;
    uint64_t n_sets = 0;  // SBWT rows (k-mers + dummy rows)
    uint64_t n_kmers = 0; // distinct real k-mers
    uint64_t C[4] = {0, 0, 0, 0};
    std::vector<uint64_t> rows[4];
    std::vector<uint8_t> lcs; // n_sets entries, LCS[0] = 0
};

struct BuildParams {
    uint32_t k = 31;
    bool add_revcomp = false;
    uint32_t num_threads = 1;
    // sharded indexes (kbo_capi.cpp: an index too large for 32-bit row numbers is built as several, MS = the maximum over
    // them): only the reverse-complement strand of the input; the sorted distinct k-mers as 64-bit words, key_words per
    // k-mer (colex keys: equal k-mers have equal keys in every shard), for counting the k-mers of the union
    bool revcomp_only = false;
    std::vector<uint64_t> *keys_out = nullptr;
    uint32_t *key_words_out = nullptr;
};

// Sort-based construction (colex-sorted padded k-mer rows).  Throws std::runtime_error.
void build_host_index(const uint8_t *const *seqs, const size_t *lens, size_t n_seqs,
                      const BuildParams &p, HostIndex &out);

// ---- device layout (32-bit positions: n_sets < 2^32 - 1) -------------------------
//
// Rank blocks: for each character c an array of 16-byte blocks, block b covering
// rows [96 b, 96 b + 96):
//     { cum, w0, w1, w2 }   cum = C[c] + popcount(B_c[0 .. 96 b)),  w* = the 96 row bits
// so that   C[c] + rank_c(i) = cum + popcount(bits below (i - 96 b))   costs ONE
// aligned 16-byte load.  n_blocks = n_sets / 96 + 2 (rank at i = n_sets is legal).
//
// Contraction entries: for every row i (plus a sentinel at i = n_sets) 12 bytes
//     { lcs[i], psv[i], nsv[i] }
// with psv[i] = largest j < i with lcs[j] < lcs[i] and nsv[i] = smallest j > i with
// lcs[j] < lcs[i] (lcs[0] = 0 and the sentinel lcs[n_sets] = 0 bound both searches).
// contract_left([l,r), m) for m = max(lcs[l], lcs[r]) is then
//     l' = lcs[l] == m ? psv[l] : l,   r' = lcs[r] == m ? nsv[r] : r
// — two 12-byte loads and two compares instead of the reference's linear LCS scans.
constexpr uint32_t kRankRowsPerBlock = 96;

struct DeviceLayout {
    uint64_t n_blocks = 0;            // per character
    std::vector<uint32_t> rank[4];    // 4 * n_blocks words each
    std::vector<uint32_t> ent;        // 3 * (n_sets + 1) words: {lcs, psv, nsv} per row
    // optional two-base extension blocks, same format as `rank`: pair p = 4*c1 + c2 occupies
    // blocks [p * n_blocks, (p+1) * n_blocks); bit i of its bit-vector is
    // B_c1[i] & B_c2[C[c1] + rank_c1(i)], its base count C[c2] + rank_c2(C[c1])
    std::vector<uint32_t> pair;
};
void make_device_layout(const HostIndex &h, DeviceLayout &out, bool with_pairs = false);

// Recovery lines (the guided walk's view of the index: plan_kernels.hip).  Where a read leaves the reference the walk
// alternates failed extensions, contractions and extensions on rows that are random from one base to the next, so what
// a base needs should come with ONE 128-byte line: line b covers rows [64 b, 64 b + 64) and holds
//     bytes  0 ..  63   four 16-byte rank blocks { C[c] + rank_c(64 b), row bits 0..31, row bits 32..63, 0 }, c = A,C,G,T
//     bytes 64 .. 127   LCS[64 b .. 64 b + 64)  (0 beyond row n_sets - 1: the sentinel of the contraction searches)
// A failed extension has fetched the line that also holds the LCS values around its rows; the contraction scans them
// (previous / next smaller value inside a 16-row window) and falls back to the {lcs, psv, nsv} entries when the
// window does not hold the answer.  n_sets / 64 + 2 lines and one all-zero line behind them; 2 bytes per row.
constexpr uint32_t kFatRows = 64;
void make_recovery_lines(const HostIndex &h, std::vector<uint8_t> &out);

// Path cover of the index's de Bruijn graph, laid out as one text (path_cover.cpp): every row sits at
// exactly one position; text[p] is the label of the edge node_at[p-1] -> node_at[p], 0 where a path starts.
struct PathCover {
    static constexpr size_t kPad = 64;  // zero bytes in front of and behind the text
    std::vector<uint8_t> text;          // kPad + n_sets + kPad bytes
    std::vector<uint32_t> pos;          // row -> position
    std::vector<uint32_t> node_at;      // position -> row
};
void make_path_cover(const HostIndex &h, PathCover &out);

// flat file (own format, see kbo_capi.cpp)
// (cover: optional path cover written behind / read from behind the index; a cover from a file is validated against the
// subset matrix before it is used)
void save_host_index(const HostIndex &h, const std::string &path, const PathCover *cover = nullptr);
void load_host_index(const std::string &path, HostIndex &h, PathCover *cover = nullptr, bool *have_cover = nullptr);
void validate_path_cover(const HostIndex &h, const PathCover &pc); // throws std::runtime_error
void save_sbwt_pair(const HostIndex &h, const std::string &prefix); // <prefix>.sbwt + <prefix>.lcs (sbwt_build.cpp)
bool load_sbwt_pair(const std::string &prefix, HostIndex &h);       // false: payload written by the sbwt crate itself
void validate_host_index(const HostIndex &h); // throws std::runtime_error on inconsistent k / C / edge bits / LCS

} // namespace kbo
