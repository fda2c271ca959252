// path_cover.cpp — a node-disjoint path cover of the SBWT's de Bruijn graph, laid out as one text.
//
// Why: while the interval of the walk (A1, reference index.rs:251-252) is a single row u, extending by
// c succeeds exactly when the node of u has an edge labelled c, and leads to that edge's target, at
// depth min(d+1, k) (an edge of the group of u when d == k: the reference contracts to the (k-1)-suffix
// group first and lands on the same row at the same depth).  Along a path of the graph this is a plain
// comparison of the query with the path's labels, so the walk can skip those bases.
// The cover puts every row on exactly one path:
//     node_at[p]  row at text position p            pos[row]  its position
//     text[p]     label of the edge node_at[p-1] -> node_at[p] ('A','C','G','T'), 0 where a path starts
// so "text[p+1] == c" proves that extending the single-row interval {node_at[p]} by c gives {node_at[p+1]}.
// Built from the subset matrix and the LCS array alone (indexes adopted through kbo_index_from_parts have no
// text): one streaming pass matches the rows of every (k-1)-suffix group with the group's successors, one
// pass over the resulting chains numbers them.
#include "sbwt_index.hpp"

#include <cstring>

#include <stdexcept>

namespace kbo {

void make_path_cover(const HostIndex &h, PathCover &out)
{
    const uint64_t n = h.n_sets;
    if (n >= 0xFFFFFFF0ull) throw std::runtime_error("path cover: n_sets >= 2^32");
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    auto bit = [&](int c, uint64_t i) -> bool { return (h.rows[c][i >> 6] >> (i & 63)) & 1; };
    // ---- chains: next[u] = the successor matched to row u
    std::vector<uint32_t> next(n, NONE);
    std::vector<uint8_t> has_pred(n, 0);
    {
        uint64_t cnt[4] = {h.C[0], h.C[1], h.C[2], h.C[3]}; // extend-right bijection: set bits of B_c, in row
                                                            // order, map onto rows C[c], C[c]+1, ...
        uint32_t succ[4];
        unsigned ns = 0, used = 0;
        for (uint64_t i = 0; i < n; i++) {
            const bool first = i == 0 || h.lcs[i] + 1u < h.k; // opens its (k-1)-suffix group
            if (first) {
                ns = used = 0;
                for (int c = 0; c < 4; c++)
                    if (bit(c, i)) {
                        if (cnt[c] >= n) throw std::runtime_error("path cover: edge bits exceed n_sets");
                        succ[ns++] = (uint32_t)cnt[c]++;
                    }
            }
            if (used < ns) {
                const uint32_t s = succ[used++];
                if (s != i) { // a self-loop (AAA -> AAA) cannot be a path step
                    next[i] = s;
                    has_pred[s] = 1;
                }
            }
        }
    }
    // ---- layout: heads first (row order), then whatever is left (cycles), each broken where it is met
    out.pos.assign(n, NONE);
    out.node_at.assign(n, 0);
    out.text.assign(n + 2 * PathCover::kPad, 0);
    uint8_t *text = out.text.data() + PathCover::kPad;
    auto label = [&](uint32_t row) -> uint8_t { // last character of the row: rows are in colex order
        for (int c = 3; c >= 0; c--)
            if (row >= h.C[c]) return (uint8_t)"ACGT"[c];
        return 0; // the root
    };
    uint64_t p = 0;
    auto lay = [&](uint32_t u) {
        bool start = true;
        while (u != NONE && out.pos[u] == NONE) {
            out.pos[u] = (uint32_t)p;
            out.node_at[p] = u;
            text[p] = start ? 0 : label(u);
            start = false;
            p++;
            u = next[u];
        }
    };
    for (uint64_t i = 0; i < n; i++)
        if (!has_pred[i]) lay((uint32_t)i);
    for (uint64_t i = 0; i < n; i++)
        if (out.pos[i] == NONE) lay((uint32_t)i);
    if (p != n) throw std::runtime_error("path cover: rows left without a position");
}

// Recovery lines: sbwt_index.hpp.
void make_recovery_lines(const HostIndex &h, std::vector<uint8_t> &out)
{
    const uint64_t n = h.n_sets, n_lines = n / kFatRows + 2, n_words = (n + 63) / 64;
    out.assign((n_lines + 1) * 128, 0);
    uint64_t cum[4] = {h.C[0], h.C[1], h.C[2], h.C[3]};
    for (uint64_t b = 0; b < n_lines; b++) {
        uint8_t *line = out.data() + b * 128;
        for (int c = 0; c < 4; c++) {
            uint64_t w = b < n_words ? h.rows[c][b] : 0;
            if (b * 64 + 64 > n) w &= b * 64 >= n ? 0 : (~0ull >> (64 - (n - b * 64))); // (rows beyond the last one: no bits)
            const uint32_t blk[4] = {(uint32_t)cum[c], (uint32_t)w, (uint32_t)(w >> 32), 0u};
            std::memcpy(line + 16 * c, blk, 16);
            cum[c] += (uint64_t)__builtin_popcountll(w);
        }
        if (b * 64 < n) std::memcpy(line + 64, h.lcs.data() + b * 64, (size_t)std::min<uint64_t>(64, n - b * 64));
    }
}

} // namespace kbo
