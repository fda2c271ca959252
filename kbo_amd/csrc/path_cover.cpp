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
// text): one streaming pass matches the rows of every (k-1)-suffix group with the group's successors, two
// parallel passes over the resulting chains (cut into segments) number them.
#include "sbwt_index.hpp"

#include <cstring>

#include <algorithm>
#include <atomic>
#include <functional>
#include <stdexcept>
#include <thread>

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
    // The chains are followed pointer by pointer - a DRAM miss a row, and a genome without repeats is ONE chain - so they are
    // cut into segments at "splitters" (the heads, and every row that is a multiple of kSplit: a row is as good as random
    // along a chain), the segments are measured in parallel, one short sequential pass over the splitters hands every
    // segment of a head's chain its position - heads in row order, each chain to its end, exactly the layout one thread
    // following the chains would make - and the segments are laid in parallel.
    constexpr uint32_t kSplit = 1024;
    auto is_split = [&](uint32_t u) { return (u & (kSplit - 1)) == 0; }; // (for a row that has a predecessor)
    std::vector<uint32_t> splitters;
    for (uint64_t i = 0; i < n; i++)
        if (!has_pred[i] || is_split((uint32_t)i)) splitters.push_back((uint32_t)i);
    const size_t ns = splitters.size();
    std::vector<uint32_t> seg_len(ns), seg_next(ns);
    std::vector<uint64_t> seg_start(ns, ~0ull);
    auto index_of = [&](uint32_t row) { return (size_t)(std::lower_bound(splitters.begin(), splitters.end(), row) - splitters.begin()); };
    auto in_parallel = [&](size_t n_tasks, const std::function<void(size_t)> &fn) {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)32, (size_t)hw, n_tasks / 64 + 1}));
        std::atomic<size_t> nxt{0};
        auto work = [&] {
            for (;;) {
                const size_t a = nxt.fetch_add(16);
                if (a >= n_tasks) return;
                for (size_t i = a; i < std::min(n_tasks, a + 16); i++) fn(i);
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    };
    in_parallel(ns, [&](size_t s) { // length of the segment that starts at splitter s, and the splitter it runs into
        uint32_t len = 1, u = next[splitters[s]];
        while (u != NONE && !is_split(u)) {
            len++;
            u = next[u];
        }
        seg_len[s] = len;
        seg_next[s] = u;
    });
    uint64_t p = 0;
    for (size_t s = 0; s < ns; s++) {
        if (has_pred[splitters[s]]) continue; // heads, in row order
        size_t at = s;
        for (;;) {
            seg_start[at] = p;
            p += seg_len[at];
            if (seg_next[at] == NONE) break;
            at = index_of(seg_next[at]);
        }
    }
    in_parallel(ns, [&](size_t s) {
        uint64_t q = seg_start[s];
        if (q == ~0ull) return; // (a splitter on a cycle: below)
        uint32_t u = splitters[s];
        for (uint32_t j = 0; j < seg_len[s]; j++) {
            out.pos[u] = (uint32_t)q;
            out.node_at[q] = u;
            text[q] = has_pred[u] ? label(u) : 0;
            q++;
            u = next[u];
        }
    });
    // whatever is left lies on cycles: each is broken where the row order meets it (few rows, if any: one thread)
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
