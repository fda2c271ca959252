// sbwt_build.cpp — sort-based construction of the SBWT subset matrix + LCS array.
//
// Produces the same abstract index as the sbwt crate does for kbo::build (reference
// index.rs:56-99, with build_lcs(true)): rows = distinct k-mers of every ACGT-run of
// length >= k plus, for each k-mer without a predecessor, its $-padded proper
// prefixes (and the root $^k); colexicographic order with $ < A < C < G < T; edge bit
// B_c[i] set iff row i opens its (k-1)-suffix group and row[1:]+c is a row;
// C[c] = 1 + sum_{c'<c} popcount(B_c'); LCS[i] = common suffix of rows i-1,i without $.
// The algorithm here is this project's own (the crate's bit-packed k-mer sorting is not
// replicated): k-mers are packed into colex keys, bucket-sorted on worker threads, and
// the edge bits come from four linear merge-joins instead of per-row searches.
#include "sbwt_index.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace kbo {
namespace {

inline int code_of(uint8_t ch)
{
    switch (ch) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1; // splits an ACGT run
    }
}

// Colex key: digit t (2 bits) = code of the t-th char counted from the END of the
// k-mer, stored most-significant first, left-aligned in W 64-bit words.  Lexicographic
// order of the words == colex order of the k-mers; '$' padding is digit value 0 and is
// disambiguated by the row's count of real characters (rows sort by (key, real)).
template <int W> struct Key {
    uint64_t w[W];
    bool operator<(const Key &o) const
    {
        for (int i = 0; i < W; i++)
            if (w[i] != o.w[i]) return w[i] < o.w[i];
        return false;
    }
    bool operator==(const Key &o) const
    {
        for (int i = 0; i < W; i++)
            if (w[i] != o.w[i]) return false;
        return true;
    }
    bool operator!=(const Key &o) const { return !(*this == o); }
};

template <int W> inline Key<W> key_zero()
{
    Key<W> k;
    for (int i = 0; i < W; i++) k.w[i] = 0;
    return k;
}
template <int W> inline Key<W> shr2(Key<W> a)
{
    for (int i = W - 1; i > 0; i--) a.w[i] = (a.w[i] >> 2) | (a.w[i - 1] << 62);
    a.w[0] >>= 2;
    return a;
}
template <int W> inline Key<W> shl_bits(Key<W> a, unsigned bits)
{
    unsigned ws = bits / 64, bs = bits % 64;
    Key<W> r = key_zero<W>();
    for (int i = 0; i + (int)ws < W; i++) {
        uint64_t v = a.w[i + ws] << bs;
        if (bs && i + (int)ws + 1 < W) v |= a.w[i + ws + 1] >> (64 - bs);
        r.w[i] = v;
    }
    return r;
}
template <int W> inline void set_digit(Key<W> &a, unsigned p, uint64_t v)
{
    unsigned bit = 2 * p, wi = bit / 64, sh = 62 - (bit % 64);
    a.w[wi] = (a.w[wi] & ~(3ull << sh)) | (v << sh);
}
template <int W> inline unsigned get_digit(const Key<W> &a, unsigned p)
{
    unsigned bit = 2 * p, wi = bit / 64, sh = 62 - (bit % 64);
    return (unsigned)((a.w[wi] >> sh) & 3);
}
// number of equal leading digits (capped at 32*W)
template <int W> inline unsigned common_digits(const Key<W> &a, const Key<W> &b)
{
    for (int i = 0; i < W; i++) {
        uint64_t x = a.w[i] ^ b.w[i];
        if (x) return (unsigned)(32 * i + (__builtin_clzll(x) >> 1));
    }
    return 32 * W;
}

template <int W> struct Row {
    Key<W> key;
    uint32_t real; // number of non-$ characters (k for a real k-mer)
    bool operator<(const Row &o) const
    {
        if (key != o.key) return key < o.key;
        return real < o.real;
    }
    bool operator==(const Row &o) const { return key == o.key && real == o.real; }
};

template <typename F> void parallel_for(size_t n_tasks, unsigned n_threads, F f)
{
    if (n_threads <= 1 || n_tasks <= 1) {
        for (size_t t = 0; t < n_tasks; t++) f(t);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    std::exception_ptr err;
    std::mutex err_mu;
    unsigned nt = (unsigned)std::min<size_t>(n_threads, n_tasks);
    for (unsigned i = 0; i < nt; i++)
        th.emplace_back([&] {
            try {
                for (;;) {
                    size_t t = next.fetch_add(1);
                    if (t >= n_tasks) break;
                    f(t);
                }
            } catch (...) {
                std::lock_guard<std::mutex> g(err_mu);
                if (!err) err = std::current_exception();
            }
        });
    for (auto &t : th) t.join();
    if (err) std::rethrow_exception(err);
}

// Bucket on the 5 leading digits, then std::sort each bucket on a worker thread.
template <int W> void sort_keys(std::vector<Key<W>> &v, unsigned n_threads)
{
    const size_t n = v.size();
    if (n_threads <= 1 || n < (1u << 16)) {
        std::sort(v.begin(), v.end());
        return;
    }
    constexpr unsigned B = 1024;
    const unsigned nt = n_threads;
    std::vector<std::vector<size_t>> hist(nt, std::vector<size_t>(B, 0));
    auto bucket = [](const Key<W> &k) { return (unsigned)(k.w[0] >> 54); };
    parallel_for(nt, nt, [&](size_t t) {
        size_t a = n * t / nt, b = n * (t + 1) / nt;
        for (size_t i = a; i < b; i++) hist[t][bucket(v[i])]++;
    });
    std::vector<size_t> start(B + 1, 0);
    for (unsigned b = 0; b < B; b++) {
        size_t s = 0;
        for (unsigned t = 0; t < nt; t++) s += hist[t][b];
        start[b + 1] = start[b] + s;
    }
    std::vector<std::vector<size_t>> pos(nt, std::vector<size_t>(B));
    for (unsigned b = 0; b < B; b++) {
        size_t s = start[b];
        for (unsigned t = 0; t < nt; t++) { pos[t][b] = s; s += hist[t][b]; }
    }
    std::vector<Key<W>> tmp(n);
    parallel_for(nt, nt, [&](size_t t) {
        size_t a = n * t / nt, b = n * (t + 1) / nt;
        for (size_t i = a; i < b; i++) tmp[pos[t][bucket(v[i])]++] = v[i];
    });
    parallel_for(B, nt, [&](size_t b) { std::sort(tmp.begin() + start[b], tmp.begin() + start[b + 1]); });
    v.swap(tmp);
}

template <int W>
void build_impl(const uint8_t *const *seqs, const size_t *lens, size_t n_seqs, const BuildParams &p,
                HostIndex &out)
{
    const uint32_t k = p.k;
    const unsigned nt = std::max(1u, p.num_threads);

    // keep only the 2k leading bits of a key
    Key<W> mask2k = key_zero<W>();
    for (unsigned d = 0; d < k; d++) set_digit(mask2k, d, 3);

    // ---- 1. k-mers of every ACGT run of length >= k (forward and, optionally, revcomp)
    size_t total = 0;
    for (size_t s = 0; s < n_seqs; s++) {
        size_t run = 0;
        for (size_t i = 0; i < lens[s]; i++) {
            run = code_of(seqs[s][i]) >= 0 ? run + 1 : 0;
            if (run >= k) total++;
        }
    }
    std::vector<Key<W>> kmers;
    const bool want_fw = !p.revcomp_only, want_rc = p.add_revcomp || p.revcomp_only;
    kmers.reserve(total * ((want_fw ? 1 : 0) + (want_rc ? 1 : 0)));
    for (size_t s = 0; s < n_seqs; s++) {
        size_t run = 0;
        Key<W> fw = key_zero<W>(), rc = key_zero<W>();
        for (size_t i = 0; i < lens[s]; i++) {
            int c = code_of(seqs[s][i]);
            if (c < 0) { run = 0; fw = key_zero<W>(); rc = key_zero<W>(); continue; }
            run++;
            fw = shr2(fw);                  // older chars move away from the end
            set_digit(fw, 0, (uint64_t)c);
            for (int j = 0; j < W; j++) fw.w[j] &= mask2k.w[j];
            if (want_rc) {                  // revcomp k-mer ends with comp(first char)
                rc = shl_bits(rc, 2);
                set_digit(rc, k - 1, (uint64_t)(3 - c));
            }
            if (run >= k) {
                if (want_fw) kmers.push_back(fw);
                if (want_rc) kmers.push_back(rc);
            }
        }
    }
    sort_keys(kmers, nt);
    kmers.erase(std::unique(kmers.begin(), kmers.end()), kmers.end());
    const size_t N = kmers.size();
    if (p.keys_out) { // (sharded build: the caller merges the shards' sorted k-mers to count those of the union)
        p.keys_out->resize(N * W);
        for (size_t i = 0; i < N; i++)
            for (int j = 0; j < W; j++) (*p.keys_out)[i * W + j] = kmers[i].w[j];
        if (p.key_words_out) *p.key_words_out = W;
    }

    // ---- 2. k-mers without a predecessor -> dummy rows.
    // x has a predecessor iff some y has y[1:] == x[:-1].  In key space
    // P(x) = key(x) << 2 (drop the last char) and S(y) = key(y) with digit k-1 cleared.
    // The k-mers ending with a given char form a contiguous range in which P(x) is
    // ascending, and S(.) is ascending over all k-mers: four linear merge-joins.
    std::vector<Row<W>> dummies;
    dummies.push_back(Row<W>{key_zero<W>(), 0}); // root $^k, always a row
    {
        size_t range_start[5];
        for (unsigned c = 0; c < 4; c++) {
            Key<W> lo = key_zero<W>();
            set_digit(lo, 0, c);
            range_start[c] = (size_t)(std::lower_bound(kmers.begin(), kmers.end(), lo) - kmers.begin());
        }
        range_start[4] = N;
        std::vector<std::vector<size_t>> orphans(4);
        parallel_for(4, nt, [&](size_t c) {
            size_t y = 0;
            for (size_t x = range_start[c]; x < range_start[c + 1]; x++) {
                Key<W> P = shl_bits(kmers[x], 2);
                bool found = false;
                if (k == 1) found = true;
                while (!found && y < N) {
                    Key<W> S = kmers[y];
                    set_digit(S, k - 1, 0);
                    if (S < P) { y++; continue; }
                    found = (S == P);
                    break;
                }
                if (!found) orphans[c].push_back(x);
            }
        });
        for (unsigned c = 0; c < 4; c++)
            for (size_t x : orphans[c])
                for (uint32_t j = 1; j < k; j++) // $^(k-j) x[0..j): the j first chars of x
                    dummies.push_back(Row<W>{shl_bits(kmers[x], 2 * (k - j)), j});
    }
    std::sort(dummies.begin(), dummies.end());
    dummies.erase(std::unique(dummies.begin(), dummies.end()), dummies.end());

    // ---- 3. merge into colex row order
    const size_t n = N + dummies.size();
    if (n >= 0xFFFFFFF0ull) throw std::runtime_error("n_sets >= 2^32: 64-bit positions not built yet");
    std::vector<Key<W>> rkey(n);
    std::vector<uint8_t> rreal(n);
    {
        size_t a = 0, b = 0, o = 0;
        while (a < N || b < dummies.size()) {
            bool take_dummy;
            if (a == N) take_dummy = true;
            else if (b == dummies.size()) take_dummy = false;
            else take_dummy = dummies[b] < Row<W>{kmers[a], k};
            if (take_dummy) { rkey[o] = dummies[b].key; rreal[o] = (uint8_t)dummies[b].real; b++; }
            else { rkey[o] = kmers[a]; rreal[o] = (uint8_t)k; a++; }
            o++;
        }
    }
    std::vector<Key<W>>().swap(kmers);

    out.k = k;
    out.n_sets = n;
    out.n_kmers = N;
    const size_t nw = (n + 63) / 64;
    for (int c = 0; c < 4; c++) out.rows[c].assign(nw, 0);
    out.lcs.assign(n, 0);

    // ---- 4. edge bits.  Rows ending with c are the contiguous range [first[c], first[c+1]);
    // walking them in order against the (k-1)-suffix groups in order is a merge-join:
    // row y receives its single incoming edge from the first row of the group whose
    // (k-1)-suffix equals y[:-1].
    size_t first[5];
    first[0] = 1; // row 0 is the root ($ as last char)
    for (unsigned c = 1; c < 4; c++) {
        Key<W> lo = key_zero<W>();
        set_digit(lo, 0, c);
        // first row with key >= lo and real >= 1
        size_t a = 1, b = n;
        while (a < b) {
            size_t m = a + (b - a) / 2;
            if (rkey[m] < lo) a = m + 1; else b = m;
        }
        first[c] = a;
    }
    first[4] = n;
    // root-only corner: rows with digit0 == 0 (A) and real >= 1 start at 1 by construction.
    auto suffix_frame = [&](size_t z, Key<W> &key, uint32_t &real) { // z[1:]
        key = rkey[z];
        set_digit(key, k - 1, 0);
        real = std::min<uint32_t>(rreal[z], k - 1);
    };
    parallel_for(4, nt, [&](size_t c) {
        size_t g = 0; // current group start
        Key<W> gk; uint32_t gr;
        suffix_frame(0, gk, gr);
        for (size_t y = first[c]; y < first[c + 1]; y++) {
            Key<W> P = shl_bits(rkey[y], 2);
            uint32_t pr = rreal[y] - 1;
            // advance to the group whose frame == (P, pr)
            for (;;) {
                if (gk == P && gr == pr) break;
                // next group start
                size_t z = g + 1;
                Key<W> zk; uint32_t zr;
                for (;; z++) {
                    if (z >= n) throw std::runtime_error("sbwt build: row without incoming edge");
                    suffix_frame(z, zk, zr);
                    if (zk != gk || zr != gr) break;
                }
                g = z; gk = zk; gr = zr;
            }
            out.rows[c][g >> 6] |= 1ull << (g & 63); // distinct c -> distinct vectors, no race
        }
    });
    if (k == 1) { /* every row is in the single empty-suffix group: handled above */ }

    uint64_t acc = 1;
    for (int c = 0; c < 4; c++) {
        out.C[c] = acc;
        for (size_t w = 0; w < nw; w++) acc += (uint64_t)__builtin_popcountll(out.rows[c][w]);
    }
    if (acc != n) throw std::runtime_error("sbwt build: edge count != n_sets - 1");
    for (int c = 0; c < 4; c++)
        if (first[c] != out.C[c] && first[c] < first[c + 1])
            throw std::runtime_error("sbwt build: C array inconsistent with row order");

    // ---- 5. LCS
    const size_t chunk = 1 << 16;
    parallel_for((n + chunk - 1) / chunk, nt, [&](size_t t) {
        size_t a = std::max<size_t>(1, t * chunk), b = std::min(n, (t + 1) * chunk);
        for (size_t i = a; i < b; i++) {
            unsigned cd = common_digits(rkey[i], rkey[i - 1]);
            unsigned m = std::min<unsigned>(rreal[i], rreal[i - 1]);
            out.lcs[i] = (uint8_t)std::min(cd, m);
        }
    });
}

} // namespace

void build_host_index(const uint8_t *const *seqs, const size_t *lens, size_t n_seqs,
                      const BuildParams &p, HostIndex &out)
{
    if (p.k == 0 || p.k > 255) throw std::runtime_error("k must be in 1..255");
    if (p.k <= 32) build_impl<1>(seqs, lens, n_seqs, p, out);
    else if (p.k <= 64) build_impl<2>(seqs, lens, n_seqs, p, out);
    else if (p.k <= 128) build_impl<4>(seqs, lens, n_seqs, p, out);
    else build_impl<8>(seqs, lens, n_seqs, p, out);
}

void make_device_layout(const HostIndex &h, DeviceLayout &out, bool with_pairs)
{
    const uint64_t n = h.n_sets;
    out.n_blocks = n / kRankRowsPerBlock + 2;
    auto bit = [&](int c, uint64_t i) -> uint32_t {
        return i < n ? (uint32_t)((h.rows[c][i >> 6] >> (i & 63)) & 1) : 0u;
    };
    for (int c = 0; c < 4; c++) {
        out.rank[c].assign(out.n_blocks * 4, 0);
        uint64_t cum = h.C[c];
        for (uint64_t b = 0; b < out.n_blocks; b++) {
            uint32_t *blk = &out.rank[c][b * 4];
            blk[0] = (uint32_t)cum;
            for (unsigned w = 0; w < 3; w++) {
                uint32_t v = 0;
                uint64_t base = b * kRankRowsPerBlock + 32 * w;
                if (base < n)
                    for (unsigned j = 0; j < 32; j++) v |= bit(c, base + j) << j;
                blk[1 + w] = v;
                cum += (uint64_t)__builtin_popcount(v);
            }
        }
    }
    // previous / next strictly-smaller LCS value (monotone stack, O(n))
    out.ent.assign(3 * (n + 1) + 4, 0);
    auto lcs_at = [&](uint64_t i) -> uint32_t { return i < n ? h.lcs[i] : 0u; };
    std::vector<uint32_t> stack;
    stack.reserve(256);
    for (uint64_t i = 0; i <= n; i++) {
        const uint32_t v = lcs_at(i);
        while (!stack.empty() && lcs_at(stack.back()) >= v) stack.pop_back();
        out.ent[3 * i + 0] = v;
        out.ent[3 * i + 1] = stack.empty() ? 0u : stack.back();
        stack.push_back((uint32_t)i);
    }
    stack.clear();
    for (uint64_t ii = n + 1; ii-- > 0;) {
        const uint32_t v = lcs_at(ii);
        while (!stack.empty() && lcs_at(stack.back()) >= v) stack.pop_back();
        out.ent[3 * ii + 2] = stack.empty() ? (uint32_t)n : stack.back();
        stack.push_back((uint32_t)ii);
    }
    out.pair.clear();
    if (!with_pairs) return;
    // two-base extension: the set bits of B_c1, in row order, map one to one onto the rows
    // C[c1], C[c1]+1, ... (the extend-right bijection), so one sweep per c1 with a running target row
    out.pair.assign(16 * out.n_blocks * 4, 0);
    std::vector<std::thread> th;
    for (int c1 = 0; c1 < 4; c1++)
        th.emplace_back([&, c1] {
            uint64_t target = h.C[c1];
            for (uint64_t i = 0; i < n; i++) {
                if (!bit(c1, i)) continue;
                for (int c2 = 0; c2 < 4; c2++)
                    if (bit(c2, target)) {
                        uint32_t *blk = &out.pair[((uint64_t)(4 * c1 + c2) * out.n_blocks + i / kRankRowsPerBlock) * 4];
                        const unsigned o = (unsigned)(i % kRankRowsPerBlock);
                        blk[1 + o / 32] |= 1u << (o % 32);
                    }
                target++;
            }
            for (int c2 = 0; c2 < 4; c2++) {
                uint64_t cum = h.C[c2];
                for (uint64_t j = 0; j < h.C[c1]; j++) cum += bit(c2, j); // rank_c2(C[c1])
                for (uint64_t b = 0; b < out.n_blocks; b++) {
                    uint32_t *blk = &out.pair[((uint64_t)(4 * c1 + c2) * out.n_blocks + b) * 4];
                    blk[0] = (uint32_t)cum;
                    cum += (uint64_t)(__builtin_popcount(blk[1]) + __builtin_popcount(blk[2]) + __builtin_popcount(blk[3]));
                }
            }
        });
    for (auto &t : th) t.join();
}

// ---- flat index file: magic, k, n_sets, n_kmers, C[4], rows[4], lcs [, "KBOPCOV1", text, pos, node_at] -------------
// The optional tail is the path cover of the plan-guided walk (9 bytes per row): laying the chains out is one long
// pointer chase (26 s per 10^8 rows, 13 minutes for a human genome), the one part of a device copy worth keeping on disk;
// rank blocks, contraction entries, two-base blocks, recovery lines and the seed table are streaming passes.
static const char kMagic[8] = {'K', 'B', 'O', 'H', 'I', 'P', '0', '1'};
static const char kCoverTag[8] = {'K', 'B', 'O', 'P', 'C', 'O', 'V', '1'};

void save_host_index(const HostIndex &h, const std::string &path, const PathCover *cover)
{
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open " + path + " for writing");
    uint64_t hdr[7] = {h.k, h.n_sets, h.n_kmers, h.C[0], h.C[1], h.C[2], h.C[3]};
    bool ok = std::fwrite(kMagic, 1, 8, f) == 8 && std::fwrite(hdr, 8, 7, f) == 7;
    size_t nw = (h.n_sets + 63) / 64;
    for (int c = 0; c < 4 && ok; c++) ok = std::fwrite(h.rows[c].data(), 8, nw, f) == nw;
    ok = ok && std::fwrite(h.lcs.data(), 1, h.n_sets, f) == h.n_sets;
    if (ok && cover) {
        const size_t n = h.n_sets;
        ok = cover->text.size() == n + 2 * PathCover::kPad && cover->pos.size() == n && cover->node_at.size() == n &&
             std::fwrite(kCoverTag, 1, 8, f) == 8 && std::fwrite(cover->text.data() + PathCover::kPad, 1, n, f) == n &&
             std::fwrite(cover->pos.data(), 4, n, f) == n && std::fwrite(cover->node_at.data(), 4, n, f) == n;
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("short write to " + path);
}

// A cover read from a file is trusted no further than the walk can check for itself: positions and rows must be inverse
// permutations of each other and every claimed edge node_at[p-1] -> node_at[p] labelled text[p] must be in the subset
// matrix (the same test tests/test_path_cover.py runs); anything else is an inconsistent file.
void validate_path_cover(const HostIndex &h, const PathCover &pc)
{
    auto bad = [](const char *what) { throw std::runtime_error(std::string("inconsistent path cover: ") + what); };
    const uint64_t n = h.n_sets;
    if (pc.text.size() != n + 2 * PathCover::kPad || pc.pos.size() != n || pc.node_at.size() != n) bad("array sizes");
    const uint8_t *text = pc.text.data() + PathCover::kPad;
    for (uint64_t p = 0; p < n; p++) {
        const uint32_t u = pc.node_at[p];
        if (u >= n || pc.pos[u] != p) bad("pos / node_at are not inverse permutations");
    }
    // edge check: the successor of row u by c is C[c] + rank_c(first row of u's (k-1)-suffix group); one streaming pass over
    // the rows gives every row its group's first row, a second one checks the claimed edges through a rank directory
    std::vector<uint32_t> cum[4];
    for (int c = 0; c < 4; c++) {
        cum[c].resize(h.rows[c].size() + 1);
        uint32_t a = 0;
        for (size_t w = 0; w < h.rows[c].size(); w++) {
            cum[c][w] = a;
            a += (uint32_t)__builtin_popcountll(h.rows[c][w]);
        }
        cum[c][h.rows[c].size()] = a;
    }
    auto rank = [&](int c, uint64_t i) -> uint64_t {
        const uint64_t w = i >> 6, o = i & 63;
        return cum[c][w] + (o ? (uint64_t)__builtin_popcountll(h.rows[c][w] & ((1ull << o) - 1)) : 0);
    };
    std::vector<uint32_t> first(n);
    for (uint64_t i = 0, f0 = 0; i < n; i++) {
        if (i == 0 || h.lcs[i] + 1u < h.k) f0 = i;
        first[i] = (uint32_t)f0;
    }
    for (uint64_t p = 0; p < n; p++) {
        const uint8_t t = text[p];
        if (t == 0) continue;
        const int c = t == 'A' ? 0 : t == 'C' ? 1 : t == 'G' ? 2 : t == 'T' ? 3 : -1;
        if (c < 0 || p == 0) bad("text byte");
        const uint64_t f0 = first[pc.node_at[p - 1]];
        if (!((h.rows[c][f0 >> 6] >> (f0 & 63)) & 1)) bad("claimed edge is not in the subset matrix");
        // the rows of a group share its successors: the one claimed must be among the group's successors by c
        // (exactly one successor per (group, c): C[c] + rank_c(f0))
        if (pc.node_at[p] != h.C[c] + rank(c, f0)) bad("claimed edge leads elsewhere");
    }
}

// Consistency of an index that did not come out of build_host_index (a file, kbo_index_from_parts): the walk kernels
// trust C[] and the rank data to keep every interval inside [0, n_sets].
void validate_host_index(const HostIndex &h)
{
    auto bad = [](const char *what) { throw std::runtime_error(std::string("inconsistent index: ") + what); };
    if (h.k < 1 || h.k > 255) bad("k outside 1..255");
    if (h.n_sets < 1) bad("n_sets == 0");
    const size_t nw = (h.n_sets + 63) / 64;
    uint64_t acc = 1;
    for (int c = 0; c < 4; c++) {
        if (h.rows[c].size() != nw) bad("subset-matrix row of the wrong length");
        if (h.n_sets & 63)
            if (h.rows[c][nw - 1] >> (h.n_sets & 63)) bad("bits set beyond n_sets");
        if (h.C[c] != acc) bad("C[c] != 1 + number of edge bits of smaller characters");
        for (size_t w = 0; w < nw; w++) acc += (uint64_t)__builtin_popcountll(h.rows[c][w]);
    }
    if (acc != h.n_sets) bad("edge bits != n_sets - 1");
    if (h.lcs.size() != h.n_sets) bad("LCS array of the wrong length");
    if (h.lcs[0] != 0) bad("LCS[0] != 0");
    for (uint64_t i = 0; i < h.n_sets; i++)
        if (h.lcs[i] >= h.k) bad("LCS value >= k");
    if (h.n_kmers > h.n_sets) bad("n_kmers > n_sets");
}

void load_host_index(const std::string &path, HostIndex &h, PathCover *cover, bool *have_cover)
{
    if (have_cover) *have_cover = false;
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    char magic[8];
    uint64_t hdr[7];
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, kMagic, 8) == 0 &&
              std::fread(hdr, 8, 7, f) == 7;
    if (ok) {
        h.k = (uint32_t)hdr[0]; h.n_sets = hdr[1]; h.n_kmers = hdr[2];
        for (int c = 0; c < 4; c++) h.C[c] = hdr[3 + c];
        if (hdr[0] < 1 || hdr[0] > 255 || h.n_sets < 1 || h.n_sets >= (1ull << 40)) { // (sizes come from the file)
            std::fclose(f);
            throw std::runtime_error("bad index file " + path + ": header out of range");
        }
        size_t nw = (h.n_sets + 63) / 64;
        for (int c = 0; c < 4 && ok; c++) {
            h.rows[c].resize(nw);
            ok = std::fread(h.rows[c].data(), 8, nw, f) == nw;
        }
        h.lcs.resize(h.n_sets);
        ok = ok && std::fread(h.lcs.data(), 1, h.n_sets, f) == h.n_sets;
    }
    bool with_cover = false;
    if (ok) { // nothing may follow the LCS array but a path cover
        char tag[8];
        const size_t got = std::fread(tag, 1, 8, f);
        if (got == 8 && std::memcmp(tag, kCoverTag, 8) == 0) {
            const size_t n = h.n_sets;
            PathCover tmp, &pc = cover ? *cover : tmp;
            pc.text.assign(n + 2 * PathCover::kPad, 0);
            pc.pos.resize(n);
            pc.node_at.resize(n);
            ok = std::fread(pc.text.data() + PathCover::kPad, 1, n, f) == n && std::fread(pc.pos.data(), 4, n, f) == n &&
                 std::fread(pc.node_at.data(), 4, n, f) == n && std::fgetc(f) == EOF;
            with_cover = ok && cover != nullptr;
        } else {
            ok = got == 0;
        }
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("bad or truncated index file " + path);
    validate_host_index(h);
    if (with_cover) {
        validate_path_cover(h, *cover);
        if (have_cover) *have_cover = true;
    }
}

// ---- <prefix>.sbwt / <prefix>.lcs, the file pair of index::serialize_sbwt / load_sbwt (reference index.rs:128-151,
// 195-212).  What the reference itself writes is pinned: the u64-LE length 12 and the tag "SubsetMatrix" in front of the
// .sbwt file (index.rs:139-140).  What follows is written by the sbwt crate's own `serialize` (index.rs:143, 150), whose
// field layout is not in the reference tree and could not be checked here (no crate source, no sample file): PARITY
// UNPINNED.  This implementation therefore writes its own payload behind the pinned header, marked by a second tag, and
// refuses - loudly, with KBO_E_UNSUPPORTED - a payload it did not write instead of guessing at the crate's fields; an
// index built by kbo-cli comes in through kbo_index_from_parts.
static const char kSbwtTag[12] = {'S', 'u', 'b', 's', 'e', 't', 'M', 'a', 't', 'r', 'i', 'x'};
static const char kOwnSbwt[8] = {'K', 'B', 'O', 'S', 'B', 'W', 'T', '1'};
static const char kOwnLcs[8] = {'K', 'B', 'O', 'L', 'C', 'S', '0', '1'};

// The pair goes to <prefix>.sbwt.kbohip + <prefix>.lcs.kbohip, NOT to the reference's <prefix>.sbwt / <prefix>.lcs: a file
// with the SubsetMatrix tag under the upstream name looks like a crate-written index to kbo-cli, whose loader would read
// this payload as the crate's fields (garbage-sized allocations, a panic) instead of refusing it.
void save_sbwt_pair(const HostIndex &h, const std::string &prefix)
{
    const std::string sp = prefix + ".sbwt.kbohip", lp = prefix + ".lcs.kbohip";
    FILE *f = std::fopen(sp.c_str(), "wb");
    if (!f) throw std::runtime_error("Expected write access to " + sp);
    const uint64_t taglen = 12, hdr[7] = {h.k, h.n_sets, h.n_kmers, h.C[0], h.C[1], h.C[2], h.C[3]};
    const size_t nw = (h.n_sets + 63) / 64;
    bool ok = std::fwrite(&taglen, 8, 1, f) == 1 && std::fwrite(kSbwtTag, 1, 12, f) == 12 &&
              std::fwrite(kOwnSbwt, 1, 8, f) == 8 && std::fwrite(hdr, 8, 7, f) == 7;
    for (int c = 0; c < 4 && ok; c++) ok = std::fwrite(h.rows[c].data(), 8, nw, f) == nw;
    std::fclose(f);
    if (!ok) throw std::runtime_error("short write to " + sp);
    f = std::fopen(lp.c_str(), "wb");
    if (!f) throw std::runtime_error("Expected write access to " + lp);
    const uint64_t n = h.n_sets;
    ok = std::fwrite(kOwnLcs, 1, 8, f) == 8 && std::fwrite(&n, 8, 1, f) == 1 && std::fwrite(h.lcs.data(), 1, n, f) == n;
    std::fclose(f);
    if (!ok) throw std::runtime_error("short write to " + lp);
}

// returns false (h untouched) when the .sbwt payload was not written by save_sbwt_pair
// (looks for <prefix>.sbwt.kbohip first, then for <prefix>.sbwt: a pair this library wrote under the upstream names
// before they were moved aside is still read; a crate-written <prefix>.sbwt is reported as such)
bool load_sbwt_pair(const std::string &prefix, HostIndex &h)
{
    std::string sp = prefix + ".sbwt.kbohip", lp = prefix + ".lcs.kbohip";
    FILE *f = std::fopen(sp.c_str(), "rb");
    if (!f) {
        sp = prefix + ".sbwt";
        lp = prefix + ".lcs";
        f = std::fopen(sp.c_str(), "rb");
    }
    if (!f) throw std::runtime_error("Expected SBWT at " + prefix + ".sbwt.kbohip (or a crate-written " + sp + ")");
    uint64_t taglen = 0, hdr[7];
    char tag[12], own[8];
    bool ok = std::fread(&taglen, 8, 1, f) == 1 && taglen == 12 && std::fread(tag, 1, 12, f) == 12 &&
              std::memcmp(tag, kSbwtTag, 12) == 0;
    if (!ok) {
        std::fclose(f);
        throw std::runtime_error(sp + ": not an .sbwt file (header is not the SubsetMatrix tag, index.rs:139-140)");
    }
    if (std::fread(own, 1, 8, f) != 8 || std::memcmp(own, kOwnSbwt, 8) != 0) {
        std::fclose(f);
        return false;
    }
    HostIndex t;
    ok = std::fread(hdr, 8, 7, f) == 7 && hdr[0] >= 1 && hdr[0] <= 255 && hdr[1] >= 1 && hdr[1] < (1ull << 40);
    if (ok) {
        t.k = (uint32_t)hdr[0]; t.n_sets = hdr[1]; t.n_kmers = hdr[2];
        for (int c = 0; c < 4; c++) t.C[c] = hdr[3 + c];
        const size_t nw = (t.n_sets + 63) / 64;
        for (int c = 0; c < 4 && ok; c++) {
            t.rows[c].resize(nw);
            ok = std::fread(t.rows[c].data(), 8, nw, f) == nw;
        }
        ok = ok && std::fgetc(f) == EOF;
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("bad or truncated index file " + sp);
    f = std::fopen(lp.c_str(), "rb");
    if (!f) throw std::runtime_error("Expected LCS array at " + lp);
    uint64_t n = 0;
    ok = std::fread(own, 1, 8, f) == 8 && std::memcmp(own, kOwnLcs, 8) == 0 && std::fread(&n, 8, 1, f) == 1 && n == t.n_sets;
    if (ok) {
        t.lcs.resize(n);
        ok = std::fread(t.lcs.data(), 1, n, f) == n && std::fgetc(f) == EOF;
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("bad, truncated or mismatched LCS file " + lp);
    validate_host_index(t);
    h = std::move(t);
    return true;
}

} // namespace kbo
