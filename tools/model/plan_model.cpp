// plan_model.cpp — CPU model of the plan-guided walk (development tool, not product, not oracle).
//
//   K1  (seed + compare): find a text diagonal for every item, compare the item with the path-cover text,
//       write the predicted MS values and the list of mismatch positions ("plan")
//   K2  (guided walk): the ordinary extend/contract walk, which jumps over the stretches the plan predicts
//       once its own state proves it is on the diagonal
// The model runs both, lane by lane, on the host, checks every MS byte against the literal walk and counts the
// iterations each design needs.  Build: make -C tools/model ; run: tools/model/plan_model [genome] [reads] [sub/65536]
#include "../../kbo_amd/csrc/sbwt_index.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" void kbo_synth_genome(uint64_t seed, uint8_t *out, uint64_t len);
extern "C" void kbo_synth_reads(uint64_t seed, const uint8_t *genome, uint64_t genome_len, uint64_t first_read,
                                uint64_t n_reads, uint32_t read_len, uint32_t sub_per_65536, uint8_t *out);
using namespace kbo;

static uint64_t rng_state = 88172645463325252ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

struct Model {
    const HostIndex &h;
    PathCover pc;
    std::vector<uint32_t> cum[4]; // rank directory: set bits before word w
    uint32_t n, k;
    explicit Model(const HostIndex &hi) : h(hi)
    {
        n = (uint32_t)h.n_sets;
        k = h.k;
        for (int c = 0; c < 4; c++) {
            cum[c].resize(h.rows[c].size() + 1);
            uint32_t a = 0;
            for (size_t w = 0; w < h.rows[c].size(); w++) {
                cum[c][w] = a;
                a += (uint32_t)__builtin_popcountll(h.rows[c][w]);
            }
            cum[c][h.rows[c].size()] = a;
        }
        make_path_cover(h, pc);
    }
    // ---- shallow tables: presence of every string of 1..D bases (as suffix of a row) and the interval of every D-mer
    uint32_t D = 0;
    std::vector<std::vector<uint8_t>> present; // present[t][code]
    std::vector<uint32_t> tabD;                // {l, r} per D-mer
    void build_shallow(uint32_t d_max)
    {
        D = std::min(d_max, k);
        present.assign(D + 1, {});
        std::vector<uint32_t> cur{0u, n}, nxt;
        for (uint32_t t = 1; t <= D; t++) {
            nxt.assign(cur.size() * 4, 0);
            present[t].assign(cur.size() / 2 * 4, 0);
            for (size_t p = 0; p < cur.size() / 2; p++)
                for (int c = 0; c < 4; c++) {
                    const uint32_t l = cur[2 * p], r = cur[2 * p + 1];
                    uint32_t l2 = 0, r2 = 0;
                    if (l < r) { l2 = (uint32_t)h.C[c] + rank(c, l); r2 = (uint32_t)h.C[c] + rank(c, r); }
                    nxt[2 * (4 * p + c)] = l2; nxt[2 * (4 * p + c) + 1] = r2;
                    present[t][4 * p + c] = l2 < r2;
                }
            cur.swap(nxt);
        }
        tabD = cur;
    }
    uint32_t rank(int c, uint32_t i) const
    {
        uint32_t w = i >> 6, o = i & 63;
        uint32_t r = cum[c][w];
        if (o) r += (uint32_t)__builtin_popcountll(h.rows[c][w] & ((1ull << o) - 1));
        return r;
    }
    static int code(uint8_t ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1; }
    // one base of the reference loop; returns number of (fails, climbs) through the counters
    struct St { uint32_t l, r, d; };
    void step(St &s, uint8_t ch, uint64_t &fails, uint64_t &climbs) const
    {
        const int c = code(ch);
        auto ext = [&](uint32_t l, uint32_t r, uint32_t &l2, uint32_t &r2) {
            if (c < 0) { l2 = r2 = 0; return; }
            l2 = (uint32_t)h.C[c] + rank(c, l);
            r2 = (uint32_t)h.C[c] + rank(c, r);
        };
        uint32_t l2, r2;
        ext(s.l, s.r, l2, r2);
        bool failed = false;
        while (s.d > 0 && l2 >= r2) {
            failed = true;
            s.d--;
            uint32_t ol = s.l, orr = s.r;
            while (s.l > 0 && h.lcs[s.l] >= s.d) s.l--;
            while (s.r < n && h.lcs[s.r] >= s.d) s.r++;
            if (ol != s.l || orr != s.r) climbs++;
            ext(s.l, s.r, l2, r2);
        }
        if (failed) fails++;
        if (l2 < r2) { s.l = l2; s.r = r2; s.d = s.d + 1 < k ? s.d + 1 : k; }
    }
};

struct Item { uint64_t start; uint32_t len, warm; };
constexpr int kPlanCap = 12;
struct Plan {
    int64_t p0;      // text position of item base 0 (valid when have)
    bool have;
    uint32_t j_conv; // K1 walked bases [0, j_conv) exactly and ended on a single row on the diagonal (0: no)
    uint32_t n_mm;   // mismatches found (may exceed the list)
    uint16_t mm[kPlanCap];
};

struct Counters {
    uint64_t k1_seed_iters = 0, k1_wave_seed_iters = 0, k1_blocks = 0, unseeded = 0, clean = 0, bogus = 0;
    uint64_t k2_accept = 0, k2_fail = 0, k2_climb = 0, k2_jumps = 0, k2_pass = 0, k2_done_early = 0, overflow = 0;
    uint64_t ref_accept = 0, ref_fail = 0, ref_climb = 0;
};

static uint32_t g_dmin = 14, g_seed_cap = 40;

// ---- K1 for one item: seed (restart on failure), compare, predict
static void k1_item(const Model &M, const uint8_t *q, const Item &it, uint8_t *ms, Plan &pl, Counters &C, uint32_t &seed_iters)
{
    const uint8_t *text = M.pc.text.data() + PathCover::kPad;
    pl.have = false; pl.j_conv = 0; pl.n_mm = 0; pl.p0 = 0;
    Model::St s{0, M.n, 0};
    bool clean = true;
    uint32_t j = 0;
    const uint32_t dmin = M.k < g_dmin ? M.k : g_dmin;
    seed_iters = 0;
    for (; j < it.len && j < g_seed_cap; j++) {
        seed_iters++;
        const int c = Model::code(q[it.start + j]);
        uint32_t l2 = 0, r2 = 0;
        if (c >= 0) { l2 = (uint32_t)M.h.C[c] + M.rank(c, s.l); r2 = (uint32_t)M.h.C[c] + M.rank(c, s.r); }
        if (l2 >= r2 && s.d > 0) { // restart from the root with the same base
            clean = false;
            s = {0, M.n, 0};
            if (c >= 0) { l2 = (uint32_t)M.h.C[c] + M.rank(c, 0); r2 = (uint32_t)M.h.C[c] + M.rank(c, M.n); }
        }
        if (l2 < r2) { s.l = l2; s.r = r2; s.d = s.d + 1 < M.k ? s.d + 1 : M.k; }
        else { clean = false; s = {0, M.n, 0}; }
        if (s.r == s.l + 1 && s.d >= dmin) { pl.have = true; break; }
    }
    if (!pl.have) { C.unseeded++; return; }
    const uint32_t j0 = j;
    pl.p0 = (int64_t)M.pc.pos[s.l] - (int64_t)j0;
    // compare + predict
    int64_t i_last = -1;
    bool mism_before_j0 = false;
    for (uint32_t t = 0; t < it.len; t++) {
        const int64_t tp = pl.p0 + t;
        const uint8_t tc = (tp >= 0 && tp < (int64_t)M.n) ? text[tp] : 0;
        const bool match = tc != 0 && tc == q[it.start + t];
        if (!match) {
            if (pl.n_mm < kPlanCap) pl.mm[pl.n_mm] = (uint16_t)t;
            pl.n_mm++;
            i_last = t;
            if (t <= j0) mism_before_j0 = true;
        }
        const int64_t dp = (int64_t)t - i_last;
        if (t >= it.warm) ms[it.start + t] = (uint8_t)(dp < (int64_t)M.k ? dp : M.k);
    }
    C.k1_blocks += (it.len + 15) / 16;
    if (clean && !mism_before_j0) { pl.j_conv = j0 + 1; C.clean++; }
    if (pl.n_mm > it.len / 4) C.bogus++;
}

// ---- units: stretches of an item that are walked (plan_emit_kernel).  Mismatches closer than `gap` share a unit;
// a unit starts on the diagonal in front of its first mismatch (or at the root, for the head of an item whose first
// bases the plan kernel did not walk exactly) and runs until the walk converges after its last mismatch.  If it
// reaches the next unit's first mismatch (`bound`) without converging, the next unit's start state was a wrong guess:
// the item is flagged and walked again in full.
struct Unit {
    uint32_t item;
    uint32_t pos, out_from, bound;
    int32_t last_mm;
    uint32_t d_start;
    bool head, plain, to_end;
};
static uint32_t g_gap = 24, g_chunk = 64;

static void form_units(const Model &M, const Item &it, const Plan &pl, uint32_t item, std::vector<Unit> &out)
{
    const uint32_t k = M.k;
    if (it.len == 0) return;
    if (!pl.have || pl.n_mm > (uint32_t)kPlanCap) { // no plan: chunks with k-1 warm-up bases from the root
        for (uint32_t c0 = it.warm; c0 < it.len; c0 += g_chunk) {
            Unit u;
            u.item = item;
            u.out_from = c0;
            u.pos = c0 > (k - 1) ? c0 - (k - 1) : 0;
            u.bound = std::min(it.len, c0 + g_chunk);
            u.last_mm = -1; u.d_start = 0; u.head = true; u.plain = true; u.to_end = u.bound == it.len;
            out.push_back(u);
        }
        if (it.warm >= it.len) {} // (cannot happen: warm < len)
        return;
    }
    uint32_t t = 0;
    int32_t prev = -1; // last mismatch seen so far
    bool head = pl.j_conv == 0;
    while (head || t < pl.n_mm) {
        Unit u;
        u.item = item;
        u.head = head; u.plain = false;
        if (head) { u.pos = 0; u.d_start = 0; u.last_mm = -1; }
        else {
            u.pos = pl.mm[t];
            const int64_t dp = (int64_t)u.pos - 1 - prev;
            u.d_start = (uint32_t)(dp < (int64_t)k ? dp : k);
            prev = pl.mm[t]; u.last_mm = prev; t++;
        }
        head = false;
        while (t < pl.n_mm && (int32_t)pl.mm[t] - (u.last_mm) < (int32_t)g_gap) { prev = pl.mm[t]; u.last_mm = prev; t++; }
        u.bound = t < pl.n_mm ? pl.mm[t] : it.len;
        u.to_end = t >= pl.n_mm;
        u.out_from = std::max(it.warm, u.pos);
        out.push_back(u);
    }
}

static void k2_unit(const Model &M, const uint8_t *q, const Item &it, const Plan &pl, const Unit &u, uint8_t *ms,
                    uint8_t *redo, Counters &C)
{
    const uint32_t k = M.k;
    Model::St s{0, M.n, 0};
    if (!u.head) {
        s.l = M.pc.node_at[(uint32_t)(pl.p0 + u.pos - 1)];
        s.r = s.l + 1;
        s.d = u.d_start;
    }
    uint32_t i = u.pos, ocur = 0, wlo = u.out_from >= it.warm ? ((u.out_from - it.warm) & 3u) : 0;
    C.k2_jumps++;
    bool conv = false;
    while (i < u.bound) {
        uint64_t f = 0, cl = 0;
        M.step(s, q[it.start + i], f, cl);
        C.k2_accept++; C.k2_fail += f; C.k2_climb += cl;
        const bool fin = i + 1 == u.bound;
        const uint32_t e = i - it.warm;
        const int64_t dp = (int64_t)i - u.last_mm;
        conv = !u.plain && (int32_t)i >= u.last_mm && s.r == s.l + 1 && s.d == (uint32_t)(dp < (int64_t)k ? dp : k);
        const bool word_done = (e & 3u) == 3u || fin || conv;
        if (i >= u.out_from) {
            ocur |= s.d << (8 * (e & 3u));
            if (word_done) {
                const uint64_t a = it.start + it.warm + (e & ~3u);
                for (uint32_t b = wlo; b <= (e & 3u); b++) ms[a + b] = (uint8_t)(ocur >> (8 * b));
                ocur = 0; wlo = 0;
            }
        }
        i++;
        if (conv) break;
    }
    if (!conv && !u.plain && !u.to_end) { redo[u.item] = 1; C.k2_pass++; }
    if (conv && i < u.bound) C.k2_done_early++;
}

// ---- the hybrid walk of a unit: while the depth is at most D the state is just (d, the last d bases): the next depth
// is the longest suffix of at most d+1 bases that is present (presence is closed under taking substrings, so the
// reference's extend / contract loop ends exactly there); the interval is fetched from the D-mer table when the depth
// reaches D, and from then on the walk is the ordinary one until a contraction takes it below D again.
struct HybridCounters { uint64_t tests = 0, tab = 0, ext = 0, fail = 0, climb = 0, iters = 0; };
static HybridCounters g_hy;
static void k2_unit_hybrid(const Model &M, const uint8_t *q, const Item &it, const Plan &pl, const Unit &u, uint8_t *ms,
                           uint8_t *redo)
{
    const uint32_t k = M.k, D = M.D;
    Model::St s{0, M.n, 0};
    bool shallow = true; // root = shallow with d = 0
    uint64_t hist = 0; // the last bases as 2-bit digits, newest in the low bits (at least D + 1 of them are kept)
    uint32_t vlen = 0; // how many of them are valid (ACGT, inside the item), capped at D + 1
    if (!u.head) {
        s.l = M.pc.node_at[(uint32_t)(pl.p0 + u.pos - 1)];
        s.r = s.l + 1;
        s.d = u.d_start;
        shallow = false;
    }
    // (the last bases before pos are needed when the walk drops into shallow mode: take them from the query)
    auto reload_hist = [&](uint32_t i) { // history = bases [i - D, i) as far as they are ACGT and inside the item
        hist = 0; vlen = 0;
        for (uint32_t t = (i > D + 1 ? i - (D + 1) : 0); t < i; t++) {
            const int c = Model::code(q[it.start + t]);
            if (c < 0) { hist = 0; vlen = 0; } else { hist = (hist << 2) | (uint64_t)c; vlen = std::min(vlen + 1, D + 1); }
        }
    };
    if (!u.head) reload_hist(u.pos); // (a unit that starts at the root must not see the bases in front of it)
    uint32_t i = u.pos;
    bool conv = false;
    while (i < u.bound) {
        const int c = Model::code(q[it.start + i]);
        if (c < 0) { hist = 0; vlen = 0; } else { hist = (hist << 2) | (uint64_t)c; vlen = std::min(vlen + 1, D + 1); }
        if (shallow && s.d == D && D < k && c >= 0) { // the depth has reached the table: fetch the interval of the last D bases (before c)
            const uint32_t key = (uint32_t)((hist >> 2) & ((1ull << (2 * D)) - 1ull)); // (d == D: the D bases before c are valid)
            s.l = M.tabD[2 * key]; s.r = M.tabD[2 * key + 1];
            shallow = false;
            g_hy.tab++; g_hy.iters++;
        }
        if (!shallow) {
            // ordinary step, but a contraction that lands below D switches to shallow mode
            uint32_t l2 = 0, r2 = 0;
            auto ext = [&](uint32_t l, uint32_t r) {
                if (c < 0) { l2 = r2 = 0; return; }
                l2 = (uint32_t)M.h.C[c] + M.rank(c, l); r2 = (uint32_t)M.h.C[c] + M.rank(c, r);
            };
            ext(s.l, s.r); g_hy.ext++; g_hy.iters++;
            bool failed = false;
            while (s.d > 0 && l2 >= r2) {
                failed = true;
                // one level up: m = max(lcs[l], lcs[r]) is where the interval changes next
                const uint32_t lm = s.l < M.n ? M.h.lcs[s.l] : 0, rm = s.r < M.n ? M.h.lcs[s.r] : 0;
                const uint32_t m = std::max(lm, rm);
                g_hy.climb++; g_hy.iters++;
                if (m < D || m == 0) { // lengths m+1 and below are decided by the presence tables
                    shallow = true;
                    s.d = m; // depth before this base, as far as the candidates go
                    break;
                }
                s.d = m;
                while (s.l > 0 && M.h.lcs[s.l] >= s.d) s.l--;
                while (s.r < M.n && M.h.lcs[s.r] >= s.d) s.r++;
                ext(s.l, s.r); g_hy.ext++; g_hy.iters++;
            }
            if (failed) g_hy.fail++;
            if (!shallow) {
                if (l2 < r2) { s.l = l2; s.r = r2; s.d = s.d + 1 < k ? s.d + 1 : k; }
            }
        }
        if (shallow) {
            uint32_t t = std::min(std::min(s.d + 1, std::min(vlen, D)), std::min(D, k));
            g_hy.iters++;
            while (t > 0) {
                g_hy.tests++;
                const uint32_t key = (uint32_t)(hist & ((1ull << (2 * t)) - 1ull));
                if (M.present[t][key]) break;
                t--;
            }
            s.d = t; s.l = 0; s.r = M.n; // (no interval in shallow mode)
        }
        const bool fin = i + 1 == u.bound;
        const uint32_t e = i - it.warm;
        const int64_t dp = (int64_t)i - u.last_mm;
        // convergence needs an interval: in shallow mode at depth D look it up now (it is needed for the next base anyway)
        if (shallow && s.d == D && D < k && !u.plain && (int32_t)i >= u.last_mm && s.d == (uint32_t)(dp < (int64_t)k ? dp : k)) {
            const uint32_t key = (uint32_t)(hist & ((1ull << (2 * D)) - 1ull));
            s.l = M.tabD[2 * key]; s.r = M.tabD[2 * key + 1];
            shallow = false;
            g_hy.tab++; g_hy.iters++;
        }
        conv = !shallow && !u.plain && (int32_t)i >= u.last_mm && s.r == s.l + 1 && s.d == (uint32_t)(dp < (int64_t)k ? dp : k);
        if (i >= u.out_from) ms[it.start + i] = (uint8_t)s.d;
        (void)e; (void)fin;
        i++;
        if (conv) break;
    }
    if (!conv && !u.plain && !u.to_end) redo[u.item] = 1;
}

struct Workload {
    std::vector<uint8_t> q;
    std::vector<Item> items;
};

static Counters g_total;
static uint64_t g_units = 0, g_redone = 0, g_hist[2][16];
static int run(const Model &M, const Workload &W, const char *name, bool verbose)
{
    Counters C;
    std::vector<uint8_t> ms(W.q.size() + 16, 0xEE), ref(W.q.size() + 16, 0xEE);
    std::vector<Plan> plans(W.items.size());
    // reference
    for (const Item &it : W.items) {
        Model::St s{0, M.n, 0};
        for (uint32_t i = 0; i < it.len; i++) {
            M.step(s, W.q[it.start + i], C.ref_fail, C.ref_climb);
            C.ref_accept++;
            if (i >= it.warm) ref[it.start + i] = (uint8_t)s.d;
        }
    }
    // K1 in waves of 64 items (lock-step seed loop: the wave pays for its slowest lane)
    for (size_t w = 0; w < W.items.size(); w += 64) {
        uint32_t mx = 0;
        for (size_t x = w; x < std::min(W.items.size(), w + 64); x++) {
            uint32_t it_iters = 0;
            k1_item(M, W.q.data(), W.items[x], ms.data(), plans[x], C, it_iters);
            C.k1_seed_iters += it_iters;
            mx = std::max(mx, it_iters);
        }
        C.k1_wave_seed_iters += mx;
    }
    std::vector<Unit> units;
    std::vector<uint8_t> redo(W.items.size(), 0);
    for (size_t x = 0; x < W.items.size(); x++) {
        if (plans[x].n_mm > (uint32_t)kPlanCap) C.overflow++;
        form_units(M, W.items[x], plans[x], (uint32_t)x, units);
    }
    uint64_t max_unit = 0, redone = 0;
    for (const Unit &u : units) {
        const uint64_t before = C.k2_accept + C.k2_fail + C.k2_climb;
        if (M.D) { k2_unit_hybrid(M, W.q.data(), W.items[u.item], plans[u.item], u, ms.data(), redo.data()); continue; }
        k2_unit(M, W.q.data(), W.items[u.item], plans[u.item], u, ms.data(), redo.data(), C);
        const uint64_t iters = C.k2_accept + C.k2_fail + C.k2_climb - before;
        max_unit = std::max(max_unit, iters);
        if (verbose) { g_hist[u.plain ? 1 : 0][std::min<uint64_t>(iters / 20, 15)]++; }
    }
    if (verbose) {
        for (int ty = 0; ty < 2; ty++) {
            printf("   %s units by iterations/20:", ty ? "plain" : "group");
            for (int b = 0; b < 16; b++) { printf(" %llu", (unsigned long long)g_hist[ty][b]); g_hist[ty][b] = 0; }
            printf("\n");
        }
    }
    for (size_t x = 0; x < W.items.size(); x++)
        if (redo[x]) { // K3: the flagged items again, in full
            redone++;
            const Item &it = W.items[x];
            Model::St st{0, M.n, 0};
            uint64_t f = 0, cl = 0;
            for (uint32_t i = 0; i < it.len; i++) { M.step(st, W.q[it.start + i], f, cl); if (i >= it.warm) ms[it.start + i] = (uint8_t)st.d; }
        }
    g_units += units.size(); g_redone += redone;
    if (verbose && M.D) {
        const double nu = (double)units.size();
        printf("   hybrid (D=%u) per unit: iterations %.1f = presence rounds + table look-ups %.2f + extends %.1f + climbs %.2f; presence tests %.1f, failed extends %.2f\n",
               M.D, g_hy.iters / nu, g_hy.tab / nu, g_hy.ext / nu, g_hy.climb / nu, g_hy.tests / nu, g_hy.fail / nu);
        g_hy = HybridCounters();
    }
    if (verbose) printf("   units %.2f per item, longest unit %llu iterations, items flagged for a full walk %.3f%%\n",
                        units.size() / (double)W.items.size(), (unsigned long long)max_unit, 100.0 * redone / W.items.size());
    uint64_t bad = 0, first_bad = ~0ull;
    for (const Item &it : W.items)
        for (uint32_t i = it.warm; i < it.len; i++)
            if (ms[it.start + i] != ref[it.start + i]) { if (!bad) first_bad = it.start + i; bad++; }
    g_total.k2_jumps += C.k2_jumps; g_total.k2_done_early += C.k2_done_early; g_total.clean += C.clean;
    g_total.unseeded += C.unseeded; g_total.ref_accept += C.ref_accept; g_total.k2_accept += C.k2_accept; g_total.overflow += C.overflow;
    const double ni = (double)W.items.size();
    if (verbose || bad)
        printf("%-28s items %zu  BAD %llu (first at %llu)\n"
               "   ref: accept %.1f fail %.2f climb %.2f per item (%.1f iterations)\n"
               "   K1: seed iters/item %.1f, lock-step wave iters %.1f, blocks %.1f, unseeded %.2f%%, clean %.1f%%, bogus %.2f%%\n"
               "   K2: accept %.1f fail %.2f climb %.2f per item (%.1f iterations), jumps %.2f pass %.2f early-done %.1f%% overflow %.2f%%\n",
               name, W.items.size(), (unsigned long long)bad, (unsigned long long)first_bad,
               C.ref_accept / ni, C.ref_fail / ni, C.ref_climb / ni, (C.ref_accept + C.ref_fail + C.ref_climb) / ni,
               C.k1_seed_iters / ni, C.k1_wave_seed_iters / (ni / 64), C.k1_blocks / ni, 100.0 * C.unseeded / ni,
               100.0 * C.clean / ni, 100.0 * C.bogus / ni,
               C.k2_accept / ni, C.k2_fail / ni, C.k2_climb / ni, (C.k2_accept + C.k2_fail + C.k2_climb) / ni,
               C.k2_jumps / ni, C.k2_pass / ni, 100.0 * C.k2_done_early / ni, 100.0 * C.overflow / ni);
    return bad ? 1 : 0;
}

static void build(const std::vector<std::string> &seqs, uint32_t k, HostIndex &h)
{
    std::vector<const uint8_t *> p;
    std::vector<size_t> l;
    for (auto &s : seqs) { p.push_back((const uint8_t *)s.data()); l.push_back(s.size()); }
    BuildParams bp;
    bp.k = k;
    bp.num_threads = 8;
    build_host_index(p.data(), l.data(), p.size(), bp, h);
}

int main(int argc, char **argv)
{
    int rc = 0;
    if (argc > 1 && std::string(argv[1]) == "fuzz") {
        // small adversarial indexes: repeats, several contigs, non-ACGT, tiny k; reads with substitutions, indels, junk
        const int rounds = argc > 2 ? atoi(argv[2]) : 300;
        for (int round = 0; round < rounds; round++) {
            const uint32_t k = (uint32_t[]){1, 2, 3, 4, 5, 7, 11, 16, 31, 33, 63}[rnd() % 11];
            const int nseq = 1 + (int)(rnd() % 3);
            const int alpha = 2 + (int)(rnd() % 3); // low-complexity alphabets make repeats
            std::vector<std::string> seqs;
            for (int s = 0; s < nseq; s++) {
                std::string g;
                const size_t len = 20 + rnd() % 3000;
                for (size_t i = 0; i < len; i++) g.push_back("ACGT"[rnd() % alpha]);
                if (rnd() % 2) { // a repeat
                    const size_t a = rnd() % len, b = std::min(len, a + 10 + rnd() % 200);
                    g += g.substr(a, b - a);
                }
                if (rnd() % 3 == 0) g[rnd() % g.size()] = 'N';
                if (rnd() % 4 == 0) g += g; // whole-sequence duplicate (cycles)
                seqs.push_back(g);
            }
            HostIndex h;
            try { build(seqs, k, h); } catch (const std::exception &e) { printf("build failed: %s\n", e.what()); continue; }
            Model M(h);
            if (rnd() % 3) M.build_shallow(1 + rnd() % 8);
            Workload W;
            const int nreads = 100;
            for (int r = 0; r < nreads; r++) {
                const std::string &g = seqs[rnd() % seqs.size()];
                const size_t len = 1 + rnd() % 300;
                std::string rd;
                size_t p = rnd() % g.size();
                while (rd.size() < len) {
                    const unsigned x = rnd() % 1000;
                    if (x < 20) rd.push_back("ACGT"[rnd() % 4]);        // substitution / insertion
                    else if (x < 25) p += rnd() % 5;                    // deletion
                    else if (x < 28) rd.push_back("Nn$a\0x"[rnd() % 6]); // junk
                    else if (x < 30) p = rnd() % g.size();              // chimera
                    if (p >= g.size()) p = rnd() % g.size();
                    rd.push_back(g[p++]);
                }
                rd.resize(len);
                Item it;
                it.start = W.q.size();
                it.len = (uint32_t)len;
                it.warm = (rnd() % 4 == 0) ? (uint32_t)std::min<size_t>(len - 1, (k > 0 ? k - 1 : 0)) : 0;
                if (rnd() % 8 == 0) it.warm = (uint32_t)(rnd() % len);
                W.items.push_back(it);
                W.q.insert(W.q.end(), rd.begin(), rd.end());
            }
            W.q.resize(W.q.size() + 16, 0);
            g_dmin = 2 + rnd() % 14; g_gap = 2 + rnd() % 30; g_chunk = 4 * (1 + rnd() % 20);
            char name[64];
            snprintf(name, sizeof name, "fuzz %d k=%u n=%u", round, k, M.n);
            rc |= run(M, W, name, false);
        }
        printf("units %llu redone items %llu\n", (unsigned long long)g_units, (unsigned long long)g_redone);
        printf("fuzz done rc=%d: jumps %llu early-done %llu clean %llu unseeded %llu overflow %llu ref bases %llu k2 bases %llu\n", rc,
               (unsigned long long)g_total.k2_jumps, (unsigned long long)g_total.k2_done_early, (unsigned long long)g_total.clean,
               (unsigned long long)g_total.unseeded, (unsigned long long)g_total.overflow, (unsigned long long)g_total.ref_accept, (unsigned long long)g_total.k2_accept);
        return rc;
    }
    const uint64_t G = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
    const uint64_t R = argc > 2 ? strtoull(argv[2], 0, 10) : 100000;
    const uint32_t sub = argc > 3 ? (uint32_t)atoi(argv[3]) : 655;
    const uint32_t L = argc > 4 ? (uint32_t)atoi(argv[4]) : 150;
    std::string g(G, 'A');
    kbo_synth_genome(0x6B626F0001ull, (uint8_t *)g.data(), G);
    HostIndex h;
    build({g}, 31, h);
    Model M(h);
    {
        size_t starts = 0;
        const uint8_t *text = M.pc.text.data() + PathCover::kPad;
        for (uint32_t p = 0; p < M.n; p++) starts += text[p] == 0;
        printf("index n=%u, path cover: %zu paths\n", M.n, starts);
    }
    Workload W;
    W.q.resize(R * L + 16);
    kbo_synth_reads(0x6B626F0002ull, (const uint8_t *)g.data(), G, 0, R, L, sub, W.q.data());
    for (uint64_t r = 0; r < R; r++) W.items.push_back({r * L, L, 0});
    if (getenv("GAP")) g_gap = atoi(getenv("GAP"));
    if (getenv("CHUNK")) g_chunk = atoi(getenv("CHUNK"));
    if (getenv("HYB")) const_cast<Model &>(M).build_shallow((uint32_t)atoi(getenv("HYB")));
    for (uint32_t dm : {14u}) {
        g_dmin = dm;
        char name[64];
        snprintf(name, sizeof name, "synthetic dmin=%u", dm);
        rc |= run(M, W, name, true);
    }
    return rc;
}
