/*
 * plan_model.c — TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE (same rules as kbo_oracle.c).
 *
 * CPU model of the PRODUCT's plan-guided A1 stage (kbo_amd/csrc/plan_kernels.hip), not of the reference: what
 * plan_kernel / plan_emit_kernel / ms_walk_guided_kernel / ms_walk_recovery_kernel / the redo pass do to a batch of
 * reads, unit by unit, with the same decisions (seed table, seed restarts, diagonal, mismatch list, unit grouping,
 * convergence test, redo flags, bail-out) and the same iteration structure, on one host thread per slice of reads.
 * Two uses:
 *   1. it produces the MS values of the stage by its own means and the caller compares them with the literal walk
 *      (ora_matching_statistics, the restatement of index.rs:251-252): the skipping argument of plan_kernels.hip's
 *      header is thereby checked on the CPU for the very reads bench.py times;
 *   2. it COUNTS the stage's work on those reads - items, seeds, units, accepted / failed / contraction iterations,
 *      stream and record bytes, distinct 128-byte lines a unit touches - which is what the roofline of bench.py prices
 *      the stage by (B_plan).  tests/test_gpu_model.py pins the counts to the kernels' own.
 *
 * Work items are whole reads (one item per sequence, no warm-up bases): the shape bench.py times.  Chunked long
 * sequences and the call mode are not modelled.
 */
#define _GNU_SOURCE
#include "kbo_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ index view with the product's block geometry */

typedef struct {
    const ora_index *x;
    uint32_t n, k;
    uint64_t C[5];
    const uint64_t *bits[4];
    uint32_t *cum[4]; /* set bits before 64-bit word w */
    const uint8_t *lcs;
    const uint8_t *text; /* path cover: n bytes, 0 where a path starts */
    const uint32_t *pos, *node_at;
    uint32_t seed_d;
    uint32_t *seed_tab; /* {l, r} per string of seed_d bases */
} pm_index;

static inline uint32_t pm_rank(const pm_index *m, int c, uint32_t i)
{
    uint32_t w = i >> 6, o = i & 63u, r = m->cum[c][w];
    if (o) r += (uint32_t)__builtin_popcountll(m->bits[c][w] & ((1ull << o) - 1ull));
    return r;
}
static inline uint32_t pm_word32(const pm_index *m, int c, uint32_t row) /* the 32 row bits around `row` (one word of a rank block) */
{
    if ((uint64_t)(row >> 6) >= ((uint64_t)m->n + 63u) / 64u) return 0;
    const uint64_t w = m->bits[c][row >> 6];
    return (row & 32u) ? (uint32_t)(w >> 32) : (uint32_t)w;
}
static inline int pm_code(uint8_t ch) { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : 4; }
static inline uint32_t pm_lcs(const pm_index *m, uint32_t i) { return i < m->n ? m->lcs[i] : 0u; } /* sentinel at n */
static inline uint32_t pm_psv(const pm_index *m, uint32_t i) /* largest j < i with lcs[j] < lcs[i]; 0 when none */
{
    const uint32_t v = pm_lcs(m, i);
    uint32_t j = i;
    while (j > 0) {
        j--;
        if (pm_lcs(m, j) < v) return j;
    }
    return 0;
}
static inline uint32_t pm_nsv(const pm_index *m, uint32_t i) /* smallest j > i with lcs[j] < lcs[i]; n when none */
{
    const uint32_t v = pm_lcs(m, i);
    for (uint32_t j = i + 1; j <= m->n; j++)
        if (pm_lcs(m, j) < v) return j;
    return m->n;
}

/* ------------------------------------------------------------------ distinct-line bookkeeping of one unit */
/* a unit touches few lines: a small open-addressing set, cleared per unit */
#define PM_SET 512
typedef struct { uint64_t key[PM_SET]; uint32_t used; } pm_set;
static void pm_set_clear(pm_set *s) { memset(s->key, 0xFF, sizeof s->key); s->used = 0; }
static int pm_touch(pm_set *s, uint32_t array, uint64_t byte_off) /* 1 when the line is new to the unit */
{
    if (s->used >= PM_SET / 2) return 0; /* (longer than any unit: stop counting rather than loop) */
    const uint64_t k = ((uint64_t)array << 56) | (byte_off >> 7);
    uint32_t h = (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 55) & (PM_SET - 1);
    while (s->key[h] != ~0ull) {
        if (s->key[h] == k) return 0;
        h = (h + 1) & (PM_SET - 1);
    }
    s->key[h] = k;
    s->used++;
    return 1;
}
enum { PM_A_RANK = 1, PM_A_ENT = 5, PM_A_LINE = 6, PM_A_UNIT = 7, PM_A_NODE = 8, PM_A_Q = 9, PM_A_OUT = 10 };

/* ------------------------------------------------------------------ per-item plan (plan_kernel) */

#define PM_LIST_MAX 29
typedef struct {
    uint32_t seeded, p0, j_conv, n_mm; /* n_mm capped at 254 like the kernel's record */
    uint16_t mm[PM_LIST_MAX];
    uint8_t tab_flagged; /* depth-table form: the table could not resolve the item (dtab_resolve_kernel) */
} pm_plan;

static void pm_plan_item(const pm_index *m, const ora_plan_params *P, const uint8_t *q, uint32_t len, uint8_t *ms,
                         pm_plan *pl, ora_plan_counts *cn)
{
    const uint32_t n = m->n, k = m->k, D = m->seed_d;
    uint32_t dmin = P->seed_depth < k ? P->seed_depth : k;
    if (dmin < 1) dmin = 1;
    uint32_t l = 0, r = n, d = 0, j = 0, j0 = 0;
    int clean = 1, seeded = 0, tab = D != 0;
    memset(pl, 0, sizeof *pl);
    while (!seeded && j < len && j < P->seed_cap) {
        if (tab && j + D <= len) { /* D bases at once from the seed table */
            uint32_t w = 0, okc = 1;
            for (uint32_t t = 0; t < D; t++) {
                const int c = pm_code(q[j + t]);
                okc &= c < 4;
                w = (w << 2) | ((uint32_t)c & 3u);
            }
            uint32_t il = 0, ir = 0;
            if (okc) {
                il = m->seed_tab[2 * (size_t)w];
                ir = m->seed_tab[2 * (size_t)w + 1];
                cn->seed_lookups++;
            }
            if (il < ir) {
                l = il; r = ir; d = D; j += D; tab = 0;
                if (r == l + 1u && d >= dmin) { seeded = 1; j0 = j - 1u; }
            } else {
                clean = 0;
                j += (D + 1u) / 2u;
            }
        } else { /* one extension from the current interval */
            const int c = pm_code(q[j]);
            uint32_t l2 = 0, r2 = 0, dbase = d;
            int again = 0;
            if (c < 4) { l2 = (uint32_t)m->C[c] + pm_rank(m, c, l); r2 = (uint32_t)m->C[c] + pm_rank(m, c, r); }
            cn->seed_extensions++;
            if (l2 >= r2) { /* the seed ends here: start again behind this base (table) or with it (no table) */
                clean = 0;
                dbase = 0;
                again = D != 0;
                l2 = c < 4 ? (uint32_t)m->C[c] : 0u;
                r2 = c < 4 ? (uint32_t)m->C[c + 1] : 0u;
            }
            const int ok = l2 < r2 && !again;
            l = ok ? l2 : 0u;
            r = ok ? r2 : n;
            d = ok ? (dbase + 1u < k ? dbase + 1u : k) : 0u;
            tab = again;
            if (r == l + 1u && d >= dmin) { seeded = 1; j0 = j; }
            j++;
        }
    }
    cn->items++;
    if (!seeded) { cn->items_unseeded++; pl->n_mm = 0; return; }
    pl->seeded = 1;
    pl->p0 = m->pos[l] - j0; /* mod 2^32 like the kernel */
    cn->pos_lookups++;
    /* compare + predict: text byte 0 (path start, padding, out of range) matches nothing */
    int64_t i_last = -1;
    uint32_t cnt = 0, mm0 = 0xFFFE;
    const int64_t tp0 = (int64_t)m->pos[l] - (int64_t)j0;
    for (uint32_t t = 0; t < len; t++) {
        const int64_t tp = tp0 + t;
        const uint8_t tc = (tp >= 0 && tp < (int64_t)n) ? m->text[tp] : 0;
        if (tc == 0 || tc != q[t]) {
            if (cnt == 0) mm0 = t;
            if (cnt < PM_LIST_MAX) pl->mm[cnt] = (uint16_t)t;
            cnt++;
            i_last = t;
        }
        const int64_t dp = (int64_t)t - i_last;
        ms[t] = (uint8_t)(dp < (int64_t)k ? dp : k);
    }
    cn->compare_bases += len;
    cn->mismatches += cnt;
    pl->n_mm = cnt < 254u ? cnt : 254u;
    pl->j_conv = (clean && (cnt == 0 || mm0 > j0)) ? j0 + 1u : 0u;
    if (pl->j_conv) cn->items_clean++;
}

/* ------------------------------------------------------------------ units (plan_emit_kernel: make_units) */

typedef struct {
    uint32_t pos, out_from, bound, d_start;
    int32_t last_mm;
    uint8_t head, plain, to_end;
} pm_unit;

static uint32_t pm_units_of(const ora_plan_params *P, uint32_t k, uint32_t len, const pm_plan *pl, pm_unit *out /* >= len / chunk + PM_LIST_MAX + 2 */)
{
    uint32_t nu = 0;
    if (len == 0) return 0;
    if (!pl->seeded || pl->n_mm > P->list_cap) { /* no plan: chunks from the root, k-1 warm-up bases, no convergence test */
        for (uint32_t c0 = 0; c0 < len; c0 += P->chunk) {
            pm_unit u;
            memset(&u, 0, sizeof u);
            u.out_from = c0;
            u.pos = c0 > k - 1u ? c0 - (k - 1u) : 0u;
            u.bound = c0 + P->chunk < len ? c0 + P->chunk : len;
            u.last_mm = -1;
            u.head = 1; u.plain = 1; u.to_end = u.bound == len;
            out[nu++] = u;
        }
        return nu;
    }
    uint32_t t = 0;
    int32_t prev = -1;
    int head = pl->j_conv == 0;
    while (head || t < pl->n_mm) {
        pm_unit u;
        memset(&u, 0, sizeof u);
        u.head = (uint8_t)head;
        int32_t last = -1;
        if (!head) {
            u.pos = pl->mm[t];
            const int64_t dp = (int64_t)u.pos - 1 - prev;
            u.d_start = (uint32_t)(dp < (int64_t)k ? dp : k);
            last = prev = (int32_t)u.pos;
            t++;
        }
        head = 0;
        while (t < pl->n_mm && (int32_t)pl->mm[t] - last < (int32_t)P->gap) { last = prev = (int32_t)pl->mm[t]; t++; }
        u.last_mm = last;
        u.bound = t < pl->n_mm ? pl->mm[t] : len;
        u.to_end = t >= pl->n_mm;
        u.out_from = u.pos;
        out[nu++] = u;
    }
    return nu;
}

/* ------------------------------------------------------------------ the walk of one unit, two forms */

typedef struct { uint32_t l, r, d; } pm_state;

/* ms_walk_guided_kernel: rank blocks (96 rows) + {lcs, psv, nsv} entries; one extension OR one contraction level per
 * iteration.  Returns 1 when the unit converged (or may stop at its bound), 0 when it flags its item. */
static int pm_walk_blocks(const pm_index *m, const uint8_t *q, const pm_unit *u, pm_state s, uint8_t *ms, ora_plan_counts *cn,
                          pm_set *lines)
{
    const uint32_t n = m->n, k = m->k;
    uint32_t i = u->pos, mflag = 0, tgt_l = 0, tgt_r = 0;
    int con = 0, conv = 0;
    while (i < u->bound) {
        if (!con) {
            const int c = pm_code(q[i]);
            uint32_t l2 = 0, r2 = 0;
            if (c < 4) {
                l2 = (uint32_t)m->C[c] + pm_rank(m, c, s.l);
                r2 = (uint32_t)m->C[c] + pm_rank(m, c, s.r);
                cn->unit_distinct_rank_lines += (uint64_t)pm_touch(lines, PM_A_RANK + (uint32_t)c, (uint64_t)(s.l / 96u) * 16u);
                cn->unit_distinct_rank_lines += (uint64_t)pm_touch(lines, PM_A_RANK + (uint32_t)c, (uint64_t)(s.r / 96u) * 16u);
            }
            const int ok = l2 < r2, accept = ok || s.d == 0;
            if (!accept) { /* nearest edge bits of c below l / from r on, inside the words the extension loaded */
                uint32_t dl = 0, dr = 0;
                if (c < 4) {
                    const uint32_t pb = s.l & 31u, below = pm_word32(m, c, s.l) & ((1u << pb) - 1u);
                    if (below) dl = pb - (31u - (uint32_t)__builtin_clz(below));
                    const uint32_t pr = s.r & 31u, above = pm_word32(m, c, s.r) & (~0u << pr);
                    if (above) dr = (uint32_t)__builtin_ffs((int)above) - pr;
                }
                mflag = dl && dr;
                tgt_l = s.l - dl;
                tgt_r = s.r + dr;
                con = 1;
                cn->walk_failed++;
                continue;
            }
            if (ok) { s.l = l2; s.r = r2; s.d = s.d + 1u < k ? s.d + 1u : k; }
            cn->walk_accepted++;
            const int64_t dp = (int64_t)i - u->last_mm;
            conv = !u->plain && (int32_t)i >= u->last_mm && s.r == s.l + 1u && s.d == (uint32_t)(dp < (int64_t)k ? dp : k);
            if (i >= u->out_from) { ms[i] = (uint8_t)s.d; cn->walk_out_bytes++; }
            i++;
            if (conv) break;
        } else { /* one level up the LCS interval tree */
            const uint32_t la = pm_lcs(m, s.l), lb = pm_lcs(m, s.r), lv = la > lb ? la : lb;
            pm_touch(lines, PM_A_ENT, (uint64_t)s.l * 12u);
            pm_touch(lines, PM_A_ENT, (uint64_t)s.r * 12u);
            const int root = lv == 0;
            const uint32_t cl = root ? 0u : (la == lv ? pm_psv(m, s.l) : s.l);
            const uint32_t cr = root ? n : (lb == lv ? pm_nsv(m, s.r) : s.r);
            const int cstop = root || !mflag || cl <= tgt_l || cr >= tgt_r;
            s.l = cl; s.r = cr; s.d = lv;
            if (cstop) con = 0;
            cn->walk_contractions++;
        }
    }
    return conv || u->plain || u->to_end;
}

/* ms_walk_recovery_kernel: one 128-byte line per 64 rows (rank blocks + LCS bytes); a failing base takes up to four
 * contraction levels out of the 16-row LCS windows [.., l] and [r, ..] of the lines it has and retries inside the same
 * iteration; when a window ends first the level comes from the entries in an iteration of its own. */
static int pm_walk_lines(const pm_index *m, const uint8_t *q, const pm_unit *u, pm_state s, uint8_t *ms, ora_plan_counts *cn,
                         pm_set *lines)
{
    const uint32_t n = m->n, k = m->k;
    uint32_t i = u->pos;
    int ent = 0, conv = 0;
    while (i < u->bound) {
        cn->walk_iterations_lines++;
        if (ent) { /* one level from the entries; the extension comes with the next iteration */
            const uint32_t la = pm_lcs(m, s.l), lb = pm_lcs(m, s.r), lv = la > lb ? la : lb;
            pm_touch(lines, PM_A_ENT, (uint64_t)s.l * 12u);
            pm_touch(lines, PM_A_ENT, (uint64_t)s.r * 12u);
            const int root = lv == 0;
            const uint32_t cl = root ? 0u : (la == lv ? pm_psv(m, s.l) : s.l), cr = root ? n : (lb == lv ? pm_nsv(m, s.r) : s.r);
            s.l = cl; s.r = cr; s.d = lv;
            ent = 0;
            cn->walk_entry_levels++;
            continue;
        }
        const int c = pm_code(q[i]);
        const uint32_t bl = s.l >> 6, br = s.r >> 6, ol = s.l & 63u, orr = s.r & 63u;
        const uint32_t wl = ol > 15u ? ol - 15u : 0u, wr = orr < 48u ? orr : 48u; /* first row of each window inside its line */
        pm_touch(lines, PM_A_LINE, (uint64_t)bl << 7);
        pm_touch(lines, PM_A_LINE, (uint64_t)br << 7);
        uint32_t l2 = 0, r2 = 0;
        if (c < 4) { l2 = (uint32_t)m->C[c] + pm_rank(m, c, s.l); r2 = (uint32_t)m->C[c] + pm_rank(m, c, s.r); }
        int ok = l2 < r2, short_win = 0;
        if (!ok && s.d != 0) cn->walk_failed++;
        for (uint32_t lev = 0; lev < 4u && !ok && s.d != 0 && !short_win; lev++) {
            /* (the windows stay where the iteration's first l and r put them: bl, br, wl, wr are not recomputed) */
            const uint32_t pl = s.l - (bl << 6) - wl, pr = s.r - (br << 6) - wr;
            const uint32_t lcs_l = pm_lcs(m, (bl << 6) + wl + pl), lcs_r = pm_lcs(m, (br << 6) + wr + pr);
            const uint32_t lvw = lcs_l > lcs_r ? lcs_l : lcs_r;
            const int need_l = lcs_l == lvw, need_r = lcs_r == lvw;
            /* previous smaller value inside window A below pl, next smaller inside window B above pr */
            int32_t fl = -1, fr = -1;
            for (int32_t t = (int32_t)pl - 1; t >= 0; t--)
                if (pm_lcs(m, (bl << 6) + wl + (uint32_t)t) < lvw) { fl = t; break; }
            for (uint32_t t = pr + 1u; t < 16u; t++)
                if (pm_lcs(m, (br << 6) + wr + t) < lvw) { fr = (int32_t)t; break; }
            cn->walk_contractions++;
            if (lvw == 0) { /* the root: its extension is [C[c], C[c+1]) */
                s.l = 0; s.r = n; s.d = 0;
                l2 = c < 4 ? (uint32_t)m->C[c] : 0u;
                r2 = c < 4 ? (uint32_t)m->C[c + 1] : 0u;
                ok = l2 < r2;
            } else if ((need_l && fl < 0) || (need_r && fr < 0)) {
                short_win = 1;
                cn->walk_short_windows++;
            } else {
                if (need_l) s.l = (bl << 6) + wl + (uint32_t)fl;
                if (need_r) s.r = (br << 6) + wr + (uint32_t)fr;
                s.d = lvw;
                l2 = r2 = 0;
                if (c < 4) { l2 = (uint32_t)m->C[c] + pm_rank(m, c, s.l); r2 = (uint32_t)m->C[c] + pm_rank(m, c, s.r); }
                ok = l2 < r2;
            }
        }
        const int accept = ok || s.d == 0;
        if (short_win) ent = 1;
        if (ok) { s.l = l2; s.r = r2; s.d = s.d + 1u < k ? s.d + 1u : k; }
        if (!accept) continue; /* (levels left, or the entries: the base is tried again) */
        cn->walk_accepted++;
        const int64_t dp = (int64_t)i - u->last_mm;
        conv = !u->plain && (int32_t)i >= u->last_mm && s.r == s.l + 1u && s.d == (uint32_t)(dp < (int64_t)k ? dp : k);
        if (i >= u->out_from) { ms[i] = (uint8_t)s.d; cn->walk_out_bytes++; }
        i++;
        if (conv) break;
    }
    return conv || u->plain || u->to_end;
}

/* the literal walk of a whole item (the redo pass and a plan that was given up: ms_walk_kernel); counts its iterations
 * the way that kernel runs them (one extension or one contraction level each) */
static void pm_walk_plain(const pm_index *m, const uint8_t *q, uint32_t len, uint8_t *ms, ora_plan_counts *cn)
{
    pm_unit u;
    memset(&u, 0, sizeof u);
    u.bound = len; u.last_mm = -1; u.head = 1; u.plain = 1; u.to_end = 1;
    pm_state s = {0, m->n, 0};
    ora_plan_counts c2;
    memset(&c2, 0, sizeof c2);
    pm_set dummy;
    pm_set_clear(&dummy);
    pm_walk_blocks(m, q, &u, s, ms, &c2, &dummy);
    cn->redo_iterations += c2.walk_accepted + c2.walk_failed + c2.walk_contractions;
    cn->redo_bases += len;
}

/* ------------------------------------------------------------------ depth-table form (dtab_kernels.hip) */
#define PM_UNKNOWN 0xFFFFFFFFu
/* dtab_anchor_depth: the `order` bases ending at base i are the suffix of exactly one row (else: unknown) - every longer
 * suffix that is present is a suffix of that row, whose characters are the path-cover text in front of its position */
static uint32_t pm_anchor_depth(const pm_index *m, uint32_t order, const uint8_t *q, uint32_t i)
{
    uint32_t l = 0, r = m->n;
    for (uint32_t x = i + 1u - order; x <= i; x++) {
        const int c = pm_code(q[x]);
        if (c >= 4) return PM_UNKNOWN;
        const uint32_t l2 = (uint32_t)m->C[c] + pm_rank(m, c, l), r2 = (uint32_t)m->C[c] + pm_rank(m, c, r);
        if (l2 >= r2) return PM_UNKNOWN;
        l = l2; r = r2;
    }
    if (r != l + 1u) return PM_UNKNOWN;
    const uint32_t p = m->pos[l];
    for (uint32_t t = 0; t < m->k; t++) {
        if (t > i) return t;
        const uint8_t qc = q[i - t];
        if (pm_code(qc) >= 4) return t;
        const uint8_t tc = t <= p ? m->text[p - t] : 0;
        if (tc == 0) return PM_UNKNOWN;
        if (tc != qc) return t >= order ? t : PM_UNKNOWN;
    }
    return m->k;
}

/* the literal walk over q[from, to) from the root, depths into tmp (uncounted: it stands for the table's content) */
static void pm_window_walk(const pm_index *m, const uint8_t *q, uint32_t from, uint32_t to, uint8_t *tmp)
{
    pm_unit u;
    memset(&u, 0, sizeof u);
    u.pos = from; u.out_from = from; u.bound = to; u.last_mm = -1; u.head = 1; u.plain = 1;
    pm_state s0 = {0, m->n, 0};
    ora_plan_counts c2;
    memset(&c2, 0, sizeof c2);
    pm_set dummy;
    pm_set_clear(&dummy);
    pm_walk_blocks(m, q, &u, s0, tmp, &c2, &dummy);
}

/* dtab_resolve_kernel / the fused plan_kernel on one item.  With a plan: lane j of mismatch m has base i = m + j (up to the
 * next mismatch, the item's end and order + 1 bases); the table gives L = the longest suffix of the bases up to i (inside the
 * item, behind the last non-ACGT byte) that is a suffix of a row when that is at most `order` - modelled by a literal walk from
 * the root that starts order + 1 bases in front of the mismatch; the bases in front of the first one with L <= j that the table
 * cannot tell are read off the path-cover text (anchors); one that stays unknown flags the item, else the bases up to that
 * first one are written.  Without a plan: every base of the item the same way, no stretch logic.  abs0 = offset of the item
 * in the query buffer. */
static void pm_resolve_item(const pm_index *m, const ora_plan_params *P, const uint8_t *q, uint32_t len, uint64_t abs0, uint8_t *ms,
                            pm_plan *pl, uint8_t *tmp /* len bytes */, ora_plan_counts *cn)
{
    const uint32_t order = P->depth_table < m->k ? P->depth_table : m->k, k = m->k;
    const int anchors = order < k && P->depth_anchors;
    pl->tab_flagged = 0;
    if (len == 0) return;
    if (pl->seeded && pl->n_mm > P->list_cap) { pl->tab_flagged = 1; cn->tab_flagged++; return; } /* a wrong diagonal: the plain walk */
    for (uint32_t i = 0; i < len; i++) /* a byte that is no base: the plain walk (the fused kernel's 2-bit copy of the item) */
        if (pm_code(q[i]) >= 4) { pl->tab_flagged = 1; cn->tab_flagged++; if (!pl->seeded) cn->items_noplan++; return; }
    if (!pl->seeded) { /* no seed: every base on its own, no stretch logic */
        cn->items_noplan++;
        pm_window_walk(m, q, 0, len, tmp);
        for (uint32_t i = 0; i < len; i++) {
            uint32_t L = PM_UNKNOWN;
            if (abs0 + i + 1u >= 32u) {
                cn->tab_lookups++;
                if (order >= k || tmp[i] <= order) L = tmp[i];
                else cn->tab_anchored++; /* (deeper than the table knows: no anchors for items without a plan) */
            }
            if (L == PM_UNKNOWN) pl->tab_flagged = 1;
            else { ms[i] = (uint8_t)(L < k ? L : k); cn->tab_written++; }
        }
        if (pl->tab_flagged) cn->tab_flagged++;
        return;
    }
    for (uint32_t t = 0; t < pl->n_mm; t++) { /* (every mismatch on its own, flagged item or not) */
        const uint32_t mpos = pl->mm[t], nxt = t + 1u < pl->n_mm ? pl->mm[t + 1u] : len;
        cn->tab_stretches++;
        uint32_t hi = mpos + order + 1u;
        if (hi > nxt) hi = nxt;
        if (hi > len) hi = len;
        pm_window_walk(m, q, mpos > order + 1u ? mpos - (order + 1u) : 0u, hi, tmp);
        uint32_t first_conv = ~0u, Lv[32];
        uint8_t sat[32];
        for (uint32_t i = mpos; i < hi; i++) {
            const uint32_t j = i - mpos;
            const int nowin = abs0 + i + 1u < 32u;
            cn->tab_lookups += nowin ? 0u : 1u;
            sat[j] = (uint8_t)(nowin ? 2 : (order < k && tmp[i] > order) ? 1 : 0);
            Lv[j] = tmp[i];
            if (!sat[j] && Lv[j] <= j && first_conv == ~0u) first_conv = j;
        }
        int unknown = 0;
        for (uint32_t i = mpos; i < hi && i - mpos < first_conv; i++) { /* the table cannot tell: anchors */
            const uint32_t j = i - mpos;
            if (sat[j] == 2) unknown = 1;
            else if (sat[j] == 1) {
                Lv[j] = anchors ? pm_anchor_depth(m, order, q, i) : PM_UNKNOWN;
                cn->tab_anchored++;
                if (Lv[j] == PM_UNKNOWN) unknown = 1;
            }
        }
        if (unknown) { if (!pl->tab_flagged) cn->tab_flagged++; pl->tab_flagged = 1; continue; }
        for (uint32_t i = mpos; i < hi && i - mpos <= first_conv; i++) {
            ms[i] = (uint8_t)(Lv[i - mpos] < k ? Lv[i - mpos] : k);
            cn->tab_written++;
        }
    }
}

/* ------------------------------------------------------------------ driver */

typedef struct {
    const pm_index *m; const ora_plan_params *P; const uint8_t *concat; const uint64_t *off; size_t begin, end;
    uint8_t *ms_out; ora_plan_counts cn; pm_plan *plans; /* plans of [begin, end) kept for pass 2 */
    int pass; int gave_up;
} pm_job;

static void pm_units_walk(pm_job *j)
{
    const pm_index *m = j->m;
    const ora_plan_params *P = j->P;
    pm_set lines;
    size_t maxlen = 0;
    for (size_t r = j->begin; r < j->end; r++) {
        const size_t L = (size_t)(j->off[r + 1] - j->off[r]);
        if (L > maxlen) maxlen = L;
    }
    pm_unit *units = (pm_unit *)malloc((maxlen / (P->chunk ? P->chunk : 1) + PM_LIST_MAX + 4) * sizeof(pm_unit));
    for (size_t r = j->begin; r < j->end; r++) {
        const uint8_t *q = j->concat + j->off[r];
        const uint32_t len = (uint32_t)(j->off[r + 1] - j->off[r]);
        uint8_t *ms = j->ms_out + j->off[r];
        const pm_plan *pl = &j->plans[r - j->begin];
        if (j->gave_up) { pm_walk_plain(m, q, len, ms, &j->cn); continue; }
        if (P->depth_table) { /* no units: what the table left unresolved takes the plain walk */
            if (pl->tab_flagged) { j->cn.items_flagged++; pm_walk_plain(m, q, len, ms, &j->cn); }
            continue;
        }
        const uint32_t nu = pm_units_of(P, m->k, len, pl, units);
        int flagged = 0;
        for (uint32_t x = 0; x < nu; x++) {
            const pm_unit *u = &units[x];
            pm_state s = {0, m->n, 0};
            pm_set_clear(&lines);
            /* what every unit touches besides the index: its record, its start row, two query blocks, its output bytes */
            pm_touch(&lines, PM_A_UNIT, (uint64_t)(j->cn.units + x) * 32u);
            pm_touch(&lines, PM_A_Q, (uint64_t)(j->off[r] + (u->pos & ~15u)));
            pm_touch(&lines, PM_A_Q, (uint64_t)(j->off[r] + (u->pos & ~15u) + 16u));
            pm_touch(&lines, PM_A_OUT, (uint64_t)(j->off[r] + u->out_from));
            if (!u->head) {
                const uint32_t tp = pl->p0 + u->pos - 1u;
                s.l = m->node_at[tp]; s.r = s.l + 1u; s.d = u->d_start;
                pm_touch(&lines, PM_A_NODE, (uint64_t)tp * 4u);
                j->cn.node_lookups++;
            }
            const int fine = P->recovery_lines ? pm_walk_lines(m, q, u, s, ms, &j->cn, &lines) : pm_walk_blocks(m, q, u, s, ms, &j->cn, &lines);
            if (!fine) flagged = 1;
            j->cn.unit_distinct_lines += lines.used;
            if (u->plain) j->cn.units_plain++;
            else if (u->head) j->cn.units_head++;
        }
        j->cn.units += nu;
        if (flagged) { /* the redo pass: the item again, in full, by the plain kernel */
            j->cn.items_flagged++;
            pm_walk_plain(m, q, len, ms, &j->cn);
        }
    }
    free(units);
}

static void *pm_worker(void *arg)
{
    pm_job *j = (pm_job *)arg;
    if (j->pass == 1) { /* plan_kernel + the unit count */
        size_t maxlen = 0;
        for (size_t r = j->begin; r < j->end; r++) {
            const size_t L = (size_t)(j->off[r + 1] - j->off[r]);
            if (L > maxlen) maxlen = L;
        }
        pm_unit *units = (pm_unit *)malloc((maxlen / (j->P->chunk ? j->P->chunk : 1) + PM_LIST_MAX + 4) * sizeof(pm_unit));
        uint8_t *tmp = (uint8_t *)malloc(maxlen + 16);
        for (size_t r = j->begin; r < j->end; r++) {
            const uint32_t len = (uint32_t)(j->off[r + 1] - j->off[r]);
            pm_plan_item(j->m, j->P, j->concat + j->off[r], len, j->ms_out + j->off[r], &j->plans[r - j->begin], &j->cn);
            if (j->P->depth_table)
                pm_resolve_item(j->m, j->P, j->concat + j->off[r], len, j->off[r], j->ms_out + j->off[r], &j->plans[r - j->begin], tmp, &j->cn);
            else
                j->cn.units_counted += pm_units_of(j->P, j->m->k, len, &j->plans[r - j->begin], units);
            if (j->plans[r - j->begin].seeded && j->plans[r - j->begin].n_mm > j->P->list_cap) j->cn.items_list_overflow++;
        }
        free(tmp);
        free(units);
    } else {
        pm_units_walk(j);
    }
    return NULL;
}

static void pm_add(ora_plan_counts *a, const ora_plan_counts *b)
{
    uint64_t *x = (uint64_t *)a;
    const uint64_t *y = (const uint64_t *)b;
    for (size_t i = 0; i < sizeof *a / sizeof(uint64_t); i++) x[i] += y[i];
}

int ora_plan_model(const ora_index *x, const uint8_t *text, const uint32_t *pos, const uint32_t *node_at,
                   const ora_plan_params *P, const uint8_t *concat, const uint64_t *offsets, size_t n_reads, int n_threads,
                   uint8_t *ms_out, ora_plan_counts *counts)
{
    if (!x || !text || !pos || !node_at || !P || !concat || !offsets || !ms_out || !counts) return ORA_E_BAD_ARG;
    if (P->gap < 2 || P->chunk < 1 || P->list_cap + 1u > PM_LIST_MAX || P->seed_table_depth > 14 || P->depth_table > 17) return ORA_E_BAD_ARG;
    if (ora_index_n_sets(x) >= 0xFFFFFFF0ull) return ORA_E_BAD_ARG;
    pm_index m;
    memset(&m, 0, sizeof m);
    m.x = x;
    m.n = (uint32_t)ora_index_n_sets(x);
    m.k = ora_index_k(x);
    ora_index_C(x, m.C);
    m.C[4] = m.n;
    m.lcs = ora_index_lcs(x);
    m.text = text; m.pos = pos; m.node_at = node_at;
    const size_t nw = ((size_t)m.n + 63) / 64;
    for (int c = 0; c < 4; c++) {
        m.bits[c] = ora_index_bits(x, c);
        m.cum[c] = (uint32_t *)malloc((nw + 1) * sizeof(uint32_t));
        uint32_t a = 0;
        for (size_t w = 0; w < nw; w++) { m.cum[c][w] = a; a += (uint32_t)__builtin_popcountll(m.bits[c][w]); }
        m.cum[c][nw] = a;
    }
    /* seed table: the interval of every string of D bases, level by level from the root (device_index.cpp) */
    m.seed_d = P->seed_table_depth < m.k ? P->seed_table_depth : m.k;
    if (m.seed_d) {
        size_t cur_n = 1;
        uint32_t *cur = (uint32_t *)malloc(2 * sizeof(uint32_t)), *nxt;
        cur[0] = 0; cur[1] = m.n;
        for (uint32_t t = 0; t < m.seed_d; t++) {
            nxt = (uint32_t *)malloc(cur_n * 4 * 2 * sizeof(uint32_t));
            for (size_t p = 0; p < cur_n; p++)
                for (int c = 0; c < 4; c++) {
                    uint32_t l2 = 0, r2 = 0;
                    if (cur[2 * p] < cur[2 * p + 1]) {
                        l2 = (uint32_t)m.C[c] + pm_rank(&m, c, cur[2 * p]);
                        r2 = (uint32_t)m.C[c] + pm_rank(&m, c, cur[2 * p + 1]);
                    }
                    nxt[2 * (4 * p + (size_t)c)] = l2;
                    nxt[2 * (4 * p + (size_t)c) + 1] = r2;
                }
            free(cur);
            cur = nxt;
            cur_n *= 4;
        }
        m.seed_tab = cur;
    }
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_reads) n_threads = n_reads ? (int)n_reads : 1;
    pm_job *jobs = (pm_job *)calloc((size_t)n_threads, sizeof(pm_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    memset(counts, 0, sizeof *counts);
    uint64_t units_counted = 0;
    for (int pass = 1; pass <= 2; pass++) {
        for (int t = 0; t < n_threads; t++) {
            pm_job *j = &jobs[t];
            if (pass == 1) {
                j->m = &m; j->P = P; j->concat = concat; j->off = offsets; j->ms_out = ms_out;
                j->begin = n_reads * (size_t)t / (size_t)n_threads;
                j->end = n_reads * (size_t)(t + 1) / (size_t)n_threads;
                j->plans = (pm_plan *)malloc((j->end - j->begin + 1) * sizeof(pm_plan));
            }
            j->pass = pass;
            pthread_create(&th[t], NULL, pm_worker, j);
        }
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
        if (pass == 1) {
            for (int t = 0; t < n_threads; t++) units_counted += jobs[t].cn.units_counted;
            /* plan_emit_kernel's bail-out: more units than bail_x16 / 16 per read of 150 bases -> every item walked plainly */
            const uint64_t per = offsets[n_reads] / 150u > n_reads ? offsets[n_reads] / 150u : n_reads;
            const uint64_t bail = per * P->bail_x16 / 16u + 64u;
            uint64_t unresolved = 0;
            for (int t = 0; t < n_threads; t++) unresolved += jobs[t].cn.tab_flagged;
            /* (depth-table form: the plan is given up when more than half of the items had no plan or stayed unresolved) */
            const int gave_up = P->depth_table ? unresolved > n_reads / 2u + 64u : units_counted > bail;
            for (int t = 0; t < n_threads; t++) jobs[t].gave_up = gave_up;
            counts->gave_up = (uint64_t)gave_up;
        }
    }
    for (int t = 0; t < n_threads; t++) {
        const uint64_t keep = counts->gave_up;
        pm_add(counts, &jobs[t].cn);
        counts->gave_up = keep;
        free(jobs[t].plans);
    }
    counts->bases = offsets[n_reads];
    free(jobs); free(th);
    for (int c = 0; c < 4; c++) free(m.cum[c]);
    free(m.seed_tab);
    return ORA_OK;
}
