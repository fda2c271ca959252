/*
 * kbo_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see kbo_oracle.h).
 *
 * Plain-C restatement of the reference algorithm.  Every function cites the
 * reference file:line it follows.  The layout deliberately mimics what the
 * sbwt crate does on the CPU (four separate bit-vectors with 512-bit-block rank
 * samples, bit-packed LCS) so that timing it is a fair "port" baseline.
 *
 * Parity pinned by the reference's golden vectors (JSON files under tests/golden).
 */
#define _GNU_SOURCE
#include "kbo_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <unistd.h>
#include <string.h>

struct ora_index {
    uint32_t k;
    uint64_t n;       /* n_sets: number of SBWT rows (k-mers + dummies) */
    uint64_t n_kmers; /* distinct real k-mers                           */
    uint64_t C[4];
    uint64_t *bits[4];   /* n bits each                                */
    uint64_t *rank512[4];/* set bits before each 512-bit block         */
    uint8_t *lcs;        /* byte copy (export / cross-checks)          */
    uint64_t *lcs_packed;/* width-bit packed copy used by the walk     */
    uint32_t lcs_width;
    uint8_t *rows;       /* n*k reversed rows, symbols 0..4 ($ACGT); may be NULL */
};

/* ------------------------------------------------------------------ utils */

static inline int sym_of(uint8_t ch)
{ /* sbwt DNA alphabet ACGT; anything else splits a run (index.rs:265 uses '-') */
    switch (ch) {
    case 'A': return 1;
    case 'C': return 2;
    case 'G': return 3;
    case 'T': return 4;
    default: return 0;
    }
}

static size_t g_cmp_len;
static int cmp_rows(const void *a, const void *b) { return memcmp(a, b, g_cmp_len); }

/* first row index in [0,n) whose first `plen` bytes are >= key (memcmp order) */
static uint64_t lower_bound_prefix(const uint8_t *rows, uint64_t n, size_t k,
                                   const uint8_t *key, size_t plen)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (memcmp(rows + mid * k, key, plen) < 0) lo = mid + 1; else hi = mid;
    }
    return lo;
}

static void finish_supports(ora_index *x)
{
    uint64_t n = x->n, nblk = (n + 511) / 512 + 1;
    for (int c = 0; c < 4; c++) {
        x->rank512[c] = (uint64_t *)calloc(nblk, sizeof(uint64_t));
        uint64_t acc = 0, nw = (n + 63) / 64;
        for (uint64_t w = 0; w < nw; w++) {
            if ((w & 7) == 0) x->rank512[c][w >> 3] = acc;
            acc += (uint64_t)__builtin_popcountll(x->bits[c][w]);
        }
        if ((nw & 7) == 0) x->rank512[c][nw >> 3] = acc;
    }
    /* LcsArray: bit-packed integer vector, width = bits needed for k-1 */
    uint32_t w = 1;
    while ((1u << w) < x->k) w++;
    x->lcs_width = w;
    uint64_t nwords = (n * w + 63) / 64 + 1;
    x->lcs_packed = (uint64_t *)calloc(nwords, sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) {
        uint64_t bit = i * w, v = x->lcs[i];
        x->lcs_packed[bit >> 6] |= v << (bit & 63);
        if ((bit & 63) + w > 64) x->lcs_packed[(bit >> 6) + 1] |= v >> (64 - (bit & 63));
    }
}

static inline uint64_t lcs_get(const ora_index *x, uint64_t i)
{
    uint32_t w = x->lcs_width;
    uint64_t bit = i * w, v = x->lcs_packed[bit >> 6] >> (bit & 63);
    if ((bit & 63) + w > 64) v |= x->lcs_packed[(bit >> 6) + 1] << (64 - (bit & 63));
    return v & ((1ull << w) - 1);
}

/* rank_c(i): number of set bits of B_c in [0,i) */
static inline uint64_t rank_c(const ora_index *x, int c, uint64_t i)
{
    uint64_t blk = i >> 9, w0 = blk << 3, w1 = i >> 6;
    uint64_t r = x->rank512[c][blk];
    const uint64_t *b = x->bits[c];
    for (uint64_t w = w0; w < w1; w++) r += (uint64_t)__builtin_popcountll(b[w]);
    if (i & 63) r += (uint64_t)__builtin_popcountll(b[w1] & ((1ull << (i & 63)) - 1));
    return r;
}

/* ------------------------------------------------------------------ build */

/* Abstract content of the index (verified against the
 * goldens): rows = distinct k-mers of every ACGT-run of length >= k, plus for
 * each k-mer with no predecessor all its $-left-padded proper prefixes incl.
 * $^k; sorted colexicographically with $<A<C<G<T.  B_c[i]=1 iff row i is the
 * first row of its (k-1)-suffix group and row[1:]+c is a row.
 * C[c] = 1 + sum_{c'<c} popcount(B_c').  LCS[i] = longest common suffix of
 * rows i-1,i where $ never matches.   (built at reference index.rs:73-94) */
int ora_index_build(const uint8_t *const *seqs, const size_t *lens, size_t n_seqs,
                    uint32_t k, int add_revcomp, ora_index **out)
{
    if (!seqs || !lens || n_seqs == 0 || k == 0 || k > 255 || !out) return ORA_E_BAD_ARG;
    /* pass 1: count k-mers */
    uint64_t total = 0;
    for (size_t s = 0; s < n_seqs; s++) {
        size_t run = 0;
        for (size_t i = 0; i < lens[s]; i++) {
            run = sym_of(seqs[s][i]) ? run + 1 : 0;
            if (run >= k) total++;
        }
    }
    if (add_revcomp) total *= 2;
    /* rows are stored REVERSED (last char first) so memcmp order == colex */
    uint8_t *km = (uint8_t *)malloc((total + 1) * (size_t)k);
    if (!km) return ORA_E_NOMEM;
    uint64_t m = 0;
    for (size_t s = 0; s < n_seqs; s++) {
        size_t run = 0;
        for (size_t i = 0; i < lens[s]; i++) {
            run = sym_of(seqs[s][i]) ? run + 1 : 0;
            if (run >= k) {
                uint8_t *row = km + m * k;
                for (uint32_t t = 0; t < k; t++) row[t] = (uint8_t)sym_of(seqs[s][i - t]);
                m++;
                if (add_revcomp) { /* revcomp k-mer: reversed string is the complement of the forward string */
                    uint8_t *rc = km + m * k;
                    for (uint32_t t = 0; t < k; t++)
                        rc[t] = (uint8_t)(5 - sym_of(seqs[s][i - (k - 1) + t]));
                    m++;
                }
            }
        }
    }
    g_cmp_len = k;
    qsort(km, m, k, cmp_rows);
    uint64_t nk = 0;
    for (uint64_t i = 0; i < m; i++)
        if (i == 0 || memcmp(km + i * k, km + (i - 1) * k, k) != 0) {
            if (nk != i) memmove(km + nk * k, km + i * k, k);
            nk++;
        }
    /* dummies: k-mer x has a predecessor iff some k-mer y has y[1:] == x[:-1],
     * i.e. reversed: y_rev[0..k-1) == x_rev[1..k) */
    uint64_t cap = 1024, nd = 0;
    uint8_t *dm = (uint8_t *)malloc(cap * (size_t)k);
    memset(dm, 0, k); /* root $^k always present */
    nd = 1;
    for (uint64_t i = 0; i < nk; i++) {
        const uint8_t *x = km + i * k;
        int has_pred = 0;
        if (k == 1) has_pred = 1; /* x[:-1] is empty: every k-mer precedes it */
        else {
            uint64_t p = lower_bound_prefix(km, nk, k, x + 1, k - 1);
            has_pred = (p < nk && memcmp(km + p * k, x + 1, k - 1) == 0);
        }
        if (has_pred) continue;
        for (uint32_t j = 1; j < k; j++) { /* $^(k-j) x[0..j) */
            if (nd == cap) { cap *= 2; dm = (uint8_t *)realloc(dm, cap * (size_t)k); }
            uint8_t *row = dm + nd * k;
            memset(row, 0, k);
            /* forward x[t] = x_rev[k-1-t]; reversed dummy: (x[j-1],...,x[0],0,...) */
            for (uint32_t t = 0; t < j; t++) row[t] = x[k - 1 - (j - 1 - t)];
            nd++;
        }
    }
    uint64_t nall = nk + nd;
    uint8_t *rows = (uint8_t *)malloc(nall * (size_t)k);
    memcpy(rows, km, nk * (size_t)k);
    memcpy(rows + nk * (size_t)k, dm, nd * (size_t)k);
    free(km); free(dm);
    qsort(rows, nall, k, cmp_rows);
    uint64_t n = 0;
    for (uint64_t i = 0; i < nall; i++)
        if (i == 0 || memcmp(rows + i * k, rows + (i - 1) * k, k) != 0) {
            if (n != i) memmove(rows + n * k, rows + i * k, k);
            n++;
        }

    ora_index *x = (ora_index *)calloc(1, sizeof(*x));
    x->k = k; x->n = n; x->n_kmers = nk; x->rows = rows;
    uint64_t nw = (n + 63) / 64 + 1;
    for (int c = 0; c < 4; c++) x->bits[c] = (uint64_t *)calloc(nw, sizeof(uint64_t));
    x->lcs = (uint8_t *)calloc(n + 1, 1);
    uint8_t *succ = (uint8_t *)malloc(k);
    for (uint64_t i = 0; i < n; i++) {
        const uint8_t *r = rows + i * k;
        int first = (i == 0) || (k > 1 && memcmp(r, r - k, k - 1) != 0);
        if (k == 1) first = (i == 0);
        if (first) {
            memcpy(succ + 1, r, k - 1);
            for (int c = 0; c < 4; c++) {
                succ[0] = (uint8_t)(c + 1);
                uint64_t p = lower_bound_prefix(rows, n, k, succ, k);
                if (p < n && memcmp(rows + p * k, succ, k) == 0)
                    x->bits[c][i >> 6] |= 1ull << (i & 63);
            }
        }
        if (i > 0) {
            uint32_t t = 0;
            while (t < k && r[t] == r[(ptrdiff_t)t - (ptrdiff_t)k] && r[t] != 0) t++;
            x->lcs[i] = (uint8_t)t;
        }
    }
    free(succ);
    uint64_t acc = 1;
    for (int c = 0; c < 4; c++) {
        x->C[c] = acc;
        for (uint64_t w = 0; w < nw; w++) acc += (uint64_t)__builtin_popcountll(x->bits[c][w]);
    }
    finish_supports(x);
    *out = x;
    return ORA_OK;
}

int ora_index_from_parts(uint32_t k, uint64_t n_sets, uint64_t n_kmers,
                         const uint64_t *const rows[4], const uint64_t C[4],
                         const uint8_t *lcs, ora_index **out)
{
    if (!rows || !C || !lcs || !out || k == 0 || k > 255 || n_sets == 0) return ORA_E_BAD_ARG;
    ora_index *x = (ora_index *)calloc(1, sizeof(*x));
    x->k = k; x->n = n_sets; x->n_kmers = n_kmers;
    uint64_t nw = (n_sets + 63) / 64;
    for (int c = 0; c < 4; c++) {
        x->C[c] = C[c];
        x->bits[c] = (uint64_t *)calloc(nw + 1, sizeof(uint64_t));
        memcpy(x->bits[c], rows[c], nw * sizeof(uint64_t));
        if (n_sets & 63) x->bits[c][nw - 1] &= (1ull << (n_sets & 63)) - 1;
    }
    x->lcs = (uint8_t *)calloc(n_sets + 1, 1);
    memcpy(x->lcs, lcs, n_sets);
    finish_supports(x);
    *out = x;
    return ORA_OK;
}

void ora_index_free(ora_index *x)
{
    if (!x) return;
    for (int c = 0; c < 4; c++) { free(x->bits[c]); free(x->rank512[c]); }
    free(x->lcs); free(x->lcs_packed); free(x->rows); free(x);
}

uint32_t ora_index_k(const ora_index *x) { return x->k; }
uint64_t ora_index_n_sets(const ora_index *x) { return x->n; }
uint64_t ora_index_n_kmers(const ora_index *x) { return x->n_kmers; }
void ora_index_C(const ora_index *x, uint64_t C[4]) { memcpy(C, x->C, sizeof(x->C)); }
const uint64_t *ora_index_bits(const ora_index *x, int c) { return x->bits[c & 3]; }
const uint8_t *ora_index_lcs(const ora_index *x) { return x->lcs; }

/* row of a full-length k-mer (chars ACGT$) by binary search over the stored sorted rows;
 * used by kbo_oracle_refine.c as sbwt's search() for k-length patterns */
int ora_index_find_kmer(const ora_index *x, const uint8_t *kmer, uint64_t *row)
{
    if (!x->rows) return 0;
    uint8_t key[256];
    for (uint32_t t = 0; t < x->k; t++) {
        uint8_t ch = kmer[x->k - 1 - t];
        int sy = ch == '$' ? 0 : sym_of(ch);
        if (ch != '$' && sy == 0) return 0;
        key[t] = (uint8_t)sy;
    }
    uint64_t p = lower_bound_prefix(x->rows, x->n, x->k, key, x->k);
    if (p < x->n && memcmp(x->rows + p * x->k, key, x->k) == 0) { if (row) *row = p; return 1; }
    return 0;
}

/* position of the (j+1)-th set bit of B_c (j < popcount(B_c)) */
static uint64_t select_c(const ora_index *x, int c, uint64_t j)
{
    uint64_t lo = 0, hi = ((x->n + 63) / 64 - 1) >> 3; /* last 512-bit block (of those that hold words) whose sample is <= j */
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo + 1) / 2;
        if (x->rank512[c][mid] <= j) lo = mid; else hi = mid - 1;
    }
    uint64_t r = x->rank512[c][lo], w = lo << 3;
    for (;; w++) {
        uint64_t pc = (uint64_t)__builtin_popcountll(x->bits[c][w]);
        if (r + pc > j) break;
        r += pc;
    }
    uint64_t word = x->bits[c][w];
    for (uint64_t t = r; t < j; t++) word &= word - 1; /* drop the set bits in front of the wanted one */
    return (w << 6) + (uint64_t)__builtin_ctzll(word);
}

int ora_index_access_kmer(const ora_index *x, uint64_t colex, uint8_t *out_k)
{
    if (colex >= x->n) return ORA_E_BAD_ARG;
    if (x->rows) {
        const uint8_t *r = x->rows + colex * x->k;
        for (uint32_t t = 0; t < x->k; t++) out_k[t] = (uint8_t)"$ACGT"[r[x->k - 1 - t]];
        return ORA_OK;
    }
    /* an index adopted from its parts has no row table: spell the row from the subset matrix alone.  Row i >= 1 with
     * C[c] <= i < C[c+1] ends in c and is (row p)[1..] + c for the row p that holds the (i - C[c])-th edge bit of B_c
     * (the extend-right bijection read backwards); row 0 is $^k.  k steps give the k characters, last one first. */
    uint64_t i = colex;
    for (uint32_t t = x->k; t-- > 0;) {
        if (i == 0) { out_k[t] = '$'; continue; }
        int c = 3;
        while (c > 0 && i < x->C[c]) c--;
        out_k[t] = (uint8_t)"ACGT"[c];
        i = select_c(x, c, i - x->C[c]);
    }
    return ORA_OK;
}

/* --------------------------------------------------- A1: matching statistics
 * sbwt::StreamingIndex::matching_statistics as called at reference
 * index.rs:251-252 (semantics: SURVEY.md §8(a) A1):
 *   for each byte c:  Ic = extend_right(I,c)
 *                     while d>0 && Ic empty: I = contract_left(I,d-1); d-=1; Ic = extend_right(I,c)
 *                     if Ic non-empty: I=Ic; d=min(d+1,k)
 *                     push (d, I)
 * extend_right([l,r),c) = [C[c]+rank_c(l), C[c]+rank_c(r)); empty for non-ACGT c
 * (build's documented choice; unpinned by the reference).
 * contract_left([l,r),t): while l>0 && LCS[l]>=t: l--; while r<n && LCS[r]>=t: r++. */
typedef struct { uint64_t l, r; } ival;

static inline ival extend_right(const ora_index *x, ival I, int c, ora_counters *ctr)
{
    ival o = {0, 0};
    if (ctr) ctr->extend_calls++;
    if (c < 0) return o;
    if (ctr) {
        ctr->rank_calls += 2;
        ctr->rank_blocks += ((I.l >> 9) == (I.r >> 9)) ? 1 : 2;
    }
    o.l = x->C[c] + rank_c(x, c, I.l);
    o.r = x->C[c] + rank_c(x, c, I.r);
    return o;
}

static inline ival contract_left(const ora_index *x, ival I, uint64_t target, ora_counters *ctr)
{
    if (ctr) ctr->contracts++;
    while (I.l > 0) {
        if (ctr) ctr->lcs_reads++;
        if (lcs_get(x, I.l) >= target) I.l--; else break;
    }
    while (I.r < x->n) {
        if (ctr) ctr->lcs_reads++;
        if (lcs_get(x, I.r) >= target) I.r++; else break;
    }
    return I;
}

int ora_matching_statistics(const ora_index *x, const uint8_t *q, size_t len,
                            uint64_t *d_out, uint64_t *lo, uint64_t *hi, ora_counters *ctr)
{
    if (len == 0) return ORA_E_EMPTY_QUERY; /* index.rs:248 */
    uint64_t d = 0;
    ival I = {0, x->n};
    for (size_t i = 0; i < len; i++) {
        int c = sym_of(q[i]) - 1;
        ival Ic = extend_right(x, I, c, ctr);
        while (d > 0 && Ic.l >= Ic.r) {
            I = contract_left(x, I, d - 1, ctr);
            d -= 1;
            Ic = extend_right(x, I, c, ctr);
        }
        if (Ic.l < Ic.r) {
            I = Ic;
            d = d + 1 < x->k ? d + 1 : x->k;
        }
        if (d_out) d_out[i] = d;
        if (lo) lo[i] = I.l;
        if (hi) hi[i] = I.r;
    }
    if (ctr) ctr->bases += len;
    return ORA_OK;
}

/* --------------------------------------------------- A3: derandomize.rs:91-145 */

/* Rust f64::powi lowers to llvm.powi -> compiler-rt __powidf2 */
static double powi_f64(double a, int b)
{
    const int recip = b < 0;
    double r = 1;
    while (1) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1 / r : r;
}

double ora_log_rm_max_cdf(size_t t, size_t alphabet_size, size_t n_kmers)
{ /* derandomize.rs:99 */
    return (double)n_kmers *
           log1p(-powi_f64(exp(log(1.0) - log((double)alphabet_size)), (int)t + 1));
}

size_t ora_random_match_threshold(size_t k, size_t n_kmers, size_t alphabet_size,
                                  double max_error_prob)
{ /* derandomize.rs:139-144 */
    for (size_t i = 1; i < k; i++)
        if (ora_log_rm_max_cdf(i, alphabet_size, n_kmers) > log1p(-max_error_prob)) return i;
    return k;
}

/* --------------------------------------------------- A4/A5: derandomize.rs:221-288 */

int64_t ora_derandomize_ms_val(size_t curr, int64_t next, size_t threshold, size_t k)
{ /* derandomize.rs:232-246 */
    int64_t run = next - 1;
    if (curr == k) run = (int64_t)k;
    if (curr > threshold && next < (int64_t)curr) run = (int64_t)curr;
    return run;
}

int ora_derandomize_ms_vec(const uint64_t *noisy, size_t len, size_t k, size_t threshold,
                           int64_t *out)
{
    if (k == 0) return ORA_E_BAD_ARG; /* derandomize.rs:274 */
    if (threshold <= 1) return ORA_E_THRESHOLD; /* :275 */
    if (len <= 2) return ORA_E_LEN_LE_2;        /* :276 */
    out[len - 1] = noisy[len - 1] > threshold ? (int64_t)noisy[len - 1] : 0; /* :282 */
    for (size_t i = 2; i < len + 1; i++) /* :283-285 */
        out[len - i] = ora_derandomize_ms_val(noisy[len - i], out[len - i + 1], threshold, k);
    return ORA_OK;
}

/* --------------------------------------------------- A6: translate.rs:180-293 */

void ora_translate_ms_val(int64_t curr, int64_t next, int64_t prev, size_t threshold,
                          uint32_t *aln_curr, uint32_t *aln_next)
{ /* translate.rs:188-215 */
    *aln_next = ' ';
    if (curr > (int64_t)threshold && next > 0 && next < (int64_t)threshold) {
        *aln_curr = 'R';
        *aln_next = 'R';
    } else if (curr <= 0) {
        *aln_curr = (next == 1 && prev > 0) ? 'X' : '-';
    } else {
        *aln_curr = 'M';
    }
}

int ora_translate_ms_vec(const int64_t *ms, size_t len, size_t k, size_t threshold,
                         uint32_t *res)
{
    if (k == 0) return ORA_E_BAD_ARG;           /* translate.rs:268 */
    if (threshold <= 1) return ORA_E_THRESHOLD; /* :269 */
    if (len <= 2) return ORA_E_LEN_LE_2;        /* :270 */
    for (size_t i = 0; i < len; i++) res[i] = ' ';
    for (size_t pos = 0; pos < len; pos++) { /* :275-290 */
        int64_t prev = pos > 1 ? ms[pos - 1] : (int64_t)k;
        int64_t curr = ms[pos];
        int64_t next = pos < len - 1 ? ms[pos + 1] : ms[pos];
        if (!(pos > 1 && res[pos - 1] == 'R' && res[pos] == 'R')) {
            uint32_t a, b;
            ora_translate_ms_val(curr, next, prev, threshold, &a, &b);
            res[pos] = a;
            if (pos + 1 < len - 1 && b != ' ') res[pos + 1] = b;
        }
    }
    return ORA_OK;
}

/* --------------------------------------------------- A7: lib.rs:612-628 */

static int matches_one(const ora_index *x, const uint8_t *q, size_t len, size_t threshold,
                       uint64_t *d, int64_t *der, uint32_t *tr, uint8_t *chars_out,
                       uint8_t *d_out, ora_counters *ctr)
{
    int rc = ora_matching_statistics(x, q, len, d, NULL, NULL, ctr); /* lib.rs:624 */
    if (rc) return rc;
    if (d_out) for (size_t i = 0; i < len; i++) d_out[i] = (uint8_t)d[i];
    rc = ora_derandomize_ms_vec(d, len, x->k, threshold, der);       /* lib.rs:625 */
    if (rc) return rc;
    rc = ora_translate_ms_vec(der, len, x->k, threshold, tr);        /* lib.rs:627 */
    if (rc) return rc;
    if (chars_out) for (size_t i = 0; i < len; i++) chars_out[i] = (uint8_t)tr[i];
    return ORA_OK;
}

int ora_matches(const ora_index *x, const uint8_t *q, size_t len, double max_error_prob,
                uint8_t *chars_out)
{
    size_t threshold = ora_random_match_threshold(x->k, x->n_kmers, 4, max_error_prob); /* lib.rs:620 */
    uint64_t *d = (uint64_t *)malloc((len + 1) * sizeof(uint64_t));
    int64_t *der = (int64_t *)malloc((len + 1) * sizeof(int64_t));
    uint32_t *tr = (uint32_t *)malloc((len + 1) * sizeof(uint32_t));
    int rc = matches_one(x, q, len, threshold, d, der, tr, chars_out, NULL, NULL);
    free(d); free(der); free(tr);
    return rc;
}

typedef struct {
    const ora_index *x; const uint8_t *concat; const uint64_t *off;
    size_t begin, end, threshold; uint8_t *chars_out, *d_out; ora_counters ctr; int rc; int count;
} batch_job;

static void *batch_worker(void *arg)
{
    batch_job *j = (batch_job *)arg;
    size_t maxlen = 0;
    for (size_t r = j->begin; r < j->end; r++) {
        size_t L = (size_t)(j->off[r + 1] - j->off[r]);
        if (L > maxlen) maxlen = L;
    }
    uint64_t *d = (uint64_t *)malloc((maxlen + 1) * sizeof(uint64_t));
    int64_t *der = (int64_t *)malloc((maxlen + 1) * sizeof(int64_t));
    uint32_t *tr = (uint32_t *)malloc((maxlen + 1) * sizeof(uint32_t));
    for (size_t r = j->begin; r < j->end; r++) {
        size_t o = (size_t)j->off[r], L = (size_t)(j->off[r + 1] - j->off[r]);
        int rc = matches_one(j->x, j->concat + o, L, j->threshold, d, der, tr,
                             j->chars_out ? j->chars_out + o : NULL,
                             j->d_out ? j->d_out + o : NULL, j->count ? &j->ctr : NULL);
        if (rc && !j->rc) j->rc = rc;
    }
    free(d); free(der); free(tr);
    return NULL;
}

int ora_matches_batch(const ora_index *x, const uint8_t *concat, const uint64_t *offsets,
                      size_t n_reads, double max_error_prob, int n_threads,
                      uint8_t *chars_out, uint8_t *d_out, ora_counters *ctr)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_reads) n_threads = n_reads ? (int)n_reads : 1;
    size_t threshold = ora_random_match_threshold(x->k, x->n_kmers, 4, max_error_prob);
    batch_job *jobs = (batch_job *)calloc((size_t)n_threads, sizeof(batch_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].x = x; jobs[t].concat = concat; jobs[t].off = offsets;
        jobs[t].begin = n_reads * (size_t)t / (size_t)n_threads;
        jobs[t].end = n_reads * (size_t)(t + 1) / (size_t)n_threads;
        jobs[t].threshold = threshold; jobs[t].chars_out = chars_out; jobs[t].d_out = d_out;
        jobs[t].count = ctr != NULL;
        pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    int rc = ORA_OK;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc && !rc) rc = jobs[t].rc;
        if (ctr) {
            ctr->bases += jobs[t].ctr.bases; ctr->extend_calls += jobs[t].ctr.extend_calls;
            ctr->rank_calls += jobs[t].ctr.rank_calls; ctr->rank_blocks += jobs[t].ctr.rank_blocks;
            ctr->contracts += jobs[t].ctr.contracts; ctr->lcs_reads += jobs[t].ctr.lcs_reads;
        }
    }
    free(jobs); free(th);
    return rc;
}

/* ---- timed batch driver for bench.py's cpu_baseline leg: the same A1->A5->A6 chain per read, but with what a
 * production host loop would have: a thread pool created once (pinned, one thread per core), outputs and scratch
 * allocated and touched before the clock starts (an untimed warm-up pass), reads handed out dynamically in chunks of
 * 256, and a barrier around every timed pass.  *seconds_out = wall time of the `passes` timed passes. */
#include <sched.h>
#include <stdatomic.h>
#include <time.h>

typedef struct {
    const ora_index *x; const uint8_t *concat; const uint64_t *off; size_t n_reads, threshold, maxlen;
    uint8_t *chars_out, *d_out; int passes, n_threads;
    atomic_size_t next; pthread_barrier_t bar; atomic_int rc;
} pool_job;
typedef struct { pool_job *j; int tid; } pool_arg;

static void *pool_worker(void *arg)
{
    pool_arg *a = (pool_arg *)arg;
    pool_job *j = a->j;
    /* pinned to the tid-th CPU of the PROCESS's affinity mask (inside a cpuset CPU number tid may not be ours) */
    cpu_set_t allowed, set;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
        int n_allowed = CPU_COUNT(&allowed);
        if (n_allowed > 0) {
            int want = a->tid % n_allowed, seen = 0;
            for (int c = 0; c < CPU_SETSIZE; c++) {
                if (!CPU_ISSET(c, &allowed)) continue;
                if (seen++ == want) {
                    CPU_ZERO(&set);
                    CPU_SET(c, &set);
                    pthread_setaffinity_np(pthread_self(), sizeof set, &set);
                    break;
                }
            }
        }
    }
    uint64_t *d = (uint64_t *)malloc((j->maxlen + 1) * sizeof(uint64_t));
    int64_t *der = (int64_t *)malloc((j->maxlen + 1) * sizeof(int64_t));
    uint32_t *tr = (uint32_t *)malloc((j->maxlen + 1) * sizeof(uint32_t));
    for (int pass = 0; pass <= j->passes; pass++) { /* pass 0 is the warm-up */
        pthread_barrier_wait(&j->bar); /* main has reset `next` */
        for (;;) {
            size_t b = atomic_fetch_add(&j->next, 256), e = b + 256 < j->n_reads ? b + 256 : j->n_reads;
            if (b >= j->n_reads) break;
            for (size_t r = b; r < e; r++) {
                size_t o = (size_t)j->off[r], L = (size_t)(j->off[r + 1] - j->off[r]);
                int rc = matches_one(j->x, j->concat + o, L, j->threshold, d, der, tr,
                                     j->chars_out ? j->chars_out + o : NULL, j->d_out ? j->d_out + o : NULL, NULL);
                if (rc) atomic_store(&j->rc, rc);
            }
        }
        pthread_barrier_wait(&j->bar); /* pass done */
    }
    free(d); free(der); free(tr);
    return NULL;
}

int ora_matches_batch_timed(const ora_index *x, const uint8_t *concat, const uint64_t *offsets, size_t n_reads,
                            double max_error_prob, int n_threads, int passes, uint8_t *chars_out, uint8_t *d_out,
                            double *seconds_out)
{
    if (n_threads < 1) n_threads = 1;
    if (passes < 1) passes = 1;
    pool_job j;
    memset(&j, 0, sizeof j);
    j.x = x; j.concat = concat; j.off = offsets; j.n_reads = n_reads; j.chars_out = chars_out; j.d_out = d_out;
    j.passes = passes; j.n_threads = n_threads;
    j.threshold = ora_random_match_threshold(x->k, x->n_kmers, 4, max_error_prob);
    for (size_t r = 0; r < n_reads; r++) {
        size_t L = (size_t)(offsets[r + 1] - offsets[r]);
        if (L > j.maxlen) j.maxlen = L;
    }
    atomic_init(&j.next, 0);
    atomic_init(&j.rc, 0);
    pthread_barrier_init(&j.bar, NULL, (unsigned)n_threads + 1);
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    pool_arg *args = (pool_arg *)calloc((size_t)n_threads, sizeof(pool_arg));
    for (int t = 0; t < n_threads; t++) {
        args[t].j = &j; args[t].tid = t;
        pthread_create(&th[t], NULL, pool_worker, &args[t]);
    }
    struct timespec t0, t1;
    for (int pass = 0; pass <= passes; pass++) {
        atomic_store(&j.next, 0);
        if (pass == 1) clock_gettime(CLOCK_MONOTONIC, &t0);
        pthread_barrier_wait(&j.bar); /* start */
        pthread_barrier_wait(&j.bar); /* done */
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    pthread_barrier_destroy(&j.bar);
    free(th); free(args);
    if (seconds_out) *seconds_out = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    return atomic_load(&j.rc);
}

/* ---- the first pass of call_variants over a batch of reads (variant_calling.rs:266-273, statement for statement per
 * read): the breakpoint predicate on the MS values and the search for the closest unique match to its right.  Records
 * {read, i, j, ref_colex} in read order (threads own contiguous ranges of reads, put together afterwards).  Returns the
 * number of sites (records beyond `cap` are counted, not written) or a negative ORA_E_* code. */
typedef struct {
    const ora_index *x; const uint8_t *concat; const uint64_t *off; size_t begin, end, threshold;
    uint64_t *recs; size_t n, cap; int rc;
} sites_job;

static void *sites_worker(void *arg)
{
    sites_job *j = (sites_job *)arg;
    size_t maxlen = 0;
    for (size_t r = j->begin; r < j->end; r++) {
        size_t L = (size_t)(j->off[r + 1] - j->off[r]);
        if (L > maxlen) maxlen = L;
    }
    uint64_t *d = (uint64_t *)malloc((maxlen + 1) * sizeof(uint64_t));
    uint64_t *lo = (uint64_t *)malloc((maxlen + 1) * sizeof(uint64_t));
    uint64_t *hi = (uint64_t *)malloc((maxlen + 1) * sizeof(uint64_t));
    const size_t k = j->x->k, t = j->threshold;
    for (size_t r = j->begin; r < j->end; r++) {
        const size_t o = (size_t)j->off[r], len = (size_t)(j->off[r + 1] - j->off[r]);
        int rc = ora_matching_statistics(j->x, j->concat + o, len, d, lo, hi, NULL); /* variant_calling.rs:266 */
        if (rc) { j->rc = rc; break; }
        for (size_t i = 1; i < len; i++) {                                       /* :268 */
            if (d[i] < d[i - 1] && d[i - 1] >= t && d[i] < t) {                    /* :269 */
                const size_t jend = i + k + 1 < len ? i + k + 1 : len;           /* :271 */
                for (size_t q = i + 1; q < jend; q++) {
                    if (d[q] >= t && hi[q] - lo[q] == 1) {                        /* :272 */
                        if (j->n == j->cap) {
                            j->cap = j->cap ? 2 * j->cap : 1024;
                            j->recs = (uint64_t *)realloc(j->recs, j->cap * 4 * sizeof(uint64_t));
                        }
                        uint64_t *w = j->recs + 4 * j->n++;
                        w[0] = r; w[1] = i; w[2] = q; w[3] = lo[q];              /* :273 ref_colex */
                        break;                                                   /* :289 */
                    }
                }
            }
        }
    }
    free(d); free(lo); free(hi);
    return NULL;
}

long ora_call_sites_batch(const ora_index *x, const uint8_t *concat, const uint64_t *offsets, size_t n_reads,
                          size_t threshold, int n_threads, uint64_t *recs_out, size_t cap)
{
    if (n_threads < 1) n_threads = 1;
    if ((size_t)n_threads > n_reads) n_threads = n_reads ? (int)n_reads : 1;
    sites_job *jobs = (sites_job *)calloc((size_t)n_threads, sizeof(sites_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].x = x; jobs[t].concat = concat; jobs[t].off = offsets; jobs[t].threshold = threshold;
        jobs[t].begin = n_reads * (size_t)t / (size_t)n_threads;
        jobs[t].end = n_reads * (size_t)(t + 1) / (size_t)n_threads;
        pthread_create(&th[t], NULL, sites_worker, &jobs[t]);
    }
    long total = 0;
    int rc = ORA_OK;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc && !rc) rc = jobs[t].rc;
        for (size_t s = 0; s < jobs[t].n; s++, total++)
            if ((size_t)total < cap && recs_out) memcpy(recs_out + 4 * total, jobs[t].recs + 4 * s, 4 * sizeof(uint64_t));
        free(jobs[t].recs);
    }
    free(jobs); free(th);
    return rc ? (long)rc : total;
}

/* --------------------------------------------------- format.rs:98-193 */

size_t ora_run_lengths_gapped(const uint8_t *aln, size_t len, size_t max_gap_len,
                              ora_rle *out, size_t cap)
{ /* format.rs:147-192, statement for statement */
    size_t n_out = 0, i = 0;
    int match_start = 0;
    while (i < len) {
        match_start = (aln[i] != '-' && aln[i] != ' ') && !match_start;
        if (match_start) {
            ora_rle rle = {i, 0, 0, 0, 0, 0, 0};
            size_t within_gap_bases = 0;
            int within_gap_start = 0;
            while (i < len && aln[i] != ' ') {
                int is_true_gap = aln[i] == '-';
                if (is_true_gap && !within_gap_start) {
                    within_gap_start = 1;
                    rle.gap_opens += 1;
                    within_gap_bases = 0;
                }
                if (!is_true_gap && within_gap_start) within_gap_start = 0;
                int is_match = aln[i] == 'M' || aln[i] == 'R' || aln[i] == 'I';
                int is_gap = is_true_gap || aln[i] == 'D';
                rle.matches += (uint64_t)is_match;
                rle.gap_bases += (uint64_t)is_gap;
                rle.mismatches += (uint64_t)(!is_match && !is_gap);
                rle.end = (is_match || !is_gap) ? i + 1 : rle.end;
                rle.jumps += (uint64_t)(aln[i] == 'R' && i > 0 && aln[i - 1] == 'R');
                within_gap_bases += (aln[i] == '-');
                i += 1;
                if (within_gap_bases > max_gap_len ||
                    (is_gap && i == len && rle.gap_opens > 0)) {
                    rle.gap_opens -= 1;
                    rle.gap_bases -= within_gap_bases;
                    break;
                }
            }
            if (n_out < cap && out) out[n_out] = rle;
            n_out++;
            match_start = 0;
        } else {
            i += 1;
        }
    }
    return n_out;
}

/* format::run_lengths_gapped (format.rs:143-193) of every alignment of a batch, the literal loop above per alignment:
 * rle_offsets[s] .. [s+1] = the runs of alignment s in recs_out (seven u64 each, capacity cap records).  Returns the
 * number of runs (those beyond cap are counted, not written). */
size_t ora_run_lengths_batch(const uint8_t *aln_concat, const uint64_t *offsets, size_t n_seqs, size_t max_gap_len,
                             ora_rle *recs_out, size_t cap, uint64_t *rle_offsets)
{
    size_t total = 0;
    for (size_t s = 0; s < n_seqs; s++) {
        rle_offsets[s] = total;
        total += ora_run_lengths_gapped(aln_concat + offsets[s], (size_t)(offsets[s + 1] - offsets[s]), max_gap_len,
                                        total < cap ? recs_out + total : NULL, total < cap ? cap - total : 0);
    }
    rle_offsets[n_seqs] = total;
    return total;
}

void ora_relative_to_ref(const uint8_t *ref_seq, const uint8_t *aln, size_t len, uint8_t *out)
{ /* format.rs:270-286 */
    for (size_t i = 0; i < len; i++) {
        uint8_t a = aln[i];
        if (a == 'M' || a == 'R' || a == 'I') out[i] = ref_seq[i];
        else if (a == 'X') out[i] = '-';
        else if (a == 'D') out[i] = '-';
        else if (a != '-') out[i] = a;
        else out[i] = '-';
    }
}
