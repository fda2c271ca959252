/*
 * index_check.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as kbo_oracle.h).
 *
 * A check of an SBWT's row ORDER that passes through neither the product's builder nor the oracle's: the rank of a
 * k-mer among the k-mers of the input, counted straight off the input text.  The abstract content of the index the
 * reference builds at index.rs:56-99 has its rows in colexicographic order ($ < A < C < G < T,
 * compared from the LAST character backwards), so for an input whose k-mers are all distinct the row of a k-mer is
 * (number of input k-mers that are colex-smaller) + (number of $-padded dummy rows that are colex-smaller) - a number
 * this file computes by one threaded pass over the text, and the tests compare with the interval the product's walk of
 * that k-mer ends in, and with the rows the subset matrix spells (tests/test_gpu_configs.py).
 */
#include "kbo_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static inline int code_of(uint8_t ch)
{
    switch (ch) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}

/* colex key of the k characters at s (k <= 64): the last character is the most significant digit; -1 in *ok when a
 * character is not A, C, G or T */
static u128 key_of(const uint8_t *s, uint32_t k, int *ok)
{
    u128 key = 0;
    *ok = 1;
    for (uint32_t j = 0; j < k; j++) {
        int c = code_of(s[j]);
        if (c < 0) { *ok = 0; return 0; }
        key |= (u128)(unsigned)c << (2u * j);
    }
    return key;
}

typedef struct {
    const uint8_t *seq; size_t lo, hi; uint32_t k; /* k-mer start positions [lo, hi) */
    const u128 *sorted; size_t m;                  /* the queries' keys, ascending */
    uint64_t *less_hist, *eq_hist;                 /* m + 1 / m counters of this thread */
} rank_job;

static void *rank_worker(void *arg)
{
    rank_job *j = (rank_job *)arg;
    const uint32_t k = j->k;
    u128 key = 0;
    uint32_t valid = 0; /* characters of the current window that are bases, counted from its end */
    /* rolling: the window ending at position e = p + k - 1; key' = key >> 2 | code(e) << 2 (k - 1) */
    size_t e0 = j->lo, e1 = j->hi + k - 1; /* characters [lo, hi + k - 1) */
    for (size_t e = e0; e < e1; e++) {
        int c = code_of(j->seq[e]);
        if (c < 0) { valid = 0; key = 0; continue; }
        key = (key >> 2) | ((u128)(unsigned)c << (2u * (k - 1u)));
        if (valid < k) valid++;
        if (valid < k) continue;
        /* first query key >= key */
        size_t a = 0, b = j->m;
        while (a < b) {
            size_t mid = a + (b - a) / 2;
            if (j->sorted[mid] < key) a = mid + 1; else b = mid;
        }
        /* this k-mer is smaller than every query from a on that is not equal to it */
        if (a < j->m && j->sorted[a] == key) {
            j->eq_hist[a]++;   /* (equal queries: counted at the first of them, spread below) */
            j->less_hist[a]++; /* smaller than the queries behind the equal ones: fixed up by the caller */
        } else {
            j->less_hist[a]++;
        }
    }
    return NULL;
}

/* For each of the m query k-mers (k bytes each, back to back; k <= 64): less_out = number of k-mer OCCURRENCES in seq
 * (windows of k bases; windows with a non-ACGT byte do not count) that are colex-smaller, equal_out = occurrences equal
 * to it.  A query with a non-ACGT byte gets (UINT64_MAX, 0). */
int ora_kmer_colex_ranks(const uint8_t *seq, size_t len, uint32_t k, const uint8_t *kmers, size_t m, int n_threads,
                         uint64_t *less_out, uint64_t *equal_out)
{
    if (k == 0 || k > 64 || !seq || (!kmers && m)) return ORA_E_BAD_ARG;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    u128 *keys = (u128 *)malloc((m + 1) * sizeof(u128));
    size_t *order = (size_t *)malloc((m + 1) * sizeof(size_t));
    u128 *sorted = (u128 *)malloc((m + 1) * sizeof(u128));
    uint8_t *okv = (uint8_t *)malloc(m + 1);
    if (!keys || !order || !sorted || !okv) { free(keys); free(order); free(sorted); free(okv); return ORA_E_NOMEM; }
    size_t mv = 0;
    for (size_t i = 0; i < m; i++) {
        int ok;
        keys[i] = key_of(kmers + i * k, k, &ok);
        okv[i] = (uint8_t)ok;
        if (ok) order[mv++] = i;
    }
    /* sort the valid queries by key (shell sort over the index array: m is a few 10^4) */
    for (size_t gap = mv / 2; gap > 0; gap /= 2)
        for (size_t i = gap; i < mv; i++) {
            size_t t = order[i], jx = i;
            while (jx >= gap && keys[order[jx - gap]] > keys[t]) { order[jx] = order[jx - gap]; jx -= gap; }
            order[jx] = t;
        }
    for (size_t i = 0; i < mv; i++) sorted[i] = keys[order[i]];
    const size_t n_pos = len >= k ? len - k + 1 : 0;
    uint64_t *hist = (uint64_t *)calloc((size_t)n_threads * (2 * mv + 2), sizeof(uint64_t));
    rank_job *jobs = (rank_job *)calloc((size_t)n_threads, sizeof(rank_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!hist || !jobs || !th) { free(keys); free(order); free(sorted); free(okv); free(hist); free(jobs); free(th); return ORA_E_NOMEM; }
    for (int t = 0; t < n_threads; t++) {
        jobs[t].seq = seq; jobs[t].k = k; jobs[t].sorted = sorted; jobs[t].m = mv;
        jobs[t].lo = n_pos * (size_t)t / (size_t)n_threads;
        jobs[t].hi = n_pos * (size_t)(t + 1) / (size_t)n_threads;
        jobs[t].less_hist = hist + (size_t)t * (2 * mv + 2);
        jobs[t].eq_hist = jobs[t].less_hist + mv + 1;
        pthread_create(&th[t], NULL, rank_worker, &jobs[t]);
    }
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    /* occurrences whose first query >= them is query a: smaller than the queries a .. (unless equal to them) */
    uint64_t run = 0;
    for (size_t a = 0; a < mv;) {
        uint64_t h = 0, eq = 0;
        for (int t = 0; t < n_threads; t++) { h += jobs[t].less_hist[a]; eq += jobs[t].eq_hist[a]; }
        /* the queries equal to sorted[a] share these counts */
        size_t b = a;
        while (b < mv && sorted[b] == sorted[a]) b++;
        for (size_t x = a; x < b; x++) {
            less_out[order[x]] = run + (h - eq); /* occurrences that sorted to a but are strictly smaller */
            equal_out[order[x]] = eq;
        }
        for (size_t x = a + 1; x < b; x++) /* (nothing sorts to the later duplicates) */
            for (int t = 0; t < n_threads; t++) { h += jobs[t].less_hist[x]; }
        run += h;
        a = b;
    }
    for (size_t i = 0; i < m; i++)
        if (!okv[i]) { less_out[i] = UINT64_MAX; equal_out[i] = 0; }
    free(keys); free(order); free(sorted); free(okv); free(hist); free(jobs); free(th);
    return ORA_OK;
}
