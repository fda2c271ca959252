/*
 * kbo_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the k-bounded matching-statistics path of
 * tmaklin/kbo v0.5.1:  matching_statistics -> derandomize_ms_vec ->
 * translate_ms_vec (+ format::run_lengths_gapped / relative_to_ref).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libkbo_hip.so) never links or calls it.
 *
 * PARITY PINNING: the upstream crate is Rust and its hot loop lives in the
 * un-vendored dependency `sbwt = "0.3.4"` (reference Cargo.toml:18); neither
 * can be compiled in this image (no rustc/cargo).  The restatement is pinned
 * instead by every golden vector the reference's own tests/doctests hold for
 * this path (tests/golden/, see tests/test_oracle_golden.py): MS values
 * (index.rs:265-273), derandomize (derandomize.rs:298-379), translate
 * (translate.rs:396-532), matches/map/find doctests (lib.rs:600-609, 647-717,
 * 786-805), run_lengths (format.rs:295-330), plus the call/add_variants
 * goldens that exercise intervals and access_kmer.
 */
#ifndef KBO_ORACLE_H
#define KBO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_index ora_index;

/* Operation counters of the reference algorithm:
 * used to derive the algorithmic bytes per query base. */
typedef struct {
    uint64_t bases;        /* query bases processed                        */
    uint64_t extend_calls; /* extend_right invocations                     */
    uint64_t rank_calls;   /* rank queries (2 per extend)                  */
    uint64_t rank_blocks;  /* distinct 512-bit rank blocks touched/extend  */
    uint64_t contracts;    /* contract_left invocations                    */
    uint64_t lcs_reads;    /* LCS elements read by contract_left           */
} ora_counters;

/* ---- index (semantics of sbwt::SbwtIndex<SubsetMatrix> + LcsArray as built at
 *      reference index.rs:56-99; abstract content per SURVEY.md §8(a) A0) ---- */
int ora_index_build(const uint8_t *const *seqs, const size_t *lens, size_t n_seqs,
                    uint32_t k, int add_revcomp, ora_index **out);
int ora_index_from_parts(uint32_t k, uint64_t n_sets, uint64_t n_kmers,
                         const uint64_t *const rows[4], const uint64_t C[4],
                         const uint8_t *lcs, ora_index **out);
void ora_index_free(ora_index *idx);
uint32_t ora_index_k(const ora_index *idx);
uint64_t ora_index_n_sets(const ora_index *idx);
uint64_t ora_index_n_kmers(const ora_index *idx);
void ora_index_C(const ora_index *idx, uint64_t C[4]);
const uint64_t *ora_index_bits(const ora_index *idx, int c); /* ceil(n/64) words */
const uint8_t *ora_index_lcs(const ora_index *idx);          /* n bytes          */
/* sbwt access_kmer: writes k chars ('$','A','C','G','T') of row `colex`;
 * only available for indexes made by ora_index_build. */
int ora_index_access_kmer(const ora_index *idx, uint64_t colex, uint8_t *out_k);

/* ---- A1/A2: StreamingIndex::matching_statistics via index::query_sbwt
 *      (reference index.rs:243-256) ---- */
int ora_matching_statistics(const ora_index *idx, const uint8_t *query, size_t len,
                            uint64_t *d, uint64_t *lo, uint64_t *hi,
                            ora_counters *ctr /* nullable, accumulated */);

/* ---- A3: derandomize.rs:91-100, 127-145 ---- */
double ora_log_rm_max_cdf(size_t t, size_t alphabet_size, size_t n_kmers);
size_t ora_random_match_threshold(size_t k, size_t n_kmers, size_t alphabet_size,
                                  double max_error_prob);
/* ---- A4/A5: derandomize.rs:221-247, 269-288 ---- */
int64_t ora_derandomize_ms_val(size_t curr_noisy_ms, int64_t next_derand_ms,
                               size_t threshold, size_t k);
int ora_derandomize_ms_vec(const uint64_t *noisy_ms, size_t len, size_t k,
                           size_t threshold, int64_t *out);
/* ---- A6: translate.rs:180-216, 263-293 (chars as Rust `char` = u32) ---- */
void ora_translate_ms_val(int64_t curr, int64_t next, int64_t prev, size_t threshold,
                          uint32_t *aln_curr, uint32_t *aln_next);
int ora_translate_ms_vec(const int64_t *derand_ms, size_t len, size_t k,
                         size_t threshold, uint32_t *out);

/* ---- A7 glue: lib.rs:612-628 (matches) ; chars as bytes ---- */
int ora_matches(const ora_index *idx, const uint8_t *query, size_t len,
                double max_error_prob, uint8_t *chars_out);

/* ---- format.rs:18-193, 266-287 ---- */
typedef struct {
    uint64_t start, end, matches, mismatches, jumps, gap_bases, gap_opens;
} ora_rle;
/* returns number of RLEs (writes at most cap) */
size_t ora_run_lengths_gapped(const uint8_t *aln, size_t len, size_t max_gap_len,
                              ora_rle *out, size_t cap);
size_t ora_run_lengths_batch(const uint8_t *aln_concat, const uint64_t *offsets, size_t n_seqs, size_t max_gap_len,
                             ora_rle *recs_out, size_t cap, uint64_t *rle_offsets);
void ora_relative_to_ref(const uint8_t *ref_seq, const uint8_t *aln, size_t len,
                         uint8_t *out);

/* ---- batch driver used by bench.py's cpu_baseline leg: the same A1->A5->A6
 *      chain over many reads, split over `n_threads` pthreads (reads are
 *      independent; kbo itself is single-threaded per call, lib.rs:612-628).
 *      offsets has n_reads+1 entries into concat.  chars_out/d_out nullable. */
int ora_matches_batch(const ora_index *idx, const uint8_t *concat,
                      const uint64_t *offsets, size_t n_reads, double max_error_prob,
                      int n_threads, uint8_t *chars_out, uint8_t *d_out,
                      ora_counters *ctr);

/* the same with a pinned thread pool, pre-touched outputs, dynamic hand-out of reads and `passes` timed passes after an
 * untimed warm-up pass; *seconds_out = wall time of the timed passes */
int ora_matches_batch_timed(const ora_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_reads,
                            double max_error_prob, int n_threads, int passes, uint8_t *chars_out, uint8_t *d_out,
                            double *seconds_out);

/* error codes (mirror the reference's asserts) */
#define ORA_OK 0
#define ORA_E_EMPTY_QUERY (-1) /* index.rs:248 */
#define ORA_E_LEN_LE_2 (-2)    /* derandomize.rs:276, translate.rs:270 */
#define ORA_E_THRESHOLD (-3)   /* derandomize.rs:275, translate.rs:269 */
#define ORA_E_BAD_ARG (-4)
#define ORA_E_NOMEM (-5)

#ifdef __cplusplus
}
#endif

/* ---- refinement stages ("next" rows of SURVEY.md §8(f)); see kbo_oracle_refine.c ---- */
#ifdef __cplusplus
extern "C" {
#endif
typedef struct {
    uint64_t query_pos;
    uint8_t query_chars[256]; uint32_t query_len; /* a variant is at most k <= 255 characters; longer would set overflow */
    uint8_t ref_chars[256]; uint32_t ref_len;
    uint32_t overflow;
} ora_variant;
/* kbo::call (lib.rs:547-573) = variant_calling::call_variants (variant_calling.rs:249-294):
 * returns the number of variants (writes at most cap), or a negative ORA_E_* code. */
/* first pass of call_variants over a batch (variant_calling.rs:266-273 per read): records {read, i, j, ref_colex} as
 * four u64, in read order; returns the number of sites (those beyond cap are not written) or a negative code */
long ora_call_sites_batch(const ora_index *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_reads,
                          size_t threshold, int n_threads, uint64_t *recs_out, size_t cap);
long ora_call(const ora_index *query_idx, const uint8_t *ref_seq, size_t len, uint32_t k, double max_error_prob,
              ora_variant *out, size_t cap);
/* translate::add_variants (translate.rs:350-386), in place on byte chars */
int ora_add_variants(uint8_t *translation, size_t len, const ora_variant *v, size_t n);
/* gap_filling::fill_gaps (gap_filling.rs:444-526) on byte chars; ms = (d, lo, hi) arrays */
int ora_fill_gaps(const ora_index *idx, const uint8_t *translation, const uint64_t *d, const uint64_t *lo,
                  const uint64_t *hi, const uint8_t *ref_seq, size_t len, size_t threshold, double max_err_prob,
                  uint8_t *out);
/* kbo::map (lib.rs:720-761) with all options */
int ora_map(const ora_index *query_idx, const uint8_t *ref_seq, size_t len, uint32_t k, double max_error_prob,
            int fill_gaps, int call_variants, int format, uint8_t *out);
#define ORA_E_PANIC (-6) /* the reference would panic (index out of bounds / usize underflow / assert!) */
/* ---- plan_model.c: CPU model of the PRODUCT's plan-guided A1 stage (kbo_amd/csrc/plan_kernels.hip).  It repeats the
 * stage's decisions read by read, produces the stage's MS values by the stage's own means (the caller checks them against
 * ora_matching_statistics) and counts its work; bench.py's roofline prices the stage by these counts. ---- */
typedef struct {
    uint32_t seed_table_depth; /* bases per seed-table entry (0 = no table): what the device copy carries            */
    uint32_t seed_depth;       /* a single-row seed must be this deep (capped at k): log4(rows) + 3 by default        */
    uint32_t seed_cap;         /* bases of an item in which a seed may start (64)                                     */
    uint32_t gap;              /* mismatches closer than this share a unit: log4(rows) + 9 by default                 */
    uint32_t chunk;            /* bases per unit of an item without a plan (32)                                       */
    uint32_t list_cap;         /* mismatches an item's record + list hold (13 for reads)                              */
    uint32_t bail_x16;         /* more units than this / 16 per read of 150 bases: the plan is given up (50)          */
    uint32_t recovery_lines;   /* 0: rank blocks + entries (ms_walk_guided_kernel); 1: recovery lines (>= 24 Mi rows) */
    uint32_t depth_table;      /* 0: units + guided walk; else the order of the depth table (dtab_kernels.hip): the bases
                                * behind mismatches are looked up, no units                                             */
    uint32_t depth_anchors;    /* ... with anchors: a base deeper than the table knows is read off the path-cover text     */
} ora_plan_params;

typedef struct { /* all u64; per launch (the batch handed in) */
    uint64_t bases, items, items_unseeded, items_clean, items_list_overflow, items_flagged, gave_up;
    uint64_t seed_lookups, seed_extensions, pos_lookups, compare_bases, mismatches;
    uint64_t units_counted, units, units_head, units_plain, node_lookups;
    uint64_t walk_accepted, walk_failed, walk_contractions, walk_entry_levels, walk_short_windows, walk_iterations_lines;
    uint64_t walk_out_bytes, unit_distinct_lines; /* distinct 128-byte lines a unit touches, summed over the units */
    uint64_t unit_distinct_rank_lines;            /* ... of which rank-block lines (3.3 MB at C2: they stay in a 4 MiB L2) */
    uint64_t redo_bases, redo_iterations;
    uint64_t tab_lookups, tab_written, tab_flagged; /* depth-table form: look-ups, values written from them, items it could not resolve */
    uint64_t tab_anchored, items_noplan;            /* ... bases deeper than the table knows (tried through the anchors), items without a plan */
    uint64_t tab_stretches;                         /* ... mismatches whose stretch was looked up (model only) */
} ora_plan_counts;

/* text / pos / node_at: the path cover of the index (n_sets entries each: kbo_index_path_cover of the product, whose
 * claims tests/test_path_cover.py checks against the subset matrix).  ms_out: offsets[n_reads] bytes. */
int ora_plan_model(const ora_index *idx, const uint8_t *text, const uint32_t *pos, const uint32_t *node_at,
                   const ora_plan_params *params, const uint8_t *concat, const uint64_t *offsets, size_t n_reads,
                   int n_threads, uint8_t *ms_out, ora_plan_counts *counts);

/* ---- index_check.c: ranks of k-mers among the k-mers of a text, counted straight off the text (a check of an index's row
 * order that passes through no index builder).  For each of the m query k-mers (k <= 64 bytes each, back to back):
 * less_out = occurrences of colex-smaller k-mers in seq (windows with a non-ACGT byte do not count), equal_out =
 * occurrences of the k-mer itself; a query with a non-ACGT byte gets (UINT64_MAX, 0). */
int ora_kmer_colex_ranks(const uint8_t *seq, size_t len, uint32_t k, const uint8_t *kmers, size_t m, int n_threads,
                         uint64_t *less_out, uint64_t *equal_out);

#ifdef __cplusplus
}
#endif
#endif /* KBO_ORACLE_H */
