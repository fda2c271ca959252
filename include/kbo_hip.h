/*
 * kbo_hip.h — C ABI of the MI355X-native k-bounded matching-statistics path of kbo.
 *
 * This is the drop-in boundary: every entry point below is what a binding of the
 * reference crate (tmaklin/kbo v0.5.1, Rust) would call instead of its CPU path.
 * Each declaration cites the reference interface it replaces (file:line under the
 * reference's src/).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - return 0 (KBO_OK) or a negative KBO_E_* code; each code mirrors one of the
 *     reference's assert!/panic! sites.  kbo_last_error() gives a thread-local text.
 *   - the caller owns all in/out buffers; the index handle is immutable after
 *     construction, so concurrent calls on one handle are allowed.
 *   - "host" entry points take host pointers and do H2D/D2H themselves;
 *     "*_dev" entry points take pointers that are already resident in the HBM of
 *     the current HIP device plus a hipStream_t (passed as void*), enqueue their
 *     kernels on that stream and return without synchronising.
 *   - ALL matching-statistics work runs in the HIP kernels (gfx950).  There is no
 *     CPU fallback: without a usable GPU the compute entry points fail with KBO_E_HIP.
 */
#ifndef KBO_HIP_H
#define KBO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KBO_OK 0
#define KBO_E_EMPTY_QUERY (-1)    /* index.rs:248   assert!(!query.is_empty())          */
#define KBO_E_LEN_LE_2 (-2)       /* derandomize.rs:276, translate.rs:270  len > 2      */
#define KBO_E_THRESHOLD_LE_1 (-3) /* derandomize.rs:275, translate.rs:269  threshold > 1*/
#define KBO_E_BAD_ARG (-4)        /* derandomize.rs:96-97,133-137,274; null/invalid arg */
#define KBO_E_NOMEM (-5)
#define KBO_E_K_MISMATCH (-6)     /* lib.rs:559,729  k of index != k of opts            */
#define KBO_E_HIP (-7)            /* HIP runtime error / no device / kernel image missing */
#define KBO_E_UNSUPPORTED (-8)    /* valid in the reference, not (yet) built here        */
#define KBO_E_MS_RANGE (-9)       /* derandomize.rs:229-230  noisy/derand value > k     */
#define KBO_E_IO (-10)            /* index.rs:137,202 file open/read/write failure      */
#define KBO_E_REF_PANIC (-11)     /* the reference would panic here (index/usize underflow, assert!) */

const char *kbo_last_error(void);
const char *kbo_version(void);

/* ------------------------------------------------------------------ options */

/* kbo::BuildOpts (lib.rs:259-313).  Only k and add_revcomp change the index content;
 * the remaining fields steer the sbwt crate's construction algorithm and are accepted
 * for signature compatibility (num_threads is honoured by this builder too). */
typedef struct {
    uint32_t k;              /* 31    */
    int32_t add_revcomp;     /* false */
    uint32_t num_threads;    /* 1     */
    uint32_t prefix_precalc; /* 8     */
    int32_t build_select;    /* false */
    uint32_t mem_gb;         /* 4     */
    int32_t dedup_batches;   /* false */
    const char *temp_dir;    /* NULL  */
} kbo_build_opts;
void kbo_build_opts_default(kbo_build_opts *o); /* lib.rs:300-313 */

/* kbo::FindOpts (lib.rs:358-382) */
typedef struct {
    double max_error_prob; /* 1e-7 */
    size_t max_gap_len;    /* 0    */
} kbo_find_opts;
void kbo_find_opts_default(kbo_find_opts *o);

/* kbo::MapOpts (lib.rs:412-466).  fill_gaps / call_variants select the host-side
 * refinement stages that follow the hot path (lib.rs:743-754). */
typedef struct {
    double max_error_prob; /* 1e-7 */
    int32_t fill_gaps;     /* true */
    int32_t call_variants; /* true */
    int32_t format;        /* true */
    kbo_build_opts sbwt_build_opts; /* build_select = true */
} kbo_map_opts;
void kbo_map_opts_default(kbo_map_opts *o);

/* kbo::CallOpts (lib.rs:318-353) */
typedef struct {
    double max_error_prob;          /* 1e-7 */
    kbo_build_opts sbwt_build_opts; /* build_select = true */
} kbo_call_opts;
void kbo_call_opts_default(kbo_call_opts *o);

/* kbo::variant_calling::Variant (variant_calling.rs:8-26).  Arrays returned by kbo_call live
 * in one allocation: release the whole result with a single kbo_free(variants). */
typedef struct {
    uint64_t query_pos;
    const uint8_t *query_chars;
    size_t query_len;
    const uint8_t *ref_chars;
    size_t ref_len;
} kbo_variant;

/* kbo::format::RLE (format.rs:18-33) */
typedef struct {
    uint64_t start, end, matches, mismatches, jumps, gap_bases, gap_opens;
} kbo_rle;

/* ------------------------------------------------------------------ index
 * Opaque stand-in for (sbwt::SbwtIndexVariant, sbwt::LcsArray) — the pair returned by
 * kbo::build (lib.rs:501-506) and consumed by every query function.  Owns the host
 * copy and one device-resident copy per GPU it has been used on. */
typedef struct kbo_index kbo_index_t;

/* kbo::build / index::build_sbwt_from_vecs (lib.rs:501-506, index.rs:56-99). */
int kbo_index_build(const uint8_t *const *seqs, const size_t *lens, size_t n_seqs,
                    const kbo_build_opts *opts, kbo_index_t **out);

/* Adopt an index built elsewhere (e.g. by the sbwt crate on the Rust side): the four
 * SubsetMatrix rows as little-endian 64-bit words (bit i of word i/64 = row i), the C
 * array and the LCS array as bytes.  Replaces handing &SbwtIndexVariant/&LcsArray to
 * index::query_sbwt (index.rs:243-247). */
int kbo_index_from_parts(uint32_t k, uint64_t n_sets, uint64_t n_kmers,
                         const uint64_t *const rows[4], const uint64_t C[4],
                         const uint8_t *lcs, kbo_index_t **out);
/* Inverse of the above; rows[c] must hold ceil(n_sets/64) words, lcs n_sets bytes. */
int kbo_index_export_parts(const kbo_index_t *idx, uint64_t *const rows[4], uint64_t C[4],
                           uint8_t *lcs);
void kbo_index_free(kbo_index_t *idx);

size_t kbo_index_k(const kbo_index_t *idx);         /* SbwtIndex::k()       lib.rs:620 */
uint64_t kbo_index_n_kmers(const kbo_index_t *idx); /* SbwtIndex::n_kmers() lib.rs:620 */
uint64_t kbo_index_n_sets(const kbo_index_t *idx);  /* SbwtIndex::n_sets()             */

/* index::serialize_sbwt / load_sbwt (index.rs:128-151, 195-212) — own flat,
 * device-ready file format "<prefix>.kbohip" (NOT the sbwt crate's .sbwt/.lcs).  While the plan-guided walk is enabled
 * (default) the file also carries the index's path cover (9 bytes per row; computed by the save if the handle has none
 * yet): laying it out is the one slow, serial part of making a device copy (26 s per 10^8 rows), so a loaded index uploads
 * in the time of a few streaming passes.  A cover read from a file is validated against the subset matrix (KBO_E_IO). */
int kbo_index_save(const kbo_index_t *idx, const char *path);
int kbo_index_load(const char *path, kbo_index_t **out);
/* index::serialize_sbwt / load_sbwt (index.rs:128-151, 195-212).  The reference writes <prefix>.sbwt + <prefix>.lcs: a
 * u64-LE length and the tag "SubsetMatrix" (index.rs:139-140), then the sbwt crate's own serialize() payload, which
 * nothing in the reference tree pins: PARITY UNPINNED.  These functions therefore keep their
 * own payload (pinned header + second tag "KBOSBWT1") under their OWN names, <prefix>.sbwt.kbohip + <prefix>.lcs.kbohip,
 * so that kbo-cli never mistakes the pair for a crate-written index; kbo_index_load_sbwt reads that pair and returns
 * KBO_E_UNSUPPORTED (not a guess at the crate's fields) when all it finds is a crate-written <prefix>.sbwt: such an index
 * comes in through kbo_index_from_parts (INTEGRATION.md has the Rust side).  KBO_E_IO when a file is missing, truncated
 * or inconsistent. */
int kbo_index_save_sbwt(const kbo_index_t *idx, const char *prefix);
int kbo_index_load_sbwt(const char *prefix, kbo_index_t **out);

/* Sharded indexes.  Row numbers are 32 bits on the device, so an index of 2^32 rows or more - a human genome WITH its reverse
 * complements, 6.2 * 10^9 rows - cannot be one index here.  kbo_index_build builds such an input as SHARDS: ordinary indexes
 * over groups of whole sequences, the forward and the reverse-complement strand of every group apart.  The strings (of at
 * most k characters) that are suffixes of an SBWT's rows are the substrings of its input's ACGT-runs of at least k characters
 * - a property of the sequences one by one - so the DEPTH of the walk against the index of everything is the maximum of the
 * depths against the shards, and n_kmers (what the threshold of the derandomisation needs) is counted over the union.
 * Everything that only needs depths therefore gives the union index's results, bit for bit, by walking every shard:
 * kbo_matches[_batch][_packed], kbo_map[_batch] without fill_gaps / call_variants, kbo_find[_batch][_packed], kbo_ms_batch /
 * kbo_matching_statistics without intervals, and the device-resident entry points (d_work: kbo_index_work_bytes).  What needs
 * ROWS of the union - intervals, kbo_call*, kbo_fill_gaps, full kbo_map, export / save, the path-cover getters - returns
 * KBO_E_UNSUPPORTED for such a handle.  kbo_index_shards: 1 for an ordinary index; kbo_index_n_sets counts the rows of all
 * shards. */
int kbo_index_shards(const kbo_index_t *idx);

/* Upload (idempotent) the device layout to HIP device `device` (-1 = current). */
int kbo_index_to_device(kbo_index_t *idx, int device);
/* Bytes of the device-resident layout: rank blocks / contraction entries {lcs,psv,nsv}. */
int kbo_index_device_bytes(const kbo_index_t *idx, uint64_t *rank_bytes, uint64_t *lcs_bytes);
/* bytes of two-base extension blocks a device copy of this index carries (0 = none, see kbo_set_pair_steps) */
uint64_t kbo_index_device_pair_bytes(const kbo_index_t *idx);
/* What the copy of `idx` on `device` (-1 = current) holds and what making it cost - the index is built once and queried many
 * times (index.rs:56-99 is not part of any query), but the plan structures are sized by log4(rows), not by the index, so a
 * caller that serves few queries per index wants to see the bill.  Bytes by part; entries_64bit = the contraction entries sit
 * in their own allocation behind 64-bit offsets (rank blocks + entries >= 4 GiB); seconds of host work + upload by part
 * (cover: laying out the path cover, 0 when the handle had one already - from an index file or an earlier copy).
 * KBO_E_BAD_ARG when there is no such copy, KBO_E_UNSUPPORTED for a sharded handle (ask its shards). */
typedef struct {
    uint64_t rank_bytes, entry_bytes, pair_bytes;                           /* what every walk needs */
    uint64_t cover_bytes, lines_bytes, seed_bytes, dtab_bytes, anchor_bytes; /* the plan structures (0: the copy has none) */
    uint32_t entries_64bit, seed_depth, dtab_order, dtab_grouped;
    double layout_seconds, upload_seconds, cover_seconds, lines_seconds, seed_seconds, dtab_seconds;
} kbo_device_layout;
int kbo_index_device_layout(kbo_index_t *idx, int device, kbo_device_layout *out);

/* ------------------------------------------------------------------ A3 (host, f64)
 * derandomize::log_rm_max_cdf (derandomize.rs:91-100) and
 * derandomize::random_match_threshold (derandomize.rs:127-145). */
int kbo_log_rm_max_cdf(size_t t, size_t alphabet_size, size_t n_kmers, double *out);
int kbo_random_match_threshold(size_t k, size_t n_kmers, size_t alphabet_size,
                               double max_error_prob, size_t *out);

/* ------------------------------------------------------------------ single-sequence
 * parity entry points with the reference's element widths (host pointers). */

/* index::query_sbwt -> Vec<(usize, Range<usize>)> (index.rs:243-256).
 * d/lo/hi each hold len elements; lo/hi may both be NULL. */
int kbo_matching_statistics(kbo_index_t *idx, const uint8_t *query, size_t len, uint64_t *d,
                            uint64_t *lo, uint64_t *hi);
/* derandomize::derandomize_ms_vec (derandomize.rs:269-288). */
int kbo_derandomize_ms_vec(const uint64_t *noisy_ms, size_t len, size_t k, size_t threshold,
                           int64_t *out);
/* derandomize::derandomize_ms_val (derandomize.rs:221-247) — scalar, host. */
int kbo_derandomize_ms_val(size_t curr_noisy_ms, int64_t next_derand_ms, size_t threshold,
                           size_t k, int64_t *out);
/* translate::translate_ms_vec (translate.rs:263-293); chars as Rust `char` (u32). */
int kbo_translate_ms_vec(const int64_t *derand_ms, size_t len, size_t k, size_t threshold,
                         uint32_t *out);
/* translate::translate_ms_val (translate.rs:180-216) — scalar, host. */
int kbo_translate_ms_val(int64_t ms_curr, int64_t ms_next, int64_t ms_prev, size_t threshold,
                         uint32_t *aln_curr, uint32_t *aln_next);
/* kbo::matches (lib.rs:612-628): chars as Rust `char` (u32), len elements. */
int kbo_matches(kbo_index_t *idx, const uint8_t *query, size_t len, double max_error_prob,
                uint32_t *chars_out);
/* kbo::map (lib.rs:720-761): out holds len bytes. */
int kbo_map(kbo_index_t *query_idx, const uint8_t *ref_seq, size_t len, const kbo_map_opts *opts,
            uint8_t *out);
/* kbo::call (lib.rs:547-573): builds an index of ref_seq, runs variant_calling::call_variants
 * (variant_calling.rs:249-294; all MS passes on the GPU, k-mer walks of the sites batched). */
int kbo_call(kbo_index_t *query_idx, const uint8_t *ref_seq, size_t len, const kbo_call_opts *opts,
             kbo_variant **out, size_t *n_out);
/* kbo::call over a batch: every sequence of (concat, offsets) plays ref_seq against the same query index.  The first
 * pass of call_variants (MS walk + the breakpoint scan, variant_calling.rs:266-273) runs on the device for the whole
 * batch and only the sites come back; the second pass walks all query-side k-mers in one batch and, per sequence, the
 * reference-side k-mers against that sequence's own index (lib.rs:553).  *out holds var_offsets[n_seqs] variants, those
 * of sequence s at [var_offsets[s], var_offsets[s+1]); one allocation, kbo_free(*out).  opts->sbwt_build_opts.k must
 * equal the index's k (lib.rs:559). */
int kbo_call_batch(kbo_index_t *query_idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                   const kbo_call_opts *opts, kbo_variant **out, uint64_t *var_offsets /* n_seqs + 1 */);
/* The same with the variants as flat arrays in the order of (sequence, query position) - the form they leave the device in
 * (call_emit_kernels.hip: the device sorts a slab's sites, runs resolve_variant's case analysis, variant_calling.rs:139-201, and slices
 * the characters; 10 bytes per variant cross PCIe instead of 164 per site): no record with two pointers per variant to fill, nothing for
 * a binding to copy.  Variant v of sequence s (v in [var_offsets[s], var_offsets[s + 1])) has query_pos[v], query_len[v] query characters
 * followed by ref_len[v] reference characters in `chars`, the variants' characters back to back in variant order.  Released by
 * kbo_call_flat_free(result).  kbo_call_batch is this + the reference's records made from it. */
typedef struct {
    uint64_t n_variants, n_chars;
    uint32_t *query_pos;
    uint16_t *query_len, *ref_len;
    uint8_t *chars;
} kbo_call_flat;
int kbo_call_batch_flat(kbo_index_t *query_idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                        const kbo_call_opts *opts, kbo_call_flat *result, uint64_t *var_offsets /* n_seqs + 1 */);
void kbo_call_flat_free(kbo_call_flat *result);
/* The first pass of call_variants in one go, device-resident: the walk of kbo_ms_batch_dev (MS values to d_ms_out, no
 * intervals) whose lanes run the breakpoint scan on the values they produce.  Sites are 16-byte records {offset of i in
 * d_concat, offset of j, row of ms[j], 0} in KBO_CALL_LISTS lists as below; d_count needs KBO_CALL_LISTS * 64 + 64 bytes:
 * the last counter is non-zero when some read had more than four breakpoints waiting within k bases - then (and when a
 * list overflowed) use kbo_ms_batch_dev with intervals + kbo_call_sites_dev instead.  A record whose first word is
 * 0xFFFFFFFF is void and to be skipped (plan-guided walk: a site of a read that was afterwards scanned again in full,
 * where the same site appears once more).  Asynchronous on `stream`. */
int kbo_call_walk_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs,
                      uint64_t total_bases, size_t max_seq_len, size_t threshold, uint8_t *d_ms_out, void *d_sites,
                      size_t capacity, uint32_t *d_count, void *d_work, size_t work_bytes, void *stream);
/* The breakpoint scan alone, device-resident: d_ms / d_lo / d_hi as kbo_ms_batch_dev wrote them (intervals requested).
 * Sites are 16-byte records {sequence, i, j, row of ms[j]} in KBO_CALL_LISTS lists (one counter would serialise the
 * appends): list g occupies d_sites[g * (capacity / KBO_CALL_LISTS) ...] and has d_count[16 g] records, in arrival
 * order; d_count is KBO_CALL_LISTS counters 64 bytes apart (16 KiB).  A counter above capacity / KBO_CALL_LISTS means
 * that list overflowed: repeat with more room.  Asynchronous on `stream`. */
#define KBO_CALL_LISTS 256
int kbo_call_sites_dev(const uint8_t *d_ms, const uint32_t *d_lo, const uint32_t *d_hi, const uint64_t *d_offsets,
                       size_t n_seqs, uint64_t total_bases, size_t k, size_t threshold, void *d_sites, size_t capacity,
                       uint32_t *d_count, void *stream);
/* translate::add_variants (translate.rs:350-386) on Rust-char (u32) alignment strings, in place. */
int kbo_add_variants(uint32_t *translation, size_t len, const kbo_variant *variants, size_t n_variants);
/* gap_filling::fill_gaps (gap_filling.rs:444-526) preceded by the steps its callers run
 * (lib.rs:735-747): query_sbwt, derandomize_ms_vec and translate_ms_vec with the GIVEN threshold
 * on the GPU, then the host-side gap filling.  out holds len Rust chars. */
int kbo_fill_gaps(kbo_index_t *query_idx, const uint8_t *ref_seq, size_t len, size_t threshold,
                  double max_err_prob, uint32_t *out);
/* gap_filling::nearest_unique_context (gap_filling.rs:127-151): kmer_out holds k bytes;
 * *kmer_len is 0 when no unique interval was found. */
int kbo_nearest_unique_context(kbo_index_t *idx, const uint8_t *ref_seq, size_t len, size_t range_start,
                               size_t range_end, size_t *kmer_idx, uint8_t *kmer_out, size_t *kmer_len);
/* kbo::find (lib.rs:808-821): *out is allocated by the library (kbo_free). */
int kbo_find(kbo_index_t *idx, const uint8_t *query, size_t len, const kbo_find_opts *opts,
             kbo_rle **out, size_t *n_out);
/* format::run_lengths_gapped (format.rs:143-193) and relative_to_ref (format.rs:266-287)
 * on byte-wide alignment strings (host; sequential variable-length output).  KBO_E_REF_PANIC for an alignment that starts
 * with 'R': the reference evaluates aln[i - 1] with i = 0 there (format.rs:175) and panics; translate_ms_vec never produces
 * one (an 'R' needs a derandomised value above the threshold, the first base's is at most 1), so the pipeline's entry points
 * cannot run into it. */
int kbo_run_lengths_gapped(const uint8_t *aln, size_t len, size_t max_gap_len, kbo_rle **out,
                           size_t *n_out);
/* format::run_lengths_gapped over a batch of alignments on the GPU (one lane per alignment): runs of
 * all sequences concatenated, rle_offsets[i]..[i+1] = alignment i (n_seqs+1 entries, caller-allocated;
 * *rles library-allocated, release with kbo_free).  Same records as kbo_run_lengths_gapped. */
int kbo_run_lengths_gapped_batch(const uint8_t *aln_concat, const uint64_t *offsets, size_t n_seqs,
                                 size_t max_gap_len, kbo_rle **rles, uint64_t *rle_offsets);
int kbo_relative_to_ref(const uint8_t *ref_seq, const uint8_t *aln, size_t len, uint8_t *out);
void kbo_free(void *p);

/* ------------------------------------------------------------------ batched entry points
 * (new surface: the reference takes ONE sequence per call, lib.rs:612-617; batching over
 * reads/contigs lives in kbo-cli).  `concat` holds all sequences back to back,
 * offsets[i]..offsets[i+1] delimits sequence i (n_seqs+1 entries).  Compact element
 * widths: d as u8 (d <= k <= 255), intervals as u32 (n_sets < 2^32), chars as u8. */
int kbo_ms_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                 uint8_t *d_out, uint32_t *lo_out, uint32_t *hi_out);
int kbo_matches_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets,
                      size_t n_seqs, double max_error_prob, uint8_t *chars_out);
/* map with fill_gaps=false, call_variants=false (lib.rs:735-738, 756-760) over a batch. */
int kbo_map_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                  double max_error_prob, int format, uint8_t *out);
/* find over a batch: RLEs of all sequences concatenated, rle_offsets[i]..[i+1] = sequence i
 * (rle_offsets has n_seqs+1 entries, caller-allocated; *rles library-allocated, kbo_free).  The
 * run lengths are computed on the device (the translated characters never leave it). */
int kbo_find_batch(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                   const kbo_find_opts *opts, kbo_rle **rles, uint64_t *rle_offsets);
/* The same into caller-owned memory (buffers reused from call to call cost no page faults): `rles`
 * holds `capacity` records; *n_runs receives the number of runs of the batch.  When that exceeds
 * `capacity` the call fails with KBO_E_NOMEM after filling rle_offsets and *n_runs (records beyond
 * the capacity are not written), so a second call with a large enough buffer succeeds. */
int kbo_find_batch_into(kbo_index_t *idx, const uint8_t *concat, const uint64_t *offsets, size_t n_seqs,
                        const kbo_find_opts *opts, kbo_rle *rles, size_t capacity, uint64_t *rle_offsets,
                        size_t *n_runs);

/* ------------------------------------------------------------------ packed batches
 * The batch entry points above move one byte per base over PCIe in each direction, which is what bounds them (about
 * 40 Gbp/s host -> host against 140 Gbp/s on the device).  Reads carry 2 bits per base and the alphabet of kbo::matches is
 * exactly { M, -, X, R } (translate.rs:180-216), so these entry points take and return 2-bit words: a quarter of the bytes,
 * the same kernels in between (the words are unpacked and packed on the device).
 * Layout, input and output alike: sequence s of len_s bases occupies ceil(len_s / 16) little-endian u32 words, the
 * sequences back to back in order (kbo_packed_words() in all); base i of a sequence sits in bits 2 (i mod 16), + 1 of its
 * word i / 16.  Input: A, C, G, T = 0 .. 3; every other byte of the reads (N, lower case, ...: they break matches like in
 * the reference) travels in a side list { position in base coordinates (offsets[] space), byte }, positions ascending,
 * and the 2 bits at such a position are ignored.  Output: M, -, X, R = 0 .. 3.
 * offsets[] counts BASES as everywhere else (n_seqs + 1 entries).  kbo_pack_reads / kbo_unpack_matches are host helpers
 * (threaded) for callers that hold bytes; KBO_E_NOMEM from kbo_pack_reads when the reads hold more than exc_cap
 * non-ACGT bases (*n_exc is set to their number). */
size_t kbo_packed_words(const uint64_t *offsets, size_t n_seqs);
int kbo_pack_reads(const uint8_t *concat, const uint64_t *offsets, size_t n_seqs, uint32_t *words_out, uint64_t *exc_pos,
                   uint8_t *exc_byte, size_t exc_cap, size_t *n_exc);
int kbo_unpack_matches(const uint32_t *words, const uint64_t *offsets, size_t n_seqs, uint8_t *chars_out);
/* kbo::matches (lib.rs:612-628) over a packed batch */
int kbo_matches_batch_packed(kbo_index_t *idx, const uint32_t *words, const uint64_t *offsets, size_t n_seqs,
                             const uint64_t *exc_pos, const uint8_t *exc_byte, size_t n_exc, double max_error_prob,
                             uint32_t *words_out);
/* kbo::find (lib.rs:808-821) over a packed batch.  The run lengths come back as the seven u32 the device writes (the fields of
 * format::RLE, format.rs:18-33; a sequence is shorter than 2^32 bases): 28 bytes per run instead of kbo_rle's 56, and no
 * widening pass on the host - at about 1.15 runs per 150-base read the records are as many bytes as the packed alignments
 * themselves.  *rles is library-allocated (kbo_free); rle_offsets as for kbo_find_batch. */
typedef struct {
    uint32_t start, end, matches, mismatches, jumps, gap_bases, gap_opens;
} kbo_rle32;
int kbo_find_batch_packed(kbo_index_t *idx, const uint32_t *words, const uint64_t *offsets, size_t n_seqs,
                          const uint64_t *exc_pos, const uint8_t *exc_byte, size_t n_exc, const kbo_find_opts *opts,
                          kbo_rle32 **rles, uint64_t *rle_offsets);

/* ------------------------------------------------------------------ device-resident path
 * Everything already in the HBM of the current device; kernels are enqueued on `stream`
 * (a hipStream_t) and the call returns immediately.  d_concat must be 16-byte aligned,
 * the other buffers 4-byte aligned.
 * SLACK: the kernels move bytes in 16-byte blocks relative to each sequence, so every per-base
 * buffer handed to these entry points - d_concat, d_ms_out / d_ms, d_ref, d_chars_out / d_chars -
 * must have at least 16 readable (for outputs: writable) bytes behind its last base, i.e. be
 * allocated with total_bases + 16 bytes or more (the in-repo callers round up to 16 and add 64).
 * The calls check what they can: they refuse total_bases + 16 > 2^32.  Sequences shorter than 3 are skipped by the fused
 * derandomize/translate kernel (the host entry points reject them like the reference).
 * d_work is device scratch of at least kbo_work_bytes(...) bytes for the batch (16-byte aligned).  Since round 5 that figure holds, for
 * batches with sequences of more than 160 bases (max_seq_len 0 or > 160), the regions of the kernels for long sequences as well (about
 * 0.5 B per base more): what kbo_map_batch_dev[_tail], kbo_find_batch_dev and kbo_map_stream_* check.  The walks alone - kbo_ms_batch_dev,
 * kbo_call_walk_dev - never touch those regions and check kbo_ms_work_bytes(...) only (the round-4 figure; <= kbo_work_bytes). */
size_t kbo_work_bytes(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k);
size_t kbo_ms_work_bytes(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k);
/* the same for a given index: a sharded index needs total_bases + 32 bytes more (one further shard's MS values) */
size_t kbo_index_work_bytes(const kbo_index_t *idx, size_t n_seqs, uint64_t total_bases, size_t max_seq_len);
/* A1 over a batch.  total_bases = offsets[n_seqs] (known to the caller; avoids a device read-back);
 * max_seq_len = length of the longest sequence if the caller knows it, 0 = unknown.  Batches of reads
 * get one work item per sequence; when max_seq_len is unknown or long, the item list is built on the
 * device from the offsets: sequences are cut into chunks that restart the walk k-1 bases upstream
 * (the MS of a base depends only on the k bases ending at it), so a few long sequences still fill the
 * device.  Same results either way - provided max_seq_len, when given, is not SMALLER than the longest sequence: the
 * kernels size their work items (16-bit lengths) and LDS stretches from it; pass 0 when in doubt.
 * Reads (max_seq_len <= 160) over a copy with a depth table, intervals not asked for: ONE kernel puts the values together (k where
 * nothing happened, the ramps behind the mismatches, the table's values behind them) and a second kernel walks the reads it leaves -
 * 305 Gbp/s at C2, 392 with resident batches in turn on two streams of the caller's (the plan-guided walk it replaces there: 242);
 * every other batch - chunks of long sequences, intervals (d_lo_out / d_hi_out: the colexicographic intervals, index.rs:243-256),
 * sharded indexes - takes the plan-guided walk. */
int kbo_ms_batch_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets,
                     size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint8_t *d_ms_out,
                     uint32_t *d_lo_out, uint32_t *d_hi_out, void *d_work, size_t work_bytes, void *stream);
/* A5+A6 fused (+ optional format::relative_to_ref when d_ref != NULL): u8 MS -> u8 chars.
 * max_seq_len = length of the longest sequence in the batch if the caller knows it (selects
 * the LDS-staged kernel for short reads), 0 = unknown.  d_work (optional, NULL = none): scratch of
 * kbo_derand_work_bytes() bytes, 16-byte aligned; with it long reads / contigs are processed in
 * pieces of 256 positions, one lane each, instead of one lane per sequence. */
size_t kbo_derand_work_bytes(size_t n_seqs, uint64_t total_bases);
int kbo_derand_translate_dev(const uint8_t *d_ms, const uint64_t *d_offsets, size_t n_seqs,
                             uint64_t total_bases, size_t k, size_t threshold, const uint8_t *d_ref,
                             uint8_t *d_chars_out, size_t max_seq_len, void *d_work, size_t work_bytes,
                             void *stream);
/* kbo::map with fill_gaps = false and call_variants = false (lib.rs:726-738; format != 0: + relative_to_ref, lib.rs:756-757) or
 * kbo::matches (lib.rs:612-628; format = 0) over a device-resident batch, the whole chain MS -> derandomize_ms_vec ->
 * translate_ms_vec enqueued on `stream`; the threshold comes from the index and max_error_prob (lib.rs:620, 731).  Batches of
 * reads (max_seq_len <= 160) over an index copy that carries a depth table run as ONE kernel (kbo_amd/csrc/map_kernels.hip): the
 * MS values never leave the chip unless want_ms != 0.  Batches with LONGER sequences - contigs, whole reference sequences, long reads:
 * what kbo::map / matches / find are called with (lib.rs:612-628, 720-761) - run as one kernel too when want_ms == 0 (one wave per
 * piece of a sequence, kbo_amd/csrc/long_kernels.hip: max_seq_len > 160 or 0 = unknown; any length below 4 GiB per launch).  Other
 * batches run as kbo_ms_batch_dev + kbo_derand_translate_dev (d_work carries the scratch of both).
 * d_ms: total_bases bytes + 16, 4-byte aligned: the MS value of every base when want_ms != 0 or the batch takes the two-kernel
 * route, otherwise scratch (unspecified contents).  d_work / work_bytes as for
 * kbo_ms_batch_dev; d_concat needs 16 readable bytes of slack behind the batch.  Sequences of fewer than 3 bases (the
 * reference asserts, derandomize.rs:276) have no alignment: their bytes of d_chars_out (and d_ms) are unspecified - left unwritten by the
 * two-kernel route, overwritten by the one kernel, whose stores are whole lines - and kbo_find_batch_dev reports no run for them.  *fused (optional) = 1 when the
 * batch took the one kernel. */
int kbo_map_batch_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                      size_t max_seq_len, double max_error_prob, int format, int want_ms, uint8_t *d_ms, uint8_t *d_chars_out,
                      void *d_work, size_t work_bytes, void *stream, int *fused);
/* The same with the second pass - one kernel that walks the few reads the first leaves (two in ten thousand at 1 % substitutions), a wave
 * a read: a chain of dependent look-ups that takes 0.07 ms however few they are - enqueued on `tail_stream`, ordered behind the kernel on `stream` by an
 * event: a caller with several batches in flight (own d_ms / d_chars_out / d_work each) keeps `stream` busy with the next batch's
 * kernel meanwhile.  The batch's outputs are complete when BOTH streams have reached this point; whatever touches this batch's
 * buffers next - on either stream - has to be ordered behind `tail_stream`.  tail_stream == stream: kbo_map_batch_dev.  The two-kernel
 * route (*fused = 0) runs on `stream` alone.  Two such pairs of streams that take the batches in turn, two batches in flight on
 * each, keep the device fuller still (INTEGRATION.md "Several batches in flight"; bench.py: 1 034 against 679 Gbp/s at C2). */
int kbo_map_batch_dev_tail(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                           size_t max_seq_len, double max_error_prob, int format, int want_ms, uint8_t *d_ms, uint8_t *d_chars_out,
                           void *d_work, size_t work_bytes, void *stream, void *tail_stream, int *fused);
/* A (stream, tail_stream) pair for kbo_map_batch_dev_tail / kbo_find_batch_dev made the way kbo_map_stream_* makes its own: `stream` -
 * the kernels' - kept off `tail_cus` compute units of the current device (-1: the library's default, 32; 0: two plain streams), the tail
 * stream a plain one: a second pass is a chain of dependent look-ups by a few hundred waves that otherwise waits for wave slots behind
 * kernels that hold them all; with units the kernels cannot take its workgroups start at once (C2, two such pairs: 757 -> 967 Gbp/s),
 * and a second pass that is work still has the whole device.  hipStream_t values; kbo_stream_pair_destroy when they are idle. */
int kbo_stream_pair_create(int tail_cus, void **stream, void **tail_stream);
void kbo_stream_pair_destroy(void *stream, void *tail_stream);
/* ---- several batches in flight, the library's own arrangement (what bench.py's headline is measured with): `pipelines` pairs of
 * (kernel stream, second-pass stream) that take the batches in turn, two slots - work and MS buffers, a completion event - per
 * pipeline, so that a batch's second pass runs beside the next batches' kernels.  max_* size the slots' buffers: a batch may not
 * exceed them.  The batch's own buffers (d_concat, d_offsets, d_chars_out) stay the caller's and must stay valid and untouched
 * until the batch is complete.
 *   kbo_map_stream_submit   enqueues kbo::map (format != 0) / kbo::matches of one device-resident batch and returns at once;
 *                           d_ms_out (optional, padded as d_chars_out): the matching statistics of every base too, one byte a base, as
 *                           kbo_ms_batch_dev gives them (index.rs:243-256: the raw k-bounded values, NOT derandomized);
 *                           ready_stream (optional): the stream whose work so far produces the batch's inputs - the pipeline waits
 *                           for it on the device; *ticket (optional) names the batch; *fused (optional) as kbo_map_batch_dev's
 *   kbo_map_stream_wait     blocks the calling thread until that batch is complete (submit never blocks, so more batches than the
 *                           pipelines have slots may be queued: a ticket whose slot a later batch has taken is waited for through the
 *                           next batch of its pipeline that still holds one - a pipeline's batches complete in order);
 *                           kbo_map_stream_wait_on makes `stream` wait for it on the device instead;  kbo_map_stream_sync: every
 *                           batch submitted so far */
typedef struct kbo_map_stream kbo_map_stream_t;
int kbo_map_stream_create(kbo_index_t *idx, int pipelines, size_t max_seqs, uint64_t max_bases, size_t max_seq_len, kbo_map_stream_t **out);
int kbo_map_stream_submit(kbo_map_stream_t *ms, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                          size_t max_seq_len, double max_error_prob, int format, uint8_t *d_ms_out, uint8_t *d_chars_out, void *ready_stream,
                          uint64_t *ticket, int *fused);
int kbo_map_stream_wait(kbo_map_stream_t *ms, uint64_t ticket);
int kbo_map_stream_wait_on(kbo_map_stream_t *ms, uint64_t ticket, void *stream);
int kbo_map_stream_sync(kbo_map_stream_t *ms);
void kbo_map_stream_free(kbo_map_stream_t *ms);
/* kbo::find (lib.rs:808-821) over a device-resident batch: kbo_map_batch_dev_tail with format = 0, then format::run_lengths_gapped
 * (format.rs:143-193) of the characters - kbo_run_lengths_dev's buffers and record layout (d_rle_work: kbo_run_lengths_work_bytes()) -
 * enqueued behind the second pass on `tail_stream` (pass `stream` for one stream).  With max_gap_len = 0 (FindOpts' default) the one
 * kernel counts the runs of every read it finishes while the characters are in LDS, and the run lengths are ONE pass over the
 * characters instead of two (count, then emit). */
int kbo_find_batch_dev(kbo_index_t *idx, const uint8_t *d_concat, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                       size_t max_seq_len, double max_error_prob, size_t max_gap_len, uint8_t *d_ms, uint8_t *d_chars_out, void *d_work,
                       size_t work_bytes, void *d_rle_work, uint32_t *d_records, size_t capacity, void *stream, void *tail_stream, int *fused);
/* kbo::matches (lib.rs:612-628) over a device-resident PACKED batch of reads (the layout of kbo_matches_batch_packed: kbo_packed_words()
 * words, every read starts a word; d_offsets counts bases), through the one kernel's packed-native form: the words go into the kernel
 * as they are and the characters leave it as words (M, -, X, R = 0 .. 3) - a quarter of a byte per base each way; only the reads it
 * leaves to the second pass are ever unpacked.  uniform_len = the length of every read if they are all equally long, else 0;
 * d_exc_pos / d_exc_byte / n_exc: the non-ACGT list on the device (positions in base coordinates, ascending) or NULL / 0;
 * d_scratch: kbo_matches_packed_dev_scratch_bytes() bytes, 16-byte aligned; d_work / work_bytes as for kbo_ms_batch_dev;
 * tail_stream as for kbo_map_batch_dev_tail (pass `stream` for one stream).  KBO_E_UNSUPPORTED when the batch or this copy of
 * the index cannot take that kernel (reads longer than 160 bases, no depth table, a threshold below the table's order):
 * kbo_matches_batch_packed takes any batch. */
size_t kbo_matches_packed_dev_scratch_bytes(size_t n_seqs, uint64_t total_bases);
int kbo_matches_packed_dev(kbo_index_t *idx, const uint32_t *d_words, const uint64_t *d_offsets, size_t n_seqs, uint64_t total_bases,
                           size_t max_seq_len, size_t uniform_len, const uint64_t *d_exc_pos, const uint8_t *d_exc_byte, size_t n_exc,
                           double max_error_prob, uint32_t *d_words_out, void *d_scratch, void *d_work, size_t work_bytes, void *stream,
                           void *tail_stream);
/* format::run_lengths_gapped over device-resident characters (the output of kbo_derand_translate_dev
 * without d_ref), enqueued on `stream`.  d_work: kbo_run_lengths_work_bytes(n_seqs) bytes; afterwards
 * word s of d_work plus word (n_seqs + 1 + s / 1024) is the index of sequence s's first run, and the
 * last word of d_work the total number of runs.  Records are seven u32 {start, end, matches,
 * mismatches, jumps, gap_bases, gap_opens}; runs beyond `capacity` are counted but not written.
 * max_seq_len = length of the longest sequence if known (reads take LDS-staged kernels), 0 = unknown. */
size_t kbo_run_lengths_work_bytes(size_t n_seqs);
int kbo_run_lengths_dev(const uint8_t *d_chars, const uint64_t *d_offsets, size_t n_seqs, size_t max_seq_len,
                        size_t max_gap_len, void *d_work, uint32_t *d_records, size_t capacity, void *stream);
/* Options of ONE index handle: what the process-wide setters below and in kbo_hip_tuning.h (kbo_set_devices, kbo_set_slab_bytes,
 * kbo_set_plan, kbo_set_depth_table, kbo_set_depth_table_anchors) decide for every index, decided for this one - two indexes of one
 * process (a small reference next to a large one; a service with one handle per tenant) no longer share them.  A field left at its
 * "inherit" value follows the process-wide setting at the time of use.  Set them before the handle's first batch / before
 * kbo_index_to_device: like the process-wide setters, depth_table* shape device copies made from then on (an existing copy keeps
 * its table; depth_table = -1 makes launches ignore it), plan = 0 stops planned launches at once, devices and slab_bytes apply to
 * the next host batch.  Not to be changed while a batch of this handle is in flight on another thread.  A sharded index hands its
 * options on to its shards. */
#define KBO_OPT_INHERIT (-2147483647 - 1)
#define KBO_OPT_MAX_DEVICES 16
typedef struct kbo_index_opts {
    uint32_t struct_size;        /* sizeof(kbo_index_opts_t), as filled in by kbo_index_opts_default: the struct may grow */
    int32_t plan;                /* KBO_OPT_INHERIT | 0 = plain walk only | 1 = path cover + tables on its device copies, planned launches */
    int32_t depth_table;         /* KBO_OPT_INHERIT | 0 = order by index size | -1 = none | 1 .. 17 = bases per entry */
    int32_t depth_table_anchors; /* KBO_OPT_INHERIT | -1 = by the table's margin over log4(rows) | 0 = no | 1 = yes */
    uint64_t slab_bytes;         /* 0 = inherit | query bytes per slab of its host batches (clamped to 64 KiB .. 3.75 GiB) */
    int32_t n_devices;           /* -1 = inherit | 0 = the current device | 1 .. 16 = devices[0 .. n) */
    int32_t devices[KBO_OPT_MAX_DEVICES];
} kbo_index_opts_t;
int kbo_index_opts_default(kbo_index_opts_t *opts); /* every field "inherit" */
int kbo_index_set_opts(kbo_index_t *idx, const kbo_index_opts_t *opts);
int kbo_index_get_opts(const kbo_index_t *idx, kbo_index_opts_t *opts);
/* Devices the host batch entry points (kbo_matches_batch / kbo_map_batch / kbo_find_batch) spread
 * their slabs over: index replicated per device, one submitting + one completing host thread and
 * three stage streams (upload, kernels, download) per device, disjoint output slices, no collective.
 * n = 0 restores the default (the current device). */
int kbo_set_devices(const int *devices, int n);
/* Host helper threads used by the host batch entry points for the staging copies between pageable
 * user buffers and pinned memory (two teams of this size; default min(8, cores)). */
int kbo_set_host_threads(int n);
/* host batches are processed in slabs of at most this many query bytes (default 16 MiB) */
int kbo_set_slab_bytes(size_t bytes);
/* Frees the per-device scratch (streams, device buffers, pinned staging) the host batch entry
 * points keep between calls, and the calling thread's own caches (kbo_call / kbo_call_batch keep a
 * device arena for their per-sequence indexes per host thread). */
int kbo_release_scratch(void);
/* The path cover the plan-guided walk uses (host computation, for inspection and tests): every row of the index sits at
 * exactly one text position; text[p] ('A','C','G','T') labels the edge node_at[p-1] -> node_at[p] of the index's de
 * Bruijn graph, 0 where a path starts.  All three arrays have n_sets entries. */
int kbo_index_path_cover(const kbo_index_t *idx, uint8_t *text, uint32_t *pos, uint32_t *node_at);
/* The recovery lines the guided walk reads large indexes through (host computation, for inspection and tests): line b
 * (128 bytes) covers rows [64 b, 64 b + 64): four 16-byte rank blocks { C[c] + rank_c(64 b), row bits 0..31, row bits
 * 32..63, 0 } for c = A, C, G, T, then the 64 LCS bytes of those rows (0 beyond the last row).  n_sets / 64 + 2 lines and
 * one all-zero line; *n_bytes receives the size, lines == NULL only asks for it. */
int kbo_index_recovery_lines(const kbo_index_t *idx, uint8_t *lines, size_t *n_bytes);
/* bytes of path cover + recovery lines + seed table + depth table (and its anchors) a device copy of this index carries (0 =
 * none).  The two tables are sized by log4(rows), not by the index: up to 2 GiB + 64 GiB (kbo_hip_tuning.h: kbo_set_depth_table,
 * INTEGRATION.md "Device memory: the depth table"). */
uint64_t kbo_index_device_plan_bytes(const kbo_index_t *idx);

/* Tuning knobs, experiment switches and test hooks (none of them changes a result) are declared in kbo_hip_tuning.h. */

#ifdef __cplusplus
}
#endif
#endif /* KBO_HIP_H */
