/*
 * kbo_hip_tuning.h — tuning knobs, experiment switches and test hooks of libkbo_hip.so.
 *
 * NOT part of the drop-in boundary (include/kbo_hip.h): nothing here has a counterpart in the reference crate, no
 * setting changes a result (every combination is parity-tested against the oracle), and a binding of the reference
 * never needs this file.  The in-repo tests, tools/ and bench.py use it to force code paths through their corners at
 * small sizes and to record the measurements behind DESIGN.md section 6.  All settings are process-wide and apply to
 * launches (or, where stated, to device copies of an index) made after the call.
 */
#ifndef KBO_HIP_TUNING_H
#define KBO_HIP_TUNING_H

#include "kbo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ launch geometry of the walk kernels */
/* upper bound on resident waves of the plain walk, and threads per workgroup */
int kbo_walk_geometry(int *max_waves, int *threads);
int kbo_set_walk_waves_per_cu(int waves_per_cu); /* 0 = default (32) */
int kbo_set_walk_threads(int threads);           /* workgroup size 64 / 128 / 256 */
int kbo_set_walk_rare(int period);               /* hot-loop iterations between item-bookkeeping visits (default 8) */
/* The guided walk (the plan-guided form of A1).  waves_per_cu: resident waves per CU, 0 = default (8 or 12, scaled with
 * kbo_set_walk_waves_per_cu; few, so that the lines of the lanes in flight stay in L2: DESIGN.md 4.2), < 0 keeps.
 * recovery_lines: 1 / 0 = the walk reads the index through the recovery lines (one 128-byte line per 64 rows with rank
 * blocks and LCS values) / through the rank blocks and contraction entries, -1 = by index size (default), < -1 keeps. */
int kbo_set_guided_walk(int waves_per_cu, int recovery_lines);

/* ------------------------------------------------------------------ index layout on the device */
/* Two-base extension blocks (2.7 B per index row on the device): built for indexes with at least min_rows rows
 * (default 24 Mi; 0 = always, UINT64_MAX = never; applies to device copies made after the call), used by the walk from
 * matches at least min_depth deep (default 16; < 0 keeps it). */
int kbo_set_pair_steps(uint64_t min_rows, int min_depth);
/* tests: the rank blocks, contraction entries and two-base blocks of the index's copy on `device` (-1: the current one) - made on the
 * device from the row bit-vectors and the LCS bytes (layout_kernels.hip) - against the host's single-threaded construction of the same
 * layout: *n_diff = the bytes that differ.  KBO_DEVICE_LAYOUT=0 in the environment makes copies from the host's instead. */
int kbo_index_layout_check(kbo_index_t *idx, int device, uint64_t *n_diff);
/* tests: the handle's path cover - laid out on the device by the first copy that needed it (cover_kernels.hip) - against the host's
 * construction (path_cover.cpp): *n_diff = entries of text / pos / node_at that differ.  KBO_DEVICE_COVER=0: the host's from the start. */
int kbo_index_cover_check(kbo_index_t *idx, uint64_t *n_diff);
/* Host batches (kbo_matches_batch / kbo_map_batch / kbo_find_batch and the packed forms): caller's buffers that are pinned
 * already (hipHostMalloc / hipHostRegister) are used in place - no staging copies, no host threads busy - instead of being
 * staged through the slots' own pinned buffers like pageable ones.  Default 0: staged is the faster of the two on the MI355X
 * boxes this was measured on (600 Mbp: 34.9 against 27.7 Gbp/s, packed 123 against 104).  Environment: KBO_HOST_INPLACE. */
int kbo_set_host_in_place(int on);
/* tests: force the 64-bit-offset contraction-entry layout (device copies made after the call) */
int kbo_set_force_big_layout(int on);
/* tests: kbo_index_build makes at least this many shards whatever the size of its input (0 = by size: one index below
 * 3.76 * 10^9 rows); see kbo_hip.h "Sharded indexes" */
int kbo_set_index_shards(int shards);
/* tests: shard i of a sharded index - an ordinary index over its part of the input (n_kmers = its own), BORROWED: it lives as
 * long as `idx` and must not be freed.  An ordinary index is its own shard 0.  NULL when i is out of range. */
const kbo_index_t *kbo_index_shard(const kbo_index_t *idx, int i);
/* The depth table of device copies made after the call (kbo_amd/csrc/dtab_kernels.hip): for every string of `order` bases the
 * longest suffix of it that is a suffix of a row of the index, 4^order bytes of device memory.  With it, the stretches behind
 * a read's mismatches against the plan's diagonal - where the matching statistic is the length of a random match, about
 * log4(rows) - cost one independent byte look-up per base instead of a chain of dependent rank look-ups; values deeper than
 * `order` send their read to the plain walk.  0 = by index size (log4(rows) + 3.2 rounded up, at most 17 and k; none from
 * about 1.2 * 10^9 rows on - 17 bases are then too few - or when it would take more than half of the free device memory),
 * 1 .. 17 = that order (capped at k),
 * < 0 = none: new copies get no table and launches over copies that have one do not use it (until the knob is >= 0 again).
 * Results are identical with and without it. */
int kbo_set_depth_table(int order);
/* ... with ANCHORS (device copies made after the call): a hash of the strings of `order` bases that are the suffix of exactly
 * one row, with that row's place in the path cover; a base deeper than the table knows is then read off the path-cover text
 * (by a kernel of its own behind the plan kernel) instead of sending its read to the plain walk; a quarter of those reads
 * remain.  Measured: slower at C2 (the redo pass is bound by its longest chain, not by its size), 4 % faster at C3.
 * -1 = on indexes of 24 Mi rows and more whose table's margin over log4(rows) is below 3.75 bases (default: there the reads left to the plain walk are many
 * enough for it to pay - C4 +12 %, C3 +4 % - and from 3 * 10^8 rows on the table only wins with them), 0 = never, 1 = always. */
int kbo_set_depth_table_anchors(int mode);
/* bytes the seed + depth tables of a device copy may take while they are built (device copies made after the call): 0 = half of the
 * device memory that is free at that moment (default).  A table that does not fit is degraded - grouped layout -> plain layout
 * (a quarter of the bytes) -> fewer bases -> none (the copy then gets the host-built seed table and the guided walk) - never an
 * error, also when its order was forced with kbo_set_depth_table. */
int kbo_set_plan_table_budget(uint64_t bytes);
/* When a device copy makes its plan structures (path cover, recovery lines, seed and depth tables, 2-bit text): kbo_index_to_device
 * makes them at once; a copy made implicitly by a first query makes them once the batches that went through it add up to `bases`
 * bases - until then MS-only batches take the plain walk, with the same results.  -1 = by index size (default): at once when they
 * cost under a tenth of a second (up to ~ 1.5 * 10^6 rows), else after the bases whose saving pays for them (65 ns per row without
 * a stored path cover, 15 ns with one, against 7 ps saved per base: ~ 50 Gbp for a 5 Mbp index); 0 = always at once. */
int kbo_set_plan_lazy(int64_t bases);
/* inspection / tests: the depth table of the copy of `idx` on `device` (-1 = current; the copy is made if there is none):
 * *order bases per entry (0: the copy has no table), 4^*order bytes; entry of a string (2-bit digits A, C, G, T = 0 .. 3, the
 * last base least significant) = length of its longest suffix that is a suffix of a row of the index, or 0x80 | e when the
 * whole string is, bit c of e set when the string with base c in front of it is one too.  On the device the entries of
 * three consecutive bases of a read share a 64-byte line, so every entry is there three times; `view` (0 .. 2) says which
 * copy to hand back, in key order (ignored for tables too small to be grouped).  table == NULL only asks for *n_bytes and
 * *order; tables of more than 14 bases are not handed back (KBO_E_UNSUPPORTED). */
int kbo_index_depth_table(kbo_index_t *idx, int device, int view, uint8_t *table, size_t *n_bytes, int *order);
/* tests: depth of the seed table of device copies made after the call (0 = by index size: 8 / 10 / 12 / 13 bases;
 * 1 .. 13 = that many, capped at k).  Large tables are what large indexes get: 12 bases = 128 MiB, 13 = 512 MiB. */
int kbo_set_seed_table_depth(int bases);

/* ------------------------------------------------------------------ the plan-guided walk */
/* Device copies made while it is enabled (default) carry a path cover of the index's de Bruijn graph (9 B per row) and
 * kbo_ms_batch[_dev] / kbo_matches_batch / kbo_map_batch / kbo_find_batch skip the stretches of every read that match it;
 * results are identical either way.  enabled < 0 keeps the setting, > 0 also clears every copy's hold-off; seed_depth
 * (> 0: fixed; < 0: automatic = log4(rows) + 3, the default; 0 keeps) and seed_cap (default 64 = the most; <= 0 keeps)
 * tune the diagonal search. */
int kbo_set_plan(int enabled, int seed_depth, int seed_cap);
/* (0 keeps a value) mismatches closer than `gap` bases (>= 2; < 0: automatic = log4(rows) + 9, the default: 20 on a
 * 5 Mbp index, 22 on 100 Mbp) are walked by one unit; reads without a diagonal are walked in chunks of `chunk` bases
 * (default 32); a launch with more than bail_x16 / 16 units per read (default 50 / 16: the break-even is near 4 %
 * substitutions) gives the plan up and takes the plain walk, and the following 16 launches over that device copy of that
 * index do not plan at all.  bail_x16 = 0 forces that path (tests), < 0 keeps. */
int kbo_set_plan_tuning(int gap, int chunk, int bail_x16);
/* tests: the unit array of a launch gets 1 / divisor of its normal capacity (reads whose units do not fit take the plain
 * walk); 1 = normal */
int kbo_set_plan_unit_cap_divisor(int divisor);
/* inspection / tests: launches over the copy of `idx` on `device` (-1 = current) that gave the plan up so far, and the
 * launches the copy will still skip planning for; either pointer may be NULL.  KBO_E_BAD_ARG when no such copy exists. */
int kbo_index_plan_holdoff(kbo_index_t *idx, int device, uint32_t *bails, int *holdoff);

/* instrumentation (default off): launches of the plan-guided stage made while it is on count their own work - per-lane adds
 * in the walk's hot loop, one atomic per counter and wave at exit: about 1 % of the stage's time, which is why the default
 * instantiations of the kernels carry none of it */
int kbo_set_plan_stats(int on);
/* inspection: the work counters the plan-guided stage kept about its last launch over a device-resident batch (made with
 * kbo_set_plan_stats(1)) - what
 * the CPU model of the stage (oracle/plan_model.c), whose counts bench.py's roofline prices the stage by, is pinned to
 * (tests/test_gpu_model.py).  Arguments as for the kbo_ms_batch_dev call that ran on d_work; synchronises `stream`.
 * out[0..7]: units walked, accepted extensions, failed extensions, contraction levels, levels taken from the entries
 * (recovery-line form), seed-table look-ups, seed extensions, mismatches against the diagonals;  out[8]: units emitted,
 * [9]: entries of the redo list, [10]: 1 = the plan was given up, [11]: 1 = a walk left through its guard;  depth-table form
 * (kbo_set_depth_table) out[12]: table look-ups, [13]: values written from the table, [14] = [15]: items the table could not
 * resolve (counted by the lanes / by the launch's control word; out[8] is meaningless in that form), [16]: bases read off the
 * path-cover text (anchors), [17]: items resolved without a plan.  Meaningless
 * (stale or zero) when that launch did not plan (plain walk, hold-off, intervals requested) or did not count. */
#define KBO_PLAN_STATS 20
int kbo_plan_stats_dev(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k, const void *d_work,
                       uint64_t out[KBO_PLAN_STATS], void *stream);

/* inspection: which reads the last planned launch over a device-resident batch of READS (kbo_ms_batch_dev, kbo_map_batch_dev: one
 * item per sequence) left to the plain walk - flags_out[s] != 0 - read out of the launch's work buffer.  Arguments as for that call;
 * synchronises `stream`. */
int kbo_plan_flags_dev(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k, const void *d_work, uint8_t *flags_out, void *stream);
/* the kernel for sequences of more than 160 bases (long_kernels.hip: one wave per piece of a sequence): 0 = never (such batches
 * take the walk + the derandomize / translate kernels), 1 = wherever the index copy has what it needs (default), 2 = as 1 with
 * EVERY piece sent to its second pass (plain walk + literal recurrences) - a test hook, exact like the others. */
int kbo_set_map_long(int mode);
/* kbo_ms_batch_dev over a batch of reads (<= 160 bases, no intervals) and a copy with a depth table: 1 (default) = map_reads_kernel in its
 * MS-emitting form, stopped behind the values; 0 = the plan-guided walk as for every other batch (also taken while kbo_set_plan_stats is
 * on: the work counters are that walk's). */
int kbo_set_ms_one_kernel(int on);
/* kbo_call_batch[_flat]: 1 (default) = the variants are put in order, resolved and sliced on the device (call_emit_kernels.hip: two slots
 * on two streams, ten bytes per variant over PCIe); 0 = rounds 3 - 5's route for every slab (a record + a window per site to the host,
 * host threads sort, resolve and slice) - what the default falls back to for a slab it cannot finish; 2 = as 1, but the second pass's
 * depths for k <= 64 by the kernel that serves k > 64 (call_second_kernels.hip: every lane extends its own matches); same results (tests). */
int kbo_set_call_device_emit(int on);
/* inspection: what the last kbo_map_batch_dev / kbo_find_batch_dev call over sequences of more than 160 bases did when it took
 * the one kernel for sequences of any length (long_kernels.hip); arguments as for that call, synchronises `stream`.
 * out[0]: pieces, [1]: pieces whose proof failed (plain walk + literal recurrences), [2]: their sub-items; with kbo_set_plan_stats(1) [3]: seed look-ups,
 * [4]: filter look-ups, [5]: depth-table look-ups, [6]: of those, second look-ups behind a window that was present;
 * [8..12] (experiments, KBO_LONG_X & 128): shader cycles of the waves' first lanes by phase - staging, stretches, planes to
 * characters, proof, output; [13], [14] (kbo_set_plan_stats): pieces that tried all their words on a band of diagonals at once /
 * that kept the result; [16..24] (kbo_set_plan_stats): flagged pieces by what flagged them - other, list too long, a window one base
 * on present and extended, the window one base back, the one inside the next stretch; [21..24]: the last four after the band pass. */
#define KBO_LONG_STATS 25
int kbo_long_stats_dev(size_t n_seqs, uint64_t total_bases, size_t max_seq_len, uint32_t k, const void *d_work, uint64_t out[KBO_LONG_STATS],
                       void *stream);

/* instrumentation (default off): while on, every kbo_map_batch_dev call that takes the one-kernel route records HIP events on its
 * stream around map_reads_kernel and behind the redo pass (three event records per call).  kbo_stage_timing_read waits for the
 * recorded calls, returns their number and the sums of the two intervals in milliseconds - the kernel itself / the reads it left
 * to the plain walk (list, walk, their derandomize + translate) - and forgets them.  bench.py prices its roofline by the first.
 * on > 1: on, and the events of that many calls are made now instead of inside the first calls that record them. */
int kbo_set_stage_timing(int on);
int kbo_stage_timing_read(double *kernel_ms_sum, double *redo_ms_sum, int *n_calls);

/* test hook: what kbo_call_batch answers the reference-side walks of its sites with (variant_calling.rs:280: the matched
 * row's k-mer against the index the reference builds of the sequence itself, lib.rs:553) - the depths of the walk of each
 * of n_kmers k-mers (k bytes each, back to back) against the SBWT of `seq` built with (k, add_revcomp), computed from a
 * suffix automaton of the sequence's ACGT-runs of at least k characters instead of that index.  Host only (no GPU).
 * depths_out: n_kmers * k values.  tests/test_capi_host.py compares them with the oracle's walk of a real one-sequence index. */
int kbo_run_automaton_depths(const uint8_t *seq, size_t len, uint32_t k, int add_revcomp, const uint8_t *kmers,
                             size_t n_kmers, uint32_t *depths_out);

/* ------------------------------------------------------------------ experiments recorded in DESIGN.md section 6 */
/* plain walk kernel: only the first lane_limit lanes of every wave take reads (64 = all; what a sub-wave tiling would
 * have to beat), and every workgroup reserves dummy_lds_bytes of LDS it never touches (what staging a wave's MS values
 * in LDS for a fused A5/A6 would cost in occupancy) */
int kbo_set_walk_experiment(int lane_limit, int dummy_lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* KBO_HIP_TUNING_H */
