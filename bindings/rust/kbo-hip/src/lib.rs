//! Bindings of the reference crate's matching-statistics path to `libkbo_hip.so` (MI355X, gfx950).
//!
//! SOURCE ONLY: no Rust toolchain exists in the image this repository is built in, so this file has never been
//! compiled; the C ABI it declares (`include/kbo_hip.h`) is what the test suite drives through Python/ctypes.
//! Function names and signatures follow the reference (`kbo` 0.5.1): `find`, `matches`, `map`, `call`,
//! `index::query_sbwt`; each wrapper cites the reference lines it replaces.  The reference panics on precondition
//! failures (`assert!` at index.rs:248, derandomize.rs:274-276, translate.rs:268-270, lib.rs:559-560, 729), so the
//! wrappers turn the library's error codes back into panics.
#![allow(unsafe_code)] // kbo has #![warn(unsafe_code)] (lib.rs:242); an FFI module has to opt out

use std::ops::Range;
use std::os::raw::{c_char, c_int, c_void};

pub mod ffi {
    use super::*;
    #[repr(C)] pub struct KboIndex { _private: [u8; 0] }
    #[repr(C)] pub struct KboMapStream { _private: [u8; 0] }
    #[repr(C)] #[derive(Clone, Copy, Debug, PartialEq)]
    pub struct KboRle { pub start: u64, pub end: u64, pub matches: u64, pub mismatches: u64,
                        pub jumps: u64, pub gap_bases: u64, pub gap_opens: u64 }
    #[repr(C)] #[derive(Clone, Copy, Debug, PartialEq)]
    pub struct KboRle32 { pub start: u32, pub end: u32, pub matches: u32, pub mismatches: u32,
                          pub jumps: u32, pub gap_bases: u32, pub gap_opens: u32 }
    #[repr(C)] pub struct KboBuildOpts { pub k: u32, pub add_revcomp: i32, pub num_threads: u32, pub prefix_precalc: u32,
                                         pub build_select: i32, pub mem_gb: u32, pub dedup_batches: i32, pub temp_dir: *const c_char }
    #[repr(C)] pub struct KboFindOpts { pub max_error_prob: f64, pub max_gap_len: usize }
    #[repr(C)] pub struct KboMapOpts { pub max_error_prob: f64, pub fill_gaps: i32, pub call_variants: i32, pub format: i32,
                                       pub sbwt_build_opts: KboBuildOpts }
    #[repr(C)] pub struct KboCallOpts { pub max_error_prob: f64, pub sbwt_build_opts: KboBuildOpts }
    #[repr(C)] pub struct KboVariant { pub query_pos: u64, pub query_chars: *const u8, pub query_len: usize,
                                       pub ref_chars: *const u8, pub ref_len: usize }
    #[repr(C)] pub struct KboCallFlat { pub n_variants: u64, pub n_chars: u64, pub query_pos: *mut u32, pub query_len: *mut u16,
                                        pub ref_len: *mut u16, pub chars: *mut u8 }
    pub const KBO_OPT_INHERIT: i32 = i32::MIN;
    #[repr(C)] pub struct KboIndexOpts { pub struct_size: u32, pub plan: i32, pub depth_table: i32, pub depth_table_anchors: i32,
                                         pub slab_bytes: u64, pub n_devices: i32, pub devices: [i32; 16] }
    extern "C" {
        pub fn kbo_last_error() -> *const c_char;
        pub fn kbo_free(p: *mut c_void);
        // kbo::build (lib.rs:501-506) / an index the sbwt crate built, handed over by its parts
        pub fn kbo_index_build(seqs: *const *const u8, lens: *const usize, n_seqs: usize, opts: *const KboBuildOpts,
                               out: *mut *mut KboIndex) -> c_int;
        pub fn kbo_index_from_parts(k: u32, n_sets: u64, n_kmers: u64, rows: *const *const u64, c: *const u64, lcs: *const u8,
                                    out: *mut *mut KboIndex) -> c_int;
        pub fn kbo_index_free(idx: *mut KboIndex);
        pub fn kbo_index_shards(idx: *const KboIndex) -> c_int;
        pub fn kbo_index_save(idx: *const KboIndex, path: *const c_char) -> c_int;   // .kbohip: index + path cover
        pub fn kbo_index_load(path: *const c_char, out: *mut *mut KboIndex) -> c_int;
        // index::query_sbwt (index.rs:243-256), kbo::matches / map / find / call (lib.rs:612-628, 720-761, 808-821, 547-573)
        pub fn kbo_matching_statistics(idx: *mut KboIndex, query: *const u8, len: usize, d: *mut u64, lo: *mut u64, hi: *mut u64) -> c_int;
        pub fn kbo_matches(idx: *mut KboIndex, query: *const u8, len: usize, p: f64, chars_out: *mut u32) -> c_int;
        pub fn kbo_map(idx: *mut KboIndex, ref_seq: *const u8, len: usize, opts: *const KboMapOpts, out: *mut u8) -> c_int;
        pub fn kbo_find(idx: *mut KboIndex, query: *const u8, len: usize, opts: *const KboFindOpts, out: *mut *mut KboRle,
                        n_out: *mut usize) -> c_int;
        pub fn kbo_call(idx: *mut KboIndex, ref_seq: *const u8, len: usize, opts: *const KboCallOpts, out: *mut *mut KboVariant,
                        n_out: *mut usize) -> c_int;
        // batches: what kbo-cli's loop over the reads / contigs of a file calls once
        pub fn kbo_matches_batch(idx: *mut KboIndex, concat: *const u8, offsets: *const u64, n_seqs: usize, p: f64, chars_out: *mut u8) -> c_int;
        pub fn kbo_map_batch(idx: *mut KboIndex, concat: *const u8, offsets: *const u64, n_seqs: usize, p: f64, format: c_int, out: *mut u8) -> c_int;
        pub fn kbo_find_batch(idx: *mut KboIndex, concat: *const u8, offsets: *const u64, n_seqs: usize, opts: *const KboFindOpts,
                              rles: *mut *mut KboRle, rle_offsets: *mut u64) -> c_int;
        pub fn kbo_call_batch(idx: *mut KboIndex, concat: *const u8, offsets: *const u64, n_seqs: usize, opts: *const KboCallOpts,
                              out: *mut *mut KboVariant, var_offsets: *mut u64) -> c_int;
        pub fn kbo_call_batch_flat(idx: *mut KboIndex, concat: *const u8, offsets: *const u64, n_seqs: usize, opts: *const KboCallOpts,
                                   result: *mut KboCallFlat, var_offsets: *mut u64) -> c_int;
        pub fn kbo_call_flat_free(result: *mut KboCallFlat);
        pub fn kbo_stream_pair_create(tail_cus: c_int, stream: *mut *mut c_void, tail_stream: *mut *mut c_void) -> c_int;
        pub fn kbo_stream_pair_destroy(stream: *mut c_void, tail_stream: *mut c_void);
        // 2-bit packed batches: a quarter of the bytes over PCIe (kbo_hip.h "packed batches")
        pub fn kbo_packed_words(offsets: *const u64, n_seqs: usize) -> usize;
        pub fn kbo_pack_reads(concat: *const u8, offsets: *const u64, n_seqs: usize, words_out: *mut u32, exc_pos: *mut u64,
                              exc_byte: *mut u8, exc_cap: usize, n_exc: *mut usize) -> c_int;
        pub fn kbo_unpack_matches(words: *const u32, offsets: *const u64, n_seqs: usize, chars_out: *mut u8) -> c_int;
        pub fn kbo_matches_batch_packed(idx: *mut KboIndex, words: *const u32, offsets: *const u64, n_seqs: usize, exc_pos: *const u64,
                                        exc_byte: *const u8, n_exc: usize, p: f64, words_out: *mut u32) -> c_int;
        pub fn kbo_find_batch_packed(idx: *mut KboIndex, words: *const u32, offsets: *const u64, n_seqs: usize, exc_pos: *const u64,
                                     exc_byte: *const u8, n_exc: usize, opts: *const KboFindOpts, rles: *mut *mut KboRle32,
                                     rle_offsets: *mut u64) -> c_int;
        // device-resident batches on a hipStream_t (kbo_hip.h): kbo::map / matches for reads as ONE kernel; `_tail`: its second
        // pass on a second stream, beside the next batch's kernel; packed: 2-bit words in and out
        pub fn kbo_work_bytes(n_seqs: usize, total_bases: u64, max_seq_len: usize, k: u32) -> usize;
        pub fn kbo_ms_work_bytes(n_seqs: usize, total_bases: u64, max_seq_len: usize, k: u32) -> usize;
        pub fn kbo_index_to_device(idx: *mut KboIndex, device: c_int) -> c_int;
        pub fn kbo_map_batch_dev(idx: *mut KboIndex, d_concat: *const u8, d_offsets: *const u64, n_seqs: usize, total_bases: u64,
                                 max_seq_len: usize, p: f64, format: c_int, want_ms: c_int, d_ms: *mut u8, d_chars_out: *mut u8,
                                 d_work: *mut c_void, work_bytes: usize, stream: *mut c_void, fused: *mut c_int) -> c_int;
        pub fn kbo_map_batch_dev_tail(idx: *mut KboIndex, d_concat: *const u8, d_offsets: *const u64, n_seqs: usize, total_bases: u64,
                                      max_seq_len: usize, p: f64, format: c_int, want_ms: c_int, d_ms: *mut u8, d_chars_out: *mut u8,
                                      d_work: *mut c_void, work_bytes: usize, stream: *mut c_void, tail_stream: *mut c_void,
                                      fused: *mut c_int) -> c_int;
        // several batches in flight through the library's own pipelines (kbo_hip.h kbo_map_stream_*): tickets instead of streams
        pub fn kbo_map_stream_create(idx: *mut KboIndex, pipelines: c_int, max_seqs: usize, max_bases: u64, max_seq_len: usize,
                                     out: *mut *mut KboMapStream) -> c_int;
        pub fn kbo_map_stream_submit(ms: *mut KboMapStream, d_concat: *const u8, d_offsets: *const u64, n_seqs: usize, total_bases: u64,
                                     max_seq_len: usize, p: f64, format: c_int, d_ms_out: *mut u8, d_chars_out: *mut u8,
                                     ready_stream: *mut c_void, ticket: *mut u64, fused: *mut c_int) -> c_int;
        pub fn kbo_map_stream_wait(ms: *mut KboMapStream, ticket: u64) -> c_int;
        pub fn kbo_map_stream_wait_on(ms: *mut KboMapStream, ticket: u64, stream: *mut c_void) -> c_int;
        pub fn kbo_map_stream_sync(ms: *mut KboMapStream) -> c_int;
        pub fn kbo_map_stream_free(ms: *mut KboMapStream);
        // kbo::find for a device-resident batch: the characters, then format::run_lengths_gapped of them (records of seven u32)
        pub fn kbo_run_lengths_work_bytes(n_seqs: usize) -> usize;
        pub fn kbo_find_batch_dev(idx: *mut KboIndex, d_concat: *const u8, d_offsets: *const u64, n_seqs: usize, total_bases: u64,
                                  max_seq_len: usize, p: f64, max_gap_len: usize, d_ms: *mut u8, d_chars_out: *mut u8, d_work: *mut c_void,
                                  work_bytes: usize, d_rle_work: *mut c_void, d_records: *mut u32, capacity: usize, stream: *mut c_void,
                                  tail_stream: *mut c_void, fused: *mut c_int) -> c_int;
        pub fn kbo_matches_packed_dev_scratch_bytes(n_seqs: usize, total_bases: u64) -> usize;
        pub fn kbo_matches_packed_dev(idx: *mut KboIndex, d_words: *const u32, d_offsets: *const u64, n_seqs: usize, total_bases: u64,
                                      max_seq_len: usize, uniform_len: usize, d_exc_pos: *const u64, d_exc_byte: *const u8, n_exc: usize,
                                      p: f64, d_words_out: *mut u32, d_scratch: *mut c_void, d_work: *mut c_void, work_bytes: usize,
                                      stream: *mut c_void, tail_stream: *mut c_void) -> c_int;
        // options of one handle (what kbo_set_plan / kbo_set_depth_table* / kbo_set_devices / kbo_set_slab_bytes set process-wide)
        pub fn kbo_index_opts_default(opts: *mut KboIndexOpts) -> c_int;
        pub fn kbo_index_set_opts(idx: *mut KboIndex, opts: *const KboIndexOpts) -> c_int;
        pub fn kbo_index_get_opts(idx: *const KboIndex, opts: *mut KboIndexOpts) -> c_int;
    }
}

fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(ffi::kbo_last_error()) }.to_string_lossy().into_owned();
        panic!("kbo_hip error {}: {}", rc, msg);
    }
}

/// Stands in for `(&SbwtIndexVariant, &LcsArray)`: the index resident in HBM.  Immutable after construction.
pub struct GpuIndex(*mut ffi::KboIndex);
unsafe impl Send for GpuIndex {}
unsafe impl Sync for GpuIndex {}
impl Drop for GpuIndex { fn drop(&mut self) { unsafe { ffi::kbo_index_free(self.0) } } }

/// The abstract content of an `sbwt` index in the form `kbo_index_from_parts` takes: the four SubsetMatrix rows as
/// u64 words (bit `i & 63` of word `i >> 6` = row `i`), the C array, one LCS byte per row.
pub struct SbwtParts { pub k: usize, pub n_sets: usize, pub n_kmers: usize, pub rows: [Vec<u64>; 4], pub c: [u64; 4], pub lcs: Vec<u8> }

/// Reads an index built by the `sbwt` crate out through the calls the reference itself makes on it - `k()`,
/// `n_sets()`, `n_kmers()` (lib.rs:620), `access_kmer(colex)` (variant_calling.rs:276; needs `build_select = true`) -
/// and nothing else: the crate's source is not in the reference tree, so its bit-vector accessors could not be looked
/// up, let alone tested, where this was written.  O(n k log n) time and n k bytes: fine for bacterial genomes, too slow
/// for a human one - there a maintainer replaces the body by reads of the SubsetMatrix rows and the LCS int-vector
/// (one line each, once the accessor names are at hand).  Whatever fills it, `kbo_index_from_parts` validates the parts
/// (C against the edge bits, edge bits == n_sets - 1, LCS < k) and refuses anything inconsistent with KBO_E_BAD_ARG.
///
/// Definitions (verified against the reference's goldens): rows are in colex order;
/// `B_c[i] = 1` iff row i is the first row of its (k-1)-suffix group and `row[1..] + c` is a row;
/// `C[c] = 1 + sum_{c' < c} popcount(B_c')`; `LCS[i]` = longest common suffix of rows i-1 and i, `$` never matching.
pub fn export_sbwt_parts(sbwt: &sbwt::SbwtIndex<sbwt::SubsetMatrix>) -> SbwtParts {
    let (k, n) = (sbwt.k(), sbwt.n_sets());
    let kmers: Vec<Vec<u8>> = (0..n).map(|i| sbwt.access_kmer(i)).collect(); // '$'-padded on the left
    let colex = |a: &[u8], b: &[u8]| a.iter().rev().cmp(b.iter().rev()); // ('$' = 36 sorts below 'A')
    let words = (n + 63) / 64;
    let mut rows: [Vec<u64>; 4] = [vec![0; words], vec![0; words], vec![0; words], vec![0; words]];
    let mut lcs = vec![0u8; n];
    for i in 0..n {
        if i > 0 {
            let common = kmers[i - 1].iter().rev().zip(kmers[i].iter().rev()).take_while(|(a, b)| a == b && **a != b'$').count();
            lcs[i] = common as u8;
        }
        let first_of_group = i == 0 || kmers[i - 1][1..] != kmers[i][1..];
        if !first_of_group { continue; }
        for (ci, c) in b"ACGT".iter().enumerate() {
            let mut target = kmers[i][1..].to_vec();
            target.push(*c);
            if kmers.binary_search_by(|row| colex(row, &target)).is_ok() { rows[ci][i >> 6] |= 1u64 << (i & 63); }
        }
    }
    let mut c = [1u64; 4];
    for ci in 1..4 { c[ci] = c[ci - 1] + rows[ci - 1].iter().map(|w| w.count_ones() as u64).sum::<u64>(); }
    SbwtParts { k, n_sets: n, n_kmers: sbwt.n_kmers(), rows, c, lcs }
}

impl GpuIndex {
    /// `kbo::build` (lib.rs:501-506) with this library's own builder.
    pub fn build(seqs: &[Vec<u8>], opts: &kbo::BuildOpts) -> Self {
        let ptrs: Vec<*const u8> = seqs.iter().map(|s| s.as_ptr()).collect();
        let lens: Vec<usize> = seqs.iter().map(|s| s.len()).collect();
        let o = ffi::KboBuildOpts { k: opts.k as u32, add_revcomp: opts.add_revcomp as i32, num_threads: opts.num_threads as u32,
                                    prefix_precalc: opts.prefix_precalc as u32, build_select: opts.build_select as i32,
                                    mem_gb: opts.mem_gb as u32, dedup_batches: opts.dedup_batches as i32, temp_dir: std::ptr::null() };
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::kbo_index_build(ptrs.as_ptr(), lens.as_ptr(), seqs.len(), &o, &mut h) });
        GpuIndex(h)
    }
    /// 1 for an ordinary index; more when the input had 3.76e9 rows or more (a human genome with `add_revcomp`) and was built as
    /// shards: `matches` / `find` / `map` without refinement give the one index's results, `call` / gap filling / save are refused.
    pub fn shards(&self) -> usize { unsafe { ffi::kbo_index_shards(self.0) as usize } }
    /// An index the sbwt crate already built (what `kbo::build` returns, what kbo-cli loads from `.sbwt` / `.lcs`).
    pub fn from_sbwt(index: &sbwt::SbwtIndexVariant) -> Self {
        let sbwt::SbwtIndexVariant::SubsetMatrix(ref sbwt) = index;
        let p = export_sbwt_parts(sbwt);
        let ptrs: Vec<*const u64> = p.rows.iter().map(|r| r.as_ptr()).collect();
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::kbo_index_from_parts(p.k as u32, p.n_sets as u64, p.n_kmers as u64, ptrs.as_ptr(), p.c.as_ptr(),
                                                 p.lcs.as_ptr(), &mut h) });
        GpuIndex(h)
    }
}

/// `kbo::index::query_sbwt` (index.rs:243-256)
pub fn query_sbwt(query: &[u8], idx: &GpuIndex) -> Vec<(usize, Range<usize>)> {
    let n = query.len();
    let (mut d, mut lo, mut hi) = (vec![0u64; n], vec![0u64; n], vec![0u64; n]);
    check(unsafe { ffi::kbo_matching_statistics(idx.0, query.as_ptr(), n, d.as_mut_ptr(), lo.as_mut_ptr(), hi.as_mut_ptr()) });
    (0..n).map(|i| (d[i] as usize, lo[i] as usize..hi[i] as usize)).collect()
}

/// `kbo::matches` (lib.rs:612-628)
pub fn matches(query_seq: &[u8], idx: &GpuIndex, opts: kbo::MatchOpts) -> Vec<char> {
    let mut out = vec![0u32; query_seq.len()];
    check(unsafe { ffi::kbo_matches(idx.0, query_seq.as_ptr(), query_seq.len(), opts.max_error_prob, out.as_mut_ptr()) });
    out.into_iter().map(|c| char::from_u32(c).unwrap()).collect()
}

/// `kbo::find` (lib.rs:808-821)
pub fn find(query_seq: &[u8], idx: &GpuIndex, opts: kbo::FindOpts) -> Vec<kbo::format::RLE> {
    let (mut p, mut n) = (std::ptr::null_mut(), 0usize);
    let o = ffi::KboFindOpts { max_error_prob: opts.max_error_prob, max_gap_len: opts.max_gap_len };
    check(unsafe { ffi::kbo_find(idx.0, query_seq.as_ptr(), query_seq.len(), &o, &mut p, &mut n) });
    let v = unsafe { std::slice::from_raw_parts(p, n) }.iter().map(|r| kbo::format::RLE {
        start: r.start as usize, end: r.end as usize, matches: r.matches as usize, mismatches: r.mismatches as usize,
        jumps: r.jumps as usize, gap_bases: r.gap_bases as usize, gap_opens: r.gap_opens as usize }).collect();
    unsafe { ffi::kbo_free(p as *mut c_void) };
    v
}

/// `kbo::find` over all reads of a file at once, 2-bit packed over PCIe (`kbo_find_batch_packed`): reads[i] -> its runs.
pub fn find_batch(reads: &[Vec<u8>], idx: &GpuIndex, opts: kbo::FindOpts) -> Vec<Vec<kbo::format::RLE>> {
    let mut offsets = vec![0u64; reads.len() + 1];
    for (i, r) in reads.iter().enumerate() { offsets[i + 1] = offsets[i] + r.len() as u64; }
    let concat: Vec<u8> = reads.concat();
    let mut words = vec![0u32; unsafe { ffi::kbo_packed_words(offsets.as_ptr(), reads.len()) }];
    let cap = concat.iter().filter(|b| !matches!(**b, b'A' | b'C' | b'G' | b'T')).count();
    let (mut exc_pos, mut exc_byte, mut n_exc) = (vec![0u64; cap], vec![0u8; cap], 0usize);
    check(unsafe { ffi::kbo_pack_reads(concat.as_ptr(), offsets.as_ptr(), reads.len(), words.as_mut_ptr(), exc_pos.as_mut_ptr(),
                                       exc_byte.as_mut_ptr(), cap, &mut n_exc) });
    let o = ffi::KboFindOpts { max_error_prob: opts.max_error_prob, max_gap_len: opts.max_gap_len };
    let (mut p, mut ro) = (std::ptr::null_mut(), vec![0u64; reads.len() + 1]);
    check(unsafe { ffi::kbo_find_batch_packed(idx.0, words.as_ptr(), offsets.as_ptr(), reads.len(), exc_pos.as_ptr(), exc_byte.as_ptr(),
                                              n_exc, &o, &mut p, ro.as_mut_ptr()) });
    let all = unsafe { std::slice::from_raw_parts(p, ro[reads.len()] as usize) };
    let out = (0..reads.len()).map(|i| all[ro[i] as usize..ro[i + 1] as usize].iter().map(|r| kbo::format::RLE {
        start: r.start as usize, end: r.end as usize, matches: r.matches as usize, mismatches: r.mismatches as usize,
        jumps: r.jumps as usize, gap_bases: r.gap_bases as usize, gap_opens: r.gap_opens as usize }).collect()).collect();
    unsafe { ffi::kbo_free(p as *mut c_void) };
    out
}

/// `kbo::call` (lib.rs:547-573) with every sequence of a batch as `ref_seq` (`kbo_call_batch`).
pub fn call_batch(idx: &GpuIndex, seqs: &[Vec<u8>], opts: &kbo::CallOpts) -> Vec<Vec<kbo::variant_calling::Variant>> {
    let mut offsets = vec![0u64; seqs.len() + 1];
    for (i, r) in seqs.iter().enumerate() { offsets[i + 1] = offsets[i] + r.len() as u64; }
    let concat: Vec<u8> = seqs.concat();
    let b = &opts.sbwt_build_opts;
    let o = ffi::KboCallOpts { max_error_prob: opts.max_error_prob, sbwt_build_opts: ffi::KboBuildOpts {
        k: b.k as u32, add_revcomp: b.add_revcomp as i32, num_threads: b.num_threads as u32, prefix_precalc: b.prefix_precalc as u32,
        build_select: b.build_select as i32, mem_gb: b.mem_gb as u32, dedup_batches: b.dedup_batches as i32, temp_dir: std::ptr::null() } };
    // the flat form (`kbo_call_batch_flat`): the device's own order and layout, one allocation, 10 bytes per variant - the two
    // `Vec<u8>` of every `Variant` are made from slices of `chars` here, where the reference's type asks for them
    let mut flat = ffi::KboCallFlat { n_variants: 0, n_chars: 0, query_pos: std::ptr::null_mut(), query_len: std::ptr::null_mut(),
                                      ref_len: std::ptr::null_mut(), chars: std::ptr::null_mut() };
    let mut vo = vec![0u64; seqs.len() + 1];
    check(unsafe { ffi::kbo_call_batch_flat(idx.0, concat.as_ptr(), offsets.as_ptr(), seqs.len(), &o, &mut flat, vo.as_mut_ptr()) });
    let nv = flat.n_variants as usize;
    let (pos, ql, rl, chars) = unsafe { (std::slice::from_raw_parts(flat.query_pos, nv), std::slice::from_raw_parts(flat.query_len, nv),
                                         std::slice::from_raw_parts(flat.ref_len, nv), std::slice::from_raw_parts(flat.chars, flat.n_chars as usize)) };
    let mut c = 0usize;
    let mut out = Vec::with_capacity(seqs.len());
    for s in 0..seqs.len() {
        let mut vs = Vec::with_capacity((vo[s + 1] - vo[s]) as usize);
        for v in vo[s] as usize..vo[s + 1] as usize {
            let (a, b) = (ql[v] as usize, rl[v] as usize);
            vs.push(kbo::variant_calling::Variant { query_pos: pos[v] as usize, query_chars: chars[c..c + a].to_vec(),
                                                    ref_chars: chars[c + a..c + a + b].to_vec() });
            c += a + b;
        }
        out.push(vs);
    }
    unsafe { ffi::kbo_call_flat_free(&mut flat) };
    out
}

/// The recipe that turns SURVEY.md section 8(f) 4 from "unpinned" into "pinned" on a machine that has the `sbwt` crate and an MI355X
/// (neither is in the image this was written in: never compiled, never run).  The reference's own test inputs (index.rs:262-275,
/// lib.rs:600-609, lib.rs:786-805): the index the crate builds, handed over by `export_sbwt_parts` -> `kbo_index_from_parts`, must
/// give what `sbwt::StreamingIndex::matching_statistics` gives - depths AND intervals - and `matches` / `find` must give the crate's.
// (never compiled - no Rust toolchain in the image this was written in - so behind a feature of its own: a plain `cargo test` must not
// break on a signature this module assumes wrongly.  `cargo test --features pin-tests` on a machine with the crate and a GPU.)
#[cfg(all(test, feature = "pin-tests"))]
mod pin_against_the_crate {
    use super::*;

    fn crate_ms(query: &[u8], index: &sbwt::SbwtIndexVariant, lcs: &sbwt::LcsArray) -> Vec<(usize, Range<usize>)> {
        kbo::index::query_sbwt(query, index, lcs) // index.rs:243-256: StreamingIndex::new + matching_statistics
    }

    #[test]
    fn query_sbwt_of_the_handed_over_index_equals_the_crates() {
        // index.rs:265-268
        let reference: Vec<Vec<u8>> = vec![b"AAAGAACCA-TCAGGGCG".to_vec()];
        let query = b"CAAGCCACTCATTGGGTC".to_vec();
        let mut opts = kbo::BuildOpts::default();
        opts.k = 3;
        let (sbwt, lcs) = kbo::index::build_sbwt_from_vecs(&reference, &Some(opts));
        let gpu = GpuIndex::from_sbwt(&sbwt);
        let want = crate_ms(&query, &sbwt, &lcs);
        assert_eq!(want.iter().map(|x| x.0).collect::<Vec<_>>(), vec![1, 2, 2, 3, 2, 2, 3, 2, 1, 2, 3, 1, 1, 1, 2, 3, 1, 2]); // index.rs:270
        assert_eq!(query_sbwt(&query, &gpu), want); // depths and intervals
    }

    #[test]
    fn random_genome_every_position() {
        // 50 kbp, k = 31, 1 % substitutions, a few N: depths and intervals of 200 reads of 150 bases and of one 20 kbp sequence
        let mut x = 0x6B626F0001u64;
        let mut next = || { x = x.wrapping_add(0x9E3779B97F4A7C15); let mut z = x; z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
                            z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB); z ^ (z >> 31) };
        let genome: Vec<u8> = (0..50_000).map(|_| b"ACGT"[(next() >> 62) as usize]).collect();
        let (sbwt, lcs) = kbo::index::build_sbwt_from_vecs(&[genome.clone()], &Some(kbo::BuildOpts::default()));
        let gpu = GpuIndex::from_sbwt(&sbwt);
        let mut reads: Vec<Vec<u8>> = (0..200).map(|_| { let s = (next() % (50_000 - 150)) as usize; genome[s..s + 150].to_vec() }).collect();
        reads.push(genome[10_000..30_000].to_vec());
        for r in reads.iter_mut() {
            for b in r.iter_mut() { let v = next(); if (v & 0xFFFF) < 655 { *b = b"ACGT"[(v >> 62) as usize]; } else if (v & 0xFFFFF) == 7 { *b = b'N'; } }
            assert_eq!(query_sbwt(r, &gpu), crate_ms(r, &sbwt, &lcs));
        }
    }

    #[test]
    fn matches_and_find_equal_the_crates() {
        // lib.rs:600-609
        let reference: Vec<Vec<u8>> = vec![b"AAAGAACCA-TCAGGGCG".to_vec()];
        let query = b"GTGACTATGAGGAT".to_vec();
        let mut opts = kbo::BuildOpts::default();
        opts.k = 3;
        let (sbwt, lcs) = kbo::build(&reference, opts);
        let gpu = GpuIndex::from_sbwt(&sbwt);
        assert_eq!(matches(&query, &gpu, kbo::MatchOpts::default()), kbo::matches(&query, &sbwt, &lcs, kbo::MatchOpts::default()));
        assert_eq!(matches(&query, &gpu, kbo::MatchOpts::default()).iter().collect::<String>(), "---------MMM--"); // lib.rs:607
        assert_eq!(find(&query, &gpu, kbo::FindOpts::default()), kbo::find(&query, &sbwt, &lcs, kbo::FindOpts::default()));
    }
}
