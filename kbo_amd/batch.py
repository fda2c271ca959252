"""Batched and device-resident entry points (new surface; the reference API takes one
sequence per call, lib.rs:612-617, and leaves batching to kbo-cli)."""
import ctypes as C

import numpy as np

from . import _capi, derandomize
from ._capi import check, lib


def _prep(concat, offsets):
    concat = np.ascontiguousarray(concat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    return concat, offsets, len(offsets) - 1


def ms_batch(sbwt, concat, offsets, want_intervals=False):
    """kbo_ms_batch -> (d uint8, lo uint32 | None, hi uint32 | None)"""
    concat, offsets, n = _prep(concat, offsets)
    total = int(offsets[-1]) if n > 0 else 0
    d = np.zeros(max(total, 1), dtype=np.uint8)
    lo = np.zeros(max(total, 1), dtype=np.uint32) if want_intervals else None
    hi = np.zeros(max(total, 1), dtype=np.uint32) if want_intervals else None
    check(lib().kbo_ms_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, n, d.ctypes.data,
                             lo.ctypes.data if want_intervals else None,
                             hi.ctypes.data if want_intervals else None))
    return d[:total], (lo[:total] if want_intervals else None), (hi[:total] if want_intervals else None)


def matches_batch(sbwt, concat, offsets, max_error_prob=1e-7):
    """kbo::matches over every sequence of the batch -> uint8 chars"""
    concat, offsets, n = _prep(concat, offsets)
    total = int(offsets[-1]) if n > 0 else 0
    out = np.zeros(max(total, 1), dtype=np.uint8)
    check(lib().kbo_matches_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, n, max_error_prob,
                                  out.ctypes.data))
    return out[:total]


def map_batch(sbwt, concat, offsets, max_error_prob=1e-7, format=True):  # noqa: A002
    """kbo::map with fill_gaps=false, call_variants=false over the batch -> uint8"""
    concat, offsets, n = _prep(concat, offsets)
    total = int(offsets[-1]) if n > 0 else 0
    out = np.zeros(max(total, 1), dtype=np.uint8)
    check(lib().kbo_map_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, n, max_error_prob,
                              int(format), out.ctypes.data))
    return out[:total]


def find_batch(sbwt, concat, offsets, find_opts=None):
    """kbo::find over the batch -> (list of RLE tuples, rle_offsets uint64[n+1])"""
    from . import FindOpts
    o = find_opts if find_opts is not None else FindOpts()
    co = _capi.FindOpts(o.max_error_prob, o.max_gap_len)
    concat, offsets, n = _prep(concat, offsets)
    ro = np.zeros(n + 1, dtype=np.uint64)
    p = C.POINTER(_capi.RLE)()
    check(lib().kbo_find_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, n, C.byref(co),
                               C.byref(p), ro.ctypes.data))
    total = int(ro[-1])
    if total > 100_000:  # big batches: hand back the records as an [n_runs, 7] uint64 array
        rles = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(total, 7)).copy()
    else:
        rles = [p[i].as_tuple() for i in range(total)]
    lib().kbo_free(p)
    return rles, ro


def call_batch(sbwt, concat, offsets, call_opts=None):
    """kbo::call with every sequence of the batch as ref_seq -> list (per sequence) of lists of Variant"""
    from . import CallOpts, variant_calling
    o = call_opts if call_opts is not None else CallOpts()
    co = _capi.CallOpts(o.max_error_prob, o.sbwt_build_opts._to_c())
    concat, offsets, n = _prep(concat, offsets)
    vo = np.zeros(n + 1, dtype=np.uint64)
    p = C.POINTER(_capi.Variant)()
    check(lib().kbo_call_batch(sbwt._h, concat.ctypes.data, offsets.ctypes.data, n, C.byref(co), C.byref(p), vo.ctypes.data))
    allv = variant_calling._from_c(p, int(vo[-1]))  # (frees the allocation)
    return [allv[int(vo[s]):int(vo[s + 1])] for s in range(n)]


def pack_reads(concat, offsets):
    """bytes -> (u32 words, exception positions u64, exception bytes u8): the layout of kbo_hip.h's packed entry points"""
    concat, offsets, n = _prep(concat, offsets)
    words = np.zeros(int(lib().kbo_packed_words(offsets.ctypes.data, n)), dtype=np.uint32)
    cap = 1024
    while True:
        pos, byt, ne = np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint8), C.c_size_t(0)
        rc = lib().kbo_pack_reads(concat.ctypes.data, offsets.ctypes.data, n, words.ctypes.data, pos.ctypes.data, byt.ctypes.data,
                                  cap, C.byref(ne))
        if rc == -5 and ne.value > cap:  # KBO_E_NOMEM: more non-ACGT bases than the list held
            cap = ne.value
            continue
        check(rc)
        return words, pos[:ne.value].copy(), byt[:ne.value].copy()


def unpack_matches(words, offsets):
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = np.zeros(int(offsets[-1]), dtype=np.uint8)
    check(lib().kbo_unpack_matches(words.ctypes.data, offsets.ctypes.data, len(offsets) - 1, out.ctypes.data))
    return out


def matches_batch_packed(sbwt, words, offsets, exc_pos, exc_byte, max_error_prob=1e-7):
    """kbo::matches over 2-bit packed reads -> 2-bit packed characters (M, -, X, R = 0 .. 3), same word layout"""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = np.zeros(len(words), dtype=np.uint32)
    check(lib().kbo_matches_batch_packed(sbwt._h, words.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                         exc_pos.ctypes.data if len(exc_pos) else None, exc_byte.ctypes.data if len(exc_byte) else None,
                                         len(exc_pos), max_error_prob, out.ctypes.data))
    return out


def find_batch_packed(sbwt, words, offsets, exc_pos, exc_byte, find_opts=None):
    from . import FindOpts
    o = find_opts if find_opts is not None else FindOpts()
    co = _capi.FindOpts(o.max_error_prob, o.max_gap_len)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    ro = np.zeros(n + 1, dtype=np.uint64)
    p = C.c_void_p()
    check(lib().kbo_find_batch_packed(sbwt._h, words.ctypes.data, offsets.ctypes.data, n, exc_pos.ctypes.data if len(exc_pos) else None,
                                      exc_byte.ctypes.data if len(exc_byte) else None, len(exc_pos), C.byref(co), C.byref(p), ro.ctypes.data))
    total = int(ro[-1])
    rles = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(max(1, total), 7))[:total].astype(np.uint64)  # (kbo_rle32)
    lib().kbo_free(p)
    return rles, ro


class _FlatOwner:
    """keeps a kbo_call_flat alive for the numpy views into it (freed with the last of them)"""

    def __init__(self, flat):
        self.flat = flat

    def __del__(self):
        try:
            lib().kbo_call_flat_free(C.byref(self.flat))
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def call_batch_arrays(sbwt, concat, offsets, call_opts=None):
    """kbo::call over a batch without a Python object per variant (kbo_hip.h kbo_call_batch_flat) -> dict of numpy arrays:
    var_offsets (n_seqs + 1), query_pos, query_len, ref_len (one entry per variant) and chars (per variant its query characters,
    then its reference characters, back to back in variant order) - views of the library's one allocation, nothing copied.
    variants_of(result, s) turns one sequence's slice into (query_pos, query_chars, ref_chars) tuples."""
    from . import CallOpts
    o = call_opts if call_opts is not None else CallOpts()
    co = _capi.CallOpts(o.max_error_prob, o.sbwt_build_opts._to_c())
    concat, offsets, n = _prep(concat, offsets)
    vo = np.zeros(n + 1, dtype=np.uint64)
    flat = _capi.CallFlat()
    check(lib().kbo_call_batch_flat(sbwt._h, concat.ctypes.data, offsets.ctypes.data, n, C.byref(co), C.byref(flat), vo.ctypes.data))
    owner = _FlatOwner(flat)
    nv, nc = int(flat.n_variants), int(flat.n_chars)

    def view(ptr, count, dtype):
        if count == 0:
            return np.zeros(0, dtype=dtype)
        return np.ctypeslib.as_array(ptr, shape=(count,))

    return {"var_offsets": vo, "query_pos": view(flat.query_pos, nv, np.uint32), "query_len": view(flat.query_len, nv, np.uint16),
            "ref_len": view(flat.ref_len, nv, np.uint16), "chars": view(flat.chars, nc, np.uint8), "_owner": owner}


def variants_of(res, s):
    """the variants of sequence s of a call_batch_arrays result as (query_pos, query_chars bytes, ref_chars bytes) tuples"""
    a, b = int(res["var_offsets"][s]), int(res["var_offsets"][s + 1])
    start = (np.concatenate([[0], np.cumsum(res["query_len"].astype(np.int64) + res["ref_len"].astype(np.int64))])
             if "_starts" not in res else res["_starts"])
    res["_starts"] = start
    out = []
    for v in range(a, b):
        q0, ql, rl = int(start[v]), int(res["query_len"][v]), int(res["ref_len"][v])
        out.append((int(res["query_pos"][v]), res["chars"][q0:q0 + ql].tobytes(), res["chars"][q0 + ql:q0 + ql + rl].tobytes()))
    return out


class PackedDeviceBatch:
    """A batch of reads resident in HBM as 2-bit words (kbo_hip.h kbo_matches_packed_dev): run() = kbo::matches through the one
    kernel's packed-native form, the characters as 2-bit words in self.words_out (unpack_matches() for bytes)."""

    def __init__(self, sbwt, concat, offsets, device, max_error_prob=1e-7):
        import torch
        self.torch, self.sbwt, self.device = torch, sbwt, device
        concat, offsets, n = _prep(concat, offsets)
        self.offsets = offsets
        self.n_seqs, self.total = n, int(offsets[-1])
        lens = np.diff(offsets.astype(np.int64))
        self.max_len = int(lens.max()) if n else 0
        self.uniform_len = self.max_len if n and int(lens.min()) == self.max_len else 0
        self.max_error_prob = max_error_prob
        words, pos, byt = pack_reads(concat, offsets)
        self.n_exc = len(pos)
        with torch.cuda.device(device):
            sbwt.to_device(-1)
            self.words = torch.from_numpy(words.view(np.int32)).to(device)
            self.words_out = torch.zeros(len(words) + 4, dtype=torch.int32, device=device)
            self.off = torch.from_numpy(offsets.view(np.int64)).to(device)
            self.exc_pos = torch.from_numpy(pos.view(np.int64)).to(device) if self.n_exc else None
            self.exc_byte = torch.from_numpy(byt).to(device) if self.n_exc else None
            self.scratch = torch.zeros(int(lib().kbo_matches_packed_dev_scratch_bytes(n, self.total)) // 8 + 2, dtype=torch.int64, device=device)
            self.work_bytes = int(lib().kbo_index_work_bytes(sbwt._h, n, self.total, self.max_len))
            self.work = torch.zeros(self.work_bytes // 8 + 2, dtype=torch.int64, device=device)

    def run(self, stream=None, tail_stream=None):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        t = tail_stream if tail_stream is not None else s
        check(lib().kbo_matches_packed_dev(self.sbwt._h, self.words.data_ptr(), self.off.data_ptr(), self.n_seqs, self.total, self.max_len,
                                           self.uniform_len, self.exc_pos.data_ptr() if self.n_exc else None,
                                           self.exc_byte.data_ptr() if self.n_exc else None, self.n_exc, self.max_error_prob,
                                           self.words_out.data_ptr(), self.scratch.data_ptr(), self.work.data_ptr(), self.work_bytes,
                                           s.cuda_stream, t.cuda_stream))

    def chars(self):
        """the characters as bytes, on the host"""
        return unpack_matches(self.words_out[:len(self.words)].cpu().numpy().view(np.uint32), self.offsets)


class DeviceBatch:
    """A batch of reads resident in HBM (torch owns the memory, the C ABI gets raw
    pointers and the torch stream).  run() = A1 walk kernel, then fused A5+A6 kernel."""

    def __init__(self, sbwt, concat, offsets, device, max_error_prob=1e-7, want_intervals=False,
                 format=False, want_ms=True):  # noqa: A002
        import torch
        self.torch = torch
        self.sbwt = sbwt
        self.device = device
        concat, offsets, n = _prep(concat, offsets)
        self.n_seqs = n
        self.total = int(offsets[-1])
        self.max_len = int(np.max(np.diff(offsets.astype(np.int64)))) if n else 0
        self.k = sbwt.k()
        self.threshold = derandomize.random_match_threshold(self.k, sbwt.n_kmers(), 4, max_error_prob)
        pad = (self.total + 15) // 16 * 16 + 64
        with torch.cuda.device(device):
            sbwt.to_device(-1)
            self.q = torch.zeros(pad, dtype=torch.uint8, device=device)
            self.q[:self.total] = torch.from_numpy(concat if concat.flags.writeable else concat.copy()).to(device)
            self.off = torch.from_numpy(offsets.view(np.int64)).to(device)
            self.ms = torch.zeros(pad, dtype=torch.uint8, device=device)
            self.chars = torch.zeros(pad, dtype=torch.uint8, device=device)
            self.work_bytes = int(lib().kbo_index_work_bytes(sbwt._h, n, self.total, self.max_len))
            self.work = torch.zeros(self.work_bytes // 8 + 2, dtype=torch.int64, device=device)
            self.lo = torch.zeros(self.total, dtype=torch.int32, device=device) if want_intervals else None
            self.hi = torch.zeros(self.total, dtype=torch.int32, device=device) if want_intervals else None
        self.format = format
        self.want_ms = want_ms
        self.max_error_prob = max_error_prob
        self.fused = None  # set by run(): True when the last run() took the one-kernel route (map_kernels.hip)

    def walk(self, stream=None):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        check(lib().kbo_ms_batch_dev(self.sbwt._h, self.q.data_ptr(), self.off.data_ptr(), self.n_seqs,
                                     self.total, self.max_len, self.ms.data_ptr(),
                                     self.lo.data_ptr() if self.lo is not None else None,
                                     self.hi.data_ptr() if self.hi is not None else None,
                                     self.work.data_ptr(), self.work_bytes, s.cuda_stream))

    def plan_stats(self, stream=None):
        """Work counters of the last walk() when it took the plan-guided stage (kbo_hip_tuning.h kbo_plan_stats_dev)."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        out = np.zeros(20, dtype=np.uint64)
        check(lib().kbo_plan_stats_dev(self.n_seqs, self.total, self.max_len, self.k, self.work.data_ptr(), out.ctypes.data,
                                       s.cuda_stream))
        names = ("units_walked", "accepted", "failed", "levels", "entry_levels", "seed_lookups", "seed_extensions",
                 "mismatches", "units", "redo_entries", "gave_up", "guard", "tab_lookups", "tab_written", "tab_flagged",
                 "tab_unresolved", "tab_anchored", "items_noplan")
        return {n: int(v) for n, v in zip(names, out)}

    def plan_flags(self, stream=None):
        """uint8[n_seqs]: which reads the last planned launch left to the plain walk (kbo_hip_tuning.h kbo_plan_flags_dev)"""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        out = np.zeros(self.n_seqs, dtype=np.uint8)
        check(lib().kbo_plan_flags_dev(self.n_seqs, self.total, self.max_len, self.k, self.work.data_ptr(), out.ctypes.data, s.cuda_stream))
        return out

    def long_stats(self, stream=None):
        """What the last run() / run_find() did when it took the kernel for sequences of any length (kbo_hip_tuning.h kbo_long_stats_dev)."""
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        out = np.zeros(25, dtype=np.uint64)
        check(lib().kbo_long_stats_dev(self.n_seqs, self.total, self.max_len, self.k, self.work.data_ptr(), out.ctypes.data, s.cuda_stream))
        names = ("pieces", "flagged", "sub_items", "seed_lookups", "filter_lookups", "table_lookups", "second_lookups", "_",
                 "cyc_staging", "cyc_stretches", "cyc_planes", "cyc_proof", "cyc_output", "band_tried", "band_taken", "_",
                 "why_other", "why_list", "why_ext", "why_back", "why_on", "why_band_list", "why_band_ext", "why_band_back", "why_band_on")
        return {n: int(v) for n, v in zip(names, out) if n != "_"}

    def derand_translate(self, stream=None):
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        if self.max_len == 0 or self.max_len > 480:  # long reads / contigs: scratch for the piece-wise kernel
            if getattr(self, "dt_work", None) is None:
                self.dt_work_bytes = int(lib().kbo_derand_work_bytes(self.n_seqs, self.total))
                self.dt_work = self.torch.zeros(self.dt_work_bytes // 8 + 2, dtype=self.torch.int64, device=self.device)
            work, work_bytes = self.dt_work.data_ptr(), self.dt_work_bytes
        else:
            work, work_bytes = None, 0
        check(lib().kbo_derand_translate_dev(self.ms.data_ptr(), self.off.data_ptr(), self.n_seqs, self.total, self.k,
                                             self.threshold, self.q.data_ptr() if self.format else None,
                                             self.chars.data_ptr(), self.max_len, work, work_bytes, s.cuda_stream))

    def run_lengths(self, max_gap_len=0, stream=None, runs_per_seq=2):
        """format::run_lengths_gapped of the (unformatted) characters, on the device: fills self.rle_records
        ([capacity, 7] int32) and self.rle_work (first-run indices; last word = number of runs)."""
        torch = self.torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        if getattr(self, "rle_work", None) is None:
            wb = int(lib().kbo_run_lengths_work_bytes(self.n_seqs))
            self.rle_work = torch.zeros(wb // 4 + 1, dtype=torch.int32, device=self.device)
            self.rle_capacity = runs_per_seq * self.n_seqs + 16
            self.rle_records = torch.zeros((self.rle_capacity, 7), dtype=torch.int32, device=self.device)
        check(lib().kbo_run_lengths_dev(self.chars.data_ptr(), self.off.data_ptr(), self.n_seqs, self.max_len, max_gap_len,
                                        self.rle_work.data_ptr(), self.rle_records.data_ptr(), self.rle_capacity,
                                        s.cuda_stream))

    def run_find(self, max_gap_len=0, stream=None, tail_stream=None, runs_per_seq=2):
        """kbo::find over the batch (kbo_hip.h kbo_find_batch_dev): the characters (unformatted) and their run lengths, as
        run() + run_lengths() leave them; with max_gap_len = 0 the one kernel counts the runs itself"""
        torch = self.torch
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        t = tail_stream if tail_stream is not None else s
        if getattr(self, "rle_work", None) is None:
            wb = int(lib().kbo_run_lengths_work_bytes(self.n_seqs))
            self.rle_work = torch.zeros(wb // 4 + 1, dtype=torch.int32, device=self.device)
            self.rle_capacity = runs_per_seq * self.n_seqs + 16
            self.rle_records = torch.zeros((self.rle_capacity, 7), dtype=torch.int32, device=self.device)
        fused = C.c_int(0)
        check(lib().kbo_find_batch_dev(self.sbwt._h, self.q.data_ptr(), self.off.data_ptr(), self.n_seqs, self.total, self.max_len,
                                       self.max_error_prob, max_gap_len, self.ms.data_ptr(), self.chars.data_ptr(), self.work.data_ptr(),
                                       self.work_bytes, self.rle_work.data_ptr(), self.rle_records.data_ptr(), self.rle_capacity,
                                       s.cuda_stream, t.cuda_stream, C.byref(fused)))
        self.fused = bool(fused.value)

    def run_lengths_host(self):
        """-> (records [n_runs, 7] uint32, first-run index per sequence uint32 [n_seqs + 1]) on the host"""
        w = self.rle_work.cpu().numpy().view(np.uint32)
        n = self.n_seqs
        words = n + 1 + (n + 1 + 1023) // 1024
        total = int(w[words])
        first = w[n + 1 + np.arange(n + 1) // 1024] + w[:n + 1]
        return self.rle_records[:total].cpu().numpy().view(np.uint32), first

    def run(self, stream=None, tail_stream=None):
        """kbo::map (format) / kbo::matches over the batch: kbo_map_batch_dev - one kernel for reads over an index copy with a
        depth table (self.ms then holds every MS value only when want_ms), one wave per piece for longer sequences when the MS
        values are not asked for (long_kernels.hip), else the walk + the derandomize / translate kernels."""
        if self.lo is not None:
            self.walk(stream)
            self.derand_translate(stream)
            self.fused = False
            return
        s = stream if stream is not None else self.torch.cuda.current_stream(self.device)
        fused = C.c_int(0)
        if tail_stream is not None:  # the second pass on another stream (kbo_hip.h kbo_map_batch_dev_tail): complete when both have drained
            check(lib().kbo_map_batch_dev_tail(self.sbwt._h, self.q.data_ptr(), self.off.data_ptr(), self.n_seqs, self.total, self.max_len,
                                               self.max_error_prob, int(self.format), int(self.want_ms), self.ms.data_ptr(),
                                               self.chars.data_ptr(), self.work.data_ptr(), self.work_bytes, s.cuda_stream,
                                               tail_stream.cuda_stream, C.byref(fused)))
        else:
            check(lib().kbo_map_batch_dev(self.sbwt._h, self.q.data_ptr(), self.off.data_ptr(), self.n_seqs, self.total, self.max_len,
                                          self.max_error_prob, int(self.format), int(self.want_ms), self.ms.data_ptr(), self.chars.data_ptr(),
                                          self.work.data_ptr(), self.work_bytes, s.cuda_stream, C.byref(fused)))
        self.fused = bool(fused.value)


def stream_pair(device, tail_cus=-1):
    """(stream, tail_stream) as torch streams, made by the library (kbo_hip.h kbo_stream_pair_create): the tail stream - a batch's second
    pass - on compute units of its own.  The pair lives as long as the process (the torch wrappers do not own the streams)."""
    import torch
    with torch.cuda.device(device):
        ks, ts = C.c_void_p(), C.c_void_p()
        check(lib().kbo_stream_pair_create(tail_cus, C.byref(ks), C.byref(ts)))
        return torch.cuda.ExternalStream(ks.value, device=device), torch.cuda.ExternalStream(ts.value, device=device)


class MapStream:
    """Several device-resident batches in flight through the library's own pipelines (kbo_hip.h kbo_map_stream_*): pairs of (kernel
    stream, second-pass stream) that take the batches in turn, two slots of work memory each."""

    def __init__(self, sbwt, max_seqs, max_bases, max_seq_len, pipelines=2):
        self.sbwt = sbwt
        h = C.c_void_p()
        check(lib().kbo_map_stream_create(sbwt._h, pipelines, max_seqs, max_bases, max_seq_len, C.byref(h)))
        self._h = h

    def submit(self, dev, ready_stream=None):
        """kbo::map (dev.format) / kbo::matches of a DeviceBatch's sequences into dev.chars (and the matching statistics into dev.ms
        when the batch was made with want_ms) -> ticket"""
        t = C.c_uint64(0)
        fused = C.c_int(0)
        check(lib().kbo_map_stream_submit(self._h, dev.q.data_ptr(), dev.off.data_ptr(), dev.n_seqs, dev.total, dev.max_len, dev.max_error_prob,
                                          int(dev.format), dev.ms.data_ptr() if dev.want_ms else None, dev.chars.data_ptr(),
                                          ready_stream.cuda_stream if ready_stream is not None else None, C.byref(t), C.byref(fused)))
        dev.fused = bool(fused.value)
        return int(t.value)

    def wait(self, ticket):
        check(lib().kbo_map_stream_wait(self._h, ticket))

    def wait_on(self, ticket, stream):
        check(lib().kbo_map_stream_wait_on(self._h, ticket, stream.cuda_stream))

    def sync(self):
        check(lib().kbo_map_stream_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib().kbo_map_stream_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
