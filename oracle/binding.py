"""ctypes binding of the CPU oracle (oracle/libkbo_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Product code (kbo_amd/) must never import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkbo_oracle.so")


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("bases", "extend_calls", "rank_calls", "rank_blocks", "contracts", "lcs_reads")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PlanParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("seed_table_depth", "seed_depth", "seed_cap", "gap", "chunk", "list_cap", "bail_x16",
                                          "recovery_lines", "depth_table", "depth_anchors")]


class PlanCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "bases", "items", "items_unseeded", "items_clean", "items_list_overflow", "items_flagged", "gave_up",
        "seed_lookups", "seed_extensions", "pos_lookups", "compare_bases", "mismatches",
        "units_counted", "units", "units_head", "units_plain", "node_lookups",
        "walk_accepted", "walk_failed", "walk_contractions", "walk_entry_levels", "walk_short_windows", "walk_iterations_lines",
        "walk_out_bytes", "unit_distinct_lines", "unit_distinct_rank_lines", "redo_bases", "redo_iterations",
        "tab_lookups", "tab_written", "tab_flagged", "tab_anchored", "items_noplan", "tab_stretches")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RLE(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("start", "end", "matches", "mismatches", "jumps", "gap_bases", "gap_opens")]

    def as_tuple(self):
        return tuple(int(getattr(self, n)) for n, _ in self._fields_)


class OVariant(C.Structure):
    _fields_ = [("query_pos", C.c_uint64), ("query_chars", C.c_uint8 * 256), ("query_len", C.c_uint32),
                ("ref_chars", C.c_uint8 * 256), ("ref_len", C.c_uint32), ("overflow", C.c_uint32)]

    def as_tuple(self):
        return (int(self.query_pos), bytes(self.query_chars[:self.query_len]).decode(),
                bytes(self.ref_chars[:self.ref_len]).decode())


def build_lib():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _stale():
    """the library is missing or older than one of its sources"""
    if not os.path.exists(_LIB_PATH):
        return True
    t = os.path.getmtime(_LIB_PATH)
    return any(os.path.getmtime(os.path.join(_HERE, f)) > t for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile")


def _load():
    if _stale():
        build_lib()
    lib = C.CDLL(_LIB_PATH)
    vp, u8p, u64p, i64p, u32p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), \
        C.POINTER(C.c_int64), C.POINTER(C.c_uint32)
    lib.ora_index_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t,
                                    C.c_uint32, C.c_int, C.POINTER(vp)]
    lib.ora_index_from_parts.argtypes = [C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(vp), u64p,
                                         vp, C.POINTER(vp)]
    lib.ora_index_free.argtypes = [vp]
    lib.ora_index_free.restype = None
    lib.ora_index_k.argtypes = [vp]; lib.ora_index_k.restype = C.c_uint32
    lib.ora_index_n_sets.argtypes = [vp]; lib.ora_index_n_sets.restype = C.c_uint64
    lib.ora_index_n_kmers.argtypes = [vp]; lib.ora_index_n_kmers.restype = C.c_uint64
    lib.ora_index_C.argtypes = [vp, u64p]; lib.ora_index_C.restype = None
    lib.ora_index_bits.argtypes = [vp, C.c_int]; lib.ora_index_bits.restype = vp
    lib.ora_index_lcs.argtypes = [vp]; lib.ora_index_lcs.restype = vp
    lib.ora_index_access_kmer.argtypes = [vp, C.c_uint64, vp]
    lib.ora_matching_statistics.argtypes = [vp, vp, C.c_size_t, vp, vp, vp, C.POINTER(Counters)]
    lib.ora_log_rm_max_cdf.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t]
    lib.ora_log_rm_max_cdf.restype = C.c_double
    lib.ora_random_match_threshold.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, C.c_double]
    lib.ora_random_match_threshold.restype = C.c_size_t
    lib.ora_derandomize_ms_val.argtypes = [C.c_size_t, C.c_int64, C.c_size_t, C.c_size_t]
    lib.ora_derandomize_ms_val.restype = C.c_int64
    lib.ora_derandomize_ms_vec.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]
    lib.ora_translate_ms_val.argtypes = [C.c_int64, C.c_int64, C.c_int64, C.c_size_t, u32p, u32p]
    lib.ora_translate_ms_val.restype = None
    lib.ora_translate_ms_vec.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_size_t, vp]
    lib.ora_matches.argtypes = [vp, vp, C.c_size_t, C.c_double, vp]
    lib.ora_run_lengths_gapped.argtypes = [vp, C.c_size_t, C.c_size_t, C.POINTER(RLE), C.c_size_t]
    lib.ora_run_lengths_gapped.restype = C.c_size_t
    lib.ora_run_lengths_batch.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, C.c_size_t, vp]
    lib.ora_run_lengths_batch.restype = C.c_size_t
    lib.ora_relative_to_ref.argtypes = [vp, vp, C.c_size_t, vp]
    lib.ora_relative_to_ref.restype = None
    lib.ora_matches_batch.argtypes = [vp, vp, vp, C.c_size_t, C.c_double, C.c_int, vp, vp,
                                      C.POINTER(Counters)]
    lib.ora_matches_batch_timed.argtypes = [vp, vp, vp, C.c_size_t, C.c_double, C.c_int, C.c_int, vp, vp,
                                            C.POINTER(C.c_double)]
    lib.ora_plan_model.argtypes = [vp, vp, vp, vp, C.POINTER(PlanParams), vp, vp, C.c_size_t, C.c_int, vp, C.POINTER(PlanCounts)]
    lib.ora_call_sites_batch.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_int, vp, C.c_size_t]
    lib.ora_call_sites_batch.restype = C.c_long
    lib.ora_call.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_double, C.POINTER(OVariant), C.c_size_t]
    lib.ora_call.restype = C.c_long
    lib.ora_add_variants.argtypes = [vp, C.c_size_t, C.POINTER(OVariant), C.c_size_t]
    lib.ora_fill_gaps.argtypes = [vp, vp, vp, vp, vp, vp, C.c_size_t, C.c_size_t, C.c_double, vp]
    lib.ora_map.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_double, C.c_int, C.c_int, C.c_int, vp]
    lib.ora_kmer_colex_ranks.argtypes = [vp, C.c_size_t, C.c_uint32, vp, C.c_size_t, C.c_int, vp, vp]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


class OracleError(AssertionError):
    """Mirrors a reference assert!/panic! (code in .code)."""

    def __init__(self, code):
        super().__init__(f"oracle error {code}")
        self.code = code


def _chk(rc):
    if rc != 0:
        raise OracleError(rc)


def _bytes(x):
    if isinstance(x, str):
        x = x.encode()
    return np.frombuffer(bytes(x), dtype=np.uint8) if not isinstance(x, np.ndarray) else \
        np.ascontiguousarray(x, dtype=np.uint8)


class Index:
    """sbwt::SbwtIndexVariant + sbwt::LcsArray as produced by kbo::build (lib.rs:501-506)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def build(cls, seqs, k=31, add_revcomp=False):
        seqs = [bytes(_bytes(s)) for s in seqs]
        arr = (C.c_char_p * len(seqs))(*seqs)
        lens = (C.c_size_t * len(seqs))(*[len(s) for s in seqs])
        h = C.c_void_p()
        _chk(lib().ora_index_build(arr, lens, len(seqs), k, int(add_revcomp), C.byref(h)))
        return cls(h)

    @classmethod
    def from_parts(cls, k, n_sets, n_kmers, rows, Carr, lcs):
        rows = [np.ascontiguousarray(r, dtype=np.uint64) for r in rows]
        ptrs = (C.c_void_p * 4)(*[r.ctypes.data for r in rows])
        Cc = (C.c_uint64 * 4)(*[int(v) for v in Carr])
        lcs = np.ascontiguousarray(lcs, dtype=np.uint8)
        h = C.c_void_p()
        _chk(lib().ora_index_from_parts(k, n_sets, n_kmers, ptrs, Cc, lcs.ctypes.data, C.byref(h)))
        return cls(h)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().ora_index_free(self._h)
            except Exception:  # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    @property
    def k(self):
        return int(lib().ora_index_k(self._h))

    @property
    def n_sets(self):
        return int(lib().ora_index_n_sets(self._h))

    @property
    def n_kmers(self):
        return int(lib().ora_index_n_kmers(self._h))

    @property
    def C(self):
        out = (C.c_uint64 * 4)()
        lib().ora_index_C(self._h, out)
        return [int(v) for v in out]

    def bits(self, c):
        nw = (self.n_sets + 63) // 64
        p = lib().ora_index_bits(self._h, c)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(nw,)).copy()

    def lcs(self):
        p = lib().ora_index_lcs(self._h)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.n_sets,)).copy()

    def access_kmer(self, colex):
        out = np.zeros(self.k, dtype=np.uint8)
        _chk(lib().ora_index_access_kmer(self._h, colex, out.ctypes.data))
        return out.tobytes()

    def matching_statistics(self, query, counters=None):
        """index::query_sbwt (index.rs:243-256) -> (d, lo, hi) uint64 arrays."""
        q = _bytes(query)
        n = len(q)
        d = np.zeros(n, dtype=np.uint64); lo = np.zeros(n, dtype=np.uint64); hi = np.zeros(n, dtype=np.uint64)
        _chk(lib().ora_matching_statistics(self._h, q.ctypes.data, n, d.ctypes.data, lo.ctypes.data,
                                           hi.ctypes.data, C.byref(counters) if counters is not None else None))
        return d, lo, hi

    def matches(self, query, max_error_prob=1e-7):
        q = _bytes(query)
        out = np.zeros(len(q), dtype=np.uint8)
        _chk(lib().ora_matches(self._h, q.ctypes.data, len(q), max_error_prob, out.ctypes.data))
        return out.tobytes()

    def call(self, ref_seq, k, max_error_prob=1e-7):
        """kbo::call (lib.rs:547-573) -> list of (query_pos, query_chars, ref_chars)"""
        r = _bytes(ref_seq)
        cap = len(r) // 2 + 16
        buf = (OVariant * cap)()
        n = lib().ora_call(self._h, r.ctypes.data, len(r), k, max_error_prob, buf, cap)
        if n < 0:
            raise OracleError(n)
        return [buf[i].as_tuple() for i in range(n)], buf, n

    def fill_gaps(self, translation, ref_seq, threshold, max_err_prob):
        """gap_filling::fill_gaps on the oracle's own MS of ref_seq -> bytes"""
        r = _bytes(ref_seq)
        d, lo, hi = self.matching_statistics(r)
        t = _bytes(translation)
        out = np.zeros(len(r), dtype=np.uint8)
        _chk(lib().ora_fill_gaps(self._h, t.ctypes.data, d.ctypes.data, lo.ctypes.data, hi.ctypes.data,
                                 r.ctypes.data, len(r), threshold, max_err_prob, out.ctypes.data))
        return out.tobytes()

    def map(self, ref_seq, k, max_error_prob=1e-7, fill_gaps=True, call_variants=True, format=True):  # noqa: A002
        """kbo::map (lib.rs:720-761) -> bytes"""
        r = _bytes(ref_seq)
        out = np.zeros(len(r), dtype=np.uint8)
        _chk(lib().ora_map(self._h, r.ctypes.data, len(r), k, max_error_prob, int(fill_gaps), int(call_variants),
                           int(format), out.ctypes.data))
        return out.tobytes()

    def matches_batch(self, concat, offsets, max_error_prob=1e-7, n_threads=1, want_d=False,
                      counters=None):
        concat = np.ascontiguousarray(concat, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        chars = np.zeros(len(concat), dtype=np.uint8)
        d = np.zeros(len(concat), dtype=np.uint8) if want_d else None
        _chk(lib().ora_matches_batch(self._h, concat.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                     max_error_prob, n_threads, chars.ctypes.data,
                                     d.ctypes.data if want_d else None,
                                     C.byref(counters) if counters is not None else None))
        return (chars, d) if want_d else chars


def _matches_batch_timed(self, concat, offsets, max_error_prob=1e-7, n_threads=1, passes=1, want_d=True):
    """-> (chars, d | None, seconds of the `passes` timed passes): pinned pool, warm-up pass first (kbo_oracle.h)"""
    concat = np.ascontiguousarray(concat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    chars = np.zeros(len(concat), dtype=np.uint8)
    d = np.zeros(len(concat), dtype=np.uint8) if want_d else None
    sec = C.c_double()
    _chk(lib().ora_matches_batch_timed(self._h, concat.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                       max_error_prob, n_threads, passes, chars.ctypes.data,
                                       d.ctypes.data if want_d else None, C.byref(sec)))
    return chars, d, float(sec.value)


Index.matches_batch_timed = _matches_batch_timed


def _call_sites_batch(self, concat, offsets, threshold, n_threads=1):
    """First pass of call_variants (variant_calling.rs:266-273) for every read of a batch -> u64 array (n_sites, 4) of
    {read, i, j, ref_colex}, in read order."""
    concat = np.ascontiguousarray(concat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    cap = max(1024, len(concat) // 16)
    while True:
        recs = np.zeros((cap, 4), dtype=np.uint64)
        n = int(lib().ora_call_sites_batch(self._h, concat.ctypes.data, offsets.ctypes.data, len(offsets) - 1, threshold,
                                           n_threads, recs.ctypes.data, cap))
        if n < 0:
            raise OracleError(n)
        if n <= cap:
            return recs[:n]
        cap = n


Index.call_sites_batch = _call_sites_batch


def shipped_depth_table_order(k, n_sets):
    """device_index.cpp: order of the depth table a device copy of an index of n_sets rows gets by default (0 = none)"""
    import math
    lg = math.log2(max(n_sets, 4)) / 2.0
    order = min(int(math.ceil(lg + 3.2)), 17, k)
    if order < lg + 1.9 and order < k:
        order = 0
    return order


def shipped_depth_table_anchors(k, n_sets, order):
    """device_index.cpp: whether a device copy's depth table gets anchors by default: where its margin over log4(rows) is
    below 3.75 bases and the index has 24 Mi rows or more (C3, C4, a 1 Gbp index), and on small indexes (2 x rows <= 2^24: C2),
    where the one kernel prices a window that is present by chance with them"""
    import math
    if not order or order >= k:
        return False
    return bool((order < math.log2(max(n_sets, 4)) / 2.0 + 3.75 and n_sets >= (24 << 20)) or 2 * n_sets <= (1 << 24))


def shipped_plan_params(k, n_sets, recovery_lines=None, depth_table=None, depth_anchors=None):
    """The parameters the product's plan-guided stage runs with by default on an index of n_sets rows (device_index.cpp:
    depth table, seed table depth; plan_kernels.hip launch_plan: seed depth, gap; walk_kernels.hip: recovery lines from 24 Mi rows)."""
    import math
    half_log = int(math.floor(math.log2(max(n_sets, 4)) / 2.0 + 0.5))  # std::lround(log2(n) / 2)
    order = shipped_depth_table_order(k, n_sets) if depth_table is None else int(depth_table)
    d = 10 if (k >= 10 and n_sets >= (1 << 20)) else (8 if k >= 8 else 0)
    if d == 10 and k >= 13 and n_sets >= (512 << 20):
        d = 13
    elif d == 10 and k >= 12 and n_sets >= (32 << 20):
        d = 12
    if order > 0:  # with a depth table: as deep as a seed must be, at most 14 (device_index.cpp)
        d = min(half_log + 3, 14, k, order)
    return PlanParams(seed_table_depth=d, seed_depth=half_log + 3, seed_cap=64, gap=half_log + 9, chunk=32, list_cap=13,
                      bail_x16=50, recovery_lines=int(n_sets >= (24 << 20)) if recovery_lines is None else int(recovery_lines),
                      depth_table=order,
                      depth_anchors=int(shipped_depth_table_anchors(k, n_sets, order) if depth_anchors is None else depth_anchors))


def _plan_model(self, cover, params, concat, offsets, n_threads=1):
    """ora_plan_model -> (MS bytes the modelled stage produces, counts dict).  cover = (text u8[n], pos u32[n], node_at u32[n])."""
    text, pos, node_at = (np.ascontiguousarray(cover[0], dtype=np.uint8), np.ascontiguousarray(cover[1], dtype=np.uint32),
                          np.ascontiguousarray(cover[2], dtype=np.uint32))
    concat = np.ascontiguousarray(concat, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    ms = np.zeros(len(concat), dtype=np.uint8)
    cn = PlanCounts()
    _chk(lib().ora_plan_model(self._h, text.ctypes.data, pos.ctypes.data, node_at.ctypes.data, C.byref(params),
                              concat.ctypes.data, offsets.ctypes.data, len(offsets) - 1, n_threads, ms.ctypes.data, C.byref(cn)))
    return ms, cn.as_dict()


Index.plan_model = _plan_model


def log_rm_max_cdf(t, alphabet_size, n_kmers):
    return float(lib().ora_log_rm_max_cdf(t, alphabet_size, n_kmers))


def random_match_threshold(k, n_kmers, alphabet_size, max_error_prob):
    return int(lib().ora_random_match_threshold(k, n_kmers, alphabet_size, max_error_prob))


def derandomize_ms_val(curr, nxt, threshold, k):
    return int(lib().ora_derandomize_ms_val(curr, nxt, threshold, k))


def derandomize_ms_vec(noisy, k, threshold):
    a = np.ascontiguousarray(noisy, dtype=np.uint64)
    out = np.zeros(len(a), dtype=np.int64)
    _chk(lib().ora_derandomize_ms_vec(a.ctypes.data, len(a), k, threshold, out.ctypes.data))
    return out


def translate_ms_val(curr, nxt, prev, threshold):
    a, b = C.c_uint32(), C.c_uint32()
    lib().ora_translate_ms_val(curr, nxt, prev, threshold, C.byref(a), C.byref(b))
    return chr(a.value), chr(b.value)


def translate_ms_vec(derand, k, threshold):
    a = np.ascontiguousarray(derand, dtype=np.int64)
    out = np.zeros(len(a), dtype=np.uint32)
    _chk(lib().ora_translate_ms_vec(a.ctypes.data, len(a), k, threshold, out.ctypes.data))
    return "".join(chr(v) for v in out)


def run_lengths_gapped(aln, max_gap_len=0):
    a = _bytes(aln)
    n = lib().ora_run_lengths_gapped(a.ctypes.data, len(a), max_gap_len, None, 0)
    buf = (RLE * max(n, 1))()
    lib().ora_run_lengths_gapped(a.ctypes.data, len(a), max_gap_len, buf, n)
    return [buf[i].as_tuple() for i in range(n)]


def run_lengths_batch(aln_concat, offsets, max_gap_len=0):
    """format::run_lengths_gapped of every alignment of a batch -> (u64 array (n_runs, 7), u64 offsets (n_seqs + 1))."""
    aln = np.ascontiguousarray(aln_concat, dtype=np.uint8)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(off) - 1
    ro = np.zeros(n + 1, dtype=np.uint64)
    cap = max(1024, 2 * n)
    while True:
        recs = np.zeros((cap, 7), dtype=np.uint64)
        tot = int(lib().ora_run_lengths_batch(aln.ctypes.data, off.ctypes.data, n, max_gap_len, recs.ctypes.data, cap,
                                              ro.ctypes.data))
        if tot <= cap:
            return recs[:tot], ro
        cap = tot


def relative_to_ref(ref_seq, aln):
    r, a = _bytes(ref_seq), _bytes(aln)
    out = np.zeros(len(a), dtype=np.uint8)
    lib().ora_relative_to_ref(r.ctypes.data, a.ctypes.data, len(a), out.ctypes.data)
    return out.tobytes()


def kmer_colex_ranks(seq, k, kmers, n_threads=1):
    """index_check.c: for each k-mer of `kmers` (an [m, k] uint8 array) the number of k-mer occurrences in `seq` that are
    colex-smaller and the number equal to it - counted off the text, no index involved -> (less uint64[m], equal uint64[m])"""
    s = _bytes(seq)
    km = np.ascontiguousarray(kmers, dtype=np.uint8).reshape(-1, k)
    less = np.zeros(len(km), dtype=np.uint64)
    eq = np.zeros(len(km), dtype=np.uint64)
    _chk(lib().ora_kmer_colex_ranks(s.ctypes.data, len(s), k, km.ctypes.data, len(km), n_threads, less.ctypes.data, eq.ctypes.data))
    return less, eq


def add_variants(translation, variant_buf, n):
    t = np.frombuffer(bytes(_bytes(translation)), dtype=np.uint8).copy()
    _chk(lib().ora_add_variants(t.ctypes.data, len(t), variant_buf, n))
    return t.tobytes()
