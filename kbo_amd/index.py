"""Mirror of kbo::index (reference src/index.rs) over the C ABI."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib


def _u8(x):
    if isinstance(x, np.ndarray):
        return np.ascontiguousarray(x, dtype=np.uint8)
    if isinstance(x, str):
        x = x.encode()
    return np.frombuffer(bytes(x), dtype=np.uint8)


class SbwtIndexVariant:
    """Stand-in for sbwt::SbwtIndexVariant::SubsetMatrix (+ its LcsArray): owns a kbo_index_t."""

    def __init__(self, handle, owner=None):
        self._h = handle
        self._owner = owner  # (a borrowed handle - a shard - keeps its owner alive and is never freed)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and getattr(self, "_owner", None) is None:
            try:
                lib().kbo_index_free(h)
            except Exception:  # interpreter shutdown: module globals are already gone
                pass

    def k(self):
        return int(lib().kbo_index_k(self._h))

    def n_kmers(self):
        return int(lib().kbo_index_n_kmers(self._h))

    def shards(self):
        """1 for an ordinary index; the number of shards of an index whose rows would not fit 32-bit row numbers (kbo_hip.h)"""
        return int(lib().kbo_index_shards(self._h))

    def shard(self, i):
        """shard i as an (ordinary, borrowed) index: kbo_hip_tuning.h kbo_index_shard"""
        h = lib().kbo_index_shard(self._h, i)
        if not h:
            raise IndexError(i)
        return SbwtIndexVariant(C.c_void_p(h), owner=self)

    def n_sets(self):
        return int(lib().kbo_index_n_sets(self._h))

    def to_device(self, device=-1):
        check(lib().kbo_index_to_device(self._h, device))
        return self

    def device_bytes(self):
        a, b = C.c_uint64(), C.c_uint64()
        check(lib().kbo_index_device_bytes(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def device_pair_bytes(self):
        """Bytes of two-base extension blocks in the device copies of this index (0 = none)."""
        return int(lib().kbo_index_device_pair_bytes(self._h))

    def device_plan_bytes(self):
        """Bytes of path cover (plan-guided walk) in the device copies of this index (0 = none)."""
        return int(lib().kbo_index_device_plan_bytes(self._h))

    def device_layout(self, device=-1):
        """what the copy on `device` holds and what making it cost (kbo_hip.h kbo_device_layout) as a dict"""
        lay = _capi.DeviceLayout()
        check(lib().kbo_index_device_layout(self._h, device, C.byref(lay)))
        return lay.as_dict()

    def set_opts(self, plan=None, depth_table=None, depth_table_anchors=None, slab_bytes=None, devices=None):
        """options of this handle alone (kbo_hip.h kbo_index_opts_t); None = leave the field as it is.  To follow the process-wide
        setting again pass _capi.OPT_INHERIT (devices: -1)."""
        o = _capi.IndexOpts()
        check(lib().kbo_index_get_opts(self._h, C.byref(o)))
        if plan is not None:
            o.plan = int(plan)
        if depth_table is not None:
            o.depth_table = int(depth_table)
        if depth_table_anchors is not None:
            o.depth_table_anchors = int(depth_table_anchors)
        if slab_bytes is not None:
            o.slab_bytes = int(slab_bytes)
        if devices is not None:
            if devices == -1:
                o.n_devices = -1
            else:
                o.n_devices = len(devices)
                for i, d in enumerate(devices):
                    o.devices[i] = int(d)
        check(lib().kbo_index_set_opts(self._h, C.byref(o)))

    def get_opts(self):
        o = _capi.IndexOpts()
        check(lib().kbo_index_get_opts(self._h, C.byref(o)))
        return {"plan": o.plan, "depth_table": o.depth_table, "depth_table_anchors": o.depth_table_anchors, "slab_bytes": int(o.slab_bytes),
                "devices": None if o.n_devices < 0 else [int(o.devices[i]) for i in range(o.n_devices)]}

    def export_parts(self):
        """(rows[4] uint64 words, C[4], lcs bytes) — the abstract index content."""
        n = self.n_sets()
        nw = (n + 63) // 64
        rows = [np.zeros(nw, dtype=np.uint64) for _ in range(4)]
        ptrs = (C.c_void_p * 4)(*[r.ctypes.data for r in rows])
        Carr = (C.c_uint64 * 4)()
        lcs = np.zeros(n, dtype=np.uint8)
        check(lib().kbo_index_export_parts(self._h, ptrs, Carr, lcs.ctypes.data))
        return rows, [int(v) for v in Carr], lcs

    def path_cover(self):
        """(text uint8[n], pos uint32[n], node_at uint32[n]) — the path cover of the plan-guided walk (kbo_hip.h)."""
        n = self.n_sets()
        text, pos, node = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.uint32)
        check(lib().kbo_index_path_cover(self._h, text.ctypes.data, pos.ctypes.data, node.ctypes.data))
        return text, pos, node

    def depth_table(self, device=-1, view=0):
        """(uint8[4^order], order) - the depth table of the device copy (kbo_hip_tuning.h kbo_index_depth_table); (empty, 0): none"""
        nb, order = C.c_size_t(0), C.c_int(0)
        check(lib().kbo_index_depth_table(self._h, device, view, None, C.byref(nb), C.byref(order)))
        out = np.zeros(nb.value, dtype=np.uint8)
        if nb.value:
            check(lib().kbo_index_depth_table(self._h, device, view, out.ctypes.data, C.byref(nb), C.byref(order)))
        return out, int(order.value)

    def depth_table_order(self, device=-1):
        """order of the depth table of the device copy (0: none)"""
        nb, order = C.c_size_t(0), C.c_int(0)
        check(lib().kbo_index_depth_table(self._h, device, 0, None, C.byref(nb), C.byref(order)))
        return int(order.value)

    def recovery_lines(self):
        """uint8[n_lines, 128] - the recovery lines of the guided walk (kbo_hip.h: kbo_index_recovery_lines)."""
        nb = C.c_size_t(0)
        check(lib().kbo_index_recovery_lines(self._h, None, C.byref(nb)))
        out = np.zeros(nb.value, dtype=np.uint8)
        check(lib().kbo_index_recovery_lines(self._h, out.ctypes.data, C.byref(nb)))
        return out.reshape(-1, 128)

    @classmethod
    def from_parts(cls, k, n_sets, n_kmers, rows, Carr, lcs):
        rows = [np.ascontiguousarray(r, dtype=np.uint64) for r in rows]
        ptrs = (C.c_void_p * 4)(*[r.ctypes.data for r in rows])
        Cc = (C.c_uint64 * 4)(*[int(v) for v in Carr])
        lcs = np.ascontiguousarray(lcs, dtype=np.uint8)
        h = C.c_void_p()
        check(lib().kbo_index_from_parts(k, n_sets, n_kmers, ptrs, Cc, lcs.ctypes.data, C.byref(h)))
        return cls(h)


class LcsArray:
    """Stand-in for sbwt::LcsArray; the LCS bytes live inside the index handle."""

    def __init__(self, sbwt):
        self.sbwt = sbwt


def build_sbwt_from_vecs(slices, build_options=None):
    """index::build_sbwt_from_vecs (index.rs:56-99) -> (SbwtIndexVariant, LcsArray)."""
    from . import BuildOpts
    o = build_options if build_options is not None else BuildOpts()
    seqs = [bytes(_u8(s)) for s in slices]
    if not seqs:
        raise _capi.KboError(-4, "assert!(!slices.is_empty()) (index.rs:60)")
    arr = (C.c_char_p * len(seqs))(*seqs)
    lens = (C.c_size_t * len(seqs))(*[len(s) for s in seqs])
    co = o._to_c()
    h = C.c_void_p()
    check(lib().kbo_index_build(arr, lens, len(seqs), C.byref(co), C.byref(h)))
    sbwt = SbwtIndexVariant(h)
    return sbwt, LcsArray(sbwt)


def serialize_sbwt(outfile_prefix, sbwt, lcs=None):
    """index::serialize_sbwt (index.rs:128-151).  NOT interchangeable with the reference's files: the sbwt crate's payload
    is unpinned here, so the pair is written as `<prefix>.sbwt.kbohip` + `<prefix>.lcs.kbohip` (pinned header, own payload:
    kbo_hip.h kbo_index_save_sbwt) and kbo-cli cannot read it; indexes cross the boundary through from_parts."""
    check(lib().kbo_index_save_sbwt(sbwt._h, outfile_prefix.encode()))


def load_sbwt(index_prefix):
    """index::load_sbwt (index.rs:195-212) for a pair serialize_sbwt wrote; a crate-written `<prefix>.sbwt` raises
    KboError(KBO_E_UNSUPPORTED) instead of being guessed at."""
    h = C.c_void_p()
    check(lib().kbo_index_load_sbwt(index_prefix.encode(), C.byref(h)))
    sbwt = SbwtIndexVariant(h)
    return sbwt, LcsArray(sbwt)


def save_flat(path, sbwt):
    """the single-file cache format (`kbo_index_save`)"""
    check(lib().kbo_index_save(sbwt._h, path.encode()))


def load_flat(path):
    h = C.c_void_p()
    check(lib().kbo_index_load(path.encode(), C.byref(h)))
    sbwt = SbwtIndexVariant(h)
    return sbwt, LcsArray(sbwt)


def query_sbwt(query, sbwt, lcs=None):
    """index::query_sbwt (index.rs:243-256) -> list of (d, range(l, r)) like Vec<(usize, Range<usize>)>."""
    q = _u8(query)
    n = len(q)
    d = np.zeros(max(n, 1), dtype=np.uint64)
    lo = np.zeros(max(n, 1), dtype=np.uint64)
    hi = np.zeros(max(n, 1), dtype=np.uint64)
    check(lib().kbo_matching_statistics(sbwt._h, q.ctypes.data, n, d.ctypes.data, lo.ctypes.data,
                                        hi.ctypes.data))
    return [(int(d[i]), range(int(lo[i]), int(hi[i]))) for i in range(n)]
