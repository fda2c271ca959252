"""Mirror of kbo::format (reference src/format.rs) over the C ABI."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._capi import RLE as _CRLE
from ._capi import check, lib


@dataclass(frozen=True)
class RLE:
    """format::RLE (format.rs:18-33)"""
    start: int = 0
    end: int = 0
    matches: int = 0
    mismatches: int = 0
    jumps: int = 0
    gap_bases: int = 0
    gap_opens: int = 0


def _aln_bytes(aln):
    if isinstance(aln, (bytes, bytearray)):
        return np.frombuffer(bytes(aln), dtype=np.uint8)
    if isinstance(aln, str):
        return np.frombuffer(aln.encode(), dtype=np.uint8)
    if isinstance(aln, np.ndarray):
        return np.ascontiguousarray(aln, dtype=np.uint8)
    return np.frombuffer("".join(aln).encode(), dtype=np.uint8)


def _take_rles(ptr, n):
    out = [RLE(*ptr[i].as_tuple()) for i in range(n)]
    lib().kbo_free(ptr)
    return out


def run_lengths_gapped(aln, max_gap_len):
    """format.rs:143-193"""
    a = _aln_bytes(aln)
    p, n = C.POINTER(_CRLE)(), C.c_size_t()
    check(lib().kbo_run_lengths_gapped(a.ctypes.data, len(a), max_gap_len, C.byref(p), C.byref(n)))
    return _take_rles(p, n.value)


def run_lengths_gapped_batch(aln_concat, offsets, max_gap_len):
    """format::run_lengths_gapped of many alignments at once on the GPU
    -> (numpy structured-like array [n_runs, 7] uint64, rle_offsets uint64[n+1])"""
    a = _aln_bytes(aln_concat)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(off) - 1
    ro = np.zeros(n + 1, dtype=np.uint64)
    p = C.POINTER(_CRLE)()
    check(lib().kbo_run_lengths_gapped_batch(a.ctypes.data, off.ctypes.data, n, max_gap_len, C.byref(p), ro.ctypes.data))
    total = int(ro[-1])
    runs = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(max(total, 1), 7))[:total].copy()
    lib().kbo_free(p)
    return runs, ro


def run_lengths(aln):
    """format.rs:98-102"""
    return run_lengths_gapped(aln, 0)


def relative_to_ref(ref_seq, alignment):
    """format.rs:266-287 -> bytes"""
    from .index import _u8
    r, a = _u8(ref_seq), _aln_bytes(alignment)
    n = min(len(r), len(a))
    out = np.zeros(max(n, 1), dtype=np.uint8)
    check(lib().kbo_relative_to_ref(r.ctypes.data, a.ctypes.data, n, out.ctypes.data))
    return out[:n].tobytes()
