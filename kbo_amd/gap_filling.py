"""Mirror of kbo::gap_filling (reference src/gap_filling.rs) over the C ABI."""
import ctypes as C

import numpy as np

from ._capi import check, lib
from .index import _u8


def fill_gaps_from_sequences(ref_seq, query_sbwt, threshold, max_err_prob):
    """The steps around gap_filling::fill_gaps as the reference\'s tests run them
    (gap_filling.rs:419-441): query_sbwt -> derandomize_ms_vec -> translate_ms_vec with the
    given threshold (GPU), then fill_gaps (gap_filling.rs:444-526).  -> list of chars"""
    r = _u8(ref_seq)
    out = np.zeros(max(len(r), 1), dtype=np.uint32)
    check(lib().kbo_fill_gaps(query_sbwt._h, r.ctypes.data, len(r), threshold, max_err_prob, out.ctypes.data))
    return [chr(v) for v in out[:len(r)]]


def nearest_unique_context(ref_seq, sbwt, search_range):
    """gap_filling::nearest_unique_context (gap_filling.rs:127-151) on the MS of ref_seq."""
    r = _u8(ref_seq)
    k = sbwt.k()
    km = np.zeros(max(k, 1), dtype=np.uint8)
    idx, n = C.c_size_t(), C.c_size_t()
    check(lib().kbo_nearest_unique_context(sbwt._h, r.ctypes.data, len(r), search_range.start, search_range.stop,
                                           C.byref(idx), km.ctypes.data, C.byref(n)))
    return int(idx.value), km[:n.value].tobytes()
