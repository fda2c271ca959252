"""Mirror of kbo::variant_calling (reference src/variant_calling.rs) over the C ABI."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List

from . import _capi
from ._capi import check, lib


@dataclass
class Variant:
    """variant_calling::Variant (variant_calling.rs:8-26)"""
    query_pos: int = 0
    query_chars: List[int] = field(default_factory=list)
    ref_chars: List[int] = field(default_factory=list)


def _from_c(ptr, n):
    out = [Variant(int(ptr[i].query_pos), [ptr[i].query_chars[j] for j in range(ptr[i].query_len)],
                   [ptr[i].ref_chars[j] for j in range(ptr[i].ref_len)]) for i in range(n)]
    lib().kbo_free(ptr)
    return out


def _to_c(variants):
    """-> (array of _capi.Variant, keep-alive list)"""
    arr = (_capi.Variant * max(1, len(variants)))()
    keep = []
    for i, v in enumerate(variants):
        q = (C.c_uint8 * max(1, len(v.query_chars)))(*v.query_chars)
        r = (C.c_uint8 * max(1, len(v.ref_chars)))(*v.ref_chars)
        keep += [q, r]
        arr[i].query_pos = v.query_pos
        arr[i].query_chars = C.cast(q, C.POINTER(C.c_uint8))
        arr[i].query_len = len(v.query_chars)
        arr[i].ref_chars = C.cast(r, C.POINTER(C.c_uint8))
        arr[i].ref_len = len(v.ref_chars)
    return arr, keep
