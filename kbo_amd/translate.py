"""Mirror of kbo::translate (reference src/translate.rs) over the C ABI."""
import ctypes as C

import numpy as np

from ._capi import check, lib


def translate_ms_val(ms_curr, ms_next, ms_prev, threshold):
    """translate.rs:180-216 -> (char, char)"""
    a, b = C.c_uint32(), C.c_uint32()
    check(lib().kbo_translate_ms_val(ms_curr, ms_next, ms_prev, threshold, C.byref(a), C.byref(b)))
    return chr(a.value), chr(b.value)


def translate_ms_vec(derand_ms, k, threshold):
    """translate.rs:263-293 -> list of chars (stencil kernel on the GPU)."""
    a = np.ascontiguousarray(derand_ms, dtype=np.int64)
    out = np.zeros(max(len(a), 1), dtype=np.uint32)
    check(lib().kbo_translate_ms_vec(a.ctypes.data, len(a), k, threshold, out.ctypes.data))
    return [chr(v) for v in out[:len(a)]]


def add_variants(translation, variants):
    """translate.rs:350-386 -> list of chars"""
    from .variant_calling import _to_c
    t = np.array([ord(c) for c in translation], dtype=np.uint32)
    arr, keep = _to_c(variants)
    check(lib().kbo_add_variants(t.ctypes.data, len(t), arr, len(variants)))
    del keep
    return [chr(v) for v in t]
