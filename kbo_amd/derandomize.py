"""Mirror of kbo::derandomize (reference src/derandomize.rs) over the C ABI."""
import ctypes as C

import numpy as np

from ._capi import check, lib


def log_rm_max_cdf(t, alphabet_size, n_kmers):
    """derandomize.rs:91-100"""
    out = C.c_double()
    check(lib().kbo_log_rm_max_cdf(t, alphabet_size, n_kmers, C.byref(out)))
    return out.value


def random_match_threshold(k, n_kmers, alphabet_size, max_error_prob):
    """derandomize.rs:127-145"""
    out = C.c_size_t()
    check(lib().kbo_random_match_threshold(k, n_kmers, alphabet_size, max_error_prob, C.byref(out)))
    return int(out.value)


def derandomize_ms_val(curr_noisy_ms, next_derand_ms, threshold, k):
    """derandomize.rs:221-247"""
    out = C.c_int64()
    check(lib().kbo_derandomize_ms_val(curr_noisy_ms, next_derand_ms, threshold, k, C.byref(out)))
    return int(out.value)


def derandomize_ms_vec(noisy_ms, k, threshold):
    """derandomize.rs:269-288 (runs the right-to-left recurrence on the GPU)."""
    a = np.ascontiguousarray(noisy_ms, dtype=np.uint64)
    out = np.zeros(max(len(a), 1), dtype=np.int64)
    check(lib().kbo_derandomize_ms_vec(a.ctypes.data, len(a), k, threshold, out.ctypes.data))
    return [int(v) for v in out[:len(a)]]
