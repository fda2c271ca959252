"""Deterministic synthetic genomes/reads for tests and bench.py."""
import ctypes as C

import numpy as np

from ._capi import lib

GENOME_SEED = 0x6B626F0001
READS_SEED = 0x6B626F0002


def genome(length, seed=GENOME_SEED):
    out = np.empty(length, dtype=np.uint8)
    f = lib().kbo_synth_genome
    f.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64]
    f.restype = None
    f(seed, out.ctypes.data, length)
    return out


def reads(genome_arr, n_reads, read_len=150, sub_rate=0.01, seed=READS_SEED, first_read=0):
    """-> (concat uint8 [n_reads*read_len], offsets uint64 [n_reads+1])"""
    g = np.ascontiguousarray(genome_arr, dtype=np.uint8)
    out = np.empty(n_reads * read_len, dtype=np.uint8)
    f = lib().kbo_synth_reads
    f.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                  C.c_void_p]
    f.restype = None
    f(seed, g.ctypes.data, len(g), first_read, n_reads, read_len, int(round(sub_rate * 65536)),
      out.ctypes.data)
    offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)
    return out, offsets
