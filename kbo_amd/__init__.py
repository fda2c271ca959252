"""kbo_amd — MI355X-native k-bounded matching-statistics path of kbo.

Python host-side mirror of the reference crate's public API for this path
(tmaklin/kbo v0.5.1, src/lib.rs): same function names, argument meaning and error
behaviour (reference panics surface as KboError), implemented over the C ABI in
include/kbo_hip.h.  All matching-statistics / derandomize / translate compute runs in
hand-written gfx950 HIP kernels; there is no CPU fallback.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# The host batch pipeline keeps three streams per device (upload, kernels, download) next to whatever the caller uses; the HIP runtime
# maps a process's streams onto 4 hardware queues unless told otherwise, and streams that share a queue do not overlap
# (INTEGRATION.md "Streams and hardware queues").  More queues are the APPLICATION's to ask for - GPU_MAX_HW_QUEUES=8 in its environment
# before the runtime starts, as bench.py and the tools do - or, for a process that wants this package to do it, KBO_HW_QUEUES=8 (opt-in:
# importing a library does not change the runtime's settings for every other HIP user of the process).
if os.environ.get("KBO_HW_QUEUES"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ["KBO_HW_QUEUES"])

from . import _capi, derandomize, format, gap_filling, index, translate, variant_calling  # noqa: F401
from ._capi import KboError, check, lib  # noqa: F401
from .index import LcsArray, SbwtIndexVariant, _u8  # noqa: F401


@dataclass
class BuildOpts:
    """kbo::BuildOpts (lib.rs:259-313)"""
    k: int = 31
    add_revcomp: bool = False
    num_threads: int = 1
    prefix_precalc: int = 8
    build_select: bool = False
    mem_gb: int = 4
    dedup_batches: bool = False
    temp_dir: Optional[str] = None

    def _to_c(self):
        return _capi.BuildOpts(self.k, int(self.add_revcomp), self.num_threads, self.prefix_precalc,
                               int(self.build_select), self.mem_gb, int(self.dedup_batches),
                               self.temp_dir.encode() if self.temp_dir else None)


@dataclass
class CallOpts:
    """kbo::CallOpts (lib.rs:318-353)"""
    max_error_prob: float = 0.0000001
    sbwt_build_opts: BuildOpts = field(default_factory=lambda: BuildOpts(build_select=True))


@dataclass
class FindOpts:
    """kbo::FindOpts (lib.rs:358-382)"""
    max_error_prob: float = 0.0000001
    max_gap_len: int = 0


@dataclass
class MatchOpts:
    """kbo::MatchOpts (lib.rs:387-407)"""
    max_error_prob: float = 0.0000001


@dataclass
class MapOpts:
    """kbo::MapOpts (lib.rs:412-466)"""
    max_error_prob: float = 0.0000001
    fill_gaps: bool = True
    call_variants: bool = True
    format: bool = True
    sbwt_build_opts: BuildOpts = field(default_factory=lambda: BuildOpts(build_select=True))


def build(seq_data, build_opts=None):
    """kbo::build (lib.rs:501-506) -> (SbwtIndexVariant, LcsArray)"""
    return index.build_sbwt_from_vecs(seq_data, build_opts if build_opts is not None else BuildOpts())


def matches(query_seq, sbwt, lcs=None, match_opts=None):
    """kbo::matches (lib.rs:612-628) -> list of chars"""
    o = match_opts if match_opts is not None else MatchOpts()
    q = _u8(query_seq)
    out = np.zeros(max(len(q), 1), dtype=np.uint32)
    check(lib().kbo_matches(sbwt._h, q.ctypes.data, len(q), o.max_error_prob, out.ctypes.data))
    return [chr(v) for v in out[:len(q)]]


def map(ref_seq, query_sbwt, query_lcs=None, map_opts=None):  # noqa: A001 (reference name)
    """kbo::map (lib.rs:720-761) -> bytes"""
    o = map_opts if map_opts is not None else MapOpts()
    co = _capi.MapOpts(o.max_error_prob, int(o.fill_gaps), int(o.call_variants), int(o.format),
                       o.sbwt_build_opts._to_c())
    r = _u8(ref_seq)
    out = np.zeros(max(len(r), 1), dtype=np.uint8)
    check(lib().kbo_map(query_sbwt._h, r.ctypes.data, len(r), C.byref(co), out.ctypes.data))
    return out[:len(r)].tobytes()


def find(query_seq, sbwt, lcs=None, find_opts=None):
    """kbo::find (lib.rs:808-821) -> list of format.RLE"""
    o = find_opts if find_opts is not None else FindOpts()
    co = _capi.FindOpts(o.max_error_prob, o.max_gap_len)
    q = _u8(query_seq)
    p, n = C.POINTER(_capi.RLE)(), C.c_size_t()
    check(lib().kbo_find(sbwt._h, q.ctypes.data, len(q), C.byref(co), C.byref(p), C.byref(n)))
    return format._take_rles(p, n.value)


def call(sbwt_query, lcs_query, ref_seq, call_opts=None):
    """kbo::call (lib.rs:547-573) -> list of variant_calling.Variant"""
    o = call_opts if call_opts is not None else CallOpts()
    co = _capi.CallOpts(o.max_error_prob, o.sbwt_build_opts._to_c())
    r = _u8(ref_seq)
    p, n = C.POINTER(_capi.Variant)(), C.c_size_t()
    check(lib().kbo_call(sbwt_query._h, r.ctypes.data, len(r), C.byref(co), C.byref(p), C.byref(n)))
    return variant_calling._from_c(p, n.value)
