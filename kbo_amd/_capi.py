"""ctypes loader for kbo_amd/libkbo_hip.so (the C ABI in include/kbo_hip.h).

The product path has no CPU fallback: if the shared library (and, for compute
calls, a gfx950 device) is missing, calls fail loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KBO_HIP_LIB") or os.path.join(_HERE, "libkbo_hip.so")  # KBO_HIP_LIB: counters build (tools/)

KBO_OK = 0
ERROR_NAMES = {
    -1: "KBO_E_EMPTY_QUERY", -2: "KBO_E_LEN_LE_2", -3: "KBO_E_THRESHOLD_LE_1", -4: "KBO_E_BAD_ARG",
    -5: "KBO_E_NOMEM", -6: "KBO_E_K_MISMATCH", -7: "KBO_E_HIP", -8: "KBO_E_UNSUPPORTED",
    -9: "KBO_E_MS_RANGE", -10: "KBO_E_IO", -11: "KBO_E_REF_PANIC",
}


class KboError(AssertionError):
    """A reference assert!/panic! site (or a HIP failure) reported through the C ABI."""

    def __init__(self, code, message):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


class BuildOpts(C.Structure):
    """kbo::BuildOpts (lib.rs:259-313)."""
    _fields_ = [("k", C.c_uint32), ("add_revcomp", C.c_int32), ("num_threads", C.c_uint32),
                ("prefix_precalc", C.c_uint32), ("build_select", C.c_int32), ("mem_gb", C.c_uint32),
                ("dedup_batches", C.c_int32), ("temp_dir", C.c_char_p)]


class FindOpts(C.Structure):
    """kbo::FindOpts (lib.rs:358-382)."""
    _fields_ = [("max_error_prob", C.c_double), ("max_gap_len", C.c_size_t)]


class MapOpts(C.Structure):
    """kbo::MapOpts (lib.rs:412-466)."""
    _fields_ = [("max_error_prob", C.c_double), ("fill_gaps", C.c_int32), ("call_variants", C.c_int32),
                ("format", C.c_int32), ("sbwt_build_opts", BuildOpts)]


class CallOpts(C.Structure):
    """kbo::CallOpts (lib.rs:318-353)."""
    _fields_ = [("max_error_prob", C.c_double), ("sbwt_build_opts", BuildOpts)]


class Variant(C.Structure):
    """kbo::variant_calling::Variant (variant_calling.rs:8-26), C layout."""
    _fields_ = [("query_pos", C.c_uint64), ("query_chars", C.POINTER(C.c_uint8)), ("query_len", C.c_size_t),
                ("ref_chars", C.POINTER(C.c_uint8)), ("ref_len", C.c_size_t)]


class CallFlat(C.Structure):
    """kbo_call_flat (kbo_hip.h): the variants of a batch as flat arrays, one allocation"""
    _fields_ = [("n_variants", C.c_uint64), ("n_chars", C.c_uint64), ("query_pos", C.POINTER(C.c_uint32)),
                ("query_len", C.POINTER(C.c_uint16)), ("ref_len", C.POINTER(C.c_uint16)), ("chars", C.POINTER(C.c_uint8))]


OPT_INHERIT = -2147483648


class IndexOpts(C.Structure):
    """kbo_hip.h kbo_index_opts_t"""
    _fields_ = [("struct_size", C.c_uint32), ("plan", C.c_int32), ("depth_table", C.c_int32), ("depth_table_anchors", C.c_int32),
                ("slab_bytes", C.c_uint64), ("n_devices", C.c_int32), ("devices", C.c_int32 * 16)]


class DeviceLayout(C.Structure):  # kbo_device_layout
    _fields_ = [(n, C.c_uint64) for n in ("rank_bytes", "entry_bytes", "pair_bytes", "cover_bytes", "lines_bytes", "seed_bytes",
                                         "dtab_bytes", "anchor_bytes")] + \
               [(n, C.c_uint32) for n in ("entries_64bit", "seed_depth", "dtab_order", "dtab_grouped")] + \
               [(n, C.c_double) for n in ("layout_seconds", "upload_seconds", "cover_seconds", "lines_seconds", "seed_seconds",
                                          "dtab_seconds")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class RLE(C.Structure):
    """kbo::format::RLE (format.rs:18-33)."""
    _fields_ = [(n, C.c_uint64) for n in
                ("start", "end", "matches", "mismatches", "jumps", "gap_bases", "gap_opens")]

    def as_tuple(self):
        return tuple(int(getattr(self, n)) for n, _ in self._fields_)


# every symbol include/kbo_hip.h declares (checked by tests/test_capi_host.py) ...
SYMBOLS = [
    "kbo_last_error", "kbo_version", "kbo_build_opts_default", "kbo_find_opts_default",
    "kbo_map_opts_default", "kbo_call_opts_default", "kbo_call", "kbo_add_variants", "kbo_fill_gaps",
    "kbo_nearest_unique_context", "kbo_index_build", "kbo_index_from_parts", "kbo_index_export_parts",
    "kbo_index_free", "kbo_index_k", "kbo_index_n_kmers", "kbo_index_n_sets", "kbo_index_save",
    "kbo_index_load", "kbo_index_to_device", "kbo_index_device_bytes", "kbo_log_rm_max_cdf",
    "kbo_random_match_threshold", "kbo_matching_statistics", "kbo_derandomize_ms_vec",
    "kbo_derandomize_ms_val", "kbo_translate_ms_vec", "kbo_translate_ms_val", "kbo_matches",
    "kbo_map", "kbo_find", "kbo_run_lengths_gapped", "kbo_relative_to_ref", "kbo_free",
    "kbo_ms_batch", "kbo_matches_batch", "kbo_map_batch", "kbo_find_batch", "kbo_work_bytes", "kbo_ms_work_bytes",
    "kbo_ms_batch_dev", "kbo_derand_translate_dev", "kbo_set_slab_bytes", "kbo_set_devices", "kbo_set_host_threads",
    "kbo_release_scratch", "kbo_run_lengths_gapped_batch", "kbo_find_batch_into", "kbo_derand_work_bytes",
    "kbo_run_lengths_work_bytes", "kbo_run_lengths_dev", "kbo_index_device_pair_bytes", "kbo_index_device_plan_bytes",
    "kbo_index_path_cover", "kbo_index_recovery_lines", "kbo_call_batch", "kbo_call_batch_flat", "kbo_call_flat_free", "kbo_stream_pair_create", "kbo_stream_pair_destroy", "kbo_call_sites_dev", "kbo_call_walk_dev",
    "kbo_index_save_sbwt", "kbo_index_load_sbwt", "kbo_packed_words", "kbo_pack_reads", "kbo_unpack_matches",
    "kbo_matches_batch_packed", "kbo_find_batch_packed", "kbo_index_shards", "kbo_index_work_bytes",
    "kbo_index_device_layout", "kbo_map_batch_dev", "kbo_map_batch_dev_tail",
    "kbo_index_opts_default", "kbo_index_set_opts", "kbo_index_get_opts", "kbo_matches_packed_dev", "kbo_matches_packed_dev_scratch_bytes",
    "kbo_find_batch_dev", "kbo_map_stream_create", "kbo_map_stream_submit", "kbo_map_stream_wait", "kbo_map_stream_wait_on",
    "kbo_map_stream_sync", "kbo_map_stream_free",
]
# ... and include/kbo_hip_tuning.h (knobs, experiment switches, test hooks: not part of the drop-in boundary)
TUNING_SYMBOLS = [
    "kbo_walk_geometry", "kbo_set_walk_waves_per_cu", "kbo_set_walk_threads", "kbo_set_walk_rare", "kbo_set_guided_walk",
    "kbo_set_pair_steps", "kbo_set_force_big_layout", "kbo_set_seed_table_depth", "kbo_set_plan", "kbo_set_plan_tuning",
    "kbo_set_plan_unit_cap_divisor", "kbo_index_plan_holdoff", "kbo_set_walk_experiment", "kbo_plan_stats_dev", "kbo_set_plan_stats", "kbo_set_index_shards", "kbo_index_shard", "kbo_set_depth_table", "kbo_set_depth_table_anchors", "kbo_index_depth_table", "kbo_run_automaton_depths",
    "kbo_set_stage_timing", "kbo_stage_timing_read", "kbo_set_plan_table_budget", "kbo_set_plan_lazy", "kbo_plan_flags_dev", "kbo_long_stats_dev", "kbo_set_map_long", "kbo_set_ms_one_kernel", "kbo_set_call_device_emit", "kbo_index_layout_check", "kbo_index_cover_check",
    "kbo_set_host_in_place",
]

_lib = None


def lib():
    """Load libkbo_hip.so (raises if it has not been built: run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch bundles its own libamdhip64.so.7; the dynamic loader keeps whichever copy
    # of that SONAME arrives first for the whole process, and torch cannot initialise the
    # GPU on top of /opt/rocm's copy.  When torch is installed, let it load its runtime
    # first; the extension then binds to the same one.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                           "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, sz, u64, u32, i64, dbl = C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int64, C.c_double
    L.kbo_last_error.restype = C.c_char_p
    L.kbo_version.restype = C.c_char_p
    L.kbo_build_opts_default.argtypes = [C.POINTER(BuildOpts)]
    L.kbo_find_opts_default.argtypes = [C.POINTER(FindOpts)]
    L.kbo_map_opts_default.argtypes = [C.POINTER(MapOpts)]
    L.kbo_call_opts_default.argtypes = [C.POINTER(CallOpts)]
    L.kbo_call.argtypes = [vp, vp, sz, C.POINTER(CallOpts), C.POINTER(C.POINTER(Variant)), C.POINTER(sz)]
    L.kbo_add_variants.argtypes = [vp, sz, C.POINTER(Variant), sz]
    L.kbo_fill_gaps.argtypes = [vp, vp, sz, sz, dbl, vp]
    L.kbo_nearest_unique_context.argtypes = [vp, vp, sz, sz, sz, C.POINTER(sz), vp, C.POINTER(sz)]
    for f in (L.kbo_build_opts_default, L.kbo_find_opts_default, L.kbo_map_opts_default, L.kbo_call_opts_default,
              L.kbo_index_free, L.kbo_free):
        f.restype = None
    L.kbo_index_build.argtypes = [C.POINTER(C.c_char_p), C.POINTER(sz), sz, C.POINTER(BuildOpts), C.POINTER(vp)]
    L.kbo_index_from_parts.argtypes = [u32, u64, u64, C.POINTER(vp), C.POINTER(u64), vp, C.POINTER(vp)]
    L.kbo_index_export_parts.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), vp]
    L.kbo_index_free.argtypes = [vp]
    L.kbo_index_k.argtypes = [vp]; L.kbo_index_k.restype = sz
    L.kbo_index_n_kmers.argtypes = [vp]; L.kbo_index_n_kmers.restype = u64
    L.kbo_index_n_sets.argtypes = [vp]; L.kbo_index_n_sets.restype = u64
    L.kbo_index_save.argtypes = [vp, C.c_char_p]
    L.kbo_index_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.kbo_index_save_sbwt.argtypes = [vp, C.c_char_p]
    L.kbo_index_load_sbwt.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.kbo_index_to_device.argtypes = [vp, C.c_int]
    L.kbo_index_device_bytes.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.kbo_log_rm_max_cdf.argtypes = [sz, sz, sz, C.POINTER(dbl)]
    L.kbo_random_match_threshold.argtypes = [sz, sz, sz, dbl, C.POINTER(sz)]
    L.kbo_matching_statistics.argtypes = [vp, vp, sz, vp, vp, vp]
    L.kbo_derandomize_ms_vec.argtypes = [vp, sz, sz, sz, vp]
    L.kbo_derandomize_ms_val.argtypes = [sz, i64, sz, sz, C.POINTER(i64)]
    L.kbo_translate_ms_vec.argtypes = [vp, sz, sz, sz, vp]
    L.kbo_translate_ms_val.argtypes = [i64, i64, i64, sz, C.POINTER(u32), C.POINTER(u32)]
    L.kbo_matches.argtypes = [vp, vp, sz, dbl, vp]
    L.kbo_map.argtypes = [vp, vp, sz, C.POINTER(MapOpts), vp]
    L.kbo_find.argtypes = [vp, vp, sz, C.POINTER(FindOpts), C.POINTER(C.POINTER(RLE)), C.POINTER(sz)]
    L.kbo_run_lengths_gapped.argtypes = [vp, sz, sz, C.POINTER(C.POINTER(RLE)), C.POINTER(sz)]
    L.kbo_relative_to_ref.argtypes = [vp, vp, sz, vp]
    L.kbo_free.argtypes = [vp]
    L.kbo_ms_batch.argtypes = [vp, vp, vp, sz, vp, vp, vp]
    L.kbo_matches_batch.argtypes = [vp, vp, vp, sz, dbl, vp]
    L.kbo_map_batch.argtypes = [vp, vp, vp, sz, dbl, C.c_int, vp]
    L.kbo_find_batch.argtypes = [vp, vp, vp, sz, C.POINTER(FindOpts), C.POINTER(C.POINTER(RLE)), vp]
    L.kbo_work_bytes.argtypes = [sz, u64, sz, C.c_uint32]; L.kbo_work_bytes.restype = sz
    L.kbo_ms_work_bytes.argtypes = [sz, u64, sz, C.c_uint32]; L.kbo_ms_work_bytes.restype = sz
    L.kbo_ms_batch_dev.argtypes = [vp, vp, vp, sz, u64, sz, vp, vp, vp, vp, sz, vp]
    L.kbo_derand_translate_dev.argtypes = [vp, vp, sz, u64, sz, sz, vp, vp, sz, vp, sz, vp]
    L.kbo_derand_work_bytes.argtypes = [sz, u64]; L.kbo_derand_work_bytes.restype = sz
    L.kbo_walk_geometry.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.kbo_set_walk_waves_per_cu.argtypes = [C.c_int]
    L.kbo_set_walk_threads.argtypes = [C.c_int]
    L.kbo_set_guided_walk.argtypes = [C.c_int, C.c_int]
    L.kbo_set_walk_rare.argtypes = [C.c_int]
    L.kbo_set_slab_bytes.argtypes = [sz]
    L.kbo_set_force_big_layout.argtypes = [C.c_int]
    L.kbo_set_host_in_place.argtypes = [C.c_int]
    L.kbo_set_host_threads.argtypes = [C.c_int]
    L.kbo_release_scratch.argtypes = []
    L.kbo_set_pair_steps.argtypes = [C.c_uint64, C.c_int]
    L.kbo_run_lengths_gapped_batch.argtypes = [vp, vp, sz, sz, vp, vp]
    L.kbo_find_batch_into.argtypes = [vp, vp, vp, sz, vp, vp, sz, vp, vp]
    L.kbo_run_lengths_work_bytes.argtypes = [sz]; L.kbo_run_lengths_work_bytes.restype = sz
    L.kbo_run_lengths_dev.argtypes = [vp, vp, sz, sz, sz, vp, vp, sz, vp]
    L.kbo_index_device_pair_bytes.argtypes = [vp]
    L.kbo_index_device_pair_bytes.restype = C.c_uint64
    L.kbo_set_devices.argtypes = [C.POINTER(C.c_int), C.c_int]
    L.kbo_set_plan.argtypes = [C.c_int, C.c_int, C.c_int]
    L.kbo_set_plan_tuning.argtypes = [C.c_int, C.c_int, C.c_int]
    L.kbo_set_walk_experiment.argtypes = [C.c_int, C.c_int]
    L.kbo_plan_stats_dev.argtypes = [sz, u64, sz, C.c_uint32, vp, vp, vp]
    L.kbo_run_automaton_depths.argtypes = [vp, sz, C.c_uint32, C.c_int, vp, sz, vp]
    L.kbo_packed_words.argtypes = [vp, sz]; L.kbo_packed_words.restype = sz
    L.kbo_pack_reads.argtypes = [vp, vp, sz, vp, vp, vp, sz, C.POINTER(sz)]
    L.kbo_unpack_matches.argtypes = [vp, vp, sz, vp]
    L.kbo_matches_batch_packed.argtypes = [vp, vp, vp, sz, vp, vp, sz, dbl, vp]
    L.kbo_find_batch_packed.argtypes = [vp, vp, vp, sz, vp, vp, sz, C.POINTER(FindOpts), C.POINTER(vp), vp]
    L.kbo_set_plan_stats.argtypes = [C.c_int]
    L.kbo_set_index_shards.argtypes = [C.c_int]
    L.kbo_set_depth_table.argtypes = [C.c_int]
    L.kbo_set_depth_table_anchors.argtypes = [C.c_int]
    L.kbo_index_depth_table.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.kbo_index_shards.argtypes = [vp]
    L.kbo_index_shard.argtypes = [vp, C.c_int]; L.kbo_index_shard.restype = vp
    L.kbo_index_work_bytes.argtypes = [vp, sz, u64, sz]; L.kbo_index_work_bytes.restype = sz
    L.kbo_set_plan_unit_cap_divisor.argtypes = [C.c_int]
    L.kbo_set_seed_table_depth.argtypes = [C.c_int]
    L.kbo_index_plan_holdoff.argtypes = [vp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.kbo_index_path_cover.argtypes = [vp, vp, vp, vp]
    L.kbo_index_recovery_lines.argtypes = [vp, vp, C.POINTER(C.c_size_t)]
    L.kbo_call_batch.argtypes = [vp, vp, vp, sz, C.POINTER(CallOpts), C.POINTER(C.POINTER(Variant)), vp]
    L.kbo_call_batch_flat.argtypes = [vp, vp, vp, sz, C.POINTER(CallOpts), C.POINTER(CallFlat), vp]
    L.kbo_call_flat_free.argtypes = [C.POINTER(CallFlat)]; L.kbo_call_flat_free.restype = None
    L.kbo_stream_pair_create.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp)]
    L.kbo_stream_pair_destroy.argtypes = [vp, vp]; L.kbo_stream_pair_destroy.restype = None
    L.kbo_call_sites_dev.argtypes = [vp, vp, vp, vp, sz, u64, sz, sz, vp, sz, vp, vp]
    L.kbo_call_walk_dev.argtypes = [vp, vp, vp, sz, u64, sz, sz, vp, vp, sz, vp, vp, sz, vp]
    L.kbo_index_device_plan_bytes.argtypes = [vp]
    L.kbo_index_device_plan_bytes.restype = C.c_uint64
    L.kbo_plan_flags_dev.argtypes = [sz, u64, sz, C.c_uint32, vp, vp, vp]
    L.kbo_long_stats_dev.argtypes = [sz, u64, sz, C.c_uint32, vp, vp, vp]
    L.kbo_set_plan_table_budget.argtypes = [C.c_uint64]
    L.kbo_set_plan_lazy.argtypes = [C.c_int64]
    L.kbo_set_map_long.argtypes = [C.c_int]
    L.kbo_set_ms_one_kernel.argtypes = [C.c_int]
    L.kbo_set_call_device_emit.argtypes = [C.c_int]
    L.kbo_index_layout_check.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    L.kbo_index_cover_check.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.kbo_map_stream_create.argtypes = [vp, C.c_int, sz, u64, sz, vp]
    L.kbo_map_stream_submit.argtypes = [vp, vp, vp, sz, u64, sz, C.c_double, C.c_int, vp, vp, vp, vp, C.POINTER(C.c_int)]
    L.kbo_map_stream_wait.argtypes = [vp, u64]
    L.kbo_map_stream_wait_on.argtypes = [vp, u64, vp]
    L.kbo_map_stream_sync.argtypes = [vp]
    L.kbo_map_stream_free.argtypes = [vp]
    L.kbo_map_stream_free.restype = None
    L.kbo_set_stage_timing.argtypes = [C.c_int]
    L.kbo_stage_timing_read.argtypes = [C.POINTER(dbl), C.POINTER(dbl), C.POINTER(C.c_int)]
    L.kbo_map_batch_dev.argtypes = [vp, vp, vp, sz, u64, sz, dbl, C.c_int, C.c_int, vp, vp, vp, sz, vp, C.POINTER(C.c_int)]
    L.kbo_matches_packed_dev_scratch_bytes.argtypes = [sz, u64]
    L.kbo_matches_packed_dev_scratch_bytes.restype = sz
    L.kbo_matches_packed_dev.argtypes = [vp, vp, vp, sz, u64, sz, sz, vp, vp, sz, dbl, vp, vp, vp, sz, vp, vp]
    L.kbo_find_batch_dev.argtypes = [vp, vp, vp, sz, u64, sz, dbl, sz, vp, vp, vp, sz, vp, vp, sz, vp, vp, C.POINTER(C.c_int)]
    L.kbo_index_opts_default.argtypes = [C.POINTER(IndexOpts)]
    L.kbo_index_set_opts.argtypes = [vp, C.POINTER(IndexOpts)]
    L.kbo_index_get_opts.argtypes = [vp, C.POINTER(IndexOpts)]
    L.kbo_map_batch_dev_tail.argtypes = [vp, vp, vp, sz, u64, sz, dbl, C.c_int, C.c_int, vp, vp, vp, sz, vp, vp, C.POINTER(C.c_int)]
    L.kbo_index_device_layout.argtypes = [vp, C.c_int, C.POINTER(DeviceLayout)]
    _lib = L
    return L


def check(rc):
    if rc != KBO_OK:
        raise KboError(rc, lib().kbo_last_error().decode(errors="replace"))
