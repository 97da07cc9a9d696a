"""ctypes binding of include/searchlite_gpu.h.

There is no fallback: if libsearchlite_gpu.so is missing the import of anything that needs
it raises, and every call either runs the HIP kernels or raises SlgError.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

NO_TERM = 0xFFFFFFFF
NO_VECTOR = 0xFFFFFFFF
MAX_QUERY_TERMS = 32
MAX_K = 20001
MAX_RERANK_K = 1024
MAX_VECTOR_CLAUSES = 8
SHARD_UNIQUE_ID_BYTES = 128

OK, ERR_INVALID, ERR_DEVICE, ERR_OOM, ERR_UNSUPPORTED, ERR_INTERNAL = 0, -1, -2, -3, -4, -5
STRATEGY_BM25, STRATEGY_WAND, STRATEGY_BMW = 0, 1, 2
METRIC_COSINE, METRIC_L2 = 0, 1
PLAN_SUM, PLAN_DISMAX, PLAN_LEAF = 0, 1, 2


class SlgError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"searchlite_gpu error {code}: {msg}")
        self.code = code
        self.msg = msg


class SegmentDesc(C.Structure):
    _fields_ = [("n_docs", C.c_uint32), ("n_terms", C.c_uint32),
                ("term_offsets", C.c_void_p), ("doc_ids", C.c_void_p), ("tfs", C.c_void_p),
                ("term_field", C.c_void_p), ("n_fields", C.c_uint32),
                ("field_doc_len", C.c_void_p), ("field_avgdl", C.c_void_p),
                ("docs", C.c_float), ("k1", C.c_float), ("b", C.c_float),
                ("deleted", C.c_void_p),
                ("vec_dim", C.c_uint32), ("vec_metric", C.c_int32),
                ("vec_offsets", C.c_void_p), ("vec_values", C.c_void_p),
                ("vec_rows", C.c_uint32)]


class VectorFieldDesc(C.Structure):
    _fields_ = [("vec_dim", C.c_uint32), ("vec_metric", C.c_int32), ("vec_offsets", C.c_void_p),
                ("vec_values", C.c_void_p), ("vec_rows", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("scored_docs", C.c_uint64), ("candidates_examined", C.c_uint64),
                ("postings_advanced", C.c_uint64)]


class Tuning(C.Structure):
    """slg_tuning: planner knobs fixed per index (include/searchlite_gpu.h)."""
    _fields_ = [("struct_size", C.c_uint32), ("validate", C.c_int32), ("champions", C.c_int32),
                ("allow_any_arch", C.c_int32), ("pruning", C.c_int32),
                ("uniform_max_terms", C.c_uint32), ("uniform_round_target", C.c_uint32),
                ("multi_round_target", C.c_uint32), ("probe_target", C.c_uint32),
                ("rounds_per_slice", C.c_uint32), ("max_rounds_per_slice", C.c_uint32),
                ("slices_per_subquery", C.c_uint32), ("cand_mode", C.c_int32),
                ("slice_order", C.c_int32), ("block_max", C.c_int32), ("pool_cap_mb", C.c_uint32),
                ("uniform_kernel", C.c_uint32), ("uniform_sigma_x100", C.c_uint32),
                ("inline_cuts", C.c_int32), ("updatable", C.c_int32),
                ("uniform_plans", C.c_int32), ("score_waves_per_simd", C.c_uint32)]


class ScorePlans(C.Structure):
    """slg_score_plans: flat (leaf_group NULL) or two-level score plans."""
    _fields_ = [("q_leaf", C.c_void_p), ("q_plan", C.c_void_p), ("q_tie", C.c_void_p), ("q_nleaves", C.c_void_p),
                ("q_leaf_offsets", C.c_void_p), ("leaf_group", C.c_void_p), ("q_group_offsets", C.c_void_p),
                ("group_plan", C.c_void_p), ("group_tie", C.c_void_p),
                ("q_node_offsets", C.c_void_p), ("node_kind", C.c_void_p), ("node_tie", C.c_void_p),
                ("node_parent", C.c_void_p), ("q_min_match", C.c_void_p)]


class Ticket(C.Structure):
    """slg_ticket (slg_coalescer_submit / _wait)."""
    _fields_ = [("batch", C.c_void_p), ("row", C.c_uint32), ("k", C.c_uint32), ("kind", C.c_uint32)]


class Query(C.Structure):
    _fields_ = [("n_terms", C.c_uint32), ("term_ids", C.c_void_p), ("weights", C.c_void_p)]


_lib = None


def lib_path() -> str:
    """The product library; SLG_LIB_TAG=<tag> selects an experiment build
    libsearchlite_gpu_<tag>.so (tools/build_variant.sh, A/B timing on one box)."""
    tag = os.environ.get("SLG_LIB_TAG")
    if tag:
        return os.path.join(_build.LIBDIR, f"libsearchlite_gpu_{tag}.so")
    return _build.GPU_LIB


def load():
    """Load libsearchlite_gpu.so (built by searchlite_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "searchlite_amd has no CPU fallback.")
    if os.environ.get("SLG_NO_TORCH_PRELOAD", "0") == "0":
        # PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
        # process do not share the GPU (the second to initialise reports "no HIP GPUs"), so
        # when torch is installed load it first: libsearchlite_gpu.so then binds to the HIP
        # runtime that is already resident (same SONAME).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(path)
    vp, u32, i32, f32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float
    sigs = {
        "slg_abi_version": (u32, []),
        "slg_last_error": (C.c_char_p, []),
        "slg_last_error_code": (i32, []),
        "slg_tuning_default": (None, [vp]),
        "slg_index_create_tuned": (vp, [vp, u32, i32, vp]),
        "slg_index_get_tuning": (i32, [vp, vp]),
        "slg_device_count": (i32, []),
        "slg_index_create": (vp, [vp, u32, i32]),
        "slg_index_destroy": (None, [vp]),
        "slg_index_info": (i32, [vp, vp, vp, vp]),
        "slg_index_trim_pool": (i32, [vp, vp]),
        "slg_index_set_stream": (i32, [vp, vp]),
        "slg_index_update_deleted": (i32, [vp, u32, vp, f32]),
        "slg_index_add_segment": (i32, [vp, vp]),
        "slg_index_remove_segment": (i32, [vp, u32]),
        "slg_index_generation": (C.c_uint64, [vp]),
        "slg_index_device": (i32, [vp]),
        "slg_coalescer_create": (vp, [vp, u32, u32]),
        "slg_coalescer_destroy": (None, [vp]),
        "slg_coalescer_search": (i32, [vp, vp, u32, i32, vp, vp, vp, vp, vp]),
        "slg_coalescer_search_plan": (i32, [vp, vp, vp, i32, f32, u32, C.c_int32, u32, i32, vp, vp, vp, vp, vp]),
        "slg_coalescer_submit": (i32, [vp, vp, vp, i32, f32, u32, C.c_int32, u32, i32, i32, vp]),
        "slg_coalescer_poll": (i32, [vp, vp]),
        "slg_coalescer_wait": (i32, [vp, vp, vp, vp, vp, vp, vp]),
        "slg_coalescer_last_error": (C.c_char_p, []),
        "slg_coalescer_stats": (i32, [vp, vp, vp]),
        "slg_coalescer_phase_ms": (i32, [vp, vp, vp, vp, vp]),
        "slg_search_batch": (i32, [vp, vp, u32, u32, i32, vp, vp, vp, vp, vp]),
        "slg_index_add_filter": (i32, [vp, vp]),
        "slg_index_add_filter_terms": (i32, [vp, vp, u32, i32, vp]),
        "slg_index_add_filter_range_i64": (i32, [vp, vp, C.c_int64, C.c_int64]),
        "slg_index_add_filter_range_f64": (i32, [vp, vp, C.c_double, C.c_double]),
        "slg_index_remove_filter": (i32, [vp, i32]),
        "slg_search_batch_filtered": (i32, [vp, vp, u32, vp, u32, i32, vp, vp, vp, vp, vp]),
        "slg_batch_prepare": (vp, [vp, u32, vp, vp, vp, u32, i32]),
        "slg_batch_prepare_filtered": (vp, [vp, u32, vp, vp, vp, vp, u32, i32]),
        "slg_batch_prepare_plan": (vp, [vp, u32, vp, vp, vp, vp, vp, vp, vp, vp, u32, i32]),
        "slg_batch_prepare_plans": (vp, [vp, u32, vp, vp, vp, vp, vp, u32, i32]),
        "slg_batch_run": (i32, [vp]),
        "slg_batch_set_stream": (i32, [vp, vp]),
        "slg_batch_sync": (i32, [vp]),
        "slg_batch_fetch": (i32, [vp, vp, vp, vp, vp, vp]),
        "slg_batch_device_results": (i32, [vp, vp, vp, vp, vp]),
        "slg_batch_device_result_block": (i32, [vp, vp, vp]),
        "slg_batch_info": (i32, [vp, vp, vp, vp]),
        "slg_batch_skip_counts": (i32, [vp, vp, vp]),
        "slg_batch_destroy": (None, [vp]),
        "slg_merge_shards_device": (i32, [vp, u32, u32, u32, vp, vp, vp, vp, u32, vp, vp, vp, vp]),
        "slg_shard_unique_id": (i32, [vp, C.c_size_t]),
        "slg_shard_group_create": (vp, [vp, i32, i32, vp, u32]),
        "slg_shard_group_destroy": (None, [vp]),
        "slg_batch_run_sharded": (i32, [vp, vp, vp, vp, vp, vp]),
        "slg_batch_sharded_device_results": (i32, [vp, vp, vp, vp, vp]),
        "slg_batch_run_sharded_seq": (i32, [vp, vp, C.c_uint64, vp, vp, vp, vp]),
        "slg_shard_group_stats": (i32, [vp, vp, vp, vp, vp]),
        "slg_shard_group_skip_seq": (i32, [vp, C.c_uint64]),
        "slg_batch_fetch_sharded": (i32, [vp, vp, vp, vp, vp]),
        "slg_profile_enable": (i32, [vp, i32]),
        "slg_profile_read": (i32, [vp, vp, vp]),
        "slg_rerank_batch": (i32, [vp, u32, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp, vp, vp]),
        "slg_rerank_batch_device": (i32, [vp, u32, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp,
                                          vp, vp]),
        "slg_rerank_multi_batch": (i32, [vp, u32, u32, vp, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp,
                                         vp, vp]),
        "slg_rerank_multi_batch_device": (i32, [vp, u32, u32, vp, vp, vp, vp, vp, vp, vp, u32, u32, vp,
                                                vp, vp, vp, vp]),
        "slg_index_add_vector_field": (i32, [vp, vp, u32]),
        "slg_batch_rerank_device": (i32, [vp, u32, vp, vp, vp, u32, vp, vp, vp, vp, vp]),
        "slg_rerank_fields_batch": (i32, [vp, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, vp, vp, vp,
                                          vp, vp]),
        "slg_rerank_fields_batch_device": (i32, [vp, u32, u32, vp, vp, vp, vp, vp, vp, vp, vp, u32, u32, vp,
                                                 vp, vp, vp, vp]),
    }
    for name, (res, args) in sigs.items():
        if os.environ.get("SLG_LIB_TAG") and not hasattr(L, name):
            continue  # an experiment build of an older revision (tools/build_variant.sh): A/B timing only
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error() -> str:
    return load().slg_last_error().decode("utf-8", "replace")


def last_error_code() -> int:
    return int(load().slg_last_error_code())


def default_tuning() -> Tuning:
    """Defaults + SLG_* environment overrides (slg_tuning_default)."""
    t = Tuning()
    load().slg_tuning_default(C.addressof(t))
    return t


def check(rc: int) -> None:
    if rc != OK:
        raise SlgError(rc, last_error())
