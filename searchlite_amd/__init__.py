"""searchlite_amd — MI355X-native batched BM25 top-k scorer + vector rerank for searchlite.

The package holds only what this one hot path needs: the HIP kernels and C ABI (csrc/,
include/searchlite_gpu.h) and the host-side mirror of the scorer interface.  Importing the
package does not need a GPU; creating a GpuIndex does, and fails loudly if the HIP library
is missing (there is no CPU fallback).
"""
from .segment import (NO_TERM, NO_VECTOR, PLAN_DISMAX, PLAN_SUM, Segment, SegmentBuilder,
                      default_tokenize, fold_terms, parse_query_terms, plan_best_fields,
                      plan_dis_max_terms, plan_most_fields, plan_query_string, resolve_plan,
                      resolve_query)

__all__ = ["Segment", "SegmentBuilder", "default_tokenize", "fold_terms", "parse_query_terms",
           "resolve_query", "NO_TERM", "NO_VECTOR", "GpuIndex", "PreparedBatch", "Bm25", "Wand",
           "Bmw", "SlgError", "PLAN_SUM", "PLAN_DISMAX", "plan_query_string", "plan_best_fields",
           "plan_most_fields", "plan_dis_max_terms", "resolve_plan"]


def __getattr__(name):
    if name in ("GpuIndex", "PreparedBatch", "Bm25", "Wand", "Bmw", "device_count"):
        from . import searcher
        return getattr(searcher, name)
    if name == "SlgError":
        from ._native import SlgError
        return SlgError
    raise AttributeError(name)
