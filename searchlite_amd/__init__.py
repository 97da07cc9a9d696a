"""searchlite_amd — MI355X-native batched BM25 top-k scorer + vector rerank for searchlite.

The package holds only what this one hot path needs: the HIP kernels and C ABI (csrc/,
include/searchlite_gpu.h) and the host-side mirror of the scorer interface.  Importing the
package does not need a GPU; creating a GpuIndex does, and fails loudly if the HIP library
is missing (there is no CPU fallback).
"""
from .segment import (NO_TERM, NO_VECTOR, Segment, SegmentBuilder, default_tokenize, fold_terms,
                      parse_query_terms, resolve_query)

__all__ = ["Segment", "SegmentBuilder", "default_tokenize", "fold_terms", "parse_query_terms",
           "resolve_query", "NO_TERM", "NO_VECTOR", "GpuIndex", "PreparedBatch", "Bm25", "Wand",
           "Bmw", "SlgError"]


def __getattr__(name):
    if name in ("GpuIndex", "PreparedBatch", "Bm25", "Wand", "Bmw", "device_count"):
        from . import searcher
        return getattr(searcher, name)
    if name == "SlgError":
        from ._native import SlgError
        return SlgError
    raise AttributeError(name)
