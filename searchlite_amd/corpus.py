"""Synthetic corpora and query batches (harness; spec in SURVEY.md section 8d / DESIGN.md).

zipf_segment() wraps csrc/tools/corpus_gen.cpp: one text field, Zipf(s) term ranks, document
lengths uniform in [len_min, len_min+len_span].  zipf_queries() draws T distinct term ranks
per query uniformly from a mid-frequency band.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build
from .segment import Segment

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.build_corpus_tool()
        L = C.CDLL(path)
        L.slc_zipf_count.restype = C.c_void_p
        L.slc_zipf_count.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32,
                                     C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
        L.slc_zipf_fill.restype = C.c_int
        L.slc_zipf_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.slc_free.restype = None
        L.slc_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def zipf_segment(n_docs: int, vocab: int, s: float = 1.0, len_min: int = 128, len_span: int = 256,
                 seed: int = 42, k1: float = 0.9, b: float = 0.4, n_threads: int | None = None
                 ) -> Segment:
    L = _load()
    if n_threads is None:
        n_threads = max(1, min(16, os.cpu_count() or 1))
    offs = np.zeros(vocab + 1, dtype=np.uint64)
    doc_len = np.zeros(n_docs, dtype=np.float32)
    P = C.c_uint64(0)
    avg = C.c_double(0.0)
    h = L.slc_zipf_count(n_docs, vocab, s, len_min, len_span, seed, n_threads,
                         offs.ctypes.data, doc_len.ctypes.data, C.addressof(P), C.addressof(avg))
    if not h:
        raise RuntimeError("slc_zipf_count failed")
    try:
        doc_ids = np.empty(P.value, dtype=np.uint32)
        tfs = np.empty(P.value, dtype=np.uint32)
        rc = L.slc_zipf_fill(h, offs.ctypes.data, doc_ids.ctypes.data, tfs.ctypes.data)
        if rc != 0:
            raise RuntimeError("slc_zipf_fill failed")
    finally:
        L.slc_free(h)
    total = np.float64(doc_len.astype(np.float64).sum())
    # index/segment.rs:946-957: *sum as f32 / total_docs as f32
    avgdl = np.float32(np.float32(total) / np.float32(n_docs)) if n_docs else np.float32(0)
    return Segment(n_docs=n_docs, term_offsets=offs, doc_ids=doc_ids, tfs=tfs,
                   field_doc_len=[doc_len], field_avgdl=np.array([avgdl], dtype=np.float32),
                   docs=float(n_docs), k1=k1, b=b, fields=["body"])


def zipf_queries(nq: int, n_terms: int, rank_lo: int = 64, rank_hi: int = 8192, seed: int = 7,
                 vocab: int | None = None):
    """-> (q_offsets u32[nq+1], q_terms u32[nq*n_terms], q_weights f32[nq*n_terms]).
    Term id == term rank - 1 in zipf_segment corpora; distinct terms within a query
    (api/reader.rs:2977-2981 would fold duplicates)."""
    hi = rank_hi if vocab is None else min(rank_hi, vocab + 1)
    lo = min(rank_lo, max(1, hi - n_terms))
    rng = np.random.default_rng(seed)
    terms = np.empty((nq, n_terms), dtype=np.uint32)
    for q in range(nq):
        terms[q] = rng.choice(np.arange(lo, hi, dtype=np.uint32), size=n_terms, replace=False) - 1
    offs = (np.arange(nq + 1, dtype=np.uint32) * n_terms).astype(np.uint32)
    return offs, terms.reshape(-1), np.ones(nq * n_terms, dtype=np.float32)


def zipf_multifield_segment(n_docs: int, vocab: int, n_fields: int = 4, len_min: int = 32, len_span: int = 64,
                            seed: int = 42, k1: float = 0.9, b: float = 0.4, n_threads: int | None = None) -> Segment:
    """`n_fields` text fields over one vocabulary, each an independent zipf_segment draw (seed + f) with
    doc lengths uniform in [len_min, len_min + len_span]: term id = f * vocab + (rank - 1) (the
    "field:term" keys of index/postings.rs), per-field doc lengths and avgdl (index/segment.rs:848).
    The defaults make config 2's corpus split over 4 fields (4 x ~64 tokens per doc)."""
    parts = [zipf_segment(n_docs, vocab, len_min=len_min, len_span=len_span, seed=seed + f, k1=k1, b=b,
                          n_threads=n_threads) for f in range(n_fields)]
    offs = np.zeros(n_fields * vocab + 1, dtype=np.uint64)
    base = 0
    for f, sg in enumerate(parts):
        offs[f * vocab + 1:(f + 1) * vocab + 1] = sg.term_offsets[1:] + np.uint64(base)
        base += int(sg.term_offsets[-1])
    return Segment(n_docs=n_docs, term_offsets=offs, doc_ids=np.concatenate([sg.doc_ids for sg in parts]),
                   tfs=np.concatenate([sg.tfs for sg in parts]),
                   field_doc_len=[sg.field_doc_len[0] for sg in parts],
                   field_avgdl=np.array([sg.field_avgdl[0] for sg in parts], dtype=np.float32),
                   docs=float(n_docs), k1=k1, b=b, term_field=np.repeat(np.arange(n_fields, dtype=np.uint16), vocab),
                   fields=[f"f{f}" for f in range(n_fields)])


def multifield_queries(nq: int, n_words: int, n_fields: int, vocab: int, rank_lo: int = 64, rank_hi: int = 8192,
                       seed: int = 7):
    """Query strings of `n_words` distinct words over all `n_fields` fields (the default `fields: None`,
    api/reader.rs:2576-2586): every word is one ScorePlan leaf, its per-field terms add into it, the
    leaves are summed (query/planner.rs:354-360).  -> (q_offsets, q_terms, q_weights, q_leaf), terms in
    the reference's order (word-major, field-minor)."""
    offs, words, _ = zipf_queries(nq, n_words, rank_lo, rank_hi, seed, vocab)
    words = words.reshape(nq, n_words)
    terms = (words[:, :, None] + (np.arange(n_fields, dtype=np.uint32) * np.uint32(vocab))[None, None, :]).astype(np.uint32)
    leaf = np.broadcast_to(np.arange(n_words, dtype=np.uint32)[None, :, None], terms.shape)
    per_q = n_words * n_fields
    return ((np.arange(nq + 1, dtype=np.uint32) * per_q).astype(np.uint32), np.ascontiguousarray(terms.reshape(-1)),
            np.ones(nq * per_q, dtype=np.float32), np.ascontiguousarray(leaf.reshape(-1)))


def unit_vectors(n: int, dim: int, seed: int = 11) -> np.ndarray:
    """iid N(0,1) rows, L2-normalized in f32 (ingest normalizes cosine vectors,
    index/segment.rs:508-510)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, dim), dtype=np.float32)
    step = max(1, (1 << 24) // max(dim, 1))
    for a in range(0, n, step):
        b = min(n, a + step)
        v = rng.standard_normal((b - a, dim), dtype=np.float32)
        nrm = np.sqrt((v * v).sum(axis=1, dtype=np.float32)).astype(np.float32)
        nrm[nrm == 0] = 1
        out[a:b] = v / nrm[:, None]
    return out
