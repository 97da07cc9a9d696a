"""ctypes wrapper around oracle/libslo_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (searchlite_amd/) never does.

Segments are passed duck-typed: any object with the attributes
  n_docs, term_offsets(u64[V+1]), doc_ids(u32[P]), tfs(u32[P]), term_field(u16[V]|None),
  field_doc_len(list of f32[N]|None), field_avgdl(f32[F]), docs, k1, b, deleted(u8 bitmap|None)
works (searchlite_amd.segment.Segment has exactly these).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libslo_oracle.so")

BM25, WAND, BMW = 0, 1, 2
COSINE, L2 = 0, 1
NO_TERM = 0xFFFFFFFF


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "slo_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
             or os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "slo_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "libslo_oracle.so"])
    return _LIB_PATH


class _Term(C.Structure):
    _fields_ = [("doc_ids", C.c_void_p), ("tfs", C.c_void_p), ("len", C.c_uint32),
                ("weight", C.c_float), ("avgdl", C.c_float), ("docs", C.c_float),
                ("k1", C.c_float), ("b", C.c_float), ("leaf", C.c_uint32),
                ("doc_lengths", C.c_void_p), ("n_doc_lengths", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("scored_docs", C.c_uint64), ("candidates_examined", C.c_uint64),
                ("postings_advanced", C.c_uint64)]


class _Segment(C.Structure):
    _fields_ = [("n_docs", C.c_uint32), ("n_terms", C.c_uint32),
                ("term_offsets", C.c_void_p), ("doc_ids", C.c_void_p), ("tfs", C.c_void_p),
                ("term_field", C.c_void_p), ("n_fields", C.c_uint32),
                ("field_doc_len", C.c_void_p), ("field_avgdl", C.c_void_p),
                ("docs", C.c_float), ("k1", C.c_float), ("b", C.c_float),
                ("deleted", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        f32 = C.c_float
        L.slo_bm25.restype = f32
        L.slo_bm25.argtypes = [f32] * 7
        L.slo_score_tf.restype = f32
        L.slo_score_tf.argtypes = [f32] * 8
        L.slo_upper_bound_tf.restype = f32
        L.slo_upper_bound_tf.argtypes = [f32] * 8
        L.slo_total_cmp.restype = C.c_int
        L.slo_total_cmp.argtypes = [f32, f32]
        L.slo_execute_top_k.restype = C.c_int
        L.slo_execute_top_k.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
        L.slo_search_batch.restype = C.c_int
        L.slo_search_batch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
        L.slo_search_batch_plan.restype = C.c_int
        L.slo_search_batch_plan.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.slo_search_batch_tree.restype = C.c_int
        L.slo_search_batch_tree.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.slo_search_batch_nodes.restype = C.c_int
        L.slo_search_batch_nodes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.slo_normalize_in_place.restype = None
        L.slo_normalize_in_place.argtypes = [C.c_void_p, C.c_uint32]
        L.slo_metric_similarity.restype = f32
        L.slo_metric_similarity.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]
        L.slo_blend_scores.restype = f32
        L.slo_blend_scores.argtypes = [f32, f32, f32, C.c_int]
        L.slo_missing_vector_score.restype = f32
        L.slo_missing_vector_score.argtypes = [C.c_int]
        L.slo_rerank.restype = C.c_int
        L.slo_rerank.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                 C.c_void_p, f32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.c_void_p]
        L.slo_search_batch_faithful.restype = C.c_double
        L.slo_search_batch_faithful.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p]
        L.slo_rerank_multi.restype = C.c_int
        L.slo_rerank_multi.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data


def bm25(tf, df, doc_len, avgdl, docs, k1, b) -> float:
    return float(lib().slo_bm25(tf, df, doc_len, avgdl, docs, k1, b))


def score_tf(tf, df, doc_len, avgdl, docs, k1, b, weight) -> float:
    return float(lib().slo_score_tf(tf, df, doc_len, avgdl, docs, k1, b, weight))


def upper_bound_tf(tf, df, doc_len, avgdl, docs, k1, b, weight) -> float:
    return float(lib().slo_upper_bound_tf(tf, df, doc_len, avgdl, docs, k1, b, weight))


def total_cmp(a, b) -> int:
    return int(lib().slo_total_cmp(a, b))


class ScoredTerm:
    """query/wand.rs:65-75 (postings as two parallel arrays)."""

    def __init__(self, doc_ids, tfs, weight=1.0, avgdl=1.0, docs=1.0, k1=1.2, b=0.75, leaf=0,
                 doc_lengths=None):
        self.doc_ids = np.ascontiguousarray(doc_ids, dtype=np.uint32)
        self.tfs = np.ascontiguousarray(tfs, dtype=np.uint32)
        assert self.doc_ids.shape == self.tfs.shape
        self.weight, self.avgdl, self.docs, self.k1, self.b = weight, avgdl, docs, k1, b
        self.leaf = leaf
        self.doc_lengths = (None if doc_lengths is None
                            else np.ascontiguousarray(doc_lengths, dtype=np.float32))


def execute_top_k(terms, k, strategy=WAND, block_size=None, use_plan=False, deleted=None,
                  min_len_cache=None, want_stats=False):
    """query/wand.rs:338-456; returns [(doc, score)] (and Stats if want_stats)."""
    arr = (_Term * max(len(terms), 1))()
    for i, t in enumerate(terms):
        arr[i] = _Term(_ptr(t.doc_ids), _ptr(t.tfs), len(t.doc_ids), t.weight, t.avgdl, t.docs,
                       t.k1, t.b, t.leaf, _ptr(t.doc_lengths),
                       0 if t.doc_lengths is None else len(t.doc_lengths))
    out_doc = np.zeros(max(k, 1), dtype=np.uint32)
    out_score = np.zeros(max(k, 1), dtype=np.float32)
    st = Stats()
    mlc = None if min_len_cache is None else np.ascontiguousarray(min_len_cache, dtype=np.float32)
    n = lib().slo_execute_top_k(arr, len(terms), k, strategy, block_size or 0, int(use_plan),
                                _ptr(deleted), _ptr(mlc), _ptr(out_doc), _ptr(out_score),
                                C.addressof(st))
    hits = [(int(out_doc[i]), float(out_score[i])) for i in range(n)]
    return (hits, st) if want_stats else hits


def _pack_segments(segments):
    keep = []
    arr = (_Segment * len(segments))()
    for i, s in enumerate(segments):
        nf = len(s.field_doc_len)
        ptrs = (C.c_void_p * nf)(*[_ptr(a) for a in s.field_doc_len])
        avg = np.ascontiguousarray(s.field_avgdl, dtype=np.float32)
        keep += [ptrs, avg]
        arr[i] = _Segment(s.n_docs, len(s.term_offsets) - 1, _ptr(s.term_offsets), _ptr(s.doc_ids),
                          _ptr(s.tfs), _ptr(s.term_field), nf, C.addressof(ptrs), _ptr(avg),
                          s.docs, s.k1, s.b, _ptr(s.deleted))
    return arr, keep


PLAN_SUM, PLAN_DISMAX, PLAN_LEAF = 0, 1, 2


def search_batch(segments, q_offsets, q_terms, q_weights, k, strategy=WAND, block_size=None,
                 n_threads=1, cache_min_len=False, want_stats=False, q_leaf=None, q_plan=None,
                 q_tie=None, q_nleaves=None, q_leaf_offsets=None, leaf_group=None, q_group_offsets=None,
                 group_plan=None, group_tie=None, q_node_offsets=None, node_kind=None, node_tie=None,
                 node_parent=None):
    """api/reader.rs search() over segments for a batch of pure-disjunction queries.

    q_terms has shape [total_query_terms, n_segs] (per-segment term ids, NO_TERM if absent).
    Score plans (query/planner.rs:113-153): q_leaf[i] = leaf of query term i (default: term i of
    a query is leaf i), q_plan[q] = PLAN_SUM | PLAN_DISMAX over the leaves, q_tie[q] = DisMax
    tie breaker, q_nleaves[q] = leaves of the plan.  Two-level plans (ScoreExpr::evaluate recursion,
    planner.rs:122-153): leaf_group / group_plan / group_tie with their per-query CSR offsets — the
    root combines groups, a group combines its consecutive leaves.  Trees of any depth: per query a
    node array in pre-order (q_node_offsets CSR; node_kind PLAN_SUM / PLAN_DISMAX / PLAN_LEAF, node_tie,
    node_parent; the i-th LEAF node of a query is leaf i).
    Returns (doc[nq,k], seg[nq,k], score[nq,k], count[nq]).
    """
    segs, keep = _pack_segments(segments)
    q_offsets = np.ascontiguousarray(q_offsets, dtype=np.uint32)
    q_terms = np.ascontiguousarray(q_terms, dtype=np.uint32).reshape(-1, len(segments))
    q_weights = np.ascontiguousarray(q_weights, dtype=np.float32)
    nq = len(q_offsets) - 1
    out_doc = np.zeros((nq, k), dtype=np.uint32)
    out_seg = np.zeros((nq, k), dtype=np.uint32)
    out_score = np.zeros((nq, k), dtype=np.float32)
    out_count = np.zeros(nq, dtype=np.uint32)
    stats = (Stats * max(nq, 1))() if want_stats else None
    ql = None if q_leaf is None else np.ascontiguousarray(q_leaf, dtype=np.uint32)
    qp = None if q_plan is None else np.ascontiguousarray(q_plan, dtype=np.int32)
    qt = None if q_tie is None else np.ascontiguousarray(q_tie, dtype=np.float32)
    qn = None if q_nleaves is None else np.ascontiguousarray(q_nleaves, dtype=np.uint32)
    qlo = None if q_leaf_offsets is None else np.ascontiguousarray(q_leaf_offsets, dtype=np.uint32)
    lg = None if leaf_group is None else np.ascontiguousarray(leaf_group, dtype=np.uint32)
    qgo = None if q_group_offsets is None else np.ascontiguousarray(q_group_offsets, dtype=np.uint32)
    gp = None if group_plan is None else np.ascontiguousarray(group_plan, dtype=np.int32)
    gt = None if group_tie is None else np.ascontiguousarray(group_tie, dtype=np.float32)
    if q_node_offsets is not None:
        qno = np.ascontiguousarray(q_node_offsets, dtype=np.uint32)
        nk = np.ascontiguousarray(node_kind, dtype=np.int32)
        nt = np.ascontiguousarray(node_tie, dtype=np.float32)
        npar = np.ascontiguousarray(node_parent, dtype=np.uint32)
        rc = lib().slo_search_batch_nodes(segs, len(segments), nq, _ptr(q_offsets), _ptr(q_terms), _ptr(q_weights),
                                          _ptr(ql), _ptr(qno), _ptr(nk), _ptr(nt), _ptr(npar), k, strategy,
                                          block_size or 0, n_threads, int(cache_min_len), _ptr(out_doc),
                                          _ptr(out_seg), _ptr(out_score), _ptr(out_count),
                                          None if stats is None else C.addressof(stats))
        if rc != 0:
            raise RuntimeError(f"slo_search_batch_nodes failed: {rc}")
        if want_stats:
            return out_doc, out_seg, out_score, out_count, stats
        return out_doc, out_seg, out_score, out_count
    rc = lib().slo_search_batch_tree(segs, len(segments), nq, _ptr(q_offsets), _ptr(q_terms),
                                     _ptr(q_weights), _ptr(ql), _ptr(qp), _ptr(qt), _ptr(qn),
                                     _ptr(qlo), _ptr(lg), _ptr(qgo), _ptr(gp), _ptr(gt), k,
                                     strategy, block_size or 0, n_threads, int(cache_min_len),
                                     _ptr(out_doc), _ptr(out_seg), _ptr(out_score), _ptr(out_count),
                                     None if stats is None else C.addressof(stats))
    if rc != 0:
        raise RuntimeError(f"slo_search_batch failed: {rc}")
    if want_stats:
        return out_doc, out_seg, out_score, out_count, stats
    return out_doc, out_seg, out_score, out_count


def search_batch_filtered(segments, q_offsets, q_terms, q_weights, k, q_filter, filters,
                          strategy=WAND, **kw):
    """search_batch with a doc filter per query.  The reference's accept() is
    `!deleted && filter` for a pure disjunction (api/reader.rs:3009-3018), so a filtered query
    is the unfiltered scorer run with the tombstones OR-ed with the filter's complement (idf
    keeps using segment.docs, which is a separate field).  filters[f][s] = boolean pass mask of
    segment s or None; q_filter[q] = filter id or < 0."""
    import copy
    q_offsets = np.ascontiguousarray(q_offsets, dtype=np.uint32)
    nq = len(q_offsets) - 1
    q_terms = np.ascontiguousarray(q_terms, dtype=np.uint32).reshape(-1, len(segments))
    q_weights = np.ascontiguousarray(q_weights, dtype=np.float32)
    out = None
    for q in range(nq):
        f = int(q_filter[q]) if q_filter is not None else -1
        segs = segments
        if f >= 0:
            segs = []
            for s, seg in enumerate(segments):
                sg = copy.copy(seg)
                m = filters[f][s]
                if m is not None:
                    dead = np.zeros(seg.n_docs, dtype=bool)
                    if seg.deleted is not None:
                        dead |= np.unpackbits(seg.deleted, bitorder="little")[:seg.n_docs].astype(bool)
                    dead |= ~np.asarray(m, dtype=bool)
                    sg.deleted = np.packbits(dead, bitorder="little")
                segs.append(sg)
        a, b = int(q_offsets[q]), int(q_offsets[q + 1])
        kq = dict(kw)  # per-term / per-query plan arrays: this query's part
        if kq.get("q_leaf") is not None:
            kq["q_leaf"] = np.asarray(kq["q_leaf"])[a:b]
        for name in ("q_plan", "q_tie", "q_nleaves"):
            if kq.get(name) is not None:
                kq[name] = np.asarray(kq[name])[q:q + 1]
        if kq.get("leaf_group") is not None:  # two-level plans: this query's leaves and groups
            la, lb = int(kw["q_leaf_offsets"][q]), int(kw["q_leaf_offsets"][q + 1])
            ga, gb = int(kw["q_group_offsets"][q]), int(kw["q_group_offsets"][q + 1])
            kq["leaf_group"] = np.asarray(kw["leaf_group"])[la:lb]
            kq["group_plan"] = np.asarray(kw["group_plan"])[ga:gb]
            kq["group_tie"] = np.asarray(kw["group_tie"])[ga:gb]
            kq["q_leaf_offsets"] = np.array([0, lb - la], dtype=np.uint32)
            kq["q_group_offsets"] = np.array([0, gb - ga], dtype=np.uint32)
        if kq.get("q_node_offsets") is not None:  # trees given node by node: this query's nodes
            na, nb = int(kw["q_node_offsets"][q]), int(kw["q_node_offsets"][q + 1])
            for name in ("node_kind", "node_tie", "node_parent"):
                kq[name] = np.asarray(kw[name])[na:nb]
            kq["q_node_offsets"] = np.array([0, nb - na], dtype=np.uint32)
        r = search_batch(segs, np.array([0, b - a], dtype=np.uint32), q_terms[a:b], q_weights[a:b], k,
                         strategy=strategy, **kq)
        if out is None:
            out = [np.zeros((nq,) + x.shape[1:], dtype=x.dtype) for x in r[:4]]
        for o, x in zip(out, r[:4]):
            o[q] = x[0]
    return tuple(out) if out is not None else search_batch(segments, q_offsets, q_terms, q_weights, k,
                                                           strategy=strategy, **kw)


NO_TERM = 0xFFFFFFFF


def search_batch_min_match(segments, q_offsets, q_terms, q_weights, k, q_min_match, strategy=WAND,
                           q_filter=None, filters=None, **kw):
    """search_batch with minimum_should_match per query.  The query-string matcher accepts a doc iff
    at least `required` of its term groups hold it (api/reader.rs:1509-1517: `matched_terms >= required`,
    term_group_matches = any list of the group holds the doc, :1571-1582), and the scorer's accept() is
    `!deleted && matcher.matches(doc) && filter` (api/reader.rs:3009-3036): so a query with
    minimum_should_match = m is the unfiltered scorer run with the tombstones OR-ed with the docs that
    fewer than m LEAVES (term group = ScorePlan leaf, kw["q_leaf"]; default: term i = leaf i) hold.
    Restated with numpy over the segments' posting arrays; q_filter / filters as in search_batch_filtered."""
    q_offsets = np.ascontiguousarray(q_offsets, dtype=np.uint32)
    nq = len(q_offsets) - 1
    q_terms = np.ascontiguousarray(q_terms, dtype=np.uint32).reshape(-1, len(segments))
    q_leaf = kw.get("q_leaf")
    masks = []  # one filter per query: the docs the matcher accepts (None where minimum_should_match <= 1)
    for q in range(nq):
        m = int(q_min_match[q])
        a, b = int(q_offsets[q]), int(q_offsets[q + 1])
        per_seg = []
        for s, seg in enumerate(segments):
            if m <= 1:
                per_seg.append(None)
                continue
            leaves = np.arange(b - a) if q_leaf is None else np.asarray(q_leaf)[a:b]
            count = np.zeros(seg.n_docs, dtype=np.int32)
            for lf in np.unique(leaves):
                held = np.zeros(seg.n_docs, dtype=bool)
                for i in np.nonzero(leaves == lf)[0]:
                    t = int(q_terms[a + i, s])
                    if t != NO_TERM:
                        held[seg.doc_ids[int(seg.term_offsets[t]):int(seg.term_offsets[t + 1])]] = True
                count += held
            per_seg.append(count >= m)
        f = int(q_filter[q]) if q_filter is not None else -1
        if f >= 0:  # the request's own filter as well
            per_seg = [fm if pm is None else (pm if fm is None else (pm & np.asarray(fm, dtype=bool)))
                       for pm, fm in zip(per_seg, filters[f])]
        masks.append(per_seg)
    return search_batch_filtered(segments, q_offsets, q_terms, q_weights, k, np.arange(nq), masks,
                                 strategy=strategy, **kw)


def normalize_in_place(v):
    assert v.dtype == np.float32 and v.flags.c_contiguous
    lib().slo_normalize_in_place(_ptr(v), v.size)


def metric_similarity(metric, a, b) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    return float(lib().slo_metric_similarity(metric, _ptr(a), _ptr(b), a.size))


def blend_scores(bm25_score, vec, alpha, higher_is_better=True) -> float:
    return float(lib().slo_blend_scores(bm25_score, vec, alpha, int(higher_is_better)))


def missing_vector_score(metric) -> float:
    return float(lib().slo_missing_vector_score(metric))


def rerank(metric, vec_offsets, vec_values, qvec, alpha, cand_doc, cand_bm25, k_out):
    vec_offsets = np.ascontiguousarray(vec_offsets, dtype=np.uint32)
    vec_values = np.ascontiguousarray(vec_values, dtype=np.float32)
    qvec = np.ascontiguousarray(qvec, dtype=np.float32)
    cand_doc = np.ascontiguousarray(cand_doc, dtype=np.uint32)
    cand_bm25 = np.ascontiguousarray(cand_bm25, dtype=np.float32)
    dim = qvec.size
    out_doc = np.zeros(max(k_out, 1), dtype=np.uint32)
    out_score = np.zeros(max(k_out, 1), dtype=np.float32)
    out_vec = np.zeros(max(k_out, 1), dtype=np.float32)
    n = lib().slo_rerank(metric, dim, _ptr(vec_offsets), len(vec_offsets), _ptr(vec_values),
                         _ptr(qvec), alpha, _ptr(cand_doc), _ptr(cand_bm25), len(cand_doc), k_out,
                         _ptr(out_doc), _ptr(out_score), _ptr(out_vec))
    return out_doc[:n].copy(), out_score[:n].copy(), out_vec[:n].copy()


def rerank_multi(metric, vec_offsets, vec_values, qvecs, alpha, cand_doc, cand_bm25, k_out, boost=None):
    """compute_hybrid_score with several clauses (qvecs [n_clauses, dim], alpha [n_clauses])."""
    vec_offsets = np.ascontiguousarray(vec_offsets, dtype=np.uint32)
    vec_values = np.ascontiguousarray(vec_values, dtype=np.float32)
    qvecs = np.ascontiguousarray(qvecs, dtype=np.float32)
    nc, dim = qvecs.shape
    alpha = np.ascontiguousarray(alpha, dtype=np.float32)
    bst = None if boost is None else np.ascontiguousarray(boost, dtype=np.float32)
    cand_doc = np.ascontiguousarray(cand_doc, dtype=np.uint32)
    cand_bm25 = np.ascontiguousarray(cand_bm25, dtype=np.float32)
    out_doc = np.zeros(max(k_out, 1), dtype=np.uint32)
    out_score = np.zeros(max(k_out, 1), dtype=np.float32)
    out_vec = np.zeros(max(k_out, 1), dtype=np.float32)
    n = lib().slo_rerank_multi(metric, dim, _ptr(vec_offsets), len(vec_offsets), _ptr(vec_values), nc,
                               _ptr(qvecs), _ptr(alpha), _ptr(bst), _ptr(cand_doc), _ptr(cand_bm25),
                               len(cand_doc), k_out, _ptr(out_doc), _ptr(out_score), _ptr(out_vec))
    return out_doc[:n].copy(), out_score[:n].copy(), out_vec[:n].copy()


def rerank_fields(fields, clause_field, qvecs, alpha, cand_doc, cand_bm25, k_out, boost=None):
    """compute_hybrid_score (api/reader.rs:225-254) when the clauses name different vector fields
    (one segment).  fields: list of (metric, vec_offsets, vec_values); clause_field[c] indexes it;
    qvecs: list of the clause vectors.  Numpy restatement (small cases): similarities are f32 sums
    taken left to right (vectors/mod.rs:107-120: cosine = dot of the stored vectors, NaN -> 0;
    L2 = -sqrt(sum (x - y)^2)); a clause whose field has no vector for the doc takes
    missing_vector_score of ITS metric (:217-223); blended = mean over the clauses (:240-251);
    the vector score sums the clauses that found a vector (:236-238), and is the missing score of
    clause 0 when none did (what the single-field oracle reports).  Order: blended desc, doc asc."""
    f32 = np.float32
    nc = len(clause_field)
    old_err = np.seterr(over="ignore")  # f32::MIN sums saturate to -inf, as in the reference
    alpha = np.asarray(alpha, dtype=f32)
    bst = np.ones(nc, f32) if boost is None else np.asarray(boost, dtype=f32)
    rows = []
    for d, bm in zip(np.asarray(cand_doc, dtype=np.uint32), np.asarray(cand_bm25, dtype=f32)):
        blended_sum, vector_sum, has = f32(0), f32(0), False
        for c in range(nc):
            metric, offs, vals = fields[clause_field[c]]
            miss = f32(-1.0) if metric == 0 else f32(-3.40282347e+38)
            off = offs[d] if d < len(offs) else 0xFFFFFFFF
            if off == 0xFFFFFFFF:
                vs = miss
            else:
                x, y = np.asarray(vals[off], dtype=f32), np.asarray(qvecs[c], dtype=f32)
                if metric == 0:
                    sm = np.add.accumulate((y * x).astype(f32), dtype=f32)[-1]
                    sim = f32(0) if np.isnan(sm) else sm
                else:
                    diff = (y - x).astype(f32)
                    sim = -np.sqrt(np.add.accumulate((diff * diff).astype(f32), dtype=f32)[-1], dtype=f32)
                vs = f32(sim * bst[c])
                vector_sum = f32(vector_sum + vs)
                has = True
            a = alpha[c]
            blended = bm if a >= 1.0 else (vs if a <= 0.0 else f32(f32(a * bm) + f32(f32(f32(1.0) - a) * vs)))
            blended_sum = f32(blended_sum + blended)
        m0 = f32(-1.0) if fields[clause_field[0]][0] == 0 else f32(-3.40282347e+38)
        rows.append((f32(blended_sum / f32(nc)), int(d), vector_sum if has else m0))
    np.seterr(**old_err)
    rows.sort(key=lambda r: (-float(r[0]), r[1]))
    rows = rows[:k_out]
    return (np.array([r[1] for r in rows], np.uint32), np.array([r[0] for r in rows], f32),
            np.array([r[2] for r in rows], f32))


def search_batch_faithful(segments, q_offsets, q_terms, q_weights, k, strategy=WAND, n_threads=1):
    """BASELINE.md "Baseline A": scorer + the reference's per-query posting decode (twice) and
    doc-length rebuild.  -> ((doc, seg, score, count), seconds of the query phase)."""
    segs, keep = _pack_segments(segments)
    q_offsets = np.ascontiguousarray(q_offsets, dtype=np.uint32)
    q_terms = np.ascontiguousarray(q_terms, dtype=np.uint32).reshape(-1, len(segments))
    q_weights = np.ascontiguousarray(q_weights, dtype=np.float32)
    nq = len(q_offsets) - 1
    out_doc = np.zeros((nq, k), dtype=np.uint32)
    out_seg = np.zeros((nq, k), dtype=np.uint32)
    out_score = np.zeros((nq, k), dtype=np.float32)
    out_count = np.zeros(nq, dtype=np.uint32)
    secs = lib().slo_search_batch_faithful(segs, len(segments), nq, _ptr(q_offsets), _ptr(q_terms),
                                           _ptr(q_weights), k, strategy, n_threads, _ptr(out_doc),
                                           _ptr(out_seg), _ptr(out_score), _ptr(out_count))
    return (out_doc, out_seg, out_score, out_count), float(secs)
