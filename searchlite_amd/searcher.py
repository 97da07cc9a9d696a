"""Host mirror of the scorer interface, on top of the C ABI.

Names follow searchlite-core (query/wand.rs): `ScoredTerm`-style term lists, `execute_top_k`,
`RankedDoc`-style (doc_id, score) results, `QueryStats`.  GpuIndex owns an `slg_index`
(device-resident segments); PreparedBatch owns an `slg_batch`.

No CPU fallback lives here: every search goes through libsearchlite_gpu.so.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from .segment import Segment, fold_terms, parse_query_terms, resolve_query

Bm25, Wand, Bmw = N.STRATEGY_BM25, N.STRATEGY_WAND, N.STRATEGY_BMW  # api/types.rs:6-13


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def device_count() -> int:
    n = N.load().slg_device_count()
    if n < 0:
        raise N.SlgError(n, N.last_error())
    return n


class GpuIndex:
    """All segments of one shard, staged in HBM (slg_index_create)."""

    def __init__(self, segments: Sequence[Segment], device: int = 0, tuning: Optional[dict] = None):
        """tuning: overrides of slg_tuning fields by name (e.g. {"pruning": 1}) on top of the
        defaults / SLG_* environment."""
        self._lib = N.load()
        self.segments = list(segments)
        self.device = device
        self._batches = weakref.WeakSet()  # closed with the index
        descs = (N.SegmentDesc * len(self.segments))()
        keep = []
        for i, s in enumerate(self.segments):
            descs[i] = self._desc(s, keep)
        tune = N.default_tuning()
        for name, val in (tuning or {}).items():
            if not hasattr(tune, name):
                raise KeyError(f"slg_tuning has no field {name!r}")
            setattr(tune, name, val)
        self._h = self._lib.slg_index_create_tuned(descs, len(self.segments), device, C.addressof(tune))
        if not self._h:
            raise N.SlgError(N.last_error_code() or N.ERR_INVALID, N.last_error())

    @staticmethod
    def _desc(s: Segment, keep: list) -> "N.SegmentDesc":
        nf = len(s.field_doc_len)
        ptrs = (C.c_void_p * nf)(*[_ptr(a) for a in s.field_doc_len])
        keep.append(ptrs)
        vec_rows = 0 if s.vec_values is None else int(s.vec_values.shape[0])
        return N.SegmentDesc(
            s.n_docs, s.n_terms, _ptr(s.term_offsets), _ptr(s.doc_ids), _ptr(s.tfs),
            _ptr(s.term_field), nf, C.addressof(ptrs), _ptr(s.field_avgdl),
            s.docs, s.k1, s.b, _ptr(s.deleted),
            s.vec_dim, s.vec_metric, _ptr(s.vec_offsets), _ptr(s.vec_values), vec_rows)

    # -- index updates (the reference's commit, api/writer.rs:106-240) ------------------------
    def update_deleted(self, seg: int, deleted: Optional[np.ndarray], live_docs: float) -> None:
        """New tombstones of segment `seg` (complete bitmap, bit d&7 of byte d>>3) and its new
        live_docs (slg_index_update_deleted); the mirrored Segment object follows."""
        bm = None if deleted is None else np.ascontiguousarray(deleted, dtype=np.uint8)
        N.check(self._lib.slg_index_update_deleted(self._h, seg, _ptr(bm), float(live_docs)))
        s = self.segments[seg]
        s.deleted = bm
        s.docs = float(live_docs)

    def add_segment(self, segment: Segment) -> int:
        """Stage one more segment at the next ordinal (slg_index_add_segment) -> the ordinal."""
        keep: list = []
        d = self._desc(segment, keep)
        ord_ = self._lib.slg_index_add_segment(self._h, C.addressof(d))
        if ord_ < 0:
            raise N.SlgError(ord_, N.last_error())
        self.segments.append(segment)
        return ord_

    def remove_segment(self, seg: int) -> None:
        N.check(self._lib.slg_index_remove_segment(self._h, seg))
        del self.segments[seg]

    @property
    def generation(self) -> int:
        return int(self._lib.slg_index_generation(self._h))

    def tuning(self) -> "N.Tuning":
        t = N.Tuning()
        N.check(self._lib.slg_index_get_tuning(self._h, C.addressof(t)))
        return t

    # -- lifecycle -------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            for b in list(self._batches):  # batches die with their index
                b.close()
            self._lib.slg_index_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def n_segs(self) -> int:
        return len(self.segments)

    def info(self):
        ns, npost, nbytes = C.c_uint32(), C.c_uint64(), C.c_uint64()
        N.check(self._lib.slg_index_info(self._h, C.addressof(ns), C.addressof(npost),
                                         C.addressof(nbytes)))
        return {"n_segs": ns.value, "n_postings": npost.value, "device_bytes": nbytes.value}

    def trim_pool(self) -> int:
        """Give the pooled work buffers of finished batches back to the runtime -> bytes freed."""
        freed = C.c_uint64()
        N.check(self._lib.slg_index_trim_pool(self._h, C.addressof(freed)))
        return freed.value

    def set_stream(self, hip_stream) -> None:
        """Run on an external hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; 0 is the
        HIP null stream = PyTorch's default stream).  None restores the index's own stream."""
        h = C.c_void_p(-1) if hip_stream is None else C.c_void_p(int(hip_stream) or None)
        N.check(self._lib.slg_index_set_stream(self._h, h))

    def profile(self, on: bool) -> None:
        N.check(self._lib.slg_profile_enable(self._h, int(on)))

    def profile_read(self) -> Tuple[int, float]:
        n, ms = C.c_uint32(), C.c_float()
        N.check(self._lib.slg_profile_read(self._h, C.addressof(n), C.addressof(ms)))
        return n.value, ms.value

    # -- doc filters (accept = !deleted && filter, api/reader.rs:3009-3018) ---------
    def add_filter(self, seg_masks) -> int:
        """Register a filter: seg_masks[s] = boolean array over the docs of segment s (True =
        passes) or None (all pass).  Returns the filter id queries refer to."""
        assert len(seg_masks) == self.n_segs
        packed = [None if m is None else np.packbits(np.asarray(m, dtype=bool), bitorder="little")
                  for m in seg_masks]
        ptrs = (C.c_void_p * self.n_segs)(*[None if b is None else b.ctypes.data for b in packed])
        rc = self._lib.slg_index_add_filter(self._h, ptrs)
        if rc < 0:
            N.check(rc)
        return rc

    def add_filter_range(self, seg_columns, lo, hi) -> int:
        """Filter built on the device from one numeric fast-field column per segment (int64 or
        float64 arrays of n_docs values): doc passes iff lo <= value <= hi."""
        assert len(seg_columns) == self.n_segs
        cols = [np.ascontiguousarray(c) for c in seg_columns]
        ptrs = (C.c_void_p * self.n_segs)(*[c.ctypes.data for c in cols])
        if all(c.dtype == np.int64 for c in cols):
            rc = self._lib.slg_index_add_filter_range_i64(self._h, ptrs, int(lo), int(hi))
        elif all(c.dtype == np.float64 for c in cols):
            rc = self._lib.slg_index_add_filter_range_f64(self._h, ptrs, float(lo), float(hi))
        else:
            raise TypeError("filter columns must all be int64 or all float64")
        if rc < 0:
            N.check(rc)
        return rc

    def add_filter_terms(self, term_ids, pass_if_absent: bool = True, and_masks=None) -> int:
        """Filter built on the device from resident posting lists (slg_index_add_filter_terms): the docs
        that hold none (pass_if_absent) / at least one of the terms; term_ids [n_terms, n_segs];
        and_masks: boolean masks per segment (or None) AND-ed with it."""
        tid = np.ascontiguousarray(term_ids, dtype=np.uint32).reshape(-1, self.n_segs)
        ptrs = None
        packed = None
        if and_masks is not None:
            assert len(and_masks) == self.n_segs
            packed = [None if m is None else np.packbits(np.asarray(m, dtype=bool), bitorder="little")
                      for m in and_masks]
            ptrs = (C.c_void_p * self.n_segs)(*[None if b is None else b.ctypes.data for b in packed])
        rc = self._lib.slg_index_add_filter_terms(self._h, tid.ctypes.data if tid.size else None, tid.shape[0],
                                                  int(bool(pass_if_absent)), ptrs)
        if rc < 0:
            N.check(rc)
        return rc

    def remove_filter(self, filter_id: int) -> None:
        N.check(self._lib.slg_index_remove_filter(self._h, filter_id))

    # -- search ----------------------------------------------------------------------
    def prepare(self, q_offsets, q_terms, q_weights, k: int, strategy: int = Wand,
                q_filter=None, q_leaf=None, q_plan=None, q_tie=None, q_nleaves=None,
                q_leaf_offsets=None, leaf_group=None, q_group_offsets=None, group_plan=None,
                group_tie=None, q_node_offsets=None, node_kind=None, node_tie=None, node_parent=None,
                q_min_match=None) -> "PreparedBatch":
        """q_leaf / q_plan / q_tie / q_nleaves: score plans; leaf_group / group_plan / group_tie with
        their per-query offsets: two-level plans; q_node_offsets / node_kind / node_tie / node_parent:
        trees of any shape, node by node in pre-order (slg_batch_prepare_plans, slg_score_plans);
        q_min_match: minimum_should_match per query (leaves that must hold a doc)."""
        return PreparedBatch(self, q_offsets, q_terms, q_weights, k, strategy, q_filter,
                             q_leaf, q_plan, q_tie, q_nleaves, q_leaf_offsets, leaf_group,
                             q_group_offsets, group_plan, group_tie, q_node_offsets, node_kind, node_tie, node_parent,
                             q_min_match)

    def search_plan(self, q_offsets, q_terms, q_weights, k: int, q_leaf=None, q_plan=None,
                    q_tie=None, q_nleaves=None, strategy: int = Wand, q_filter=None, **tree):
        """Batch search with score plans (multi-field leaves / DisMax; **tree: the two-level plan
        arrays of prepare()) -> (doc, seg, score, count)."""
        b = self.prepare(q_offsets, q_terms, q_weights, k, strategy, q_filter, q_leaf, q_plan,
                         q_tie, q_nleaves, **tree)
        try:
            b.run()
            return b.fetch()
        finally:
            b.close()

    def search_batch(self, q_offsets, q_terms, q_weights, k: int, strategy: int = Wand,
                     want_stats: bool = False, q_filter=None):
        """One-shot slg_search_batch over CSR queries -> (doc, seg, score, count[, stats]).
        q_filter: optional int array, one filter id per query (< 0: none)."""
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.uint32)
        nq = len(q_offsets) - 1
        q_terms = np.ascontiguousarray(q_terms, dtype=np.uint32).reshape(-1, self.n_segs)
        q_weights = np.ascontiguousarray(q_weights, dtype=np.float32)
        qs = (N.Query * max(nq, 1))()
        for q in range(nq):
            a, b = int(q_offsets[q]), int(q_offsets[q + 1])
            qs[q] = N.Query(b - a, q_terms.ctypes.data + a * self.n_segs * 4,
                            q_weights.ctypes.data + a * 4)
        out_doc = np.zeros((nq, k), dtype=np.uint32)
        out_seg = np.zeros((nq, k), dtype=np.uint32)
        out_score = np.zeros((nq, k), dtype=np.float32)
        out_count = np.zeros(nq, dtype=np.uint32)
        stats = (N.Stats * max(nq, 1))() if want_stats else None
        qf = None if q_filter is None else np.ascontiguousarray(q_filter, dtype=np.int32)
        assert qf is None or len(qf) == nq
        N.check(self._lib.slg_search_batch_filtered(
            self._h, qs, nq, None if qf is None else _ptr(qf), k, strategy, _ptr(out_doc),
            _ptr(out_seg), _ptr(out_score), _ptr(out_count),
            None if stats is None else C.addressof(stats)))
        if want_stats:
            return out_doc, out_seg, out_score, out_count, stats
        return out_doc, out_seg, out_score, out_count

    def execute_top_k(self, terms: Sequence[Tuple[int, float]], k: int, strategy: int = Wand,
                      segment: int = 0) -> List[Tuple[int, float]]:
        """query/wand.rs:338-356 for one query against one segment: terms = [(term_id, weight)],
        term i is leaf i.  Returns [(doc_id, score)] sorted score desc, doc asc."""
        ids = np.full((len(terms), self.n_segs), N.NO_TERM, dtype=np.uint32)
        for i, (tid, _) in enumerate(terms):
            ids[i, segment] = tid
        w = np.array([t[1] for t in terms], dtype=np.float32)
        offs = np.array([0, len(terms)], dtype=np.uint32)
        d, s, sc, c = self.search_batch(offs, ids, w, k, strategy)
        return [(int(d[0, i]), float(sc[0, i])) for i in range(int(c[0]))]

    def search(self, query: str, default_field: str, limit: int = 10, strategy: int = Wand):
        """IndexReader::search for an eligible query string: k = limit + 1 is handed to the
        scorer (api/reader.rs:2615-2619), hits truncated to limit (:2836-2852).
        Returns [(segment_ord, doc_id, score)]."""
        folded = fold_terms(parse_query_terms(query, default_field))
        if not folded:
            return []
        ids, w = resolve_query(self.segments, folded)
        offs = np.array([0, len(folded)], dtype=np.uint32)
        d, s, sc, c = self.search_batch(offs, ids, w, limit + 1, strategy)
        n = min(int(c[0]), limit)
        return [(int(s[0, i]), int(d[0, i]), float(sc[0, i])) for i in range(n)]

    def search_planned(self, planned, plan: int, n_leaves: int, tie_breaker: float = 0.0,
                       limit: int = 10, strategy: int = Wand, filter_id: int = -1):
        """One request whose scored terms come from segment.plan_query_string / plan_best_fields /
        plan_most_fields / plan_dis_max_terms (the leaf assignment of query/planner.rs).
        Returns [(segment_ord, doc_id, score)] like search()."""
        from .segment import resolve_plan
        if not planned:
            return []
        ids, w, leaf = resolve_plan(self.segments, planned)
        offs = np.array([0, len(planned)], dtype=np.uint32)
        d, s, sc, c = self.search_plan(offs, ids, w, limit + 1, q_leaf=leaf, q_plan=[plan],
                                       q_tie=[tie_breaker], q_nleaves=[n_leaves], strategy=strategy,
                                       q_filter=None if filter_id < 0 else [filter_id])
        n = min(int(c[0]), limit)
        return [(int(s[0, i]), int(d[0, i]), float(sc[0, i])) for i in range(n)]

    # -- rerank ----------------------------------------------------------------------
    def rerank_batch(self, qvecs, alpha, cand_doc, cand_seg, cand_bm25, cand_count, k_out: int):
        """gpu::rerank slot (gpu/rerank.rs:3): vector similarity + alpha blend + top-k_out."""
        qvecs = np.ascontiguousarray(qvecs, dtype=np.float32)
        nq = qvecs.shape[0]
        alpha = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float32), (nq,)))
        cand_doc = np.ascontiguousarray(cand_doc, dtype=np.uint32).reshape(nq, -1)
        max_cand = cand_doc.shape[1]
        cand_seg = np.ascontiguousarray(cand_seg, dtype=np.uint32).reshape(nq, max_cand)
        cand_bm25 = np.ascontiguousarray(cand_bm25, dtype=np.float32).reshape(nq, max_cand)
        cand_count = np.ascontiguousarray(cand_count, dtype=np.uint32)
        out_doc = np.zeros((nq, k_out), dtype=np.uint32)
        out_seg = np.zeros((nq, k_out), dtype=np.uint32)
        out_score = np.zeros((nq, k_out), dtype=np.float32)
        out_vec = np.zeros((nq, k_out), dtype=np.float32)
        out_count = np.zeros(nq, dtype=np.uint32)
        N.check(self._lib.slg_rerank_batch(self._h, nq, _ptr(qvecs), _ptr(alpha), _ptr(cand_doc),
                                           _ptr(cand_seg), _ptr(cand_bm25), _ptr(cand_count),
                                           max_cand, k_out, _ptr(out_doc), _ptr(out_seg),
                                           _ptr(out_score), _ptr(out_vec), _ptr(out_count)))
        return out_doc, out_seg, out_score, out_vec, out_count

    def rerank_multi_batch(self, qvecs, alpha, cand_doc, cand_seg, cand_bm25, cand_count, k_out: int,
                           boost=None):
        """Hybrid rerank with several vector clauses (compute_hybrid_score, api/reader.rs:225-254):
        qvecs [nq, n_clauses, dim], alpha / boost [nq, n_clauses]."""
        qvecs = np.ascontiguousarray(qvecs, dtype=np.float32)
        nq, nc = qvecs.shape[0], qvecs.shape[1]
        alpha = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float32), (nq, nc)))
        bst = None if boost is None else \
            np.ascontiguousarray(np.broadcast_to(np.asarray(boost, dtype=np.float32), (nq, nc)))
        cand_doc = np.ascontiguousarray(cand_doc, dtype=np.uint32).reshape(nq, -1)
        max_cand = cand_doc.shape[1]
        cand_seg = np.ascontiguousarray(cand_seg, dtype=np.uint32).reshape(nq, max_cand)
        cand_bm25 = np.ascontiguousarray(cand_bm25, dtype=np.float32).reshape(nq, max_cand)
        cand_count = np.ascontiguousarray(cand_count, dtype=np.uint32)
        out_doc = np.zeros((nq, k_out), dtype=np.uint32)
        out_seg = np.zeros((nq, k_out), dtype=np.uint32)
        out_score = np.zeros((nq, k_out), dtype=np.float32)
        out_vec = np.zeros((nq, k_out), dtype=np.float32)
        out_count = np.zeros(nq, dtype=np.uint32)
        N.check(self._lib.slg_rerank_multi_batch(
            self._h, nq, nc, _ptr(qvecs), _ptr(alpha), _ptr(bst), _ptr(cand_doc), _ptr(cand_seg),
            _ptr(cand_bm25), _ptr(cand_count), max_cand, k_out, _ptr(out_doc), _ptr(out_seg),
            _ptr(out_score), _ptr(out_vec), _ptr(out_count)))
        return out_doc, out_seg, out_score, out_vec, out_count

    def add_vector_field(self, per_segment) -> int:
        """Stage one more vector field (vectors/mod.rs:10-17: a VectorStore per field).  per_segment[s]
        = (metric, vec_offsets u32[n_docs], vec_values f32[rows, dim]) or None (segment s has no
        vectors in it).  -> field id (>= 1; 0 is the field of the segment descriptors)."""
        descs = (N.VectorFieldDesc * len(per_segment))()
        keep = []
        for s, f in enumerate(per_segment):
            if f is None:
                continue
            metric, offs, vals = f
            offs = np.ascontiguousarray(offs, dtype=np.uint32)
            vals = np.ascontiguousarray(vals, dtype=np.float32)
            keep += [offs, vals]
            descs[s].vec_dim, descs[s].vec_metric = vals.shape[1], int(metric)
            descs[s].vec_offsets, descs[s].vec_values = offs.ctypes.data, vals.ctypes.data
            descs[s].vec_rows = vals.shape[0]
        rc = self._lib.slg_index_add_vector_field(self._h, C.addressof(descs), len(per_segment))
        if rc < 0:
            N.check(rc)
        return rc

    def rerank_fields_batch(self, clause_field, qvecs, alpha, cand_doc, cand_seg, cand_bm25, cand_count,
                            k_out: int, boost=None):
        """Hybrid rerank whose clauses name different vector fields (api/reader.rs:225-254).
        clause_field [n_clauses] field ids; qvecs [nq, sum of the clause dims] (a query's clause
        vectors one after another); alpha / boost [nq, n_clauses]."""
        cf = np.ascontiguousarray(clause_field, dtype=np.uint32)
        nc = len(cf)
        qvecs = np.ascontiguousarray(qvecs, dtype=np.float32)
        nq = qvecs.shape[0]
        alpha = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha, dtype=np.float32), (nq, nc)))
        bst = None if boost is None else \
            np.ascontiguousarray(np.broadcast_to(np.asarray(boost, dtype=np.float32), (nq, nc)))
        cand_doc = np.ascontiguousarray(cand_doc, dtype=np.uint32).reshape(nq, -1)
        max_cand = cand_doc.shape[1]
        cand_seg = np.ascontiguousarray(cand_seg, dtype=np.uint32).reshape(nq, max_cand)
        cand_bm25 = np.ascontiguousarray(cand_bm25, dtype=np.float32).reshape(nq, max_cand)
        cand_count = np.ascontiguousarray(cand_count, dtype=np.uint32)
        out_doc = np.zeros((nq, k_out), dtype=np.uint32)
        out_seg = np.zeros((nq, k_out), dtype=np.uint32)
        out_score = np.zeros((nq, k_out), dtype=np.float32)
        out_vec = np.zeros((nq, k_out), dtype=np.float32)
        out_count = np.zeros(nq, dtype=np.uint32)
        N.check(self._lib.slg_rerank_fields_batch(
            self._h, nq, nc, _ptr(cf), _ptr(qvecs), _ptr(alpha), _ptr(bst), _ptr(cand_doc), _ptr(cand_seg),
            _ptr(cand_bm25), _ptr(cand_count), max_cand, k_out, _ptr(out_doc), _ptr(out_seg),
            _ptr(out_score), _ptr(out_vec), _ptr(out_count)))
        return out_doc, out_seg, out_score, out_vec, out_count

    def rerank_batch_device(self, nq, d_qvecs, d_alpha, d_cand_doc, d_cand_seg, d_cand_bm25,
                            d_cand_count, max_cand, k_out, d_out_doc, d_out_seg, d_out_score,
                            d_out_vec, d_out_count) -> None:
        """Device-pointer form (ints), asynchronous on the index stream."""
        N.check(self._lib.slg_rerank_batch_device(
            self._h, nq, d_qvecs, d_alpha, d_cand_doc, d_cand_seg, d_cand_bm25, d_cand_count,
            max_cand, k_out, d_out_doc, d_out_seg, d_out_score, d_out_vec, d_out_count))

    def rerank_multi_batch_device(self, nq, n_clauses, d_qvecs, d_alpha, d_boost, d_cand_doc, d_cand_seg,
                                  d_cand_bm25, d_cand_count, max_cand, k_out, d_out_doc, d_out_seg,
                                  d_out_score, d_out_vec, d_out_count) -> None:
        """Device-pointer form of rerank_multi_batch (d_qvecs [nq, n_clauses, dim]; d_boost may be
        None), asynchronous on the index stream.  >= 2 cosine clauses run on the matrix cores."""
        N.check(self._lib.slg_rerank_multi_batch_device(
            self._h, nq, n_clauses, d_qvecs, d_alpha, d_boost, d_cand_doc, d_cand_seg, d_cand_bm25,
            d_cand_count, max_cand, k_out, d_out_doc, d_out_seg, d_out_score, d_out_vec, d_out_count))

    def merge_shards_device(self, n_shards, nq, k, d_doc, d_seg, d_score, d_count, seg_stride,
                            d_out_doc, d_out_seg, d_out_score, d_out_count) -> None:
        N.check(self._lib.slg_merge_shards_device(self._h, n_shards, nq, k, d_doc, d_seg, d_score,
                                                  d_count, seg_stride, d_out_doc, d_out_seg,
                                                  d_out_score, d_out_count))


def shard_unique_id() -> bytes:
    """slg_shard_unique_id: the 128-byte id rank 0 creates and hands to the other ranks of a shard
    group out of band (ncclGetUniqueId)."""
    buf = C.create_string_buffer(N.SHARD_UNIQUE_ID_BYTES)
    N.check(N.load().slg_shard_unique_id(buf, N.SHARD_UNIQUE_ID_BYTES))
    return buf.raw


class ShardGroup:
    """This rank's membership in an index-sharded search (slg_shard_group): its GpuIndex holds the
    segments of shard `rank`; the RCCL communicator lives behind the C ABI, no torch involved.
    Creation is collective: every rank constructs its ShardGroup with the same unique_id."""

    def __init__(self, index: GpuIndex, rank: int, world: int, unique_id: bytes, segs_per_rank: Optional[int] = None):
        self.index, self.rank, self.world = index, rank, world
        self.segs_per_rank = index.n_segs if segs_per_rank is None else int(segs_per_rank)
        self._lib = index._lib
        assert len(unique_id) == N.SHARD_UNIQUE_ID_BYTES
        self._h = self._lib.slg_shard_group_create(index._h, rank, world, unique_id, self.segs_per_rank)
        if not self._h:
            raise N.SlgError(N.last_error_code() or N.ERR_INVALID, N.last_error())

    def stats(self) -> dict:
        """slg_shard_group_stats (with GpuIndex.profile(True)): mean device ms per sharded run fetched
        since the last call — this rank's kernels, the all-gather, the merge."""
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_uint64()
        N.check(self._lib.slg_shard_group_stats(self._h, C.addressof(a), C.addressof(b), C.addressof(c), C.addressof(n)))
        d = max(1, n.value)
        return {"runs": int(n.value), "kernel_ms": a.value / d, "gather_ms": b.value / d, "merge_ms": c.value / d}

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.slg_shard_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class PreparedBatch:
    """A planned query batch with device-resident descriptors and work buffers."""

    def __init__(self, index: GpuIndex, q_offsets, q_terms, q_weights, k: int, strategy: int,
                 q_filter=None, q_leaf=None, q_plan=None, q_tie=None, q_nleaves=None,
                 q_leaf_offsets=None, leaf_group=None, q_group_offsets=None, group_plan=None,
                 group_tie=None, q_node_offsets=None, node_kind=None, node_tie=None, node_parent=None,
                 q_min_match=None):
        self.index = index
        self._lib = index._lib
        q_offsets = np.ascontiguousarray(q_offsets, dtype=np.uint32)
        q_terms = np.ascontiguousarray(q_terms, dtype=np.uint32)
        q_weights = np.ascontiguousarray(q_weights, dtype=np.float32)
        self.nq = len(q_offsets) - 1
        self.k = k
        qf = None if q_filter is None else np.ascontiguousarray(q_filter, dtype=np.int32)
        assert qf is None or len(qf) == self.nq
        ql = None if q_leaf is None else np.ascontiguousarray(q_leaf, dtype=np.uint32)
        qp = None if q_plan is None else np.ascontiguousarray(q_plan, dtype=np.int32)
        qt = None if q_tie is None else np.ascontiguousarray(q_tie, dtype=np.float32)
        qn = None if q_nleaves is None else np.ascontiguousarray(q_nleaves, dtype=np.uint32)
        assert ql is None or len(ql) == len(q_weights)
        assert all(x is None or len(x) == self.nq for x in (qp, qt, qn))
        opt = lambda a: None if a is None else _ptr(a)
        u32 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.uint32)
        qlo, lg, qgo = u32(q_leaf_offsets), u32(leaf_group), u32(q_group_offsets)
        gp = None if group_plan is None else np.ascontiguousarray(group_plan, dtype=np.int32)
        gt = None if group_tie is None else np.ascontiguousarray(group_tie, dtype=np.float32)
        qno, npar = u32(q_node_offsets), u32(node_parent)
        nk = None if node_kind is None else np.ascontiguousarray(node_kind, dtype=np.int32)
        ntie = None if node_tie is None else np.ascontiguousarray(node_tie, dtype=np.float32)
        qmm = u32(q_min_match)
        assert qmm is None or len(qmm) == self.nq
        plans = N.ScorePlans(opt(ql), opt(qp), opt(qt), opt(qn), opt(qlo), opt(lg), opt(qgo), opt(gp), opt(gt),
                             opt(qno), opt(nk), opt(ntie), opt(npar), opt(qmm))
        self._h = self._lib.slg_batch_prepare_plans(
            index._h, self.nq, _ptr(q_offsets), _ptr(q_terms), _ptr(q_weights), C.addressof(plans),
            opt(qf), k, strategy)
        if not self._h:
            raise N.SlgError(N.last_error_code() or N.ERR_INVALID, N.last_error())
        index._batches.add(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.slg_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self) -> None:
        N.check(self._lib.slg_batch_run(self._h))

    def run_sharded(self, group: "ShardGroup", fetch: bool = True, seq: Optional[int] = None):
        """slg_batch_run_sharded: this rank's segments, ONE ncclAllGather of the result blocks,
        device merge.  fetch=True -> merged (doc, seg, score, count) host arrays (waits);
        fetch=False -> None, the merged block stays on the device (sharded_device_results).
        seq: slg_batch_run_sharded_seq — the run's number in the group's order of collectives."""
        if not fetch:
            if seq is None:
                N.check(self._lib.slg_batch_run_sharded(self._h, group._h, None, None, None, None))
            else:
                N.check(self._lib.slg_batch_run_sharded_seq(self._h, group._h, int(seq), None, None, None, None))
            return None
        nq, k = self.nq, self.k
        out_doc = np.zeros((nq, k), dtype=np.uint32)
        out_seg = np.zeros((nq, k), dtype=np.uint32)
        out_score = np.zeros((nq, k), dtype=np.float32)
        out_count = np.zeros(nq, dtype=np.uint32)
        N.check(self._lib.slg_batch_run_sharded(self._h, group._h, _ptr(out_doc), _ptr(out_seg),
                                                _ptr(out_score), _ptr(out_count)))
        return out_doc, out_seg, out_score, out_count

    def fetch_sharded(self):
        """Waits for run_sharded(fetch=False) -> merged (doc, seg, score, count) host arrays."""
        nq, k = self.nq, self.k
        out_doc = np.zeros((nq, k), dtype=np.uint32)
        out_seg = np.zeros((nq, k), dtype=np.uint32)
        out_score = np.zeros((nq, k), dtype=np.float32)
        out_count = np.zeros(nq, dtype=np.uint32)
        N.check(self._lib.slg_batch_fetch_sharded(self._h, _ptr(out_doc), _ptr(out_seg), _ptr(out_score),
                                                  _ptr(out_count)))
        return out_doc, out_seg, out_score, out_count

    def sharded_device_results(self):
        """-> (d_doc, d_seg, d_score, d_count) raw device addresses of the merged top-k."""
        ptrs = [C.c_void_p() for _ in range(4)]
        N.check(self._lib.slg_batch_sharded_device_results(self._h, *[C.addressof(p) for p in ptrs]))
        return tuple(p.value for p in ptrs)

    def set_stream(self, hip_stream) -> None:
        """Run this batch on its own hipStream_t so several batches can be in flight at once
        (None: back to the index stream)."""
        h = C.c_void_p(-1) if hip_stream is None else C.c_void_p(int(hip_stream) or None)
        N.check(self._lib.slg_batch_set_stream(self._h, h))

    def rerank_device(self, n_clauses, d_qvecs, d_alpha, d_boost, k_out, d_out_doc, d_out_seg, d_out_score,
                      d_out_vec, d_out_count) -> None:
        """slg_batch_rerank_device: rerank this batch's own device results on the batch's stream."""
        N.check(self._lib.slg_batch_rerank_device(self._h, n_clauses, d_qvecs, d_alpha, d_boost, k_out, d_out_doc,
                                                  d_out_seg, d_out_score, d_out_vec, d_out_count))

    def sync(self) -> None:
        N.check(self._lib.slg_batch_sync(self._h))

    def info(self):
        npost, nsl, nbytes = C.c_uint64(), C.c_uint32(), C.c_uint64()
        N.check(self._lib.slg_batch_info(self._h, C.addressof(npost), C.addressof(nsl),
                                         C.addressof(nbytes)))
        return {"n_postings": npost.value, "n_slices": nsl.value,
                "algorithmic_bytes": nbytes.value}

    def skip_counts(self):
        """-> (postings of pruning-classified lists the plan covered, those never loaded) of the
        last run (block skipping, query/wand.rs:205-265)."""
        probed, skipped = C.c_uint64(), C.c_uint64()
        N.check(self._lib.slg_batch_skip_counts(self._h, C.addressof(probed), C.addressof(skipped)))
        return probed.value, skipped.value

    def device_results(self):
        """-> (d_doc, d_seg, d_score, d_count) raw device addresses."""
        ptrs = [C.c_void_p() for _ in range(4)]
        N.check(self._lib.slg_batch_device_results(self._h, *[C.addressof(p) for p in ptrs]))
        return tuple(p.value for p in ptrs)

    def device_result_block(self):
        """-> (address, n_bytes) of the contiguous doc|seg|score|count block."""
        ptr, nb = C.c_void_p(), C.c_uint64()
        N.check(self._lib.slg_batch_device_result_block(self._h, C.addressof(ptr), C.addressof(nb)))
        return ptr.value, nb.value

    def fetch(self, want_stats: bool = False):
        nq, k = self.nq, self.k
        out_doc = np.zeros((nq, k), dtype=np.uint32)
        out_seg = np.zeros((nq, k), dtype=np.uint32)
        out_score = np.zeros((nq, k), dtype=np.float32)
        out_count = np.zeros(nq, dtype=np.uint32)
        stats = (N.Stats * max(nq, 1))() if want_stats else None
        N.check(self._lib.slg_batch_fetch(self._h, _ptr(out_doc), _ptr(out_seg), _ptr(out_score),
                                          _ptr(out_count),
                                          None if stats is None else C.addressof(stats)))
        if want_stats:
            return out_doc, out_seg, out_score, out_count, stats
        return out_doc, out_seg, out_score, out_count
