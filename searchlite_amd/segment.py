"""Host-side segment model: the arrays searchlite hands its scorer, plus the small pieces
of host logic needed to build them from documents and to turn a query string into the folded
term list (so config 1 — the recipes corpus — can be replayed without the Rust crate).

Reference anchors (relative to /root/reference/searchlite-core/src):
  analysis/tokenizer.rs:7-29   default tokenizer
  api/writer.rs:126,176-181    documents ordered by external id string -> internal DocId
  index/segment.rs:641-698     doc_len, tf, "field:term" keys
  index/postings.rs:56-60      terms sorted by key
  index/segment.rs:946-957     avg field length = total tokens as f32 / total docs as f32
  api/query.rs:20-98           query-string split (plain terms, field:term)
  api/reader.rs:2971-2983      duplicate keys folded, weights summed
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

NO_TERM = 0xFFFFFFFF
NO_VECTOR = 0xFFFFFFFF


@dataclass
class Segment:
    n_docs: int
    term_offsets: np.ndarray            # u64[V+1]
    doc_ids: np.ndarray                 # u32[P]
    tfs: np.ndarray                     # u32[P]
    field_doc_len: List[Optional[np.ndarray]]   # per field f32[n_docs] (0 = missing) or None
    field_avgdl: np.ndarray             # f32[F]
    docs: float                         # live docs
    k1: float = 0.9                     # product defaults (README.md:15)
    b: float = 0.4
    term_field: Optional[np.ndarray] = None     # u16[V]
    deleted: Optional[np.ndarray] = None        # u8 bitmap
    vec_dim: int = 0
    vec_metric: int = 0
    vec_offsets: Optional[np.ndarray] = None    # u32[n_docs]
    vec_values: Optional[np.ndarray] = None     # f32[rows, dim]
    # host-side dictionary (not shipped to the device)
    fields: List[str] = field(default_factory=lambda: ["body"])
    term_dict: Optional[Dict[str, int]] = None  # "field:term" -> term id
    ext_ids: Optional[List[str]] = None

    def __post_init__(self):
        self.term_offsets = np.ascontiguousarray(self.term_offsets, dtype=np.uint64)
        self.doc_ids = np.ascontiguousarray(self.doc_ids, dtype=np.uint32)
        self.tfs = np.ascontiguousarray(self.tfs, dtype=np.uint32)
        self.field_avgdl = np.ascontiguousarray(self.field_avgdl, dtype=np.float32)
        self.field_doc_len = [None if a is None else np.ascontiguousarray(a, dtype=np.float32)
                              for a in self.field_doc_len]
        if self.term_field is not None:
            self.term_field = np.ascontiguousarray(self.term_field, dtype=np.uint16)
        if self.deleted is not None:
            self.deleted = np.ascontiguousarray(self.deleted, dtype=np.uint8)
        if self.vec_offsets is not None:
            self.vec_offsets = np.ascontiguousarray(self.vec_offsets, dtype=np.uint32)
        if self.vec_values is not None:
            self.vec_values = np.ascontiguousarray(self.vec_values, dtype=np.float32)

    @property
    def n_terms(self) -> int:
        return len(self.term_offsets) - 1

    @property
    def n_postings(self) -> int:
        return int(self.term_offsets[-1])

    def df(self, term_id: int) -> int:
        return int(self.term_offsets[term_id + 1] - self.term_offsets[term_id])

    def postings(self, term_id: int) -> Tuple[np.ndarray, np.ndarray]:
        a, b = int(self.term_offsets[term_id]), int(self.term_offsets[term_id + 1])
        return self.doc_ids[a:b], self.tfs[a:b]

    def term_id(self, key: str) -> int:
        if self.term_dict is None:
            raise KeyError("segment has no term dictionary")
        return self.term_dict.get(key, NO_TERM)

    def set_deleted(self, doc_ids: Iterable[int]) -> None:
        """Tombstone documents (index/segment.rs live_docs / is_deleted)."""
        bm = np.zeros((self.n_docs + 7) // 8, dtype=np.uint8) if self.deleted is None \
            else self.deleted.copy()
        n_new = 0
        for d in doc_ids:
            if not (bm[d >> 3] >> (d & 7)) & 1:
                bm[d >> 3] |= np.uint8(1 << (d & 7))
                n_new += 1
        self.deleted = bm
        self.docs = float(self.docs - n_new)


def default_tokenize(text: str) -> List[str]:
    """analysis/tokenizer.rs:7-29: split on non-alphanumeric chars, ASCII-lowercase."""
    out: List[str] = []
    cur: List[str] = []
    for ch in text:
        if ch.isalnum():
            cur.append(ch.lower() if ch.isascii() else ch)
        elif cur:
            out.append("".join(cur))
            cur = []
    if cur:
        out.append("".join(cur))
    return out


class SegmentBuilder:
    """Builds one segment the way SegmentWriter::write_segment does for text fields."""

    def __init__(self, fields: Sequence[str], k1: float = 0.9, b: float = 0.4):
        self.fields = list(fields)
        self.k1, self.b = k1, b
        self._docs: Dict[str, Dict[str, List[str]]] = {}

    def add_document(self, ext_id: str, values: Dict[str, object]) -> None:
        doc: Dict[str, List[str]] = {}
        for f in self.fields:
            v = values.get(f)
            if v is None:
                continue
            doc[f] = [str(x) for x in v] if isinstance(v, (list, tuple)) else [str(v)]
        self._docs[ext_id] = doc  # a later add with the same id replaces (upsert)

    def build(self) -> Segment:
        ext_ids = sorted(self._docs.keys())  # api/writer.rs:126 BTreeMap<String, Document>
        n = len(ext_ids)
        F = len(self.fields)
        lens = [np.zeros(n, dtype=np.float32) for _ in range(F)]
        present = [False] * F
        totals = [0] * F
        post: Dict[str, List[List[int]]] = {}
        for ord_, eid in enumerate(ext_ids):
            doc = self._docs[eid]
            for fi, f in enumerate(self.fields):
                if f not in doc:
                    continue
                present[fi] = True
                dl = 0
                for text in doc[f]:
                    toks = default_tokenize(text)
                    dl += len(toks)
                    for t in toks:
                        key = f"{f}:{t}"
                        lst = post.setdefault(key, [])
                        if lst and lst[-1][0] == ord_:
                            lst[-1][1] += 1   # index/postings.rs:33-41
                        else:
                            lst.append([ord_, 1])
                totals[fi] += dl
                lens[fi][ord_] = dl           # "_len:<field>" fast column
        keys = sorted(post.keys())            # index/postings.rs:56-60
        offs = np.zeros(len(keys) + 1, dtype=np.uint64)
        docs_l: List[int] = []
        tfs_l: List[int] = []
        tfield = np.zeros(len(keys), dtype=np.uint16)
        fidx = {f: i for i, f in enumerate(self.fields)}
        for i, k in enumerate(keys):
            for d, tf in post[k]:
                docs_l.append(d)
                tfs_l.append(tf)
            offs[i + 1] = len(docs_l)
            tfield[i] = fidx[k.split(":", 1)[0]]
        avg = np.array([np.float32(totals[i]) / np.float32(n) if n else np.float32(0)
                        for i in range(F)], dtype=np.float32)
        return Segment(n_docs=n, term_offsets=offs,
                       doc_ids=np.array(docs_l, dtype=np.uint32),
                       tfs=np.array(tfs_l, dtype=np.uint32),
                       field_doc_len=[lens[i] if present[i] else None for i in range(F)],
                       field_avgdl=avg, docs=float(n), k1=self.k1, b=self.b,
                       term_field=tfield, fields=list(self.fields),
                       term_dict={k: i for i, k in enumerate(keys)}, ext_ids=ext_ids)


def parse_query_terms(query: str, default_field: str) -> List[Tuple[str, float]]:
    """Plain-term subset of api/query.rs:20-98 + query-time analysis (api/reader.rs:1037-1046):
    whitespace split, optional `field:` prefix, default tokenizer.  Phrases / -terms are outside
    the GPU eligibility predicate and raise."""
    if '"' in query:
        raise ValueError("phrase queries are not GPU-eligible")
    out: List[Tuple[str, float]] = []
    for raw in query.split():
        if raw.startswith("-"):
            raise ValueError("NOT terms are not GPU-eligible")
        if ":" in raw:
            f, rest = raw.split(":", 1)
        else:
            f, rest = default_field, raw
        for tok in default_tokenize(rest):
            out.append((f"{f}:{tok}", 1.0))
    return out


def fold_terms(keys_weights: Sequence[Tuple[str, float]]) -> List[Tuple[str, float]]:
    """api/reader.rs:2971-2983: identical keys collapse to one term, weights summed (f32); the
    first occurrence fixes the leaf position."""
    order: List[str] = []
    acc: Dict[str, np.float32] = {}
    for k, w in keys_weights:
        if k not in acc:
            acc[k] = np.float32(0.0)
            order.append(k)
        acc[k] = np.float32(acc[k] + np.float32(w))
    return [(k, float(acc[k])) for k in order]


def resolve_query(segments: Sequence[Segment], folded: Sequence[Tuple[str, float]]):
    """-> (term_ids[n_terms, n_segs] u32, weights[n_terms] f32) for the C ABI."""
    ids = np.full((len(folded), len(segments)), NO_TERM, dtype=np.uint32)
    for i, (key, _) in enumerate(folded):
        for s, seg in enumerate(segments):
            ids[i, s] = seg.term_id(key)
    w = np.array([x[1] for x in folded], dtype=np.float32)
    return ids, w


# ---- score plans: how the reference's planner maps scored terms to leaves -----------
PLAN_SUM, PLAN_DISMAX = 0, 1


def plan_query_string(words: Sequence[str], fields: Sequence[Tuple[str, float]]):
    """Multi-field query string / multi_match most_fields-per-word shape (query/planner.rs:
    300-360): every word is one leaf, all its `field:word` terms add into it, the leaves are
    summed.  fields = [(field, boost)].  -> (keys_with_weight_and_leaf, plan, n_leaves)"""
    out = []
    for li, wd in enumerate(words):
        for f, boost in fields:
            out.append((f"{f}:{wd}", float(boost), li))
    return out, PLAN_SUM, len(words)


def plan_best_fields(words: Sequence[str], fields: Sequence[Tuple[str, float]]):
    """multi_match best_fields (query/planner.rs:376-396): one leaf per FIELD (all words of the
    query add into their field's leaf), DisMax over the leaves."""
    out = []
    for wd in words:
        for li, (f, boost) in enumerate(fields):
            out.append((f"{f}:{wd}", float(boost), li))
    return out, PLAN_DISMAX, len(fields)


def plan_most_fields(words: Sequence[str], fields: Sequence[Tuple[str, float]]):
    """multi_match most_fields / cross_fields scoring (query/planner.rs:397-409): ONE leaf."""
    out = [(f"{f}:{wd}", float(boost), 0) for wd in words for f, boost in fields]
    return out, PLAN_SUM, 1


def plan_dis_max_terms(terms: Sequence[Tuple[str, str, float]]):
    """dis_max over term children (query/planner.rs:445-470): one leaf per child."""
    out = [(f"{f}:{v}", float(boost), i) for i, (f, v, boost) in enumerate(terms)]
    return out, PLAN_DISMAX, len(terms)


def resolve_plan(segments: Sequence[Segment], planned):
    """planned = [(key, weight, leaf)] -> (term_ids[n, n_segs], weights[n], leaves[n])."""
    ids = np.full((len(planned), len(segments)), NO_TERM, dtype=np.uint32)
    for i, (key, _, _) in enumerate(planned):
        for s, seg in enumerate(segments):
            ids[i, s] = seg.term_id(key)
    return ids, np.array([x[1] for x in planned], dtype=np.float32), \
        np.array([x[2] for x in planned], dtype=np.uint32)
