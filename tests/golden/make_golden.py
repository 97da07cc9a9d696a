#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ (run in the build container,
where /root/reference is mounted; the fixtures travel to the GPU box, the reference does not).

Every expected result below is produced by OUR CPU oracle (oracle/slo_oracle.c), because the
reference is Rust and cannot run here: the fixtures pin the GPU path and the oracle against
each other and against accidental change, not against the Rust binary.  Inputs:
  * recipes.npz : examples/recipes/data.jsonl (the reference's own example corpus, data not
    source) indexed by searchlite_amd.segment.SegmentBuilder (default tokenizer, ids sorted),
    fields title/description/instructions, product BM25 defaults k1=0.9 b=0.4 (README.md:15);
    10 two-term OR queries, limit 10 (k = 11), "execution": wand  -> BASELINE config 1.
  * recipes_default.npz : the same corpus with all four text fields (incl. `text`), queries on the
    DEFAULT fields (multi-field leaves, Sum plan) -> BASELINE config 1 as an unmodified request.
  * pruning40.npz : a 40-doc / 7-word corpus in the shape of tests/pruning.rs:44-104
    (k1=1.2, b=0.75, bmw_block_size 4), 5 three-term queries, limit 5.
  * two_segments.npz : tests/smoke.rs:853-950 — equal scores across two segments.
  * rerank16.npz : 64 vectors x 16 dims, cosine + L2, alpha blend (tests/vector_search.rs).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from searchlite_amd.segment import (SegmentBuilder, fold_terms, parse_query_terms,  # noqa: E402
                                    resolve_query)
from tests.util import random_queries, random_segment  # noqa: E402

REF = "/root/reference/examples/recipes/data.jsonl"


def seg_arrays(seg, prefix=""):
    d = {prefix + "n_docs": np.uint32(seg.n_docs), prefix + "term_offsets": seg.term_offsets,
         prefix + "doc_ids": seg.doc_ids, prefix + "tfs": seg.tfs,
         prefix + "field_avgdl": seg.field_avgdl, prefix + "docs": np.float32(seg.docs),
         prefix + "k1": np.float32(seg.k1), prefix + "b": np.float32(seg.b),
         prefix + "n_fields": np.uint32(len(seg.field_doc_len))}
    if seg.term_field is not None:
        d[prefix + "term_field"] = seg.term_field
    for i, a in enumerate(seg.field_doc_len):
        if a is not None:
            d[prefix + f"doc_len{i}"] = a
    return d


def recipes():
    fields = ["title", "description", "instructions"]
    b = SegmentBuilder(fields, k1=0.9, b=0.4)
    with open(REF) as f:
        for line in f:
            doc = json.loads(line)
            b.add_document(doc["doc_id"], {k: doc.get(k) for k in fields})
    seg = b.build()
    queries = [("description", "tomato basil"), ("description", "quick weeknight"),
               ("title", "chicken soup"), ("title", "orzo spinach"),
               ("instructions", "simmer garlic"), ("instructions", "bake oven"),
               ("description", "vegan chili"), ("title", "shrimp curry"),
               ("instructions", "whisk butter"), ("description", "salad lemon")]
    offs, ids, ws, keys = [0], [], [], []
    for fld, q in queries:
        folded = fold_terms(parse_query_terms(q, fld))
        i, w = resolve_query([seg], folded)
        ids.append(i)
        ws.append(w)
        keys.append([k for k, _ in folded])
        offs.append(offs[-1] + len(folded))
    ids = np.concatenate(ids)
    ws = np.concatenate(ws)
    offs = np.array(offs, dtype=np.uint32)
    k = 11
    want = O.search_batch([seg], offs, ids, ws, k, strategy=O.WAND)
    bm = O.search_batch([seg], offs, ids, ws, k, strategy=O.BM25)
    assert all(np.array_equal(a, b_) for a, b_ in zip(want, bm))
    np.savez_compressed(os.path.join(HERE, "recipes.npz"), **seg_arrays(seg), q_offsets=offs,
                        q_terms=ids, q_weights=ws, k=np.uint32(k), exp_doc=want[0],
                        exp_seg=want[1], exp_score=want[2], exp_count=want[3])
    meta = {"queries": [{"field": f, "query": q, "keys": ks} for (f, q), ks in zip(queries, keys)],
            "ext_ids": seg.ext_ids,
            "top10": [[(seg.ext_ids[int(want[0][qi, i])], float(want[2][qi, i]))
                       for i in range(min(int(want[3][qi]), 10))] for qi in range(len(queries))]}
    json.dump(meta, open(os.path.join(HERE, "recipes.json"), "w"), indent=1)
    print("recipes:", seg.n_docs, "docs", seg.n_postings, "postings; first query top-3:",
          meta["top10"][0][:3])


def recipes_default_fields():
    """BASELINE config 1 on DEFAULT fields: `fields: None` means every text field of the schema
    (api/reader.rs:2576-2586; examples/recipes/schema.json: text, title, description,
    instructions).  A query string's word w becomes ONE ScorePlan leaf fed by the terms
    `<field>:w` of all four fields (query/planner.rs:300-360, weight = group boost x field boost =
    1.0), the leaves are summed (`Sum([Leaf(0..n)])`).  10 two-word queries, limit 10 (k = 11)."""
    from searchlite_amd.segment import default_tokenize, plan_query_string, resolve_plan
    fields = ["text", "title", "description", "instructions"]
    b = SegmentBuilder(fields, k1=0.9, b=0.4)
    with open(REF) as f:
        for line in f:
            doc = json.loads(line)
            b.add_document(doc["doc_id"], {k: doc.get(k) for k in fields})
    seg = b.build()
    queries = ["tomato basil", "quick weeknight", "chicken soup", "orzo spinach", "simmer garlic",
               "bake oven", "vegan chili", "shrimp curry", "whisk butter", "salad lemon"]
    offs, ids, ws, leaves, nl, keys = [0], [], [], [], [], []
    for q in queries:
        words = [t for raw in q.split() for t in default_tokenize(raw)]
        planned, plan, n_leaves = plan_query_string(words, [(f, 1.0) for f in fields])
        i, w, lf = resolve_plan([seg], planned)
        ids.append(i)
        ws.append(w)
        leaves.append(lf)
        nl.append(n_leaves)
        keys.append([p_[0] for p_ in planned])
        offs.append(offs[-1] + len(planned))
    ids, ws, leaves = np.concatenate(ids), np.concatenate(ws), np.concatenate(leaves)
    offs = np.array(offs, dtype=np.uint32)
    nl = np.array(nl, dtype=np.uint32)
    k = 11
    kw = dict(q_leaf=leaves, q_plan=np.zeros(len(queries), np.int32), q_nleaves=nl)
    want = O.search_batch([seg], offs, ids, ws, k, strategy=O.BM25, **kw)
    # Wand adds a leaf's terms in cursor-heap pop order (wand.rs:835-839), Bm25 in term order: with
    # several terms per leaf the f32 sums may differ in the last ulps (the reference's own tests
    # allow 1e-5, tests/pruning.rs:96-101) — the fixture holds the exhaustive (Bm25) result
    wand = O.search_batch([seg], offs, ids, ws, k, strategy=O.WAND, **kw)
    assert np.array_equal(want[3], wand[3]) and np.abs(want[2] - wand[2]).max() < 1e-5
    same_docs = float((want[0] == wand[0]).mean())
    print("recipes (default fields): Wand vs Bm25 identical doc positions:", same_docs,
          "max |dscore|", float(np.abs(want[2] - wand[2]).max()))
    np.savez_compressed(os.path.join(HERE, "recipes_default.npz"), **seg_arrays(seg), q_offsets=offs,
                        q_terms=ids, q_weights=ws, q_leaf=leaves, q_nleaves=nl, k=np.uint32(k),
                        exp_doc=want[0], exp_seg=want[1], exp_score=want[2], exp_count=want[3])
    meta = {"fields": fields, "queries": [{"query": q, "keys": ks} for q, ks in zip(queries, keys)],
            "top10": [[(seg.ext_ids[int(want[0][qi, i])], float(want[2][qi, i]))
                       for i in range(min(int(want[3][qi]), 10))] for qi in range(len(queries))]}
    json.dump(meta, open(os.path.join(HERE, "recipes_default.json"), "w"), indent=1)
    print("recipes (default fields):", seg.n_terms, "terms; first query top-3:", meta["top10"][0][:3])


def pruning40():
    rng = np.random.default_rng(42)
    seg = random_segment(rng, 40, 7, 6, k1=1.2, b=0.75, zipf=False)
    offs, terms, w = random_queries(rng, 5, 3, 7)
    k = 6
    want = O.search_batch([seg], offs, terms, w, k, strategy=O.BM25)
    for strat, bs in ((O.WAND, None), (O.BMW, 4)):
        got = O.search_batch([seg], offs, terms, w, k, strategy=strat, block_size=bs)
        assert all(np.array_equal(a, b_) for a, b_ in zip(got, want))
    np.savez_compressed(os.path.join(HERE, "pruning40.npz"), **seg_arrays(seg), q_offsets=offs,
                        q_terms=terms, q_weights=w, k=np.uint32(k), exp_doc=want[0],
                        exp_seg=want[1], exp_score=want[2], exp_count=want[3])


def two_segments():
    segs = []
    for s in range(2):
        b = SegmentBuilder(["body"], k1=0.9, b=0.4)
        for i in range(3):
            b.add_document(f"doc-s{s}-{i}", {"body": "rust"})
        b.add_document(f"doc-s{s}-x", {"body": "other words here"})
        segs.append(b.build())
    ids = np.array([[sg.term_id("body:rust") for sg in segs]], dtype=np.uint32)
    offs = np.array([0, 1], dtype=np.uint32)
    w = np.array([1.0], dtype=np.float32)
    want = O.search_batch(segs, offs, ids, w, 7, strategy=O.WAND)
    d = {}
    for i, sg in enumerate(segs):
        d.update(seg_arrays(sg, f"s{i}_"))
    np.savez_compressed(os.path.join(HERE, "two_segments.npz"), **d, q_offsets=offs, q_terms=ids,
                        q_weights=w, k=np.uint32(7), exp_doc=want[0], exp_seg=want[1],
                        exp_score=want[2], exp_count=want[3])


def rerank16():
    rng = np.random.default_rng(11)
    n, dim = 64, 16
    vals = rng.standard_normal((n, dim)).astype(np.float32)
    for i in range(n):
        O.normalize_in_place(vals[i])
    offsets = np.arange(n, dtype=np.uint32)
    offsets[5] = offsets[17] = 0xFFFFFFFF  # docs without a vector
    nq, ncand = 4, 32
    q = rng.standard_normal((nq, dim)).astype(np.float32)
    for i in range(nq):
        O.normalize_in_place(q[i])
    cand = np.stack([rng.choice(n, size=ncand, replace=False) for _ in range(nq)]).astype(np.uint32)
    bm25 = (rng.random((nq, ncand)) * 10).astype(np.float32)
    alpha = np.array([0.5, 0.2, 1.0, 0.0], dtype=np.float32)
    out = {}
    for metric, name in ((O.COSINE, "cos"), (O.L2, "l2")):
        docs, scores, vecs = [], [], []
        for i in range(nq):
            d_, s_, v_ = O.rerank(metric, offsets, vals, q[i], float(alpha[i]), cand[i], bm25[i], 10)
            docs.append(d_)
            scores.append(s_)
            vecs.append(v_)
        out[f"exp_doc_{name}"] = np.stack(docs)
        out[f"exp_score_{name}"] = np.stack(scores)
        out[f"exp_vec_{name}"] = np.stack(vecs)
    np.savez_compressed(os.path.join(HERE, "rerank16.npz"), vec_offsets=offsets, vec_values=vals,
                        qvecs=q, cand_doc=cand, cand_bm25=bm25, alpha=alpha, **out)


if __name__ == "__main__":
    recipes()
    recipes_default_fields()
    pruning40()
    two_segments()
    rerank16()
    print("golden fixtures written to", HERE)
