"""Shared helpers for the parity tests (test infrastructure; may use the oracle)."""
from __future__ import annotations

import numpy as np


def random_segment(rng, n_docs, vocab, avg_len, k1=1.2, b=0.75, missing_len_frac=0.0,
                   zipf=True):
    """Small random one-field segment built directly as arrays (doc ids = 0..n_docs-1)."""
    from searchlite_amd.segment import Segment
    lens = rng.integers(max(1, avg_len // 2), avg_len * 2 + 1, size=n_docs)
    if zipf:
        p = 1.0 / np.arange(1, vocab + 1)
        p /= p.sum()
    else:
        p = np.full(vocab, 1.0 / vocab)
    post = [[] for _ in range(vocab)]
    for d in range(n_docs):
        toks = rng.choice(vocab, size=int(lens[d]), p=p)
        t, c = np.unique(toks, return_counts=True)
        for ti, ci in zip(t, c):
            post[int(ti)].append((d, int(ci)))
    offs = np.zeros(vocab + 1, dtype=np.uint64)
    docs, tfs = [], []
    for t in range(vocab):
        for d, c in post[t]:
            docs.append(d)
            tfs.append(c)
        offs[t + 1] = len(docs)
    dl = lens.astype(np.float32)
    if missing_len_frac > 0:
        miss = rng.random(n_docs) < missing_len_frac
        dl[miss] = 0.0
    avg = np.float32(np.float32(lens.sum()) / np.float32(n_docs))
    return Segment(n_docs=n_docs, term_offsets=offs, doc_ids=np.array(docs, dtype=np.uint32),
                   tfs=np.array(tfs, dtype=np.uint32), field_doc_len=[dl],
                   field_avgdl=np.array([avg], dtype=np.float32), docs=float(n_docs), k1=k1, b=b)


def random_multifield_segment(rng, n_docs, vocab, n_fields, avg_len, k1=0.9, b=0.4):
    """Random segment with n_fields text fields over the same vocabulary: term id = f*vocab + w
    (the `field:word` keys of index/postings.rs), per-field doc lengths / avgdl; some docs lack a
    field (length 0)."""
    from searchlite_amd.segment import Segment
    p = 1.0 / np.arange(1, vocab + 1)
    p /= p.sum()
    V = vocab * n_fields
    post = [[] for _ in range(V)]
    lens = np.zeros((n_fields, n_docs), dtype=np.float32)
    for f in range(n_fields):
        fl = max(1, avg_len // (f + 1))
        for d in range(n_docs):
            if rng.random() < 0.1:
                continue  # field missing in this doc
            n = int(rng.integers(max(1, fl // 2), fl * 2 + 1))
            toks = rng.choice(vocab, size=n, p=p)
            t, c = np.unique(toks, return_counts=True)
            for ti, ci in zip(t, c):
                post[f * vocab + int(ti)].append((d, int(ci)))
            lens[f, d] = n
    offs = np.zeros(V + 1, dtype=np.uint64)
    docs, tfs = [], []
    for t in range(V):
        for d, c in post[t]:
            docs.append(d)
            tfs.append(c)
        offs[t + 1] = len(docs)
    avg = np.array([np.float32(lens[f].sum()) / np.float32(n_docs) for f in range(n_fields)], dtype=np.float32)
    tfield = np.repeat(np.arange(n_fields, dtype=np.uint16), vocab)
    return Segment(n_docs=n_docs, term_offsets=offs, doc_ids=np.array(docs, dtype=np.uint32),
                   tfs=np.array(tfs, dtype=np.uint32), field_doc_len=[lens[f] for f in range(n_fields)],
                   field_avgdl=avg, docs=float(n_docs), k1=k1, b=b, term_field=tfield)


def skewed_segment(rng, n_docs, n_lists):
    """One-field segment built list by list (fast for millions of doc ids): list sizes from a few
    postings to ~n_docs/8, each either spread uniformly or packed into a narrow run of doc ids."""
    from searchlite_amd.segment import Segment
    offs, docs, tfs = [0], [], []
    for _ in range(n_lists):
        size = int(min(n_docs // 4, rng.choice([3, 40, 600, 5_000, 40_000, max(8, n_docs // 8)])))
        if rng.random() < 0.35:   # clustered: (almost) consecutive doc ids
            start = int(rng.integers(0, max(1, n_docs - size * 2)))
            d = start + np.unique(rng.integers(0, size * 2, size=size))
        else:
            d = np.unique(rng.integers(0, n_docs, size=size))
        d = d.astype(np.uint32)
        docs.append(d)
        tfs.append(rng.integers(1, 4, size=len(d)).astype(np.uint32))
        offs.append(offs[-1] + len(d))
    dl = rng.integers(5, 60, size=n_docs).astype(np.float32)
    return Segment(n_docs=n_docs, term_offsets=np.array(offs, dtype=np.uint64),
                   doc_ids=np.concatenate(docs), tfs=np.concatenate(tfs), field_doc_len=[dl],
                   field_avgdl=np.array([np.float32(dl.mean())], dtype=np.float32), docs=float(n_docs),
                   k1=1.2, b=0.75)


def random_queries(rng, nq, n_terms, vocab, n_segs=1, lo=0, weights=False):
    offs = (np.arange(nq + 1) * n_terms).astype(np.uint32)
    terms = np.empty((nq * n_terms, n_segs), dtype=np.uint32)
    for q in range(nq):
        t = rng.choice(np.arange(lo, vocab), size=n_terms, replace=False)
        for s in range(n_segs):
            terms[q * n_terms:(q + 1) * n_terms, s] = t
    w = (rng.random(nq * n_terms).astype(np.float32) * 2 + 0.25) if weights \
        else np.ones(nq * n_terms, dtype=np.float32)
    return offs, terms, w


def assert_same_hits(got, want, score_tol=0.0, what=""):
    """got/want = (doc[nq,k], seg[nq,k], score[nq,k], count[nq]).  Identical (seg, doc)
    sequence; scores bit-exact when score_tol == 0, else |d| <= score_tol."""
    gd, gs, gsc, gc = got
    wd, ws, wsc, wc = want
    assert gd.shape == wd.shape, f"{what}: shape {gd.shape} vs {wd.shape}"
    bad = []
    for q in range(len(wc)):
        n = int(wc[q])
        if int(gc[q]) != n:
            bad.append((q, "count", int(gc[q]), n))
            continue
        if not (np.array_equal(gd[q, :n], wd[q, :n]) and np.array_equal(gs[q, :n], ws[q, :n])):
            i = int(np.argmax((gd[q, :n] != wd[q, :n]) | (gs[q, :n] != ws[q, :n])))
            bad.append((q, f"doc@{i}", (int(gs[q, i]), int(gd[q, i]), float(gsc[q, i])),
                        (int(ws[q, i]), int(wd[q, i]), float(wsc[q, i]))))
            continue
        if score_tol == 0.0:
            if not np.array_equal(gsc[q, :n].view(np.uint32), wsc[q, :n].view(np.uint32)):
                i = int(np.argmax(gsc[q, :n].view(np.uint32) != wsc[q, :n].view(np.uint32)))
                bad.append((q, f"score@{i}", float(gsc[q, i]), float(wsc[q, i])))
        else:
            d = np.abs(gsc[q, :n].astype(np.float64) - wsc[q, :n].astype(np.float64))
            if d.size and d.max() > score_tol:
                bad.append((q, "score", float(d.max()), score_tol))
    assert not bad, f"{what}: {len(bad)} queries differ, first: {bad[:5]}"


GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def load_golden(name):
    """-> (segments, npz) for a fixture written by tests/golden/make_golden.py."""
    import os
    from searchlite_amd.segment import Segment
    z = np.load(os.path.join(GOLDEN, name))

    def seg(prefix):
        nf = int(z[prefix + "n_fields"])
        lens = [z[prefix + f"doc_len{i}"] if (prefix + f"doc_len{i}") in z else None
                for i in range(nf)]
        return Segment(n_docs=int(z[prefix + "n_docs"]), term_offsets=z[prefix + "term_offsets"],
                       doc_ids=z[prefix + "doc_ids"], tfs=z[prefix + "tfs"], field_doc_len=lens,
                       field_avgdl=z[prefix + "field_avgdl"], docs=float(z[prefix + "docs"]),
                       k1=float(z[prefix + "k1"]), b=float(z[prefix + "b"]),
                       term_field=z[prefix + "term_field"] if (prefix + "term_field") in z else None)

    if "n_docs" in z:
        return [seg("")], z
    segs, i = [], 0
    while f"s{i}_n_docs" in z:
        segs.append(seg(f"s{i}_"))
        i += 1
    return segs, z


def golden_expected(z):
    return z["exp_doc"], z["exp_seg"], z["exp_score"], z["exp_count"]
