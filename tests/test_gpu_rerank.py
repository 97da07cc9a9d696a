"""GPU rerank (the gpu::rerank slot) vs the oracle.  The reference sums the dot product left to
right in f32 (vectors/mod.rs:111); the kernel reduces lane-parallel, so vector scores agree to
|d| <= 1e-5 (north_star tolerance 1e-4) and orderings are checked away from near-ties."""
import os

import numpy as np
import pytest

from tests.util import GOLDEN

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _segment_with_vectors(n_docs, offsets, values, metric):
    from searchlite_amd.segment import Segment
    return Segment(n_docs=n_docs, term_offsets=[0, 1], doc_ids=[0], tfs=[1],
                   field_doc_len=[np.ones(n_docs, np.float32)], field_avgdl=[1.0],
                   docs=float(n_docs), vec_dim=values.shape[1], vec_metric=metric,
                   vec_offsets=offsets, vec_values=values)


def _check(got, want_doc, want_score, want_vec, what):
    gd, gs, gsc, gv, gc = got
    for q in range(len(want_doc)):
        n = len(want_doc[q])
        assert gc[q] == n, what
        assert np.abs(gsc[q, :n] - want_score[q]).max() <= TOL, what
        # same docs; order may differ only between candidates closer than the tolerance
        assert sorted(gd[q, :n]) == sorted(want_doc[q]) or \
            np.abs(np.sort(gsc[q, :n]) - np.sort(want_score[q])).max() <= TOL, what
        for i in range(n):
            if gd[q, i] != want_doc[q][i]:
                j = list(want_doc[q]).index(gd[q, i]) if gd[q, i] in want_doc[q] else None
                assert j is not None and abs(want_score[q][j] - want_score[q][i]) <= 2 * TOL, what
        if want_vec is not None:
            m = {int(d): float(v) for d, v in zip(want_doc[q], want_vec[q])}
            for i in range(n):
                if int(gd[q, i]) in m:
                    assert abs(gv[q, i] - m[int(gd[q, i])]) <= TOL or m[int(gd[q, i])] < -1e30, what


@pytest.mark.parametrize("metric,name", [(0, "cos"), (1, "l2")])
def test_rerank_golden(metric, name):
    import searchlite_amd as sa
    z = np.load(os.path.join(GOLDEN, "rerank16.npz"))
    seg = _segment_with_vectors(64, z["vec_offsets"], z["vec_values"], metric)
    nq, nc = z["cand_doc"].shape
    with sa.GpuIndex([seg]) as ix:
        got = ix.rerank_batch(z["qvecs"], z["alpha"], z["cand_doc"], np.zeros((nq, nc), np.uint32),
                              z["cand_bm25"], np.full(nq, nc, np.uint32), 10)
    _check(got, z[f"exp_doc_{name}"], z[f"exp_score_{name}"], z[f"exp_vec_{name}"], name)


@pytest.mark.parametrize("dim,ncand,k_out", [(768, 1001, 10), (100, 300, 70), (6, 40, 40),
                                             # whole-row paths of 1 / 2 / 3 chunks with a ragged last chunk,
                                             # and the piecewise path (dim > 768, dim % 4 != 0)
                                             (256, 200, 10), (260, 200, 10), (384, 500, 20), (516, 200, 10),
                                             (1000, 200, 10), (1030, 100, 10)])
def test_rerank_random(oracle, dim, ncand, k_out):
    """config 5 shape: BM25 top-1000 -> cosine rerank -> top-10 (dim 768)."""
    import searchlite_amd as sa
    from searchlite_amd import corpus
    rng = np.random.default_rng(dim)
    n = 5000
    vals = corpus.unit_vectors(n, dim, seed=11)
    offsets = np.arange(n, dtype=np.uint32)
    offsets[rng.choice(n, size=50, replace=False)] = 0xFFFFFFFF
    rows = vals  # offsets index rows directly
    nq = 6
    q = corpus.unit_vectors(nq, dim, seed=12)
    cand = np.stack([rng.choice(n, size=ncand, replace=False) for _ in range(nq)]).astype(np.uint32)
    bm = (rng.random((nq, ncand)) * 20).astype(np.float32)
    cnt = np.full(nq, ncand, np.uint32)
    cnt[1] = ncand // 2
    alpha = np.array([0.5, 0.5, 0.2, 0.8, 1.0, 0.0], np.float32)
    for metric in (0, 1):
        seg = _segment_with_vectors(n, offsets, rows, metric)
        with sa.GpuIndex([seg]) as ix:
            got = ix.rerank_batch(q, alpha, cand, np.zeros_like(cand), bm, cnt, k_out)
        wd, ws, wv = [], [], []
        for i in range(nq):
            d_, s_, v_ = oracle.rerank(metric, offsets, rows, q[i], float(alpha[i]),
                                       cand[i, :cnt[i]], bm[i, :cnt[i]], k_out)
            wd.append(d_)
            ws.append(s_)
            wv.append(v_)
        _check(got, wd, ws, wv, f"metric {metric} dim {dim}")


def test_rerank_query_without_candidates(oracle):
    """A query whose text pass found nothing (cand_count 0) beside one that found 5."""
    import searchlite_amd as sa
    from searchlite_amd import corpus
    n, dim = 200, 768
    vals = corpus.unit_vectors(n, dim, seed=51)
    seg = _segment_with_vectors(n, np.arange(n, dtype=np.uint32), vals, 0)
    q = corpus.unit_vectors(2, dim, seed=52)
    cand = np.array([[0, 0, 0, 0, 0], [3, 9, 27, 81, 150]], np.uint32)
    bm = np.ones((2, 5), np.float32)
    with sa.GpuIndex([seg]) as ix:
        gd, gs, gsc, gv, gc = ix.rerank_batch(q, 0.5, cand, np.zeros_like(cand), bm, np.array([0, 5], np.uint32), 3)
        md, ms, msc, mv, mc = ix.rerank_multi_batch(q.reshape(2, 1, dim), 0.5, cand, np.zeros_like(cand), bm,
                                                     np.array([0, 5], np.uint32), 3, boost=np.full((2, 1), 2.0, np.float32))
    assert list(gc) == [0, 3] and list(mc) == [0, 3]
    d_, s_, v_ = oracle.rerank(0, np.arange(n, dtype=np.uint32), vals, q[1], 0.5, cand[1], bm[1], 3)
    assert list(gd[1]) == list(d_) and np.abs(gsc[1] - s_).max() <= TOL


def test_rerank_candidates_from_two_segments(oracle):
    """Candidates of one query come from several segments (api/reader.rs:2670-2778 merges them before
    the hybrid score): each row is fetched from its own segment's store; ties break by
    (segment, doc)."""
    import searchlite_amd as sa
    from searchlite_amd import corpus
    rng = np.random.default_rng(5)
    n, dim, ncand, k_out, nq = 1500, 384, 300, 25, 3
    stores, offs = [], []
    for sgi in range(2):
        stores.append(corpus.unit_vectors(n, dim, seed=41 + sgi))
        o = np.arange(n, dtype=np.uint32)
        o[rng.choice(n, size=100, replace=False)] = 0xFFFFFFFF
        offs.append(o)
    segs = [_segment_with_vectors(n, offs[i], stores[i], 0) for i in range(2)]
    q = corpus.unit_vectors(nq, dim, seed=43)
    cseg = rng.integers(0, 2, size=(nq, ncand)).astype(np.uint32)
    cdoc = np.stack([rng.choice(n, size=ncand, replace=False) for _ in range(nq)]).astype(np.uint32)
    bm = (rng.random((nq, ncand)) * 5).astype(np.float32)
    bm[:, :40] = 1.0  # ties in the text score
    cnt = np.full(nq, ncand, np.uint32)
    alpha = np.array([0.5, 1.0, 0.3], np.float32)
    with sa.GpuIndex(segs) as ix:
        got = ix.rerank_batch(q, alpha, cdoc, cseg, bm, cnt, k_out)
    # the oracle on ONE store holding both segments: combined id = seg * n + doc (same tie order)
    comb_vals = np.concatenate(stores)
    comb_offs = np.concatenate([offs[0], np.where(offs[1] == 0xFFFFFFFF, 0xFFFFFFFF, offs[1] + n)]).astype(np.uint32)
    for i in range(nq):
        d_, s_, v_ = oracle.rerank(0, comb_offs, comb_vals, q[i], float(alpha[i]),
                                   (cseg[i] * n + cdoc[i]).astype(np.uint32), bm[i], k_out)
        gd, gs, gsc, gv, gc = got
        assert gc[i] == len(d_)
        assert np.abs(gsc[i, :len(d_)] - s_).max() <= TOL
        assert list(gs[i, :len(d_)].astype(np.int64) * n + gd[i, :len(d_)]) == list(d_), f"query {i}"


def test_hybrid_blends_text_and_vector_gpu(oracle):
    """tests/vector_search.rs:201-269 through BM25 (GPU) -> rerank (GPU)."""
    import searchlite_amd as sa
    from searchlite_amd.segment import SegmentBuilder
    b = SegmentBuilder(["body"], k1=0.9, b=0.4)
    b.add_document("long", {"body": "rust rust rust"})
    b.add_document("short", {"body": "rust"})
    seg = b.build()
    seg.vec_dim, seg.vec_metric = 2, 0
    seg.vec_offsets = np.array([0, 1], dtype=np.uint32)
    seg.vec_values = np.array([[0.0, 1.0], [1.0, 0.0]], dtype=np.float32)
    with sa.GpuIndex([seg]) as ix:
        hits = ix.search("rust", "body", limit=10)
        assert [seg.ext_ids[h[1]] for h in hits] == ["long", "short"]
        d = np.array([[h[1] for h in hits]], np.uint32)
        sc = np.array([[h[2] for h in hits]], np.float32)
        od, os_, osc, ov, oc = ix.rerank_batch(np.array([[1.0, 0.0]], np.float32), 0.2, d,
                                                np.zeros_like(d), sc, np.array([2]), 2)
    assert seg.ext_ids[int(od[0, 0])] == "short"
    assert osc[0, 0] == np.float32(np.float32(0.2) * np.float32(1.1046511) + np.float32(0.8))


@pytest.mark.parametrize("n_clauses", [1, 2, 8])
@pytest.mark.parametrize("metric", [0, 1])
def test_rerank_multi_clause(oracle, n_clauses, metric):
    """compute_hybrid_score (api/reader.rs:225-254) with 1 / 2 / 8 vector clauses over one
    candidate set, at the config-5 shape (768-d, 1001 candidates -> 10): mean of the per-clause
    blends, clause boosts, missing vectors as -1.0 / f32::MIN.  Cosine runs on the f32 matrix
    cores (v_mfma_f32_16x16x4_f32: an fmaf chain, the reference sums products left to right), so
    the tolerance is the single-clause one, 1e-5."""
    import searchlite_amd as sa
    from searchlite_amd import corpus
    rng = np.random.default_rng(100 + n_clauses)
    n, dim, ncand, k_out, nq = 4000, 768, 1001, 10, 5
    vals = corpus.unit_vectors(n, dim, seed=21)
    offsets = np.arange(n, dtype=np.uint32)
    offsets[rng.choice(n, size=300, replace=False)] = 0xFFFFFFFF
    q = corpus.unit_vectors(nq * n_clauses, dim, seed=22).reshape(nq, n_clauses, dim)
    cand = np.stack([rng.choice(n, size=ncand, replace=False) for _ in range(nq)]).astype(np.uint32)
    bm = (rng.random((nq, ncand)) * 20).astype(np.float32)
    cnt = np.full(nq, ncand, np.uint32)
    cnt[2] = 333
    alpha = rng.choice(np.array([0.0, 0.2, 0.5, 0.8, 1.0], np.float32), size=(nq, n_clauses))
    boost = (rng.random((nq, n_clauses)) * 1.5 + 0.5).astype(np.float32)
    seg = _segment_with_vectors(n, offsets, vals, metric)
    with sa.GpuIndex([seg]) as ix:
        for bst in (None, boost):
            got = ix.rerank_multi_batch(q, alpha, cand, np.zeros_like(cand), bm, cnt, k_out, boost=bst)
            wd, ws, wv = [], [], []
            for i in range(nq):
                d_, s_, v_ = oracle.rerank_multi(metric, offsets, vals, q[i], alpha[i], cand[i, :cnt[i]],
                                                 bm[i, :cnt[i]], k_out, boost=None if bst is None else bst[i])
                wd.append(d_)
                ws.append(s_)
                wv.append(v_)
            global TOL
            tol_save = TOL
            TOL = 1e-5 * max(1, n_clauses // 2)  # the reported vector score is a SUM over the clauses
            try:
                _check(got, wd, ws, wv, f"{n_clauses} clauses metric {metric} boost {bst is not None}")
            finally:
                TOL = tol_save


@pytest.mark.parametrize("n_clauses", [3, 4, 8])
def test_rerank_l2_on_the_matrix_cores_with_near_duplicates(oracle, n_clauses):
    """L2 with >= 3 clauses computes |q - x|^2 = |q|^2 + |x|^2 - 2 q.x on the matrix cores; the
    identity cancels when a candidate is (nearly) the query vector, so such pairs are recomputed as the
    plain sum of squared differences (vectors/mod.rs:98-105).  Candidates here include exact copies
    of a clause vector, copies with 1e-4 / 1e-2 noise, scaled copies and ordinary vectors; non-unit
    norms on both sides."""
    import searchlite_amd as sa
    from searchlite_amd import corpus
    rng = np.random.default_rng(500 + n_clauses)
    n, dim, ncand, k_out, nq = 3000, 768, 601, 10, 4
    vals = (corpus.unit_vectors(n, dim, seed=31) * rng.uniform(0.5, 3.0, size=(n, 1))).astype(np.float32)
    q = (corpus.unit_vectors(nq * n_clauses, dim, seed=32) * 1.7).astype(np.float32).reshape(nq, n_clauses, dim)
    for i in range(nq):  # plant near-duplicates of the query's clause vectors among the rows
        for c in range(n_clauses):
            base = 40 * i + 8 * (c % 5)
            vals[base] = q[i, c]
            vals[base + 1] = q[i, c] + rng.normal(0, 1e-4, dim).astype(np.float32)
            vals[base + 2] = q[i, c] + rng.normal(0, 1e-2, dim).astype(np.float32)
            vals[base + 3] = q[i, c] * np.float32(1.001)
    offsets = np.arange(n, dtype=np.uint32)
    offsets[rng.choice(np.arange(200, n), size=100, replace=False)] = 0xFFFFFFFF
    cand = np.stack([np.concatenate([np.arange(40 * i, 40 * i + 40), rng.choice(np.arange(200, n), size=ncand - 40,
                                                                                 replace=False)])
                     for i in range(nq)]).astype(np.uint32)
    bm = (rng.random((nq, ncand)) * 5).astype(np.float32)
    cnt = np.full(nq, ncand, np.uint32)
    alpha = rng.choice(np.array([0.0, 0.3, 0.7], np.float32), size=(nq, n_clauses))
    seg = _segment_with_vectors(n, offsets, vals, 1)
    with sa.GpuIndex([seg]) as ix:
        got = ix.rerank_multi_batch(q, alpha, cand, np.zeros_like(cand), bm, cnt, k_out)
    wd, ws, wv = [], [], []
    for i in range(nq):
        d_, s_, v_ = oracle.rerank_multi(1, offsets, vals, q[i], alpha[i], cand[i], bm[i], k_out)
        wd.append(d_)
        ws.append(s_)
        wv.append(v_)
    global TOL
    tol_save = TOL
    # per clause: the identity's rounding is ~1e-6 (|q|^2 + |x|^2) / (2 d); d >= 0.22 sqrt(|q|^2 + |x|^2) off
    # the exact path, norms up to 3: within 1e-5 per clause; the vector score sums the clauses
    TOL = 1e-5 * n_clauses
    try:
        _check(got, wd, ws, wv, f"L2 on MFMA, {n_clauses} clauses, near-duplicates")
    finally:
        TOL = tol_save


@pytest.mark.parametrize("dim", [16, 48, 100, 400, 1100])
@pytest.mark.parametrize("metric", [0, 1])
def test_rerank_multi_clause_dims(oracle, dim, metric):
    """The multi-clause kernel's three paths: matrix cores (cosine, dim % 16 == 0: 16, 48, 400),
    rows cached in registers (dim <= 1024 otherwise) and the streaming loop (dim > 1024)."""
    import searchlite_amd as sa
    from searchlite_amd import corpus
    rng = np.random.default_rng(7 * dim + metric)
    n, ncand, k_out, nq, nc = 3000, 257, 12, 4, 3
    vals = corpus.unit_vectors(n, dim, seed=31)
    offsets = np.arange(n, dtype=np.uint32)
    offsets[rng.choice(n, size=200, replace=False)] = 0xFFFFFFFF
    q = corpus.unit_vectors(nq * nc, dim, seed=32).reshape(nq, nc, dim)
    cand = np.stack([rng.choice(n, size=ncand, replace=False) for _ in range(nq)]).astype(np.uint32)
    bm = (rng.random((nq, ncand)) * 20).astype(np.float32)
    cnt = np.array([ncand, 100, 17, ncand], np.uint32)
    alpha = rng.choice(np.array([0.0, 0.3, 0.5, 1.0], np.float32), size=(nq, nc))
    boost = (rng.random((nq, nc)) + 0.5).astype(np.float32)
    seg = _segment_with_vectors(n, offsets, vals, metric)
    with sa.GpuIndex([seg]) as ix:
        got = ix.rerank_multi_batch(q, alpha, cand, np.zeros_like(cand), bm, cnt, k_out, boost=boost)
    wd, ws, wv = [], [], []
    for i in range(nq):
        d_, s_, v_ = oracle.rerank_multi(metric, offsets, vals, q[i], alpha[i], cand[i, :cnt[i]],
                                         bm[i, :cnt[i]], k_out, boost=boost[i])
        wd.append(d_)
        ws.append(s_)
        wv.append(v_)
    global TOL
    tol_save = TOL
    TOL = 2e-5
    try:
        _check(got, wd, ws, wv, f"dim {dim} metric {metric}")
    finally:
        TOL = tol_save


def test_rerank_clauses_over_different_fields(oracle):
    """Clauses that name different vector fields (api/reader.rs:225-254): a 96-d cosine field
    (field 0, in the segment descriptor), a 40-d L2 field and a 256-d cosine field added with
    slg_index_add_vector_field.  Docs may have a vector in one field and not in another: only that
    clause takes its metric's missing score; the reported vector score sums the clauses found."""
    import searchlite_amd as sa
    from searchlite_amd import corpus
    rng = np.random.default_rng(77)
    n, ncand, k_out, nq = 900, 200, 15, 4
    dims, metrics = [96, 40, 256], [0, 1, 0]
    fields = []
    for f in range(3):
        vals = corpus.unit_vectors(n, dims[f], seed=60 + f)
        offs = np.arange(n, dtype=np.uint32)
        offs[rng.choice(n, size=250, replace=False)] = 0xFFFFFFFF
        fields.append((metrics[f], offs, vals))
    seg = _segment_with_vectors(n, fields[0][1], fields[0][2], 0)
    cand = np.stack([rng.choice(n, size=ncand, replace=False) for _ in range(nq)]).astype(np.uint32)
    bm = (rng.random((nq, ncand)) * 10).astype(np.float32)
    cnt = np.array([ncand, 57, ncand, 1], np.uint32)
    for clause_field in ([0, 1, 2], [2, 1], [1], [1, 1, 0, 2, 2]):
        nc = len(clause_field)
        qs = [corpus.unit_vectors(nq, dims[f], seed=70 + c) for c, f in enumerate(clause_field)]
        qcat = np.concatenate(qs, axis=1)
        alpha = rng.choice(np.array([0.0, 0.25, 0.5, 1.0], np.float32), size=(nq, nc))
        boost = (rng.random((nq, nc)) + 0.5).astype(np.float32)
        with sa.GpuIndex([seg]) as ix:
            ids = [0, ix.add_vector_field([fields[1]]), ix.add_vector_field([fields[2]])]
            assert ids == [0, 1, 2]
            got = ix.rerank_fields_batch([ids[f] for f in clause_field], qcat, alpha, cand, np.zeros_like(cand),
                                         bm, cnt, k_out, boost=boost)
        wd, ws, wv = [], [], []
        for i in range(nq):
            d_, s_, v_ = oracle.rerank_fields(fields, clause_field, [q[i] for q in qs], alpha[i],
                                              cand[i, :cnt[i]], bm[i, :cnt[i]], k_out, boost=boost[i])
            wd.append(d_)
            ws.append(s_)
            wv.append(v_)
        global TOL
        tol_save = TOL
        TOL = 1e-5 * max(1, nc // 2)
        try:
            _check(got, wd, ws, wv, f"fields {clause_field}")
        finally:
            TOL = tol_save


def test_rerank_fields_errors():
    import searchlite_amd as sa
    from searchlite_amd import corpus
    n = 50
    vals = corpus.unit_vectors(n, 8, seed=1)
    seg = _segment_with_vectors(n, np.arange(n, dtype=np.uint32), vals, 0)
    cand = np.zeros((1, 4), np.uint32)
    with sa.GpuIndex([seg]) as ix:
        with pytest.raises(sa.SlgError):  # unknown field id
            ix.rerank_fields_batch([3], np.zeros((1, 8), np.float32), 0.5, cand, cand, np.zeros((1, 4), np.float32),
                                   np.array([4], np.uint32), 2)
        with pytest.raises(sa.SlgError):  # offsets past the rows
            ix.add_vector_field([(0, np.full(n, 7, np.uint32), vals[:3])])


def test_rerank_multi_matches_single_clause_kernel():
    """One clause through the multi-clause entry (with a boost of 1.0, which takes the MFMA kernel)
    agrees with the GEMV-shaped single-clause kernel."""
    import searchlite_amd as sa
    from searchlite_amd import corpus
    rng = np.random.default_rng(5)
    n, dim, ncand, nq = 2000, 64, 200, 4
    vals = corpus.unit_vectors(n, dim, seed=3)
    offsets = np.arange(n, dtype=np.uint32)
    q = corpus.unit_vectors(nq, dim, seed=4)
    cand = np.stack([rng.choice(n, size=ncand, replace=False) for _ in range(nq)]).astype(np.uint32)
    bm = (rng.random((nq, ncand)) * 5).astype(np.float32)
    cnt = np.full(nq, ncand, np.uint32)
    seg = _segment_with_vectors(n, offsets, vals, 0)
    with sa.GpuIndex([seg]) as ix:
        a = ix.rerank_batch(q, 0.5, cand, np.zeros_like(cand), bm, cnt, 20)
        b = ix.rerank_multi_batch(q[:, None, :], np.full((nq, 1), 0.5, np.float32), cand, np.zeros_like(cand),
                                  bm, cnt, 20, boost=np.ones((nq, 1), np.float32))
    assert np.abs(a[2] - b[2]).max() <= TOL and np.abs(a[3] - b[3]).max() <= TOL
    assert (a[4] == b[4]).all()


def test_rerank_multi_limits():
    import searchlite_amd as sa
    from searchlite_amd import _native as N
    vals = np.eye(8, dtype=np.float32)
    seg = _segment_with_vectors(8, np.arange(8, dtype=np.uint32), vals, 0)
    with sa.GpuIndex([seg]) as ix:
        c = np.zeros((1, 4), np.uint32)
        with pytest.raises(N.SlgError) as e:
            ix.rerank_multi_batch(np.zeros((1, 9, 8), np.float32), 0.5, c, c, np.zeros((1, 4), np.float32),
                                  np.array([4], np.uint32), 2)
        assert e.value.code == N.ERR_UNSUPPORTED  # > MAX_VECTOR_CLAUSES (api/reader.rs:134)
