"""GPU parity: HIP path through the C ABI vs the CPU oracle on the same arrays.

Bar: identical (segment, doc) sequence and BIT-EXACT f32 scores (the kernels add per-term
partials in query-term order with contraction off, exactly the reference's leaf-order sum);
north_star only asks for 1e-4.
"""
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
