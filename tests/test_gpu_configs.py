"""BASELINE configs 3 and 4 at full size, the kernels they select pinned on small deterministic
cases, and fixed-seed regressions of mismatches seen in development (-m gpu)."""
import importlib.util
import os

import numpy as np
import pytest

from tests.util import assert_same_hits, random_queries, random_segment

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    import searchlite_amd as sa
    from searchlite_amd import searcher
    assert searcher.device_count() >= 1
    return sa


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _properties(d, s, sc, c, k, n_docs):
    """Size-independent properties of a top-k block: full rows, (score desc, seg asc, doc asc),
    distinct (seg, doc) pairs in range."""
    assert (c == k).all()
    assert (sc[:, :-1] >= sc[:, 1:]).all()
    tie = sc[:, :-1] == sc[:, 1:]
    key = s.astype(np.uint64) << np.uint64(32) | d.astype(np.uint64)
    assert (key[:, :-1][tie] < key[:, 1:][tie]).all()
    assert (d < n_docs).all()
    for r in key[:: max(1, len(key) // 64)]:
        assert len(set(r.tolist())) == k


# ---- config 3: 10M docs, 5-term OR with pruning, batch 4096, top-100 --------------------------------
def test_config3_full_size_properties(gpu, oracle):
    """BASELINE config 3 exactly as bench.py --config c3 builds it (zipf seed 43, V = 2^20, query
    seed 7, k = 101, strategy Wand => the blocked few-term kernel score_uniform4_kernel<2, 8>, the planner
    having dropped the MaxScore classification that block skipping cannot use here): determinism,
    sortedness, distinctness, and a 32-query sample bit-exact against the exhaustive oracle."""
    from searchlite_amd import corpus
    n_docs, vocab, nq, T, k = 10_000_000, 1 << 20, 4096, 5, 101
    seg = corpus.zipf_segment(n_docs, vocab, seed=43, n_threads=16)
    offs, terms, w = corpus.zipf_queries(nq, T, seed=7, vocab=vocab)
    with gpu.GpuIndex([seg]) as ix:
        b = ix.prepare(offs, terms, w, k, gpu.Wand)
        b.run()
        d, s, sc, c = b.fetch()
        b.run()
        d2, s2, sc2, c2 = b.fetch()
        b.close()
        # the exhaustive strategy on the same index returns the same block
        d3, s3, sc3, c3 = ix.search_batch(offs[:257], terms[:256 * T], w[:256 * T], k, gpu.Bm25)
    assert np.array_equal(d, d2) and np.array_equal(sc.view(np.uint32), sc2.view(np.uint32))
    assert np.array_equal(d[:256], d3) and np.array_equal(sc[:256].view(np.uint32), sc3.view(np.uint32))
    _properties(d, s, sc, c, k, n_docs)
    nchk = 32
    want = oracle.search_batch([seg], offs[:nchk + 1], terms[:nchk * T], w[:nchk * T], k,
                               strategy=oracle.BM25, n_threads=16)
    assert_same_hits((d[:nchk], s[:nchk], sc[:nchk], c[:nchk]), want, 0.0, "config 3 sample")


@pytest.mark.parametrize("tuning", [None, {"pruning": 1}, {"pruning": 1, "block_max": 0}, {"pruning": 0},
                                    {"uniform_max_terms": 4}, {"inline_cuts": 0}])
def test_five_terms_top100_pruned_kernel_small(gpu, oracle, tuning):
    """T = 5, k = 101 (two registers per lane) on a corpus small enough for the oracle to check
    every query: the kernel config 3 selects (None: score_uniform4_kernel<2, 8>, the planner drops the
    classification because block skipping has nothing to gain), the classified many-term kernel
    score_multi_kernel<2, 1> (pruning: 1), the unclassified one (uniform_max_terms: 4 -> <2, 0> is
    not reached: classification stays on; pruning: 0 with 8 lists -> few-term kernel); inline_cuts: 0 =
    cut points from partition_rounds_kernel instead of the scoring waves' own.  (The slot forms of the
    few-term kernel are no longer in the product library: -DSLG_LEGACY_KERNELS, tools/ab_uniform.sh.)"""
    from searchlite_amd import corpus
    seg = corpus.zipf_segment(300_000, 1 << 16, seed=43)
    offs, terms, w = corpus.zipf_queries(192, 5, rank_lo=8, rank_hi=4096, seed=7, vocab=1 << 16)
    want = oracle.search_batch([seg], offs, terms, w, 101, strategy=oracle.BM25, n_threads=8)
    with gpu.GpuIndex([seg], tuning=tuning) as ix:
        for strat in (gpu.Wand, gpu.Bmw, gpu.Bm25):
            got = ix.search_batch(offs, terms, w, 101, strat, want_stats=True)
            assert_same_hits(got[:4], want, 0.0, f"T=5 k=101 strategy {strat} tuning {tuning}")
        if tuning == {"pruning": 1}:  # pruning really happened: fewer docs scored than the exhaustive strategy
            pr = ix.search_batch(offs, terms, w, 101, gpu.Wand, want_stats=True)[4]
            ex = ix.search_batch(offs, terms, w, 101, gpu.Bm25, want_stats=True)[4]
            assert sum(pr[q].scored_docs for q in range(192)) < sum(ex[q].scored_docs for q in range(192))


def _skewed_queries(nq, vocab, seed):
    """One stop-word-like term (rank 1..8: in most docs) at a varying position among four rare
    terms (ranks 4096..32768): the shape WAND's skipping is made for."""
    rng = np.random.default_rng(seed)
    T = 5
    terms = np.empty((nq, T), dtype=np.uint32)
    for q in range(nq):
        rare = rng.choice(np.arange(4096, min(32768, vocab), dtype=np.uint32), size=T - 1, replace=False) - 1
        row = list(rare)
        row.insert(q % T, np.uint32(rng.integers(0, 8)))
        terms[q] = row
    offs = (np.arange(nq + 1, dtype=np.uint32) * T).astype(np.uint32)
    return offs, terms.reshape(-1), np.ones(nq * T, dtype=np.float32)


@pytest.mark.parametrize("tuning", [None, {"probe_target": 8192}, {"block_max": 0}])
def test_block_skipping_skewed_queries(gpu, oracle, tuning):
    """Block skipping (query/wand.rs:205-265): 64-posting blocks of a non-essential list that hold
    no candidate doc are not loaded.  Same hits as the exhaustive oracle, and most of the dense
    list's postings are skipped.  probe_target 8192 makes rounds of > 60 slots, which are cut into
    chunks (the skipped slots below the cut must count as consumed)."""
    from searchlite_amd import corpus
    seg = corpus.zipf_segment(200_000, 1 << 15, seed=45)
    offs, terms, w = _skewed_queries(96, 1 << 15, seed=3)
    for k in (11, 101):
        want = oracle.search_batch([seg], offs, terms, w, k, strategy=oracle.BM25, n_threads=8)
        with gpu.GpuIndex([seg], tuning=tuning) as ix:
            for strat in (gpu.Wand, gpu.Bmw):
                b = ix.prepare(offs, terms, w, k, strat)
                b.run()
                got = b.fetch(want_stats=True)
                probed, skipped = b.skip_counts()
                info = b.info()
                b.close()
                # QueryStats: postings never loaded are not counted as advanced over
                assert sum(got[4][q].postings_advanced for q in range(96)) == info["n_postings"] - skipped
                assert_same_hits(got[:4], want, 0.0, f"skewed k={k} strategy {strat} tuning {tuning}")
                if tuning and tuning.get("block_max") == 0:
                    assert (probed, skipped) == (0, 0)
                else:
                    assert probed > 0 and skipped > probed // 3, (probed, skipped)


def test_many_subqueries_are_planned_by_several_threads(gpu, oracle):
    """Batches of >= 16384 sub-queries (queries x segments) are planned by several host threads,
    each into its own vectors, stitched in query order (slg_batch_prepare_plan pass 1): 6000
    ragged queries x 3 segments with weights, an absent term here and there, stats included —
    every query against the oracle, both kernels."""
    rng = np.random.default_rng(99)
    vocab = 300
    segs = [random_segment(rng, n, vocab, 12, zipf=True) for n in (9000, 4000, 14000)]
    nq = 6000
    lens = rng.integers(1, 8, size=nq)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    terms = np.empty((int(offs[-1]), 3), dtype=np.uint32)
    for q in range(nq):
        t = rng.choice(vocab, size=int(lens[q]), replace=False)
        for s_ in range(3):
            terms[offs[q]:offs[q + 1], s_] = t
    terms[rng.random(terms.shape) < 0.02] = 0xFFFFFFFF  # SLG_NO_TERM: the segment lacks the term
    w = (rng.random(int(offs[-1])) * 2 + 0.25).astype(np.float32)
    for k in (11, 101):
        want = oracle.search_batch(segs, offs, terms.reshape(-1), w, k, strategy=oracle.BM25, n_threads=8,
                                   want_stats=True)
        with gpu.GpuIndex(segs) as ix:
            for strat in (gpu.Wand, gpu.Bm25):
                got = ix.search_batch(offs, terms.reshape(-1), w, k, strat, want_stats=True)
                assert_same_hits(got[:4], want[:4], 0.0, f"18000 sub-queries k={k} strategy {strat}")
                if strat == gpu.Bm25:
                    assert [got[4][q].postings_advanced for q in range(nq)] == \
                        [want[4][q].postings_advanced for q in range(nq)]


def test_planner_threads_report_bad_input(gpu):
    """An out-of-range term id deep inside a batch that is planned by several threads comes back
    as SLG_ERR_INVALID from the caller's thread (the worker's exception is carried over)."""
    import searchlite_amd as sa
    rng = np.random.default_rng(5)
    seg = random_segment(rng, 3000, 50, 8)
    nq = 20000
    offs, terms, w = random_queries(rng, nq, 2, 50)
    terms = terms.copy()
    terms[2 * 17001, 0] = 4000  # no such term
    with gpu.GpuIndex([seg]) as ix:
        with pytest.raises(sa.SlgError) as e:
            ix.search_batch(offs, terms.reshape(-1), w, 5, gpu.Wand)
        assert "term id out of range in query 17001" in str(e.value)
        ok = ix.search_batch(offs[:11], terms.reshape(-1)[:20], w[:20], 5, gpu.Wand)  # the index is still usable
        assert ok[3].shape == (10,)


# ---- config 4: 8 index shards, batch 8192, all-gather + merge -----------------------------------------
def test_config4_eight_shards_merge_on_one_gpu(gpu, oracle):
    """Config 4's data path on one GPU: the 8 shards bench.py --config c4 builds (1.25M docs each,
    seeds 43 + r), every shard scored alone for all 8192 5-term queries at k = 101, the result blocks
    stacked shard-major exactly as the RCCL all-gather delivers them, merged by
    slg_merge_shards_device.  Checked: properties on all queries; a sample against the oracle run
    on the 8 shards as 8 segments (api/reader.rs:2670-2778, query/sort.rs:80-93)."""
    import torch
    from searchlite_amd import corpus
    from searchlite_amd import dist as sdist
    n_shards, n_docs, vocab, nq, T, k = 8, 1_250_000, 1 << 20, 8192, 5, 101
    offs, terms, w = corpus.zipf_queries(nq, T, seed=7, vocab=vocab)
    segs, blocks = [], []
    for r in range(n_shards):
        seg = corpus.zipf_segment(n_docs, vocab, seed=43 + r, n_threads=16)
        segs.append(seg)
        ix = gpu.GpuIndex([seg])
        ix.set_stream(torch.cuda.current_stream().cuda_stream)
        b = ix.prepare(offs, terms, w, k, gpu.Wand)
        b.run()
        blocks.append(sdist.batch_result_block(b).clone())
        torch.cuda.synchronize()
        b.close()
        ix.close()
    g = torch.stack(blocks)
    g_doc, g_seg, g_score, g_count = [x.contiguous() for x in sdist.split_result_block(g, nq, k)]
    m_doc = torch.empty((nq, k), dtype=torch.int32, device="cuda")
    m_seg = torch.empty_like(m_doc)
    m_score = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    m_count = torch.empty((nq,), dtype=torch.int32, device="cuda")
    small = random_segment(np.random.default_rng(1), 64, 7, 6)  # any index of this device owns the merge
    with gpu.GpuIndex([small]) as ix:
        ix.set_stream(torch.cuda.current_stream().cuda_stream)
        ix.merge_shards_device(n_shards, nq, k, g_doc.data_ptr(), g_seg.data_ptr(), g_score.data_ptr(),
                               g_count.data_ptr(), 1, m_doc.data_ptr(), m_seg.data_ptr(),
                               m_score.data_ptr(), m_count.data_ptr())
        torch.cuda.synchronize()
    got = (m_doc.cpu().numpy().view(np.uint32), m_seg.cpu().numpy().view(np.uint32),
           m_score.cpu().numpy(), m_count.cpu().numpy().view(np.uint32))
    _properties(got[0], got[1], got[2], got[3], k, n_docs)
    assert (got[1] < n_shards).all()
    nchk = 24
    terms8 = np.repeat(terms[:nchk * T].reshape(-1, 1), n_shards, axis=1)
    want = oracle.search_batch(segs, offs[:nchk + 1], terms8, w[:nchk * T], k, strategy=oracle.BM25,
                               n_threads=16)
    assert_same_hits(tuple(x[:nchk] for x in got), want, 0.0, "config 4 sample (8 shards merged)")


# ---- fixed-seed regressions ------------------------------------------------------------------------------
@pytest.mark.parametrize("tuning", [None, {"pruning": 1}, {"uniform_max_terms": 0}, {"block_max": 0}])
def test_fuzz_seed72_batch2_regression(gpu, oracle, tuning):
    """gpurun_out/fuzz_72.log of round 1: T = 6 among the queries, k = 600, 3 multi-field segments,
    filtered and unfiltered queries in one batch, strategy Wand — three queries came back wrong while
    pruning on the many-term kernel was being built.  Replayed with the same generator and key under
    the default planner and with pruning / the many-term kernel forced."""
    _fuzz().run_case(72, 2, tuning)


@pytest.mark.parametrize("seed", [72, 73])
def test_fuzz_neighbourhood_of_seed72(gpu, oracle, seed):
    mod = _fuzz()
    for it in range(6):
        mod.run_case(seed, it, {"pruning": 1})


def test_filters_many_terms_large_k(gpu, oracle):
    """T >= 5 and k > 256 with doc filters (a shape test_doc_filters_match_accept_semantics lacks):
    mixed filtered / unfiltered queries, three segments, tombstones, strategy Wand."""
    rng = np.random.default_rng(720)
    segs = [random_segment(rng, 9000 + 1000 * i, 40, 25) for i in range(3)]
    segs[1].set_deleted(np.nonzero(rng.random(segs[1].n_docs) < 0.2)[0].tolist())
    masks = [rng.random(sg.n_docs) < 0.5 for sg in segs]
    nq, T, k = 20, 6, 600
    offs, terms, w = random_queries(rng, nq, T, 40, n_segs=3, weights=True)
    qf = np.array([0 if q % 3 else -1 for q in range(nq)], dtype=np.int32)
    want = oracle.search_batch_filtered(segs, offs, terms, w, k, qf, [masks], strategy=oracle.BM25)
    for tuning in (None, {"pruning": 1}, {"pruning": 0}):
        with gpu.GpuIndex(segs, tuning=tuning) as ix:
            fid = ix.add_filter(masks)
            assert fid == 0
            got = ix.search_batch(offs, terms, w, k, gpu.Wand, q_filter=qf)
        assert_same_hits(got, want, 0.0, f"filters T=6 k=600 tuning {tuning}")


def test_tie_breaker_outside_unit_interval_is_rejected(gpu):
    """validate_tie_breaker (query/planner.rs:850-856): tie must lie in [0, 1]; a negative tie would
    also break the threshold seed (a DisMax could fall below its largest leaf)."""
    from searchlite_amd import _native as N
    seg = random_segment(np.random.default_rng(5), 500, 20, 10)
    offs = np.array([0, 2], dtype=np.uint32)
    terms = np.array([[1], [2]], dtype=np.uint32)
    w = np.ones(2, dtype=np.float32)
    with gpu.GpuIndex([seg]) as ix:
        for tie in (-0.25, 1.5, float("nan")):
            with pytest.raises(N.SlgError) as e:
                ix.search_plan(offs, terms, w, 5, q_plan=[gpu.PLAN_DISMAX], q_tie=[tie])
            assert e.value.code == N.ERR_INVALID and "tie" in e.value.msg
        with pytest.raises(N.SlgError) as e:
            ix.search_plan(offs, terms, w, 5, q_leaf=np.array([0, 0xFFFFFFFF], dtype=np.uint32))
        assert e.value.code == N.ERR_INVALID
        ix.search_plan(offs, terms, w, 5, q_plan=[gpu.PLAN_DISMAX], q_tie=[1.0])  # the ends are legal
        ix.search_plan(offs, terms, w, 5, q_plan=[gpu.PLAN_DISMAX], q_tie=[0.0])


def test_index_closed_before_its_batches(gpu):
    """Either destruction order is safe: a batch that outlives its index is detached (calls fail
    with SLG_ERR_INVALID, destroy stays valid)."""
    from searchlite_amd import _native as N
    import ctypes as C
    seg = random_segment(np.random.default_rng(6), 800, 20, 10)
    offs, terms, w = random_queries(np.random.default_rng(7), 4, 2, 20)
    ix = gpu.GpuIndex([seg])
    b = ix.prepare(offs, terms, w, 5)
    b.run()
    b.fetch()
    # through the raw ABI: destroy the index first, then use and destroy the batch
    lib = N.load()
    h_ix, h_b = ix._h, b._h
    ix._h = None
    ix._batches.clear()
    lib.slg_index_destroy(h_ix)
    assert lib.slg_batch_run(h_b) == N.ERR_INVALID
    assert lib.slg_batch_sync(h_b) == N.ERR_INVALID
    lib.slg_batch_destroy(h_b)
    b._h = None
    # through the Python mirror: closing the index closes its batches
    ix2 = gpu.GpuIndex([seg])
    b2 = ix2.prepare(offs, terms, w, 5)
    ix2.close()
    assert b2._h is None
    b2.close()


# ---- N2: a searchlite index directory loaded from its files, through the GPU --------------------------
def test_recipes_index_files_reproduce_the_goldens(gpu, tmp_path):
    """BASELINE config 1: the recipes corpus written in searchlite's on-disk formats (restated
    writer, oracle/segfile_writer.py), loaded by the product loader (MANIFEST -> .terms -> .post ->
    _len: columns) and searched on the GPU: the committed golden top-10 lists, bit for bit."""
    from oracle import segfile_writer as W
    from searchlite_amd import index_files as IF
    from tests.test_segfile import golden_recipes_with_dictionary
    from tests.util import golden_expected
    seg, z = golden_recipes_with_dictionary()
    W.write_index(str(tmp_path), [seg], keep_positions=True)
    li = IF.load_index(str(tmp_path), k1=0.9, b=0.4)
    with gpu.GpuIndex(li.segments) as ix:
        for strat in (gpu.Bm25, gpu.Wand, gpu.Bmw):
            got = ix.search_batch(z["q_offsets"], z["q_terms"], z["q_weights"], int(z["k"]), strat)
            assert_same_hits(got, golden_expected(z), 0.0, f"recipes from index files, strategy {strat}")


@pytest.mark.parametrize("k", [1025, 3000])
def test_merge_shards_beyond_the_register_top_k(gpu, oracle, k):
    """slg_merge_shards_device for k > 1024 (an index-sharded request with limit up to 20 000,
    api/reader.rs:2595-2619): three shards scored alone, merged on the device, against the oracle
    run on the three shards as three segments (ties across shards included: small vocabulary)."""
    import torch
    from searchlite_amd import dist as sdist
    rng = np.random.default_rng(77)
    segs = [random_segment(rng, 4000 + 500 * i, 12, 8, zipf=False) for i in range(3)]
    nq = 6
    offs, terms, w = random_queries(rng, nq, 3, 12, n_segs=3)
    want = oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25)
    blocks, keep = [], []
    for i, seg in enumerate(segs):
        ix = gpu.GpuIndex([seg])
        ix.set_stream(torch.cuda.current_stream().cuda_stream)
        b = ix.prepare(offs, terms[:, i:i + 1].copy(), w, k)
        b.run()
        blocks.append(sdist.batch_result_block(b).clone())
        keep.append((ix, b))
    g = torch.stack(blocks)
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
    assert_same_hits(got, want, 0.0, f"merge_shards_device k={k}")
    for ix, b in keep:
        b.close()
        ix.close()


def test_recipes_default_fields_golden(gpu):
    """BASELINE config 1 on default fields (all four text fields incl. `text`, multi-field leaves,
    Sum plan; tests/golden/recipes_default.npz): the GPU returns the exhaustive result bit for bit
    for every strategy (8 scored terms per query => many-term kernel with leaf close)."""
    from tests.util import golden_expected, load_golden
    segs, z = load_golden("recipes_default.npz")
    nq = len(z["q_nleaves"])
    with gpu.GpuIndex(segs) as ix:
        for strat in (gpu.Bm25, gpu.Wand, gpu.Bmw):
            got = ix.search_plan(z["q_offsets"], z["q_terms"], z["q_weights"], int(z["k"]), q_leaf=z["q_leaf"],
                                 q_plan=np.zeros(nq, np.int32), q_nleaves=z["q_nleaves"], strategy=strat)
            assert_same_hits(got, golden_expected(z), 0.0, f"recipes default fields, strategy {strat}")


# ---- config 5: 1M docs + 768-d vectors, BM25 top-1000 -> cosine rerank -> top-10, as ONE pipeline ----
def test_config5_full_size_pipeline(gpu, oracle):
    """BASELINE config 5 exactly as `bench.py --config c5` chains it (corpus seed 42, 768-d store seed
    11, 1024 queries seed 7, query vectors seed 12, alpha 0.5): slg_batch_prepare(k = 1001) -> run ->
    slg_batch_device_results -> slg_rerank_batch_device -> top-10, no host round trip between the two
    stages (api/reader.rs:2786 rescoring right after the cross-segment sort; blend
    api/reader.rs:225-254, similarity vectors/mod.rs:107-129).  Properties on all 1024 queries; 16
    queries against oracle.search_batch + oracle.rerank (bit-exact BM25 block, rerank within 1e-5);
    then the SAME device candidates through slg_rerank_multi_batch_device with 2 clauses (the
    v_mfma_f32_16x16x4_f32 kernel) against oracle.rerank_multi."""
    import torch
    from searchlite_amd import corpus
    n_docs, vocab, nq, T, dim = 1_000_000, 1 << 18, 1024, 3, 768
    k, k_out = 1001, 10
    seg = corpus.zipf_segment(n_docs, vocab, seed=42, n_threads=16)
    vals = corpus.unit_vectors(n_docs, dim, seed=11)
    seg.vec_dim, seg.vec_metric = dim, 0
    seg.vec_offsets = np.arange(n_docs, dtype=np.uint32)
    seg.vec_values = vals
    offs, terms, w = corpus.zipf_queries(nq, T, seed=7, vocab=vocab)
    qh = corpus.unit_vectors(nq, dim, seed=12)
    q2h = np.stack([qh, corpus.unit_vectors(nq, dim, seed=13)], axis=1)  # [nq, 2, dim]
    a2h = np.tile(np.array([0.5, 0.3], np.float32), (nq, 1))
    stream = torch.cuda.current_stream()

    def outs():
        return (torch.empty((nq, k_out), dtype=torch.int32, device="cuda"),
                torch.empty((nq, k_out), dtype=torch.int32, device="cuda"),
                torch.empty((nq, k_out), dtype=torch.float32, device="cuda"),
                torch.empty((nq, k_out), dtype=torch.float32, device="cuda"),
                torch.empty((nq,), dtype=torch.int32, device="cuda"))

    with gpu.GpuIndex([seg]) as ix:
        ix.set_stream(stream.cuda_stream)
        qv = torch.from_numpy(qh).cuda()
        q2 = torch.from_numpy(q2h).cuda()
        alpha = torch.full((nq,), 0.5, dtype=torch.float32, device="cuda")
        alpha2 = torch.from_numpy(a2h).cuda()
        b = ix.prepare(offs, terms, w, k, gpu.Wand)
        b.run()
        d = b.device_results()
        r1, r2 = outs(), outs()
        ix.rerank_batch_device(nq, qv.data_ptr(), alpha.data_ptr(), d[0], d[1], d[2], d[3], k, k_out,
                               *[t.data_ptr() for t in r1])
        ix.rerank_multi_batch_device(nq, 2, q2.data_ptr(), alpha2.data_ptr(), None, d[0], d[1], d[2], d[3],
                                     k, k_out, *[t.data_ptr() for t in r2])
        torch.cuda.synchronize()
        cd, cs, csc, cc = b.fetch()       # the BM25 stage's block, for the checks below
        b.close()
        g1 = [t.cpu().numpy() for t in r1]
        g2 = [t.cpu().numpy() for t in r2]
    # ---- the BM25 stage: 1001 candidates per query, sorted, distinct ----
    _properties(cd, cs, csc, cc, k, n_docs)
    # ---- properties of both rerank outputs on all queries ----
    for g, nc_ in ((g1, 1), (g2, 2)):
        rd, rs, rsc, rv, rc = g[0].view(np.uint32), g[1].view(np.uint32), g[2], g[3], g[4].view(np.uint32)
        assert (rc == k_out).all() and (rs == 0).all()
        assert (rsc[:, :-1] >= rsc[:, 1:]).all()                      # blended score descending
        for q in range(nq):
            assert len(set(rd[q].tolist())) == k_out                  # distinct docs ...
            pos = {int(x): i for i, x in enumerate(cd[q])}
            assert all(int(x) in pos for x in rd[q])                  # ... out of this query's candidates
            if nc_ == 1:  # blend_scores (vectors/mod.rs:122-129): alpha * bm25 + (1 - alpha) * sim
                bm = np.array([csc[q, pos[int(x)]] for x in rd[q]], np.float32)
                want = np.float32(0.5) * bm + np.float32(0.5) * rv[q]
                assert np.abs(want - rsc[q]).max() <= 1e-5
    # ---- 16 queries against the oracle ----
    nchk = 16
    want = oracle.search_batch([seg], offs[:nchk + 1], terms[:nchk * T], w[:nchk * T], k,
                               strategy=oracle.BM25, n_threads=16)
    assert_same_hits((cd[:nchk], cs[:nchk], csc[:nchk], cc[:nchk]), want, 0.0, "config 5 BM25 top-1000")
    spec = importlib.util.spec_from_file_location("t_rerank", os.path.join(ROOT, "tests", "test_gpu_rerank.py"))
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    wd, ws, wv, wd2, ws2 = [], [], [], [], []
    for q in range(nchk):
        n = int(want[3][q])
        d_, s_, v_ = oracle.rerank(0, seg.vec_offsets, vals, qh[q], 0.5, want[0][q, :n], want[2][q, :n], k_out)
        wd.append(d_), ws.append(s_), wv.append(v_)
        d_, s_, _ = oracle.rerank_multi(0, seg.vec_offsets, vals, q2h[q], a2h[q], want[0][q, :n], want[2][q, :n], k_out)
        wd2.append(d_), ws2.append(s_)
    as_u = lambda g: (g[0].view(np.uint32), g[1].view(np.uint32), g[2], g[3], g[4].view(np.uint32))
    tr._check(as_u(g1), wd, ws, wv, "config 5: device candidates -> slg_rerank_batch_device")
    tr._check(as_u(g2), wd2, ws2, None, "config 5: device candidates -> 2-clause MFMA rerank")
