"""Index updates through the C ABI: the staged index follows searchlite's manifest
(api/writer.rs:106-240: a commit merges tombstones into segments' deleted_docs and appends at most
one segment; index/mod.rs:102+: compaction replaces segments) without being re-created.

Bar, as everywhere: identical (segment, doc) sequence and bit-exact scores against the oracle run on
the UPDATED segment descriptors — i.e. slg_index_update_deleted must re-derive every impact for the
new live_docs (api/reader.rs:2985 `docs = seg.live_docs()`) exactly as a fresh staging would.
"""
import copy

import numpy as np
import pytest

from tests.util import assert_same_hits, random_multifield_segment, random_queries, random_segment

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import searchlite_amd as sa
    from searchlite_amd import searcher
    assert searcher.device_count() >= 1
    return sa


def tombstones(rng, n_docs, frac, prev=None):
    """Complete bitmap (bit d&7 of byte d>>3) with ~frac of the docs newly deleted on top of prev."""
    bits = np.zeros(n_docs, dtype=bool) if prev is None else np.unpackbits(prev, bitorder="little")[:n_docs].astype(bool)
    bits |= rng.random(n_docs) < frac
    return np.packbits(bits, bitorder="little"), int(bits.sum())


def with_tombstones(seg, bitmap, n_deleted):
    s = copy.copy(seg)
    s.deleted = bitmap
    s.docs = float(seg.n_docs - n_deleted)  # index/segment.rs:1365-1370 live_docs
    return s


@pytest.mark.parametrize("k", [11, 400])
def test_update_deleted_rederives_impacts(gpu, oracle, k):
    """delete docs -> update -> results == oracle on the new `docs`, == a freshly created index."""
    rng = np.random.default_rng(91 + k)
    segs = [random_segment(rng, 3000, 60, 25, k1=0.9, b=0.4, missing_len_frac=0.03),
            random_segment(rng, 1500, 60, 25, k1=0.9, b=0.4)]
    offs, terms, w = random_queries(rng, 48, 3, 60, n_segs=2, weights=True)
    with gpu.GpuIndex([copy.copy(s) for s in segs]) as ix:
        assert ix.generation == 0
        before = ix.search_batch(offs, terms, w, k, gpu.Wand)
        assert_same_hits(before, oracle.search_batch(segs, offs, terms, w, k, strategy=oracle.BM25), 0.0, "before")
        bm, nd = None, 0
        for step in range(3):  # tombstones only grow (api/writer.rs:150-158)
            bm, nd = tombstones(rng, segs[0].n_docs, 0.08, bm)
            ix.update_deleted(0, bm, segs[0].n_docs - nd)
            assert ix.generation == step + 1
            cur = [with_tombstones(segs[0], bm, nd), segs[1]]
            want = oracle.search_batch(cur, offs, terms, w, k, strategy=oracle.BM25)
            for strat in (gpu.Bm25, gpu.Wand, gpu.Bmw):
                assert_same_hits(ix.search_batch(offs, terms, w, k, strat), want, 0.0, f"step {step} strategy {strat}")
        # the other segment, and a fresh index on the final descriptors
        bm1, nd1 = tombstones(rng, segs[1].n_docs, 0.2)
        ix.update_deleted(1, bm1, segs[1].n_docs - nd1)
        cur = [with_tombstones(segs[0], bm, nd), with_tombstones(segs[1], bm1, nd1)]
        want = oracle.search_batch(cur, offs, terms, w, k, strategy=oracle.BM25)
        got = ix.search_batch(offs, terms, w, k, gpu.Wand)
        assert_same_hits(got, want, 0.0, "both segments")
        with gpu.GpuIndex(cur) as fresh:
            assert_same_hits(got, fresh.search_batch(offs, terms, w, k, gpu.Wand), 0.0, "vs fresh index")
        # deleted docs are gone from the results
        dead0 = np.unpackbits(bm, bitorder="little")[:segs[0].n_docs].astype(bool)
        gd, gs, _, gc = got
        for q in range(len(gc)):
            for i in range(int(gc[q])):
                if gs[q, i] == 0:
                    assert not dead0[gd[q, i]]


def test_update_deleted_multi_field_plans(gpu, oracle):
    """Per-field doc lengths / avgdl stay resident for the re-derivation (multi-field segment,
    score plans over multi-term leaves)."""
    rng = np.random.default_rng(17)
    vocab, nf = 40, 3
    seg = random_multifield_segment(rng, 1200, vocab, nf, 18)
    nq, words = 24, 2
    offs = (np.arange(nq + 1) * words * nf).astype(np.uint32)
    terms = np.empty((nq * words * nf, 1), dtype=np.uint32)
    leaf = np.empty(nq * words * nf, dtype=np.uint32)
    for q in range(nq):
        ws = rng.choice(vocab, size=words, replace=False)
        i = q * words * nf
        for wi, wd in enumerate(ws):
            for f in range(nf):
                terms[i, 0] = f * vocab + wd
                leaf[i] = wi
                i += 1
    w = np.ones(len(leaf), dtype=np.float32)
    bm, nd = tombstones(rng, seg.n_docs, 0.15)
    with gpu.GpuIndex([copy.copy(seg)]) as ix:
        ix.update_deleted(0, bm, seg.n_docs - nd)
        cur = [with_tombstones(seg, bm, nd)]
        want = oracle.search_batch(cur, offs, terms, w, 11, strategy=oracle.BM25, q_leaf=leaf)
        got = ix.search_plan(offs, terms, w, 11, q_leaf=leaf)
        assert_same_hits(got, want, 0.0, "multi-field leaves after update")


def test_add_and_remove_segment(gpu, oracle):
    """commit appends a segment at the end of the manifest (api/writer.rs:160-192): n + 1 segments ==
    the oracle on n + 1 segments; compaction (add the merged segment, drop the old ones) shifts the
    ordinals down."""
    rng = np.random.default_rng(5)
    vocab = 50
    segs = [random_segment(rng, 2000, vocab, 20, k1=0.9, b=0.4) for _ in range(3)]
    k = 11
    with gpu.GpuIndex([copy.copy(segs[0])]) as ix:
        offs, terms1, w = random_queries(rng, 32, 3, vocab, n_segs=1, weights=True)
        assert_same_hits(ix.search_batch(offs, terms1, w, k, gpu.Wand),
                         oracle.search_batch(segs[:1], offs, terms1, w, k, strategy=oracle.BM25), 0.0, "1 segment")
        assert ix.add_segment(copy.copy(segs[1])) == 1
        assert ix.add_segment(copy.copy(segs[2])) == 2
        assert ix.generation == 2 and ix.info()["n_segs"] == 3
        terms3 = np.repeat(terms1, 3, axis=1)
        want3 = oracle.search_batch(segs, offs, terms3, w, k, strategy=oracle.BM25)
        for strat in (gpu.Bm25, gpu.Wand, gpu.Bmw):
            assert_same_hits(ix.search_batch(offs, terms3, w, k, strat), want3, 0.0, f"3 segments strategy {strat}")
        assert len({int(s) for s in want3[1][want3[3] > 0].ravel()}) > 1  # hits really span segments
        ix.remove_segment(0)
        assert ix.info()["n_segs"] == 2
        terms2 = np.repeat(terms1, 2, axis=1)
        want2 = oracle.search_batch(segs[1:], offs, terms2, w, k, strategy=oracle.BM25)
        assert_same_hits(ix.search_batch(offs, terms2, w, k, gpu.Wand), want2, 0.0, "after remove")
        with pytest.raises(gpu.SlgError):
            ix.remove_segment(5)
        ix.remove_segment(1)
        with pytest.raises(gpu.SlgError):  # the last segment stays
            ix.remove_segment(0)


def test_batches_in_flight_survive_updates(gpu, oracle):
    """A batch is bound to the index state it was prepared on: prepared before an update, run and
    fetched after it (and after the state was retired twice), it returns the OLD state's results."""
    from searchlite_amd.searcher import PreparedBatch
    rng = np.random.default_rng(23)
    vocab = 60
    seg = random_segment(rng, 4000, vocab, 25, k1=0.9, b=0.4)
    extra = random_segment(rng, 1000, vocab, 25, k1=0.9, b=0.4)
    offs, terms, w = random_queries(rng, 64, 3, vocab, weights=True)
    k = 11
    want_old = oracle.search_batch([seg], offs, terms, w, k, strategy=oracle.BM25)
    with gpu.GpuIndex([copy.copy(seg)]) as ix:
        b_old = PreparedBatch(ix, offs, terms, w, k, gpu.Wand)       # prepared, not yet run
        b_run = PreparedBatch(ix, offs, terms, w, k, gpu.Wand)
        b_run.run()                                                   # in flight during the update
        bm, nd = tombstones(rng, seg.n_docs, 0.3)
        ix.update_deleted(0, bm, seg.n_docs - nd)
        ix.add_segment(copy.copy(extra))
        # new batches see the new state
        cur = [with_tombstones(seg, bm, nd), extra]
        terms2 = np.repeat(terms, 2, axis=1)
        assert_same_hits(ix.search_batch(offs, terms2, w, k, gpu.Wand),
                         oracle.search_batch(cur, offs, terms2, w, k, strategy=oracle.BM25), 0.0, "new state")
        # the old ones their own
        assert_same_hits(b_run.fetch(), want_old, 0.0, "in flight during the update")
        b_old.run()
        assert_same_hits(b_old.fetch(), want_old, 0.0, "prepared before, run after the update")
        b_old.close()
        b_run.close()


def test_filters_follow_updates_and_ids_are_reused(gpu, oracle):
    """reject = deleted | ~filter: a registered filter takes a segment's new tombstones; a filter
    registered before slg_index_add_segment has no bitmap for the new segment and is refused; a
    removed filter's id is handed out again while a batch prepared with the old one still runs."""
    from searchlite_amd.searcher import PreparedBatch
    rng = np.random.default_rng(31)
    vocab = 40
    seg = random_segment(rng, 2500, vocab, 20, k1=0.9, b=0.4)
    offs, terms, w = random_queries(rng, 32, 3, vocab, weights=True)
    k = 11
    mask_a = rng.random(seg.n_docs) < 0.5
    mask_b = rng.random(seg.n_docs) < 0.3
    qf = np.zeros(32, dtype=np.int32)
    with gpu.GpuIndex([copy.copy(seg)]) as ix:
        fa = ix.add_filter([mask_a])
        bm, nd = tombstones(rng, seg.n_docs, 0.2)
        ix.update_deleted(0, bm, seg.n_docs - nd)
        cur = with_tombstones(seg, bm, nd)
        want_a = oracle.search_batch_filtered([cur], offs, terms, w, k, qf, [[mask_a]], strategy=oracle.BM25)
        assert_same_hits(ix.search_batch(offs, terms, w, k, gpu.Wand, q_filter=qf + fa), want_a, 0.0,
                         "filter registered before the tombstones")
        # id reuse: a batch prepared with filter A keeps A's bitmap after A is removed and B took its id
        held = PreparedBatch(ix, offs, terms, w, k, gpu.Wand, q_filter=qf + fa)
        ix.remove_filter(fa)
        fb = ix.add_filter([mask_b])
        assert fb == fa
        want_b = oracle.search_batch_filtered([cur], offs, terms, w, k, qf, [[mask_b]], strategy=oracle.BM25)
        assert_same_hits(ix.search_batch(offs, terms, w, k, gpu.Wand, q_filter=qf + fb), want_b, 0.0, "filter B")
        held.run()
        assert_same_hits(held.fetch(), want_a, 0.0, "batch prepared with the removed filter")
        held.close()
        # a filter that predates a segment is refused until it is registered again
        ix.add_segment(copy.copy(seg))
        terms2 = np.repeat(terms, 2, axis=1)
        with pytest.raises(gpu.SlgError):
            ix.search_batch(offs, terms2, w, k, gpu.Wand, q_filter=qf + fb)
        ix.remove_filter(fb)
        fc = ix.add_filter([mask_b, None])
        want_c = oracle.search_batch_filtered([cur, seg], offs, terms2, w, k, qf, [[mask_b, None]], strategy=oracle.BM25)
        assert_same_hits(ix.search_batch(offs, terms2, w, k, gpu.Wand, q_filter=qf + fc), want_c, 0.0, "re-registered")


def test_update_needs_updatable_index(gpu):
    rng = np.random.default_rng(3)
    seg = random_segment(rng, 300, 20, 10)
    with gpu.GpuIndex([seg], tuning={"updatable": 0}) as ix:
        with pytest.raises(gpu.SlgError) as e:
            ix.update_deleted(0, None, 300.0)
        assert e.value.code == -4  # SLG_ERR_UNSUPPORTED
        assert ix.add_segment(copy.copy(seg)) == 1  # staging a new segment needs nothing resident
    with gpu.GpuIndex([copy.copy(seg)]) as ix:
        with pytest.raises(gpu.SlgError):
            ix.update_deleted(3, None, 300.0)
        with pytest.raises(gpu.SlgError):
            ix.update_deleted(0, None, 301.0)
