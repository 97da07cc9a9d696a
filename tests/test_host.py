"""Host-side logic (no GPU): tokenizer, segment builder, query folding, corpus generator."""
import numpy as np
import pytest

from searchlite_amd import corpus
from searchlite_amd.segment import (NO_TERM, SegmentBuilder, default_tokenize, fold_terms,
                                    parse_query_terms, resolve_query)


def test_default_tokenizer():
    """analysis/tokenizer.rs:7-29"""
    assert default_tokenize("Mushrooms & Spinach Orzo, 2–3 min!") == \
        ["mushrooms", "spinach", "orzo", "2", "3", "min"]
    assert default_tokenize("") == [] and default_tokenize("  -- ") == []
    assert default_tokenize("Sauté ÉCLAIR") == ["sauté", "Éclair"]  # ASCII-only lowercasing


def test_ascii_only_lowercase():
    # to_ascii_lowercase leaves non-ASCII capitals alone
    assert default_tokenize("ÀB") == ["Àb"]


def test_parse_and_fold():
    """api/query.rs:20-98 (plain terms) + api/reader.rs:2971-2983 (fold duplicates)"""
    assert parse_query_terms("Tomato basil", "description") == \
        [("description:tomato", 1.0), ("description:basil", 1.0)]
    assert parse_query_terms("title:Rust body:safety", "x") == [("title:rust", 1.0), ("body:safety", 1.0)]
    assert fold_terms([("a:x", 1.0), ("a:y", 1.0), ("a:x", 1.0)]) == [("a:x", 2.0), ("a:y", 1.0)]
    with pytest.raises(ValueError):
        parse_query_terms('"a phrase"', "f")
    with pytest.raises(ValueError):
        parse_query_terms("-not", "f")


def test_builder_multi_field_and_missing_fields():
    b = SegmentBuilder(["title", "body"], k1=0.9, b=0.4)
    b.add_document("b", {"title": "Fast search", "body": "rust search engine"})
    b.add_document("a", {"body": ["tiny", "search"]})
    seg = b.build()
    assert seg.ext_ids == ["a", "b"]
    assert list(seg.field_doc_len[0]) == [0.0, 2.0]      # doc "a" has no title => 0 (missing)
    assert list(seg.field_doc_len[1]) == [2.0, 3.0]
    assert seg.field_avgdl[0] == np.float32(2) / np.float32(2)  # total tokens / ALL docs
    assert seg.field_avgdl[1] == np.float32(5) / np.float32(2)
    assert seg.term_id("body:search") != NO_TERM and seg.term_id("title:nope") == NO_TERM
    d, tf = seg.postings(seg.term_id("body:search"))
    assert list(d) == [0, 1] and list(tf) == [1, 1]
    ids, w = resolve_query([seg, seg], fold_terms(parse_query_terms("search nope", "body")))
    assert ids.shape == (2, 2) and ids[1, 0] == NO_TERM and list(w) == [1.0, 1.0]
    keys = sorted(seg.term_dict, key=seg.term_dict.get)
    assert keys == sorted(keys)  # terms sorted by key (index/postings.rs:56-60)


def test_set_deleted_updates_live_docs():
    b = SegmentBuilder(["body"])
    for i in range(10):
        b.add_document(f"d{i}", {"body": "x"})
    seg = b.build()
    seg.set_deleted([3, 3, 7])
    assert seg.docs == 8.0 and seg.deleted[0] == (1 << 3) | (1 << 7)


def test_zipf_corpus_is_deterministic_and_consistent():
    a = corpus.zipf_segment(3000, 512, seed=5, n_threads=3)
    b = corpus.zipf_segment(3000, 512, seed=5, n_threads=1)   # thread count must not matter
    for x, y in ((a.term_offsets, b.term_offsets), (a.doc_ids, b.doc_ids), (a.tfs, b.tfs),
                 (a.field_doc_len[0], b.field_doc_len[0])):
        assert np.array_equal(x, y)
    c = corpus.zipf_segment(3000, 512, seed=6, n_threads=2)
    assert not np.array_equal(a.tfs, c.tfs)
    # doc_len == sum of tfs per doc; doc ids strictly increasing inside each list
    per_doc = np.zeros(a.n_docs)
    np.add.at(per_doc, a.doc_ids, a.tfs)
    assert np.array_equal(per_doc, a.field_doc_len[0])
    assert 128 <= a.field_doc_len[0].min() and a.field_doc_len[0].max() <= 384
    for t in (0, 1, 17, 200, 511):
        d, _ = a.postings(t)
        assert (np.diff(d.astype(np.int64)) > 0).all()
    # Zipf: rank-1 term is in (nearly) every doc, df decays with rank
    assert a.df(0) > a.df(10) > a.df(100) > a.df(500)
    assert abs(float(a.field_avgdl[0]) - 256.0) < 5.0


def test_zipf_queries_distinct_terms():
    offs, terms, w = corpus.zipf_queries(50, 3, rank_lo=64, rank_hi=512, seed=1, vocab=512)
    assert list(offs[:3]) == [0, 3, 6] and len(terms) == 150 and (w == 1).all()
    t = terms.reshape(50, 3)
    assert all(len(set(r)) == 3 for r in t) and t.min() >= 63 and t.max() < 511
