"""The oracle against the committed golden fixtures (CPU).  The same fixtures are replayed
through the HIP path in tests/test_gpu_parity.py."""
import json
import os

import numpy as np
import pytest

from tests.util import GOLDEN, assert_same_hits, golden_expected, load_golden


@pytest.mark.parametrize("name", ["recipes.npz", "pruning40.npz", "two_segments.npz"])
@pytest.mark.parametrize("strategy", ["BM25", "WAND"])
def test_oracle_reproduces_golden(oracle, name, strategy):
    segs, z = load_golden(name)
    got = oracle.search_batch(segs, z["q_offsets"], z["q_terms"], z["q_weights"], int(z["k"]),
                              strategy=getattr(oracle, strategy))
    assert_same_hits(got, golden_expected(z), 0.0, f"{name} {strategy}")


def test_pruning40_bmw_block4(oracle):
    """tests/pruning.rs:44-104: Bmw with bmw_block_size 4 equals Bm25 on this corpus shape."""
    segs, z = load_golden("pruning40.npz")
    got = oracle.search_batch(segs, z["q_offsets"], z["q_terms"], z["q_weights"], int(z["k"]),
                              strategy=oracle.BMW, block_size=4)
    assert_same_hits(got, golden_expected(z), 1e-5, "bmw")


def test_recipes_config1_top10_ids():
    meta = json.load(open(os.path.join(GOLDEN, "recipes.json")))
    segs, z = load_golden("recipes.npz")
    assert len(meta["queries"]) == 10 and segs[0].n_docs == 300
    for qi, top in enumerate(meta["top10"]):
        assert [meta["ext_ids"][int(d)] for d in z["exp_doc"][qi][:len(top)]] == [t[0] for t in top]
        assert np.allclose([t[1] for t in top], z["exp_score"][qi][:len(top)], rtol=0, atol=0)
        # sorted: score desc, then doc asc (ids are already in external-id order)
        sc = z["exp_score"][qi][:int(z["exp_count"][qi])]
        dd = z["exp_doc"][qi][:int(z["exp_count"][qi])]
        for i in range(len(sc) - 1):
            assert sc[i] > sc[i + 1] or (sc[i] == sc[i + 1] and dd[i] < dd[i + 1])


def test_two_segments_tie_order():
    """tests/smoke.rs:853-950"""
    _, z = load_golden("two_segments.npz")
    n = int(z["exp_count"][0])
    assert n == 6
    assert list(zip(z["exp_seg"][0][:n], z["exp_doc"][0][:n])) == [(0, 0), (0, 1), (0, 2),
                                                                    (1, 0), (1, 1), (1, 2)]


@pytest.mark.parametrize("metric,name", [(0, "cos"), (1, "l2")])
def test_oracle_reproduces_rerank_golden(oracle, metric, name):
    z = np.load(os.path.join(GOLDEN, "rerank16.npz"))
    for i in range(len(z["alpha"])):
        d, s, v = oracle.rerank(metric, z["vec_offsets"], z["vec_values"], z["qvecs"][i],
                                float(z["alpha"][i]), z["cand_doc"][i], z["cand_bm25"][i], 10)
        assert np.array_equal(d, z[f"exp_doc_{name}"][i])
        assert np.array_equal(s.view(np.uint32), z[f"exp_score_{name}"][i].view(np.uint32))


def test_oracle_reproduces_recipes_on_default_fields(oracle):
    """BASELINE config 1 as an unmodified request: `fields: None` = all four text fields of the
    recipes schema (api/reader.rs:2576-2586), one ScorePlan leaf per query word fed by its
    `<field>:word` terms, leaves summed.  Bm25 bit-exact; Wand (heap pop order inside a leaf) within
    the reference's own 1e-5."""
    segs, z = load_golden("recipes_default.npz")
    assert len(segs[0].field_doc_len) == 4 and segs[0].n_docs == 300
    kw = dict(q_leaf=z["q_leaf"], q_plan=np.zeros(len(z["q_nleaves"]), np.int32), q_nleaves=z["q_nleaves"])
    got = oracle.search_batch(segs, z["q_offsets"], z["q_terms"], z["q_weights"], int(z["k"]),
                              strategy=oracle.BM25, **kw)
    assert_same_hits(got, golden_expected(z), 0.0, "recipes default fields, Bm25")
    wand = oracle.search_batch(segs, z["q_offsets"], z["q_terms"], z["q_weights"], int(z["k"]),
                               strategy=oracle.WAND, **kw)
    assert_same_hits(wand, golden_expected(z), 1e-5, "recipes default fields, Wand")
    meta = json.load(open(os.path.join(GOLDEN, "recipes_default.json")))
    assert meta["fields"] == ["text", "title", "description", "instructions"]
    assert all(len(q["keys"]) == 8 for q in meta["queries"])  # 2 words x 4 fields
