"""N2 — segment-file loader (searchlite_amd/index_files.py + libslg_segfile.so) against the
restated writer (oracle/segfile_writer.py) and the reference's own roundtrip values.  CPU only."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

from oracle import segfile_writer as W
from searchlite_amd import index_files as IF
from tests.util import random_multifield_segment

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_varint_roundtrip_values_of_the_reference():
    """util/varint.rs:54-61: 0, 1, 127, 128, 16384, u32::MAX; read_u32_var rejects > 5 bytes."""
    L = IF._load()
    for val in [0, 1, 127, 128, 16384, 0xFFFFFFFF]:
        buf = (C.c_uint8 * 10)()
        n = L.slf_varint_write(val, buf)
        assert bytes(buf[:n]) == W.varint(val)
        got = C.c_uint32()
        assert L.slf_varint_read_u32(buf, n, C.addressof(got)) == n and got.value == val
    bad = (C.c_uint8 * 6)(0x80, 0x80, 0x80, 0x80, 0x80, 0x01)
    got = C.c_uint32()
    assert L.slf_varint_read_u32(bad, 6, C.addressof(got)) < 0
    assert b"too long" in L.slf_last_error()
    assert L.slf_varint_read_u32(bad, 2, C.addressof(got)) < 0  # unterminated


def test_postings_roundtrip_values_of_the_reference():
    """index/postings.rs:264-310 writes_and_reads_postings: {doc 1, tf 2, positions [1, 3]},
    {doc 2, tf 1, positions [4]} with positions kept: 2 entries, max_tf >= 2, ONE block-max entry."""
    blob = W.write_term([1, 2], [2, 1], [[1, 3], [4]], keep_positions=True)
    # header: doc_freq | flag | blocks | max_doc | max_tf | block_size | max_doc[1] | max_tf[1]
    assert struct.unpack_from("<IB", blob, 0) == (2, 1)
    assert struct.unpack_from("<I", blob, 5)[0] == (1 | W.BLOCK_META_FLAG)
    dec = IF.decode_postings(blob, np.array([0], dtype=np.uint64))
    assert dec["doc_ids"].tolist() == [1, 2] and dec["tfs"].tolist() == [2, 1]
    assert dec["max_tf"][0] >= 2.0
    assert dec["blk_offsets"].tolist() == [0, 1]
    assert dec["blk_max_doc"].tolist() == [2] and dec["blk_max_tf"].tolist() == [2.0]
    assert dec["blk_size"].tolist() == [128]
    # two lists back to back, second without positions, third empty (doc_freq 0, no block meta)
    b2 = W.write_term(list(range(0, 600, 2)), [1 + (i % 3) for i in range(300)])
    b3 = W.write_term([], [])
    img = blob + b2 + b3
    dec = IF.decode_postings(img, np.array([0, len(blob), len(blob) + len(b2)], dtype=np.uint64))
    assert dec["term_offsets"].tolist() == [0, 2, 302, 302]
    assert dec["doc_ids"][2:].tolist() == list(range(0, 600, 2))
    assert dec["blk_offsets"].tolist() == [0, 1, 4, 4]               # ceil(300 / 128) = 3 blocks
    assert dec["blk_max_doc"][1:].tolist() == [254, 510, 598]
    assert dec["blk_max_tf"][1:].tolist() == [3.0, 3.0, 3.0]


def test_postings_without_block_meta_are_rebuilt_at_block_128():
    """postings.rs:188-200: a list written without block metadata gets it rebuilt by the reader."""
    docs, tfs = list(range(5, 5 + 200)), [1 + (i % 5) for i in range(200)]
    body = b"".join(W.varint(d) + W.varint(t) for d, t in zip(docs, tfs))
    blob = struct.pack("<IBIIf", 200, 0, 0, docs[-1], 5.0) + body
    dec = IF.decode_postings(blob, np.array([0], dtype=np.uint64))
    assert dec["doc_ids"].tolist() == docs
    assert dec["blk_max_doc"].tolist() == [docs[127], docs[199]]
    assert dec["blk_max_tf"].tolist() == [5.0, 5.0] and dec["max_tf"][0] == 5.0


def test_malformed_postings_fail_cleanly():
    blob = W.write_term([3, 9, 12], [1, 1, 1])
    with pytest.raises(IF.SegFileError):
        IF.decode_postings(blob[:-1], np.array([0], dtype=np.uint64))          # truncated varint
    with pytest.raises(IF.SegFileError):
        IF.decode_postings(blob, np.array([len(blob) + 4], dtype=np.uint64))   # offset outside
    bad = W.write_term([3, 3], [1, 1])
    with pytest.raises(IF.SegFileError) as e:
        IF.decode_postings(bad, np.array([0], dtype=np.uint64))
    assert "increasing" in str(e.value)


def test_terms_file_roundtrip_and_checksum():
    """index/terms.rs:79-95 roundtrips_terms_file: alpha/beta/gamma -> 10/20/30."""
    buf = W.write_terms([("alpha", 10), ("beta", 20), ("gamma", 30)])
    keys, offs = IF.read_terms(buf)
    assert keys == ["alpha", "beta", "gamma"] and offs.tolist() == [10, 20, 30]
    broken = bytearray(buf)
    broken[10] ^= 0xFF
    with pytest.raises(IF.SegFileError) as e:
        IF.read_terms(bytes(broken))
    assert "checksum" in str(e.value)


def test_fast_fields_columns():
    buf = W.write_fast_fields({"_len:body": ("i64", [3, None, 7]), "price": ("f64", [1.5, 2.5, None]),
                               "tag": ("str", ["news", None, "tech"])})
    cols = IF.read_fast_fields(buf)
    assert cols["_len:body"]["values"].tolist() == [3, 0, 7] and cols["_len:body"]["present"].tolist() == [True, False, True]
    assert cols["price"]["values"].tolist()[:2] == [1.5, 2.5]
    assert cols["tag"]["dict"] == ["news", "tech"] and cols["tag"]["values"].tolist() == [0, 0xFFFFFFFF, 1]
    assert list(IF.read_fast_fields(buf, want_prefix="_len:")) == ["_len:body"]


def golden_recipes_with_dictionary():
    """tests/golden/recipes.npz (BASELINE config 1: the reference's examples/recipes corpus, 300 docs,
    three text fields) with a term dictionary and external ids attached.  The fixture stores term ids
    only, in sorted-key order; keys are rebuilt as "<field>:<7-digit term id>", which sorts the same way."""
    from tests.util import load_golden
    segs, z = load_golden("recipes.npz")
    seg = segs[0]
    seg.fields = ["title", "description", "instructions"]
    keys = [f"{seg.fields[int(seg.term_field[t])]}:{t:07d}" for t in range(seg.n_terms)]
    assert keys == sorted(keys)
    seg.term_dict = {k: i for i, k in enumerate(keys)}
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "recipes.json")))
    seg.ext_ids = list(meta["ext_ids"])
    return seg, z


def _assert_same_segment(a, b):
    assert a.n_docs == b.n_docs and a.docs == b.docs
    for name in ("term_offsets", "doc_ids", "tfs", "term_field", "field_avgdl"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert a.term_dict == b.term_dict and a.ext_ids == b.ext_ids and a.fields == b.fields
    for x, y in zip(a.field_doc_len, b.field_doc_len):
        assert (x is None and y is None) or np.array_equal(x, y)
    assert (a.deleted is None and b.deleted is None) or np.array_equal(a.deleted, b.deleted)


@pytest.mark.parametrize("keep_positions", [False, True])
def test_index_directory_roundtrip_multi_segment(tmp_path, keep_positions):
    """Two multi-field segments (tombstones in one, vectors in the other) written as an index
    directory with the manifest paths pointing at ANOTHER machine's directory, loaded back: every
    array slg_index_create stages is identical."""
    from searchlite_amd.segment import SegmentBuilder
    rng = np.random.default_rng(8)
    words = [f"w{i}" for i in range(40)]
    segs = []
    for s in range(2):
        b = SegmentBuilder(["title", "body"], k1=1.2, b=0.75)
        for d in range(150 + 30 * s):
            doc = {"body": " ".join(rng.choice(words, size=int(rng.integers(1, 30))))}
            if rng.random() < 0.7:
                doc["title"] = " ".join(rng.choice(words, size=int(rng.integers(1, 5))))
            b.add_document(f"doc-{s}-{d:05d}", doc)
        segs.append(b.build())
    segs[0].set_deleted([3, 77, 149])
    segs[1].vec_dim, segs[1].vec_metric = 4, 1
    off = np.arange(segs[1].n_docs, dtype=np.uint32)
    off[5] = 0xFFFFFFFF
    off[6:] -= 1
    segs[1].vec_offsets = off
    segs[1].vec_values = rng.standard_normal((segs[1].n_docs - 1, 4)).astype(np.float32)
    W.write_index(str(tmp_path), segs, keep_positions=keep_positions, absolute_paths_of="/srv/elsewhere/index")
    li = IF.load_index(str(tmp_path), k1=1.2, b=0.75)
    assert li.fields == ["title", "body"] and len(li.segments) == 2
    for a, b_ in zip(segs, li.segments):
        _assert_same_segment(a, b_)
    assert li.segments[1].vec_dim == 4 and li.segments[1].vec_metric == 1
    assert np.array_equal(li.segments[1].vec_offsets, off)
    assert np.array_equal(li.segments[1].vec_values, segs[1].vec_values)
    assert li.segments[0].docs == segs[0].n_docs - 3
    # block-max metadata of the first term = last doc / max tf of each 128-posting block
    d, t = segs[0].postings(0)
    bm = li.block_max[0]
    assert bm["blk_max_doc"][:int(bm["blk_offsets"][1])].tolist() == [int(d[min(a + 128, len(d)) - 1]) for a in range(0, len(d), 128)]
    # a flipped byte in the postings file is caught by the manifest checksum
    name = [n for n in os.listdir(tmp_path) if n.endswith(".post")][0]
    raw = bytearray(open(tmp_path / name, "rb").read())
    raw[-1] ^= 0x55
    open(tmp_path / name, "wb").write(raw)
    with pytest.raises(IF.SegFileError) as e:
        IF.load_index(str(tmp_path))
    assert "checksum" in str(e.value)


def test_loader_has_no_oracle_dependency():
    src = open(os.path.join(ROOT, "searchlite_amd", "index_files.py")).read()
    assert "import oracle" not in src and "from oracle" not in src


def test_recipes_corpus_roundtrip(tmp_path):
    """BASELINE config 1's corpus written as a searchlite index directory and loaded back."""
    seg, _ = golden_recipes_with_dictionary()
    W.write_index(str(tmp_path), [seg])
    li = IF.load_index(str(tmp_path), k1=seg.k1, b=seg.b)
    _assert_same_segment(seg, li.segments[0])
    assert li.segments[0].ext_ids[0] == "recipe_0001" and li.segments[0].n_docs == 300
