"""Restated WRITER of searchlite's index files — TEST INFRASTRUCTURE ONLY (like the rest of
oracle/): the reference is Rust and cannot run here, so no index written by it exists; the
loader (searchlite_amd/index_files.py, the product) is pinned by writing the reference's byte
formats with this restatement and reading them back, and by the reference's own roundtrip values.
"parity unpinned" applies: nothing here was compared against bytes produced by searchlite itself.

Follows, line by line (searchlite-core/src/):
  util/varint.rs:5-16          write_u64 / write_u32_var
  index/postings.rs:78-129     PostingsWriter::write_term
  index/terms.rs:10-25         write_terms (crc32 = crc32fast = zlib's polynomial, util/checksum.rs)
  index/fastfields.rs:409-424, 910-1128   "FFV1" + write_field (I64 / F64 / Str columns written here)
  index/segment.rs:43-53, 898-912, 1030-1053   SegmentFileMeta JSON, write_vector_file ("VCTR")
  index/segment.rs:914-937, index/manifest.rs:14-47   SegmentMeta / Manifest JSON, collect_checksums
"""
from __future__ import annotations

import json
import os
import struct
import zlib
from typing import Dict, List, Optional, Sequence

import numpy as np

BLOCK_META_FLAG = 1 << 31  # index/postings.rs:12
DEFAULT_BLOCK_SIZE = 128   # index/postings.rs:11


def varint(v: int) -> bytes:
    """util/varint.rs:5-11."""
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def write_term(doc_ids: Sequence[int], tfs: Sequence[int], positions: Optional[Sequence[Sequence[int]]] = None,
               keep_positions: bool = False) -> bytes:
    """index/postings.rs:78-129 for one term -> its bytes."""
    n = len(doc_ids)
    out = bytearray()
    out += struct.pack("<I", n)
    out += bytes([1 if keep_positions else 0])
    block_count = -(-n // DEFAULT_BLOCK_SIZE)
    out += struct.pack("<I", (block_count | BLOCK_META_FLAG) if block_count > 0 else 0)
    out += struct.pack("<I", int(doc_ids[-1]) if n else 0)
    max_tf = np.float32(0.0)
    for t in tfs:
        max_tf = max(max_tf, np.float32(t))
    out += struct.pack("<f", float(max_tf))
    if block_count > 0:
        out += struct.pack("<I", DEFAULT_BLOCK_SIZE)
        for a in range(0, n, DEFAULT_BLOCK_SIZE):
            out += struct.pack("<I", int(doc_ids[min(a + DEFAULT_BLOCK_SIZE, n) - 1]))
        for a in range(0, n, DEFAULT_BLOCK_SIZE):
            m = np.float32(0.0)
            for t in tfs[a:a + DEFAULT_BLOCK_SIZE]:
                m = max(m, np.float32(t))
            out += struct.pack("<f", float(m))
    for i in range(n):
        out += varint(int(doc_ids[i]))
        out += varint(int(tfs[i]))
        if keep_positions:
            pos = list(positions[i]) if positions is not None else []
            out += varint(len(pos))
            prev = 0
            for p in pos:
                out += varint(p - prev)
                prev = p
    return bytes(out)


def write_terms(pairs: Sequence) -> bytes:
    """index/terms.rs:10-25: pairs = [(key, offset)] already sorted by key."""
    buf = bytearray()
    for term, off in pairs:
        b = term.encode("utf-8")
        buf += varint(len(b))
        buf += b
        buf += struct.pack("<Q", off)
    return struct.pack("<Q", len(pairs)) + bytes(buf) + struct.pack("<I", zlib.crc32(bytes(buf)) & 0xFFFFFFFF)


def write_fast_fields(columns: Dict[str, tuple]) -> bytes:
    """index/fastfields.rs:409-424 + write_field.  columns[name] = ("i64", [int | None]) |
    ("f64", [float | None]) | ("str", [str | None])."""
    out = bytearray(b"FFV1")
    out += struct.pack("<I", len(columns))
    for name, (kind, values) in columns.items():
        nb = name.encode("utf-8")
        out += struct.pack("<I", len(nb)) + nb
        if kind in ("i64", "f64"):
            out += bytes([0 if kind == "i64" else 1])
            out += struct.pack("<I", len(values))
            out += bytes(1 if v is not None else 0 for v in values)   # write_presence
            for v in values:
                out += struct.pack("<q", int(v or 0)) if kind == "i64" else struct.pack("<d", float(v or 0.0))
        elif kind == "str":
            out += bytes([2])
            out += struct.pack("<I", len(values))
            d: List[str] = []
            idx = []
            for v in values:
                if v is None:
                    idx.append(0xFFFFFFFF)
                else:
                    if v not in d:
                        d.append(v)
                    idx.append(d.index(v))
            out += struct.pack("<I", len(d))
            for s in d:
                sb = s.encode("utf-8")
                out += struct.pack("<I", len(sb)) + sb
            for i in idx:
                out += struct.pack("<I", i)
        else:
            raise ValueError(kind)
    return bytes(out)


def write_vector_file(dim: int, metric: int, offsets: np.ndarray, values: np.ndarray) -> bytes:
    """index/segment.rs:1030-1053."""
    offsets = np.ascontiguousarray(offsets, dtype="<u4")
    values = np.ascontiguousarray(values, dtype="<f4")
    rows = int((offsets != 0xFFFFFFFF).sum())
    return struct.pack("<IIIBBHII", 0x56435452, 1, dim, metric, 0, 0, len(offsets), rows) + \
        offsets.tobytes() + values.tobytes()


def write_index(path: str, segments, keep_positions: bool = False, extra_columns: Optional[dict] = None,
                vector_field: str = "embedding", absolute_paths_of: Optional[str] = None) -> dict:
    """Write `segments` (searchlite_amd.segment.Segment objects with fields / term_dict / ext_ids)
    as a searchlite index directory: what IndexWriter::commit leaves behind for the files the
    scorer path reads.  The docstore (seg_<id>.docs) is written empty (no stored fields here).
    absolute_paths_of: directory name written into the manifest paths (default: `path`), to mimic an
    index that was moved after it was built."""
    os.makedirs(path, exist_ok=True)
    root = absolute_paths_of or path
    metas = []
    for si, seg in enumerate(segments):
        sid = f"{si:08x}-0000-4000-8000-{si:012x}"
        names = {k: f"seg_{sid}.{ext}" for k, ext in (("terms", "terms"), ("postings", "post"),
                                                        ("docstore", "docs"), ("fast", "fast"), ("meta", "meta"))}
        keys = sorted(seg.term_dict, key=lambda k: seg.term_dict[k])
        assert keys == sorted(keys), "term ids must follow sorted key order (index/postings.rs:56-60)"
        post = bytearray()
        pairs = []
        for k in keys:
            d, t = seg.postings(seg.term_dict[k])
            pairs.append((k, len(post)))
            pos = [list(range(int(x))) for x in t] if keep_positions else None
            post += write_term(d.tolist(), t.tolist(), pos, keep_positions)
        blobs = {"postings": bytes(post), "terms": write_terms(pairs), "docstore": b""}
        cols: Dict[str, tuple] = {}
        for fi, f in enumerate(seg.fields):
            a = seg.field_doc_len[fi] if fi < len(seg.field_doc_len) else None
            if a is None:
                continue
            # index/segment.rs:693-697: set only for docs that have the field; absent -> None
            cols["_len:" + f] = ("i64", [int(v) if v > 0 else None for v in a.tolist()])
        for name, col in (extra_columns or {}).items():
            cols[name] = col[si] if isinstance(col, list) and col and isinstance(col[0], tuple) else col
        blobs["fast"] = write_fast_fields(cols)
        avg = {f: float(seg.field_avgdl[fi]) for fi, f in enumerate(seg.fields)
               if fi < len(seg.field_doc_len) and seg.field_doc_len[fi] is not None}
        vmeta = {}
        vec_blob = None
        if seg.vec_dim:
            vec_blob = write_vector_file(seg.vec_dim, seg.vec_metric, seg.vec_offsets, seg.vec_values)
            vmeta[vector_field] = {"dim": seg.vec_dim, "metric": "Cosine" if seg.vec_metric == 0 else "L2",
                                   "vectors": int((seg.vec_offsets != 0xFFFFFFFF).sum())}
        seg_meta = {"doc_offsets": [0] * seg.n_docs,
                    "doc_ids": list(seg.ext_ids or [f"doc-{i:08d}" for i in range(seg.n_docs)]),
                    "avg_field_lengths": avg, "vector_fields": vmeta, "use_zstd": False}
        blobs["meta"] = json.dumps(seg_meta, indent=2).encode("utf-8")
        for kind, data in blobs.items():
            with open(os.path.join(path, names[kind]), "wb") as f:
                f.write(data)
        checksums = {kind: zlib.crc32(data) & 0xFFFFFFFF for kind, data in blobs.items()}
        paths = {k: os.path.join(root, v) for k, v in names.items()}
        if vec_blob is not None:
            vdir = f"seg_{sid}_vectors"
            os.makedirs(os.path.join(path, vdir), exist_ok=True)
            with open(os.path.join(path, vdir, vector_field + ".bin"), "wb") as f:
                f.write(vec_blob)
            paths["vector_dir"] = os.path.join(root, vdir)
            checksums[f"vector_{vector_field}_bin"] = zlib.crc32(vec_blob) & 0xFFFFFFFF
        deleted = []
        if seg.deleted is not None:
            deleted = np.nonzero(np.unpackbits(seg.deleted, bitorder="little")[:seg.n_docs])[0].tolist()
        metas.append({"id": sid, "generation": si + 1, "paths": paths, "doc_count": seg.n_docs,
                      "max_doc_id": max(seg.n_docs - 1, 0), "blockmax": True, "deleted_docs": deleted,
                      "avg_field_lengths": avg, "checksums": checksums})
    fields = list(segments[0].fields) if segments else []
    manifest = {"version": 1, "uuid": "00000000-0000-4000-8000-000000000000", "segments": metas,
                "committed_at": "1970-01-01T00:00:00+00:00",
                "schema": {"doc_id_field": "_id",
                           "text_fields": [{"name": f, "analyzer": "default", "stored": True, "indexed": True,
                                            "nullable": False} for f in fields],
                           "keyword_fields": [], "numeric_fields": [], "nested_fields": [],
                           "vector_fields": []}}
    with open(os.path.join(path, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=2)
    return manifest
