"""Loader of a searchlite index directory: MANIFEST.json ->
per segment seg_<id>.terms / .post / .fast / .meta (+ seg_<id>_vectors/<field>.bin) -> the
`Segment` arrays slg_index_create stages.  Every segment keeps its own docs / df / avgdl, exactly
what `search_segment` hands the scorer (api/reader.rs:2985-3000).

File formats (searchlite-core/src/):
  MANIFEST.json          index/manifest.rs:14-47   {version, uuid, segments: [SegmentMeta], schema}
  seg_<id>.terms         index/terms.rs:10-75      u64 count | count x {varint len, bytes, u64 offset} | crc32
  seg_<id>.post          index/postings.rs:78-129  decoded by libslg_segfile.so (include/searchlite_segfile.h)
  seg_<id>.fast ("FFV1") index/fastfields.rs:409-470, 910-1128, 1166-1436; doc lengths are the I64
                         columns "_len:<field>" (:1162-1164), read as api/reader.rs:3604-3621 does
                         (absent value -> 0 -> the scorer's max(avgdl, 1) fallback)
  seg_<id>.meta          index/segment.rs:43-53    JSON {doc_offsets, doc_ids, avg_field_lengths, vector_fields, ..}
  <field>.bin ("VCTR")   index/segment.rs:960-1059 magic, version, dim, metric, doc_count, rows, offsets, values
k1 / b are IndexOptions (api/types.rs:16-26), not stored in the index: the caller passes them
(product defaults 0.9 / 0.4, README.md:15).

PARITY UNPINNED: no index written by the reference exists in this environment (Rust cannot be
built here); the formats are pinned by a restated writer (oracle/segfile_writer.py) and the
reference's own roundtrip values (index/postings.rs:264-310, util/varint.rs:54-61).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import zlib
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import build as _build
from .segment import Segment

_lib = None


class SegFileError(RuntimeError):
    pass


def _load():
    global _lib
    if _lib is None:
        path = _build.build_segfile()
        L = C.CDLL(path)
        vp, u32, u64p = C.c_void_p, C.c_uint32, C.c_void_p
        L.slf_last_error.restype = C.c_char_p
        L.slf_postings_scan.restype = C.c_int
        L.slf_postings_scan.argtypes = [vp, C.c_size_t, vp, u32, vp, vp, u64p, u64p]
        L.slf_postings_decode.restype = C.c_int
        L.slf_postings_decode.argtypes = [vp, C.c_size_t, vp, u32, vp, vp, vp, vp, vp, vp, vp, vp]
        L.slf_varint_write.restype = C.c_int
        L.slf_varint_write.argtypes = [C.c_uint64, vp]
        L.slf_varint_read_u32.restype = C.c_int
        L.slf_varint_read_u32.argtypes = [vp, C.c_size_t, vp]
        _lib = L
    return _lib


def _check(rc: int) -> None:
    if rc < 0:
        raise SegFileError(_load().slf_last_error().decode("utf-8", "replace"))


def read_terms(buf: bytes):
    """index/terms.rs:27-75 -> (keys in file order, u64 offsets).  The trailing crc32 covers the
    entries (not the count)."""
    if len(buf) < 12:
        raise SegFileError("terms file is truncated")
    (count,) = struct.unpack_from("<Q", buf, 0)
    data, (crc,) = buf[8:-4], struct.unpack_from("<I", buf, len(buf) - 4)
    if zlib.crc32(data) & 0xFFFFFFFF != crc:
        raise SegFileError("terms file failed checksum validation")
    keys: List[str] = []
    offs = np.zeros(count, dtype=np.uint64)
    cur = 0
    for i in range(count):
        ln = shift = 0
        while True:  # util/varint.rs:18-29 read_u64
            if cur >= len(data):
                raise SegFileError("unterminated varint in terms file")
            b = data[cur]
            cur += 1
            ln |= (b & 0x7F) << shift
            if not b & 0x80:
                break
            shift += 7
        if cur + ln + 8 > len(data):
            raise SegFileError("terms file ended unexpectedly")
        keys.append(data[cur:cur + ln].decode("utf-8", "replace"))
        cur += ln
        (offs[i],) = struct.unpack_from("<Q", data, cur)
        cur += 8
    return keys, offs


def decode_postings(post: bytes, offsets: np.ndarray):
    """All posting lists of one seg_<id>.post image -> dict of CSR arrays + block-max metadata."""
    L = _load()
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    V = len(offsets)
    img = np.frombuffer(post, dtype=np.uint8)
    df = np.zeros(V, dtype=np.uint32)
    nblk = np.zeros(V, dtype=np.uint32)
    P, B = C.c_uint64(0), C.c_uint64(0)
    _check(L.slf_postings_scan(img.ctypes.data, len(img), offsets.ctypes.data, V, df.ctypes.data,
                               nblk.ctypes.data, C.addressof(P), C.addressof(B)))
    out = {
        "term_offsets": np.zeros(V + 1, dtype=np.uint64),
        "doc_ids": np.zeros(P.value, dtype=np.uint32), "tfs": np.zeros(P.value, dtype=np.uint32),
        "blk_offsets": np.zeros(V + 1, dtype=np.uint64),
        "blk_max_doc": np.zeros(B.value, dtype=np.uint32), "blk_max_tf": np.zeros(B.value, dtype=np.float32),
        "blk_size": np.zeros(V, dtype=np.uint32), "max_tf": np.zeros(V, dtype=np.float32),
    }
    _check(L.slf_postings_decode(img.ctypes.data, len(img), offsets.ctypes.data, V,
                                 out["term_offsets"].ctypes.data, out["doc_ids"].ctypes.data,
                                 out["tfs"].ctypes.data, out["blk_offsets"].ctypes.data,
                                 out["blk_max_doc"].ctypes.data, out["blk_max_tf"].ctypes.data,
                                 out["blk_size"].ctypes.data, out["max_tf"].ctypes.data))
    return out


# index/fastfields.rs:27-56 column type tags
_I64, _F64, _STR, _I64L, _F64L, _STRL, _I64N, _F64N, _STRN, _NCOUNT, _NPARENT = range(11)


def read_fast_fields(buf: bytes, want_prefix: Optional[str] = None) -> Dict[str, dict]:
    """FFV1 (index/fastfields.rs:1166-1436).  Returns {name: {"type", "n", ...}}; single-valued
    i64 / f64 columns carry "present" (bool[n]) and "values"; str columns "dict" and "values"
    (u32, 0xFFFFFFFF = absent); the list / nested kinds are walked (to reach the next column)
    and kept as raw arrays.  want_prefix: only columns whose name starts with it are kept."""
    if len(buf) < 8:
        return {}
    if buf[:4] != b"FFV1":
        raise SegFileError("invalid fast field header")
    cur = 4
    (nf,) = struct.unpack_from("<I", buf, cur)
    cur += 4
    out: Dict[str, dict] = {}

    def u32s(n):
        nonlocal cur
        if cur + 4 * n > len(buf):
            raise SegFileError("unexpected end of fast field file")
        a = np.frombuffer(buf, dtype="<u4", count=n, offset=cur)
        cur += 4 * n
        return a

    def vals8(n, dt):
        nonlocal cur
        if cur + 8 * n > len(buf):
            raise SegFileError("unexpected end of fast field file")
        a = np.frombuffer(buf, dtype=dt, count=n, offset=cur)
        cur += 8 * n
        return a

    def dictionary():
        nonlocal cur
        (dl,) = u32s(1)
        d = []
        for _ in range(int(dl)):
            (sl,) = u32s(1)
            if cur + int(sl) > len(buf):
                raise SegFileError("fast-field dictionary is truncated")
            d.append(buf[cur:cur + int(sl)].decode("utf-8", "replace"))
            cur += int(sl)
        return d

    for _ in range(nf):
        (nl,) = u32s(1)
        if cur + int(nl) + 1 > len(buf):
            raise SegFileError("fast-field column header is truncated")
        name = buf[cur:cur + int(nl)].decode("utf-8", "replace")
        cur += int(nl)
        ty = buf[cur]
        cur += 1
        (n,) = u32s(1)
        n = int(n)
        col: dict = {"type": int(ty), "n": n}
        if ty in (_I64, _F64):
            if cur + n > len(buf):
                raise SegFileError("unexpected end of fast field presence")
            col["present"] = np.frombuffer(buf, dtype=np.uint8, count=n, offset=cur).astype(bool)
            cur += n
            col["values"] = vals8(n, "<i8" if ty == _I64 else "<f8")
        elif ty in (_I64L, _F64L):
            col["offsets"] = u32s(n + 1)
            col["values"] = vals8(int(col["offsets"][-1]) if n + 1 else 0, "<i8" if ty == _I64L else "<f8")
        elif ty in (_I64N, _F64N):
            col["doc_offsets"] = u32s(n + 1)
            col["object_offsets"] = u32s(int(col["doc_offsets"][-1]) + 1)
            col["values"] = vals8(int(col["object_offsets"][-1]), "<i8" if ty == _I64N else "<f8")
        elif ty == _STR:
            col["dict"] = dictionary()
            col["values"] = u32s(n)
        elif ty == _STRL:
            col["dict"] = dictionary()
            col["offsets"] = u32s(n + 1)
            col["values"] = u32s(int(col["offsets"][-1]))
        elif ty == _STRN:
            col["dict"] = dictionary()
            col["doc_offsets"] = u32s(n + 1)
            col["object_offsets"] = u32s(int(col["doc_offsets"][-1]) + 1)
            col["values"] = u32s(int(col["object_offsets"][-1]))
        elif ty == _NCOUNT:
            col["values"] = u32s(n)
        elif ty == _NPARENT:
            col["offsets"] = u32s(n + 1)
            col["values"] = u32s(int(col["offsets"][-1]))
        else:
            raise SegFileError("invalid fast field type")
        if want_prefix is None or name.startswith(want_prefix):
            out[name] = col
    return out


_VCTR_MAGIC = 0x56435452


def read_vector_file(buf: bytes, n_docs: int):
    """index/segment.rs:1034-1096 -> (dim, metric code, offsets u32[n_docs], values f32[rows, dim])."""
    if len(buf) < 24:
        raise SegFileError("vector file is truncated")
    magic, version, dim, metric, _r0, _r1, docs, rows = struct.unpack_from("<IIIBBHII", buf, 0)
    if magic != _VCTR_MAGIC or version != 1:
        raise SegFileError("invalid vector file magic / version")
    if docs != n_docs:
        raise SegFileError(f"vector doc count mismatch: expected {n_docs}, found {docs}")
    if metric not in (0, 1):
        raise SegFileError(f"unknown vector metric code {metric}")
    if 24 + 4 * docs + 4 * rows * dim > len(buf):
        raise SegFileError("vector file is truncated")
    off = np.frombuffer(buf, dtype="<u4", count=docs, offset=24)
    vals = np.frombuffer(buf, dtype="<f4", count=rows * dim, offset=24 + 4 * docs).reshape(rows, dim)
    return int(dim), int(metric), off.copy(), vals.copy()


@dataclass
class LoadedIndex:
    segments: List[Segment]
    manifest: dict
    fields: List[str]             # field id -> name (text fields, then keyword fields, then others)
    block_max: List[dict]         # per segment: blk_offsets / blk_max_doc / blk_max_tf / blk_size / max_tf


def load_index(path: str, k1: float = 0.9, b: float = 0.4, vector_field: Optional[str] = None,
               verify_checksums: bool = True) -> LoadedIndex:
    """Open a searchlite index directory as Index::open + SegmentReader::open do
    (index/segment.rs:1239-1318), in manifest (= segment_ord, api/reader.rs:2670) order."""
    with open(os.path.join(path, "MANIFEST.json"), "rb") as f:
        manifest = json.loads(f.read())
    schema = manifest.get("schema", {})
    fields = [t["name"] for t in schema.get("text_fields", [])] + \
             [t["name"] for t in schema.get("keyword_fields", [])]
    fidx = {name: i for i, name in enumerate(fields)}
    segs: List[Segment] = []
    bms: List[dict] = []
    for meta in manifest.get("segments", []):
        sid = meta["id"]

        def blob(kind, ext):
            # paths in the manifest are absolute strings of the writing machine: resolve by name
            name = os.path.basename(meta.get("paths", {}).get(kind, "")) or f"seg_{sid}.{ext}"
            with open(os.path.join(path, name), "rb") as fh:
                data = fh.read()
            want = meta.get("checksums", {}).get({"postings": "postings", "terms": "terms", "fast": "fast",
                                                  "meta": "meta"}[kind])
            if verify_checksums and want is not None and zlib.crc32(data) & 0xFFFFFFFF != int(want):
                raise SegFileError(f"segment {sid} failed checksum for {kind}")
            return data

        seg_meta = json.loads(blob("meta", "meta"))
        n_docs = int(meta["doc_count"])
        keys, offs = read_terms(blob("terms", "terms"))
        dec = decode_postings(blob("postings", "post"), offs)
        for k in keys:  # fields that only occur as term prefixes (nested paths ...)
            f = k.split(":", 1)[0]
            if f not in fidx:
                fidx[f] = len(fields)
                fields.append(f)
        term_field = np.array([fidx[k.split(":", 1)[0]] for k in keys], dtype=np.uint16)
        fast = read_fast_fields(blob("fast", "fast"), want_prefix="_len:")
        F = len(fields)
        lens: List[Optional[np.ndarray]] = [None] * F
        for fi, name in enumerate(fields):
            col = fast.get("_len:" + name)
            if col is not None and col["type"] == _I64:
                v = np.where(col["present"], col["values"], 0).astype(np.float32)  # i64_value().unwrap_or(0) as f32
                a = np.zeros(n_docs, dtype=np.float32)
                a[:min(n_docs, len(v))] = v[:n_docs]
                lens[fi] = a
        avg_map = meta.get("avg_field_lengths") or seg_meta.get("avg_field_lengths", {})
        avg = np.array([np.float32(avg_map.get(name, 0.0)) for name in fields], dtype=np.float32)
        deleted = None
        dels = sorted(set(int(d) for d in meta.get("deleted_docs", [])))
        if dels:
            bits = np.zeros(n_docs, dtype=bool)
            bits[[d for d in dels if d < n_docs]] = True
            deleted = np.packbits(bits, bitorder="little")
        seg = Segment(n_docs=n_docs, term_offsets=dec["term_offsets"], doc_ids=dec["doc_ids"], tfs=dec["tfs"],
                      field_doc_len=lens, field_avgdl=avg,
                      docs=float(max(0, n_docs - len(dels))),  # live_docs, index/segment.rs:1362-1367
                      k1=k1, b=b, term_field=term_field, deleted=deleted, fields=list(fields),
                      term_dict={k: i for i, k in enumerate(keys)}, ext_ids=list(seg_meta.get("doc_ids", [])))
        vfields = seg_meta.get("vector_fields", {})
        vname = vector_field or (sorted(vfields)[0] if vfields else None)
        if vname is not None and vname in vfields:
            vdir = os.path.basename(meta.get("paths", {}).get("vector_dir") or f"seg_{sid}_vectors")
            with open(os.path.join(path, vdir, f"{vname}.bin"), "rb") as fh:
                dim, metric, voff, vvals = read_vector_file(fh.read(), n_docs)
            seg.vec_dim, seg.vec_metric, seg.vec_offsets, seg.vec_values = dim, metric, voff, vvals
        segs.append(seg)
        bms.append({k: dec[k] for k in ("blk_offsets", "blk_max_doc", "blk_max_tf", "blk_size", "max_tf")})
    for s in segs:  # all segments share the final field table
        pad = len(fields) - len(s.field_doc_len)
        if pad:
            s.field_doc_len = list(s.field_doc_len) + [None] * pad
            s.field_avgdl = np.concatenate([s.field_avgdl, np.zeros(pad, dtype=np.float32)])
            s.fields = list(fields)
    return LoadedIndex(segments=segs, manifest=manifest, fields=fields, block_max=bms)
