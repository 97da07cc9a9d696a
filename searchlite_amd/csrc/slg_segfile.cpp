// slg_segfile.cpp — host-only decoder of searchlite's posting-list file (include/searchlite_segfile.h).
// Restates index/postings.rs:142-212 (PostingsReader::read_at) over a memory image of the file
// for all terms at once, into the CSR arrays slg_segment_desc takes.
#include "../../include/searchlite_segfile.h"

#include <cstring>
#include <string>

namespace {

thread_local std::string g_err;

constexpr uint32_t kBlockMetaFlag = 1u << 31;  // index/postings.rs:12
constexpr uint32_t kDefaultBlock = 128;        // index/postings.rs:11

struct Cursor {
  const uint8_t *p;
  size_t n, at;
  bool u32(uint32_t &v) {
    if (at + 4 > n) return false;
    std::memcpy(&v, p + at, 4);  // little endian host (x86-64 / the GPU box)
    at += 4;
    return true;
  }
  bool f32(float &v) {
    uint32_t b;
    if (!u32(b)) return false;
    std::memcpy(&v, &b, 4);
    return true;
  }
  bool u8(uint8_t &v) {
    if (at + 1 > n) return false;
    v = p[at++];
    return true;
  }
  // util/varint.rs:31-48 read_u32_var: 7 bits per byte, "varint too long" once shift > 28
  int var32(uint32_t &v) {
    uint32_t value = 0, shift = 0;
    for (;;) {
      if (at >= n) return SLF_ERR_TRUNCATED;
      const uint8_t b = p[at++];
      value |= (uint32_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) {
        v = value;
        return SLF_OK;
      }
      shift += 7;
      if (shift > 28) return SLF_ERR_FORMAT;
    }
  }
};

struct Header {
  uint32_t df, block_count, max_doc_id, block_size;
  float max_tf;
  bool has_pos, has_meta;
};

int read_header(Cursor &c, Header &h) {
  uint8_t flag;
  uint32_t raw;
  if (!c.u32(h.df) || !c.u8(flag) || !c.u32(raw) || !c.u32(h.max_doc_id) || !c.f32(h.max_tf))
    return SLF_ERR_TRUNCATED;
  h.has_pos = flag == 1;  // (positions are on disk whatever the reader's keep_positions asks)
  h.has_meta = (raw & kBlockMetaFlag) != 0;
  h.block_count = raw & ~kBlockMetaFlag;
  h.block_size = kDefaultBlock;
  return SLF_OK;
}

int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}

}  // namespace

extern "C" {

const char *slf_last_error(void) { return g_err.c_str(); }

int slf_varint_write(uint64_t v, uint8_t *out) {  // util/varint.rs:5-11
  int n = 0;
  while (v >= 0x80) {
    out[n++] = (uint8_t)((v & 0x7F) | 0x80);
    v >>= 7;
  }
  out[n++] = (uint8_t)v;
  return n;
}

int slf_varint_read_u32(const uint8_t *buf, size_t n, uint32_t *value) {
  if (!buf || !value) return fail(SLF_ERR_INVALID, "NULL argument");
  Cursor c{buf, n, 0};
  const int rc = c.var32(*value);
  if (rc != SLF_OK) return fail(rc, rc == SLF_ERR_FORMAT ? "varint too long" : "unterminated varint");
  return (int)c.at;
}

int slf_postings_scan(const uint8_t *post, size_t n_bytes, const uint64_t *offsets, uint32_t n_terms,
                      uint32_t *df_out, uint32_t *blocks_out, uint64_t *total_postings,
                      uint64_t *total_blocks) {
  if (!post || (n_terms && !offsets)) return fail(SLF_ERR_INVALID, "NULL argument");
  uint64_t P = 0, B = 0;
  for (uint32_t t = 0; t < n_terms; t++) {
    if (offsets[t] >= n_bytes) return fail(SLF_ERR_TRUNCATED, "posting offset of term " + std::to_string(t) + " is outside the file");
    Cursor c{post, n_bytes, (size_t)offsets[t]};
    Header h;
    if (read_header(c, h) != SLF_OK) return fail(SLF_ERR_TRUNCATED, "truncated posting header of term " + std::to_string(t));
    uint32_t blocks;
    if (h.has_meta && h.block_count > 0) {
      blocks = h.block_count;
    } else {  // postings.rs:188-200: rebuilt at the default block size
      blocks = (h.df + kDefaultBlock - 1) / kDefaultBlock;
    }
    if (df_out) df_out[t] = h.df;
    if (blocks_out) blocks_out[t] = blocks;
    P += h.df;
    B += blocks;
  }
  if (total_postings) *total_postings = P;
  if (total_blocks) *total_blocks = B;
  return SLF_OK;
}

int slf_postings_decode(const uint8_t *post, size_t n_bytes, const uint64_t *offsets, uint32_t n_terms,
                        uint64_t *term_offsets, uint32_t *doc_ids, uint32_t *tfs,
                        uint64_t *blk_offsets, uint32_t *blk_max_doc, float *blk_max_tf,
                        uint32_t *blk_size, float *max_tf) {
  if (!post || (n_terms && !offsets) || !term_offsets) return fail(SLF_ERR_INVALID, "NULL argument");
  uint64_t P = 0, B = 0;
  term_offsets[0] = 0;
  if (blk_offsets) blk_offsets[0] = 0;
  for (uint32_t t = 0; t < n_terms; t++) {
    const std::string where = " (term " + std::to_string(t) + ")";
    if (offsets[t] >= n_bytes) return fail(SLF_ERR_TRUNCATED, "posting offset outside the file" + where);
    Cursor c{post, n_bytes, (size_t)offsets[t]};
    Header h;
    if (read_header(c, h) != SLF_OK) return fail(SLF_ERR_TRUNCATED, "truncated posting header" + where);
    uint32_t blocks = 0;
    const bool file_meta = h.has_meta && h.block_count > 0;
    if (file_meta) {
      if (!c.u32(h.block_size)) return fail(SLF_ERR_TRUNCATED, "truncated block size" + where);
      blocks = h.block_count;
      if (c.at + (size_t)blocks * 8 > n_bytes) return fail(SLF_ERR_TRUNCATED, "truncated block-max arrays" + where);
      for (uint32_t i = 0; i < blocks; i++) {
        uint32_t d;
        c.u32(d);
        if (blk_max_doc) blk_max_doc[B + i] = d;
      }
      for (uint32_t i = 0; i < blocks; i++) {
        float f;
        c.f32(f);
        if (blk_max_tf) blk_max_tf[B + i] = f;
      }
    }
    uint32_t prev = 0;
    for (uint32_t i = 0; i < h.df; i++) {
      uint32_t d, tf;
      int rc = c.var32(d);
      if (rc == SLF_OK) rc = c.var32(tf);
      if (rc != SLF_OK) return fail(rc, "bad varint in posting " + std::to_string(i) + where);
      if (h.has_pos) {  // postings.rs:176-183: count, then deltas — skipped
        uint32_t cnt, x;
        if ((rc = c.var32(cnt)) != SLF_OK) return fail(rc, "bad position count" + where);
        for (uint32_t j = 0; j < cnt; j++)
          if ((rc = c.var32(x)) != SLF_OK) return fail(rc, "bad position delta" + where);
      }
      if (i > 0 && d <= prev) return fail(SLF_ERR_FORMAT, "doc ids not strictly increasing" + where);
      prev = d;
      if (doc_ids) doc_ids[P + i] = d;
      if (tfs) tfs[P + i] = tf;
    }
    float mt = h.max_tf;
    if (!file_meta) {  // postings.rs:188-200
      h.block_size = kDefaultBlock;
      blocks = (h.df + kDefaultBlock - 1) / kDefaultBlock;
      for (uint32_t bi = 0; bi < blocks; bi++) {
        const uint32_t a = bi * kDefaultBlock, e = a + kDefaultBlock < h.df ? a + kDefaultBlock : h.df;
        float tmax = 0.0f;
        if (tfs)
          for (uint32_t i = a; i < e; i++) tmax = (float)tfs[P + i] > tmax ? (float)tfs[P + i] : tmax;
        if (blk_max_doc) blk_max_doc[B + bi] = doc_ids ? doc_ids[P + e - 1] : h.max_doc_id;
        if (blk_max_tf) blk_max_tf[B + bi] = tmax;
      }
    }
    if (blk_max_tf)  // postings.rs:201-204: max_tf = max(header, block maxima)
      for (uint32_t bi = 0; bi < blocks; bi++) mt = blk_max_tf[B + bi] > mt ? blk_max_tf[B + bi] : mt;
    if (max_tf) max_tf[t] = mt;
    if (blk_size) blk_size[t] = h.block_size;
    P += h.df;
    B += blocks;
    term_offsets[t + 1] = P;
    if (blk_offsets) blk_offsets[t + 1] = B;
  }
  return SLF_OK;
}

}  // extern "C"
