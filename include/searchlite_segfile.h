/*
 * searchlite_segfile.h — C ABI of the segment-file decoder (host only, no GPU): reads the byte
 * formats searchlite-core writes, so a real searchlite index directory can be staged with
 * slg_index_create.
 *
 * Formats restated (paths relative to searchlite-core/src/):
 *   seg_<id>.post   index/postings.rs:78-129 (PostingsWriter::write_term), read back as
 *                   index/postings.rs:142-212 (PostingsReader::read_at) does:
 *                     u32 doc_freq | u8 keep_positions | u32 block_count (bit 31 = block meta present)
 *                     | u32 max_doc_id | f32 max_tf
 *                     | [u32 block_size | u32 block_max_doc[block_count] | f32 block_max_tf[block_count]]
 *                     | doc_freq x { varint doc_id (ABSOLUTE, :115), varint tf
 *                                    [, varint n_pos, n_pos x varint delta] }
 *   varint          util/varint.rs:5-48 (LEB128, at most 5 bytes for a u32)
 * All integers little endian (index/codec.rs:6-22).  The term dictionary (seg_<id>.terms,
 * index/terms.rs:10-75), the FFV1 fast-field file (index/fastfields.rs:409-470,1166-1180), the
 * VCTR vector file (index/segment.rs:960-1120) and MANIFEST.json are fixed-width / JSON and are
 * parsed by the host mirror (searchlite_amd/index_files.py).
 *
 * Errors: negative return, slf_last_error() describes the last failure of this thread.
 */
#ifndef SEARCHLITE_SEGFILE_H
#define SEARCHLITE_SEGFILE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SLF_OK = 0, SLF_ERR_INVALID = -1, SLF_ERR_TRUNCATED = -2, SLF_ERR_FORMAT = -3 };

const char *slf_last_error(void);

/*
 * Pass 1: for the n_terms posting lists that start at offsets[t] inside the postings file image
 * `post` (n_bytes long): df_out[t] = doc_freq, blocks_out[t] = number of block-max entries the
 * staged list has (ceil(df / block size), recomputed at block 128 when the file carries none,
 * postings.rs:188-200).  Totals through *total_postings / *total_blocks.
 */
int slf_postings_scan(const uint8_t *post, size_t n_bytes, const uint64_t *offsets, uint32_t n_terms,
                      uint32_t *df_out, uint32_t *blocks_out, uint64_t *total_postings,
                      uint64_t *total_blocks);

/*
 * Pass 2: decode every list into CSR arrays (term_offsets[n_terms + 1] postings CSR,
 * doc_ids / tfs [total_postings]) and the block-max metadata (blk_offsets[n_terms + 1],
 * blk_max_doc / blk_max_tf [total_blocks], blk_size[n_terms], max_tf[n_terms] as
 * PostingsReader exposes them).  Positions are skipped (they never reach the scorer).
 * Any of the blk_* / max_tf outputs may be NULL.  Fails with SLF_ERR_FORMAT when doc ids of a
 * list are not strictly increasing.
 */
int slf_postings_decode(const uint8_t *post, size_t n_bytes, const uint64_t *offsets, uint32_t n_terms,
                        uint64_t *term_offsets, uint32_t *doc_ids, uint32_t *tfs,
                        uint64_t *blk_offsets, uint32_t *blk_max_doc, float *blk_max_tf,
                        uint32_t *blk_size, float *max_tf);

/* util/varint.rs:5-16 / :31-48 — exposed for the tests of the reference's own roundtrip values.
 * slf_varint_write returns the bytes written (<= 10); slf_varint_read_u32 returns the bytes
 * consumed or a negative error ("varint too long" past 5 bytes, as read_u32_var). */
int slf_varint_write(uint64_t v, uint8_t *out);
int slf_varint_read_u32(const uint8_t *buf, size_t n, uint32_t *value);

#ifdef __cplusplus
}
#endif
#endif /* SEARCHLITE_SEGFILE_H */
