// searchlite_gpu.hpp — header-only C++ mirror of searchlite-core's scorer interface on top of
// the C ABI (searchlite_gpu.h).  Names and argument meaning follow
// searchlite-core/src/query/wand.rs: RankedDoc (:17-21), QueryStats (:45-50), ScoredTerm (:65-75),
// execute_top_k / execute_top_k_with_stats (:338-395), ExecutionStrategy (api/types.rs:6-13).
//
// Differences forced by the device boundary: a ScoredTerm names its posting list by term id inside
// a staged Segment (the postings, doc lengths, avgdl, docs, k1, b were handed over at staging,
// exactly the values search_segment puts into ScoredTerm, api/reader.rs:2985-3000), and the
// `accept` closure is fixed to `!is_deleted(doc)` (the GPU eligibility predicate).  Errors of the
// C ABI become searchlite::gpu::Error (the reference returns anyhow::Result).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "searchlite_gpu.h"

namespace searchlite {
namespace gpu {

using DocId = uint32_t;

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

enum class ExecutionStrategy : int { Bm25 = SLG_STRATEGY_BM25, Wand = SLG_STRATEGY_WAND, Bmw = SLG_STRATEGY_BMW };

struct RankedDoc {  // query/wand.rs:17-21
  DocId doc_id;
  float score;
};

struct QueryStats {  // query/wand.rs:45-50
  size_t scored_docs = 0, candidates_examined = 0, postings_advanced = 0;
};

struct ScoredTerm {  // query/wand.rs:65-75 (postings named by term id; leaf = position)
  uint32_t term_id;
  float weight = 1.0f;
};

// One decoded segment, the form search_segment feeds the scorer.
struct SegmentData {
  uint32_t n_docs = 0;
  std::vector<uint64_t> term_offsets;  // V+1
  std::vector<uint32_t> doc_ids, tfs;
  std::vector<float> doc_lengths;  // one field; 0 = missing
  float avgdl = 0, docs = 0, k1 = 0.9f, b = 0.4f;
  std::vector<uint8_t> deleted;  // optional bitmap
};

class Index {
 public:
  explicit Index(const std::vector<SegmentData> &segs, int device = 0) {
    std::vector<slg_segment_desc> d(segs.size());
    std::vector<const float *> lens(segs.size());
    for (size_t i = 0; i < segs.size(); i++) {
      const SegmentData &s = segs[i];
      lens[i] = s.doc_lengths.empty() ? nullptr : s.doc_lengths.data();
      d[i] = slg_segment_desc{};
      d[i].n_docs = s.n_docs;
      d[i].n_terms = (uint32_t)s.term_offsets.size() - 1;
      d[i].term_offsets = s.term_offsets.data();
      d[i].doc_ids = s.doc_ids.data();
      d[i].tfs = s.tfs.data();
      d[i].n_fields = 1;
      d[i].field_doc_len = &lens[i];
      d[i].field_avgdl = &s.avgdl;
      d[i].docs = s.docs;
      d[i].k1 = s.k1;
      d[i].b = s.b;
      d[i].deleted = s.deleted.empty() ? nullptr : s.deleted.data();
    }
    n_segs_ = (uint32_t)segs.size();
    for (const SegmentData &sd : segs) n_docs_.push_back(sd.n_docs);
    h_ = slg_index_create(d.data(), n_segs_, device);
    if (!h_) throw Error(SLG_ERR_INVALID, slg_last_error());
  }
  ~Index() { slg_index_destroy(h_); }
  Index(const Index &) = delete;
  Index &operator=(const Index &) = delete;
  slg_index *handle() const { return h_; }
  uint32_t n_segs() const { return n_segs_; }
  uint32_t n_docs(uint32_t segment) const { return n_docs_[segment]; }

 private:
  slg_index *h_ = nullptr;
  uint32_t n_segs_ = 0;
  std::vector<uint32_t> n_docs_;
};

// execute_top_k_with_stats (query/wand.rs:374-395) against one segment of the index.
inline std::vector<RankedDoc> execute_top_k_with_stats(Index &index, const std::vector<ScoredTerm> &terms,
                                                       size_t k, ExecutionStrategy strategy,
                                                       QueryStats *stats = nullptr, uint32_t segment = 0) {
  std::vector<uint32_t> ids(terms.size() * index.n_segs(), SLG_NO_TERM);
  std::vector<float> w(terms.size());
  for (size_t i = 0; i < terms.size(); i++) {
    ids[i * index.n_segs() + segment] = terms[i].term_id;
    w[i] = terms[i].weight;
  }
  slg_query q{(uint32_t)terms.size(), ids.data(), w.data()};
  std::vector<uint32_t> doc(k ? k : 1), seg(k ? k : 1);
  std::vector<float> score(k ? k : 1);
  uint32_t count = 0;
  slg_stats st{};
  const int rc = slg_search_batch(index.handle(), &q, 1, (uint32_t)k, (int)strategy, doc.data(), seg.data(),
                                  score.data(), &count, stats ? &st : nullptr);
  if (rc != SLG_OK) throw Error(rc, slg_last_error());
  if (stats) {
    stats->scored_docs += st.scored_docs;
    stats->candidates_examined += st.candidates_examined;
    stats->postings_advanced += st.postings_advanced;
  }
  std::vector<RankedDoc> out(count);
  for (uint32_t i = 0; i < count; i++) out[i] = RankedDoc{doc[i], score[i]};
  return out;
}

// execute_top_k (query/wand.rs:338-356)
inline std::vector<RankedDoc> execute_top_k(Index &index, const std::vector<ScoredTerm> &terms, size_t k,
                                            ExecutionStrategy strategy, uint32_t segment = 0) {
  return execute_top_k_with_stats(index, terms, k, strategy, nullptr, segment);
}

// The reference's `accept: FnMut(DocId, f32) -> bool` argument (wand.rs:343, called at :512 /
// :555 / :858) for predicates that depend on the doc only (deleted docs, filters, cursors:
// api/reader.rs:3009-3036).  A closure cannot cross the C ABI, so it is evaluated once per doc
// into a bitmap and registered as a doc filter for the duration of the call.
template <typename Accept>
inline std::vector<RankedDoc> execute_top_k_with_accept(Index &index, const std::vector<ScoredTerm> &terms,
                                                        size_t k, ExecutionStrategy strategy, Accept accept,
                                                        uint32_t segment = 0) {
  const uint32_t n = index.n_docs(segment);
  std::vector<uint8_t> bits(((size_t)n + 7) / 8 + 1, 0);
  for (uint32_t d = 0; d < n; d++)
    if (accept(d)) bits[d >> 3] |= (uint8_t)(1u << (d & 7));
  std::vector<const uint8_t *> per_seg(index.n_segs(), nullptr);
  per_seg[segment] = bits.data();
  const int fid = slg_index_add_filter(index.handle(), per_seg.data());
  if (fid < 0) throw Error(fid, slg_last_error());
  std::vector<uint32_t> ids(terms.size() * index.n_segs(), SLG_NO_TERM);
  std::vector<float> w(terms.size());
  for (size_t i = 0; i < terms.size(); i++) {
    ids[i * index.n_segs() + segment] = terms[i].term_id;
    w[i] = terms[i].weight;
  }
  slg_query q{(uint32_t)terms.size(), ids.data(), w.data()};
  std::vector<uint32_t> doc(k ? k : 1), seg(k ? k : 1);
  std::vector<float> score(k ? k : 1);
  uint32_t count = 0;
  const int32_t qf = fid;
  const int rc = slg_search_batch_filtered(index.handle(), &q, 1, &qf, (uint32_t)k, (int)strategy, doc.data(),
                                           seg.data(), score.data(), &count, nullptr);
  const std::string msg = rc != SLG_OK ? slg_last_error() : "";
  slg_index_remove_filter(index.handle(), fid);
  if (rc != SLG_OK) throw Error(rc, msg);
  std::vector<RankedDoc> out(count);
  for (uint32_t i = 0; i < count; i++) out[i] = RankedDoc{doc[i], score[i]};
  return out;
}

}  // namespace gpu
}  // namespace searchlite
