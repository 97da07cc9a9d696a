// slg_plan.hpp — the host planner of a query batch: a PURE function of the segments' host mirrors,
// the index's tuning and the caller's query arrays -> the descriptor image the kernels consume
// (sub-queries, term references, slices, launch order).  No HIP in here: the translation unit
// builds with g++ and is unit-tested and sanitized on a box without a GPU
// (tests/test_plan.py, tools/sanitize_cpu.sh); slg_api.hip uploads what it returns.
//
// What it mirrors: IndexReader::search_segment's preparation of the scorer call
// (searchlite-core/src/api/reader.rs:2971-3005: ScoredTerm list per segment, terms with empty
// postings dropped wand.rs:441), the score plan's shape (query/planner.rs:113-153), and the
// decisions the reference's cursors take while running (wand.rs:107-153 upper bounds ->
// here: threshold seed and MaxScore classification from the champion table).
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/searchlite_gpu.h"
#include "slg_desc.hpp"

namespace slgplan {

struct SlgError : std::runtime_error {
  int code;
  SlgError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// host mirror of one staged segment
struct SegView {
  uint32_t n_docs = 0, n_terms = 0;
  const uint64_t *term_offsets = nullptr;  // [n_terms + 1] as given (unpadded)
  const float *champ = nullptr;            // [n_terms * kChampions] or nullptr (champions off)
};

// the caller's arrays (slg_batch_prepare_plans)
struct BatchIn {
  uint32_t nq = 0;
  const uint32_t *q_offsets = nullptr;
  const uint32_t *q_term_ids = nullptr;
  const float *q_weights = nullptr;
  slg_score_plans plans{};            // all-NULL: term i of a query is leaf i, summed
  const int32_t *q_filter = nullptr;
  uint32_t k = 0;
  int strategy = SLG_STRATEGY_WAND;
  const char *filter_live = nullptr;  // [n_filters] 1 = the filter id is registered
  size_t n_filters = 0;
};

struct Plan {
  // ---- the descriptor image ----
  std::vector<slg::RoundQuery> sqs;
  std::vector<slg::TermRef> terms;
  std::vector<uint32_t> slice_sq, slice_seg, slice_order;
  std::vector<slg::QueryRef> qrefs;
  std::vector<uint32_t> bnd_coarse;  // sub-query of every 32nd round boundary
  std::vector<slg::PlanNode> nodes;  // canonical node tables of the queries with deep score trees
  std::vector<uint32_t> q_filter;    // [nq] 0 = none, f + 1 (empty when no query is filtered)
  // ---- accounting ----
  std::vector<uint64_t> q_postings;  // per query (stats.postings_advanced)
  uint64_t n_postings = 0, n_postings_essential = 0, n_postings_nonessential = 0;
  uint64_t n_rounds = 0, n_bounds = 0, n_bnd = 0, cand_total = 0;
  uint32_t max_terms = 0;
  // ---- what runs ----
  bool uniform = false;     // every sub-query fits the few-term kernel
  bool multi = false;       // many-term kernel
  bool plan_batch = false;  // some sub-query has a score plan
  bool nested = false;      // some sub-query has a two-level plan (groups of leaves)
  bool deep = false;        // some sub-query has a score tree of more than two levels
  bool pruned = false;      // some sub-query has non-essential lists (MaxScore)
  bool cand_mode = false;   // k > 256 on candidates + select
  // ---- packed image layout (pack()) ----
  size_t o_sq = 0, o_terms = 0, o_slice = 0, o_sseg = 0, o_sord = 0, o_q = 0, o_bc = 0, o_nodes = 0, image_bytes = 0;
  void layout();
  void pack(unsigned char *dst) const;  // dst: image_bytes bytes
};

// Throws SlgError (SLG_ERR_INVALID / SLG_ERR_UNSUPPORTED) on malformed input.
void plan_batch(const std::vector<SegView> &segs, const slg_tuning &tune, const BatchIn &in, Plan &out);

}  // namespace slgplan
