// slg_plan_capi.cpp — a small C ABI over the host planner (slg_plan.cpp) for the CPU unit tests
// (tests/test_plan.py) and the CPU sanitizer run (tools/sanitize_cpu.sh): the planner is pure host
// code, so its invariants are checked without a GPU.  Built as lib/libslg_plan.so with g++; NOT
// part of libsearchlite_gpu.so's exported surface (the product calls slgplan::plan_batch directly).
#include <cstring>
#include <new>

#include "slg_plan.hpp"

extern "C" {

struct slgp_segment {
  uint32_t n_docs, n_terms;
  const uint64_t *term_offsets;
  const float *champ;  // [n_terms * 68] or NULL
};

struct slgp_facts {
  uint64_t n_postings, n_postings_essential, n_postings_nonessential, n_rounds, n_bounds, n_bnd, cand_total;
  uint64_t image_bytes;
  uint32_t n_sq, n_terms, n_slices, max_terms;
  uint32_t uniform, multi, plan_batch, nested, pruned, cand_mode;
  uint32_t sizeof_round_query, sizeof_term_ref;
  uint32_t deep;
};

// -> opaque plan or NULL (err / code filled)
void *slgp_plan(const slgp_segment *segs, uint32_t n_segs, const slg_tuning *tuning, uint32_t nq,
                const uint32_t *q_offsets, const uint32_t *q_term_ids, const float *q_weights,
                const slg_score_plans *plans, const int32_t *q_filter, uint32_t k, int strategy,
                const char *filter_live, uint32_t n_filters, char *err, uint32_t err_len, int *code) {
  try {
    std::vector<slgplan::SegView> views(n_segs);
    for (uint32_t s = 0; s < n_segs; s++) {
      views[s].n_docs = segs[s].n_docs;
      views[s].n_terms = segs[s].n_terms;
      views[s].term_offsets = segs[s].term_offsets;
      views[s].champ = segs[s].champ;
    }
    slgplan::BatchIn in;
    in.nq = nq;
    in.q_offsets = q_offsets;
    in.q_term_ids = q_term_ids;
    in.q_weights = q_weights;
    if (plans) in.plans = *plans;
    in.q_filter = q_filter;
    in.k = k;
    in.strategy = strategy;
    in.filter_live = filter_live;
    in.n_filters = n_filters;
    auto *p = new slgplan::Plan();
    try {
      slgplan::plan_batch(views, *tuning, in, *p);
    } catch (...) {
      delete p;
      throw;
    }
    if (code) *code = SLG_OK;
    return p;
  } catch (const slgplan::SlgError &e) {
    if (err && err_len) {
      std::strncpy(err, e.what(), err_len - 1);
      err[err_len - 1] = 0;
    }
    if (code) *code = e.code;
  } catch (const std::exception &e) {
    if (err && err_len) {
      std::strncpy(err, e.what(), err_len - 1);
      err[err_len - 1] = 0;
    }
    if (code) *code = SLG_ERR_INTERNAL;
  }
  return nullptr;
}

void slgp_facts_of(const void *plan, slgp_facts *f) {
  const auto &p = *static_cast<const slgplan::Plan *>(plan);
  f->n_postings = p.n_postings;
  f->n_postings_essential = p.n_postings_essential;
  f->n_postings_nonessential = p.n_postings_nonessential;
  f->n_rounds = p.n_rounds;
  f->n_bounds = p.n_bounds;
  f->n_bnd = p.n_bnd;
  f->cand_total = p.cand_total;
  f->image_bytes = p.image_bytes;
  f->n_sq = (uint32_t)p.sqs.size();
  f->n_terms = (uint32_t)p.terms.size();
  f->n_slices = (uint32_t)p.slice_sq.size();
  f->max_terms = p.max_terms;
  f->uniform = p.uniform;
  f->multi = p.multi;
  f->plan_batch = p.plan_batch;
  f->nested = p.nested;
  f->pruned = p.pruned;
  f->cand_mode = p.cand_mode;
  f->sizeof_round_query = (uint32_t)sizeof(slg::RoundQuery);
  f->sizeof_term_ref = (uint32_t)sizeof(slg::TermRef);
  f->deep = p.deep;
}

// what: 0 sub-queries (RoundQuery), 1 terms (TermRef), 2 slice_sq, 3 slice_seg, 4 slice_order,
// 5 query refs (2 x u32), 6 bnd_coarse, 7 q_postings (u64), 8 the packed image, 9 deep-tree nodes (PlanNode)
uint64_t slgp_bytes(const void *plan, int what) {
  const auto &p = *static_cast<const slgplan::Plan *>(plan);
  switch (what) {
    case 0: return p.sqs.size() * sizeof(slg::RoundQuery);
    case 1: return p.terms.size() * sizeof(slg::TermRef);
    case 2: return p.slice_sq.size() * 4;
    case 3: return p.slice_seg.size() * 4;
    case 4: return p.slice_order.size() * 4;
    case 5: return p.qrefs.size() * sizeof(slg::QueryRef);
    case 6: return p.bnd_coarse.size() * 4;
    case 7: return p.q_postings.size() * 8;
    case 8: return p.image_bytes;
    case 9: return p.nodes.size() * sizeof(slg::PlanNode);
  }
  return 0;
}

void slgp_copy(const void *plan, int what, void *dst) {
  const auto &p = *static_cast<const slgplan::Plan *>(plan);
  const void *src = nullptr;
  switch (what) {
    case 0: src = p.sqs.data(); break;
    case 1: src = p.terms.data(); break;
    case 2: src = p.slice_sq.data(); break;
    case 3: src = p.slice_seg.data(); break;
    case 4: src = p.slice_order.data(); break;
    case 5: src = p.qrefs.data(); break;
    case 6: src = p.bnd_coarse.data(); break;
    case 7: src = p.q_postings.data(); break;
    case 8: p.pack(static_cast<unsigned char *>(dst)); return;
    case 9: src = p.nodes.data(); break;
  }
  const uint64_t n = slgp_bytes(plan, what);
  if (n && src) std::memcpy(dst, src, n);
}

void slgp_free(void *plan) { delete static_cast<slgplan::Plan *>(plan); }

}  // extern "C"
