// The reference's query/wand.rs unit tests (:951-1052), replayed through the C++ host mirror
// (include/searchlite_gpu.hpp) on the GPU.  Exit code 0 = all passed.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "searchlite_gpu.hpp"

using namespace searchlite::gpu;

#define CHECK(c)                                                     \
  do {                                                               \
    if (!(c)) {                                                      \
      std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      std::exit(1);                                                  \
    }                                                                \
  } while (0)

// term_from_entries (wand.rs:935-949): avgdl 10, docs 10, k1 1.2, b 0.75, doc_lengths 10
static SegmentData two_term_segment() {
  SegmentData s;
  s.n_docs = 4;
  s.term_offsets = {0, 2, 3};
  s.doc_ids = {1, 3, 3};
  s.tfs = {2, 1, 3};
  s.doc_lengths.assign(4, 10.0f);
  s.avgdl = 10.0f;
  s.docs = 10.0f;
  s.k1 = 1.2f;
  s.b = 0.75f;
  return s;
}

static void ranked_doc_ordering_prefers_smaller_id_on_tie() {  // wand.rs:951-966
  SegmentData s;
  s.n_docs = 3;
  s.term_offsets = {0, 2};
  s.doc_ids = {1, 2};
  s.tfs = {1, 1};
  s.doc_lengths.assign(3, 10.0f);
  s.avgdl = 10.0f;
  s.docs = 10.0f;
  s.k1 = 1.2f;
  s.b = 0.75f;
  Index ix({s});
  auto one = execute_top_k(ix, {{0, 1.0f}}, 1, ExecutionStrategy::Wand);
  CHECK(one.size() == 1 && one[0].doc_id == 1);
  auto two = execute_top_k(ix, {{0, 1.0f}}, 2, ExecutionStrategy::Wand);
  CHECK(two.size() == 2 && two[0].doc_id == 1 && two[1].doc_id == 2 && two[0].score == two[1].score);
}

static void brute_force_matches_wand_results() {  // wand.rs:968-1011
  Index ix({two_term_segment()});
  auto brute = execute_top_k(ix, {{0, 1.0f}, {1, 1.0f}}, 2, ExecutionStrategy::Bm25);
  auto wand = execute_top_k(ix, {{0, 1.0f}, {1, 1.0f}}, 2, ExecutionStrategy::Wand);
  CHECK(brute.size() == wand.size() && brute.size() == 2);
  for (size_t i = 0; i < brute.size(); i++) {
    CHECK(brute[i].doc_id == wand[i].doc_id);
    CHECK(std::fabs(brute[i].score - wand[i].score) < 1e-6f);
  }
  CHECK(brute[0].doc_id == 3 && brute[1].doc_id == 1);
  CHECK(std::fabs(brute[0].score - 6.6957893f) < 1e-5f && std::fabs(brute[1].score - 3.0576911f) < 1e-5f);
}

static void stats_and_k_zero() {  // wand.rs:413-416, :472, :500-503
  Index ix({two_term_segment()});
  QueryStats st;
  auto hits = execute_top_k_with_stats(ix, {{0, 1.0f}, {1, 1.0f}}, 2, ExecutionStrategy::Bm25, &st);
  CHECK(hits.size() == 2 && st.postings_advanced == 3 && st.scored_docs == 2 && st.candidates_examined == 2);
  CHECK(execute_top_k(ix, {{0, 1.0f}}, 0, ExecutionStrategy::Wand).empty());
  CHECK(execute_top_k(ix, {}, 5, ExecutionStrategy::Wand).empty());
}

static void errors_do_not_unwind_across_the_abi() {
  Index ix({two_term_segment()});
  bool threw = false;
  try {
    execute_top_k(ix, {{99, 1.0f}}, 2, ExecutionStrategy::Wand);  // no such term
  } catch (const Error &e) {
    threw = e.code == SLG_ERR_INVALID;
  }
  CHECK(threw);
}

static void accept_predicate_filters_hits() {  // wand.rs:512/:555/:858 accept(doc, score)
  Index ix({two_term_segment()});
  auto only3 = execute_top_k_with_accept(ix, {{0, 1.0f}, {1, 1.0f}}, 2, ExecutionStrategy::Wand,
                                         [](uint32_t d) { return d != 1; });
  CHECK(only3.size() == 1 && only3[0].doc_id == 3 && std::fabs(only3[0].score - 6.6957893f) < 1e-5f);
  auto none = execute_top_k_with_accept(ix, {{0, 1.0f}, {1, 1.0f}}, 2, ExecutionStrategy::Bm25,
                                        [](uint32_t) { return false; });
  CHECK(none.empty());
  auto all = execute_top_k_with_accept(ix, {{0, 1.0f}, {1, 1.0f}}, 2, ExecutionStrategy::Bm25,
                                       [](uint32_t) { return true; });
  CHECK(all.size() == 2 && all[0].doc_id == 3 && all[1].doc_id == 1);
}

int main() {
  ranked_doc_ordering_prefers_smaller_id_on_tie();
  brute_force_matches_wand_results();
  stats_and_k_zero();
  errors_do_not_unwind_across_the_abi();
  accept_predicate_filters_hits();
  std::puts("wand_tests: all passed");
  return 0;
}
