// slg_plan.cpp — host planner of a query batch (see slg_plan.hpp).  Host-only C++17.
#include "slg_plan.hpp"

#include <algorithm>
#include <cstdio>
#include <atomic>
#include <cmath>
#include <cstring>
#include <exception>
#include <thread>

namespace slgplan {

namespace {

#define PLAN_REQUIRE(cond, msg)                          \
  do {                                                   \
    if (!(cond)) throw SlgError(SLG_ERR_INVALID, (msg)); \
  } while (0)

inline uint32_t full_mask(uint32_t n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }
static_assert(slg::kMaxPlanDepth == SLG_MAX_PLAN_DEPTH, "the kernels' level arrays and the ABI's depth limit");
constexpr uint32_t kMaxPlanNodes = 255;  // leaves / groups of a two-level plan (8-bit fields of TermRef::gmeta)

// ---- validation of the caller's arrays (cheap, before anything indexes through them) ------------
struct BatchFacts {
  uint32_t total_terms = 0;
  uint32_t max_nt = 0;           // most terms of any query
  bool plans_requested = false;  // some query can need a score plan (leaf close on the device)
  bool min_match = false;        // some query has minimum_should_match > 1 (slg_score_plans::q_min_match)
  bool nested_requested = false; // some query names groups of leaves
};

BatchFacts validate_batch(const BatchIn &in, uint32_t n_segs) {
  BatchFacts f;
  PLAN_REQUIRE(in.nq == 0 || in.q_offsets != nullptr, "q_offsets is NULL");
  PLAN_REQUIRE(in.strategy == SLG_STRATEGY_BM25 || in.strategy == SLG_STRATEGY_WAND ||
                   in.strategy == SLG_STRATEGY_BMW,
               "unknown strategy");
  if (in.k > SLG_MAX_K) throw SlgError(SLG_ERR_UNSUPPORTED, "k > SLG_MAX_K (" + std::to_string(SLG_MAX_K) + ")");
  (void)n_segs;
  f.total_terms = in.nq ? in.q_offsets[in.nq] : 0;
  PLAN_REQUIRE(f.total_terms == 0 || (in.q_term_ids && in.q_weights), "q_term_ids/q_weights is NULL");
  for (uint32_t q = 0; q < in.nq; q++) {
    PLAN_REQUIRE(in.q_offsets[q + 1] >= in.q_offsets[q] && in.q_offsets[q + 1] <= f.total_terms,
                 "q_offsets not monotone");
    const uint32_t nt = in.q_offsets[q + 1] - in.q_offsets[q];
    if (nt > SLG_MAX_QUERY_TERMS)
      throw SlgError(SLG_ERR_UNSUPPORTED, "query " + std::to_string(q) + " has more than " +
                                              std::to_string(SLG_MAX_QUERY_TERMS) + " terms");
    f.max_nt = std::max(f.max_nt, nt);
  }
  const slg_score_plans &pl = in.plans;
  if (pl.q_min_match)
    for (uint32_t q = 0; q < in.nq; q++)
      if (pl.q_min_match[q] > 1u) {
        if (pl.q_min_match[q] > 255u) throw SlgError(SLG_ERR_UNSUPPORTED, "minimum_should_match > 255 in query " + std::to_string(q));
        f.min_match = true;
        f.plans_requested = true;  // (counted per leaf in the plan kernel's join)
      }
  const bool trees = pl.q_node_offsets != nullptr;
  PLAN_REQUIRE(!trees || (pl.node_kind && pl.node_tie && pl.node_parent), "score trees need node_kind, node_tie and node_parent");
  if (trees) {
    for (uint32_t q = 0; q < in.nq; q++) {
      PLAN_REQUIRE(pl.q_node_offsets[q + 1] >= pl.q_node_offsets[q], "q_node_offsets not monotone");
      const uint32_t n0 = pl.q_node_offsets[q], nn = pl.q_node_offsets[q + 1] - n0;
      PLAN_REQUIRE(nn >= 1, "a score tree has no node in query " + std::to_string(q));
      if (nn > 2 * kMaxPlanNodes)
        throw SlgError(SLG_ERR_UNSUPPORTED, "score tree of query " + std::to_string(q) + " has too many nodes");
      uint32_t depth_of[2 * kMaxPlanNodes + 1], kids[2 * kMaxPlanNodes + 1];
      uint32_t n_leaf = 0, deepest = 0;
      for (uint32_t i = 0; i < nn; i++) {
        const int kd = pl.node_kind[n0 + i];
        PLAN_REQUIRE(kd == SLG_PLAN_SUM || kd == SLG_PLAN_DISMAX || kd == SLG_PLAN_LEAF, "unknown node kind in query " + std::to_string(q));
        kids[i] = 0;
        if (i == 0) {
          depth_of[0] = 0;
        } else {
          const uint32_t pa = pl.node_parent[n0 + i];
          PLAN_REQUIRE(pa < i, "node_parent must name an earlier node (pre-order) in query " + std::to_string(q));
          PLAN_REQUIRE(pl.node_kind[n0 + pa] != SLG_PLAN_LEAF, "a leaf node has a child in query " + std::to_string(q));
          // pre-order: the parent is the last node before i whose depth is smaller
          depth_of[i] = depth_of[pa] + 1;
          PLAN_REQUIRE(pa == i - 1 || depth_of[i - 1] >= depth_of[i], "nodes are not in pre-order in query " + std::to_string(q));
          kids[pa]++;
        }
        if (kd == SLG_PLAN_LEAF) {
          n_leaf++;
          deepest = std::max(deepest, depth_of[i]);
        } else if (kd == SLG_PLAN_DISMAX) {
          const float t = pl.node_tie[n0 + i];
          PLAN_REQUIRE(t >= 0.0f && t <= 1.0f, "tie breaker outside [0, 1] in query " + std::to_string(q));
        }
      }
      for (uint32_t i = 0; i < nn; i++)
        if (pl.node_kind[n0 + i] != SLG_PLAN_LEAF && kids[i] == 0)
          throw SlgError(SLG_ERR_UNSUPPORTED, "a Sum / DisMax node without children in query " + std::to_string(q));
      if (n_leaf > kMaxPlanNodes) throw SlgError(SLG_ERR_UNSUPPORTED, "score tree of query " + std::to_string(q) + " has too many leaves");
      if (deepest > SLG_MAX_PLAN_DEPTH)
        throw SlgError(SLG_ERR_UNSUPPORTED, "score tree of query " + std::to_string(q) + " is deeper than SLG_MAX_PLAN_DEPTH");
      const uint32_t t0 = in.q_offsets[q], nt = in.q_offsets[q + 1] - t0;
      for (uint32_t i = 0; i < nt; i++)
        PLAN_REQUIRE((pl.q_leaf ? pl.q_leaf[t0 + i] : i) < n_leaf, "a term names a leaf the tree does not have in query " + std::to_string(q));
      f.plans_requested = true;  // (resolved per query: a one-level tree of single-term leaves still is the flat sum)
      if (deepest >= 2) f.nested_requested = true;
    }
    return f;
  }
  const bool groups = pl.leaf_group != nullptr;
  PLAN_REQUIRE(!groups || (pl.q_leaf_offsets && pl.q_group_offsets && pl.group_plan && pl.group_tie && pl.q_nleaves),
               "two-level plans need q_nleaves, q_leaf_offsets, q_group_offsets, group_plan and group_tie");
  for (uint32_t q = 0; q < in.nq; q++) {
    if (pl.q_plan && pl.q_plan[q] == SLG_PLAN_DISMAX) f.plans_requested = true;
    const uint32_t t0 = in.q_offsets[q], nt = in.q_offsets[q + 1] - t0;
    if (pl.q_leaf && !f.plans_requested)
      for (uint32_t i = 0; i < nt && !f.plans_requested; i++)
        for (uint32_t j = 0; j < i; j++)
          if (pl.q_leaf[t0 + i] == pl.q_leaf[t0 + j]) {
            f.plans_requested = true;
            break;
          }
    if (groups) {
      PLAN_REQUIRE(pl.q_leaf_offsets[q + 1] >= pl.q_leaf_offsets[q] && pl.q_group_offsets[q + 1] >= pl.q_group_offsets[q],
                   "q_leaf_offsets / q_group_offsets not monotone");
      const uint32_t nl = pl.q_leaf_offsets[q + 1] - pl.q_leaf_offsets[q];
      const uint32_t ng = pl.q_group_offsets[q + 1] - pl.q_group_offsets[q];
      PLAN_REQUIRE(nl == pl.q_nleaves[q], "q_leaf_offsets disagrees with q_nleaves in query " + std::to_string(q));
      // (the descriptors keep a group's id and its leaf count in 8 bits each; leaves without a
      //  term are legal — a DisMax counts them as 0.0 — so the limits are not the term limit)
      if (nl > kMaxPlanNodes || ng > kMaxPlanNodes)
        throw SlgError(SLG_ERR_UNSUPPORTED, "query " + std::to_string(q) + " has more than " +
                                                std::to_string(kMaxPlanNodes) + " leaves or groups");
      uint32_t prev = 0;
      bool is_flat = ng == nl;  // every leaf its own Sum group == the flat plan
      for (uint32_t l = 0; l < nl; l++) {
        const uint32_t g = pl.leaf_group[pl.q_leaf_offsets[q] + l];
        PLAN_REQUIRE(g < ng, "leaf_group out of range in query " + std::to_string(q));
        // leaves are numbered in the plan's traversal order, so a group's leaves are consecutive
        PLAN_REQUIRE(l == 0 || g == prev || g == prev + 1, "leaf_group must be non-decreasing without gaps");
        PLAN_REQUIRE(l != 0 || g == 0, "leaf_group must start at group 0");
        prev = g;
        if (g != l) is_flat = false;
      }
      PLAN_REQUIRE(nl == 0 || prev + 1 == ng, "a group has no leaf in query " + std::to_string(q));
      for (uint32_t g = 0; g < ng; g++) {
        const int gk = pl.group_plan[pl.q_group_offsets[q] + g];
        const float gt = pl.group_tie[pl.q_group_offsets[q] + g];
        PLAN_REQUIRE(gk == SLG_PLAN_SUM || gk == SLG_PLAN_DISMAX, "unknown group plan in query " + std::to_string(q));
        PLAN_REQUIRE(gt >= 0.0f && gt <= 1.0f, "tie breaker outside [0, 1] in query " + std::to_string(q));
        if (gk == SLG_PLAN_DISMAX) is_flat = false;
      }
      if (!is_flat && nl) {
        f.nested_requested = true;
        f.plans_requested = true;
      }
    }
  }
  return f;
}

// ---- pass 1: sub-queries (query x segment) and their terms ---------------------------------------
struct Pass1Out {
  std::vector<slg::RoundQuery> sqs;
  std::vector<slg::TermRef> terms;
  std::vector<uint64_t> sq_postings, sq_postings_all;
  std::vector<uint32_t> sq_longest_all;  // longest list of the sub-query, classification aside
  double skip_est = 0.0;                 // postings block skipping is expected to leave unread
  uint64_t n_postings = 0, n_ess = 0, n_noness = 0;
  uint32_t max_terms = 0;
  bool any_plan = false, any_filter = false, any_nested = false, any_deep = false;
  std::vector<slg::PlanNode> nodes;  // canonical node tables of this part's deep trees
  std::exception_ptr err;
};

struct Pass1Ctx {
  const std::vector<SegView> &segs;
  const slg_tuning &tn;
  const BatchIn &in;
  const BatchFacts &facts;
  bool maxscore_on;
  std::vector<uint32_t> &q_sq_begin;
  std::vector<uint64_t> &q_postings;
};

// threshold seed theta0 = max_t w_t * champ[t][rank(k)]: an exact lower bound of the k-th best score
// whenever no weight is negative (a doc's total is then >= each of its contributions: Sum, or
// DisMax with tie in [0, 1], at either level of the plan) and no doc filter can reject the champions
float threshold_seed(const SegView &sh, const slg::TermRef *t, uint32_t n, uint32_t k, uint32_t fq) {
  if (!sh.champ || k > 1024u || fq != 0) return 0.0f;
  float seed = 0.0f;
  for (uint32_t i = 0; i < n; i++) {
    if (!(t[i].weight >= 0.0f)) return 0.0f;
    if (t[i].weight > 0.0f)
      seed = std::max(seed, t[i].weight * sh.champ[(size_t)t[i].term * slg::kChampions + slg::champ_index(k)]);
  }
  return seed;
}

// MaxScore classification (strategies Wand / Bmw; exact): lists taken in ascending order of their
// maximum contribution ub_t = w_t * champ[t][0] are non-essential while the running sum of ub stays
// below theta0: a doc found only in them totals < theta0 and cannot reach the top-k
uint32_t classify_essential(const SegView &sh, const slg::TermRef *t, uint32_t n, float theta0) {
  uint32_t ess_mask = full_mask(n);
  std::vector<std::pair<float, uint32_t>> ub(n);
  for (uint32_t i = 0; i < n; i++) ub[i] = {t[i].weight * sh.champ[(size_t)t[i].term * slg::kChampions], i};
  std::sort(ub.begin(), ub.end());
  double acc = 0.0;
  for (uint32_t i = 0; i + 1 < n; i++) {  // at least one list stays essential
    acc += (double)ub[i].first;
    // margin: f32 sums of the real contributions may round up by a few ulps
    if (acc * (1.0 + 1e-5) < (double)theta0)
      ess_mask &= ~(1u << ub[i].second);
    else
      break;
  }
  return ess_mask;
}

void plan_queries(const Pass1Ctx &c, const uint32_t q_lo, const uint32_t q_hi, Pass1Out &o) {
  const BatchIn &in = c.in;
  const slg_score_plans &pl = in.plans;
  const uint32_t n_segs = (uint32_t)c.segs.size();
  const uint32_t k = in.k;
  auto &sqs = o.sqs;
  auto &terms = o.terms;
  sqs.reserve((size_t)(q_hi - q_lo) * n_segs);
  terms.reserve((size_t)(in.q_offsets[q_hi] - in.q_offsets[q_lo]) * n_segs);
  o.sq_postings.reserve(sqs.capacity());
  o.sq_postings_all.reserve(sqs.capacity());
  o.sq_longest_all.reserve(sqs.capacity());
  // the champion table is tens of MB (config 2: 71 MB) and a query touches one line of it per term
  // (cold for queries that were not planned just before): the lines of the query 8 ahead are
  // requested while this one is planned
  auto prefetch_query = [&](const uint32_t q) {
    const uint32_t t0 = in.q_offsets[q], nt = in.q_offsets[q + 1] - t0;
    for (uint32_t i = 0; i < nt; i++)
      for (uint32_t s = 0; s < n_segs; s++) {
        const uint32_t tid = in.q_term_ids[(size_t)(t0 + i) * n_segs + s];
        const SegView &sh = c.segs[s];
        if (tid == SLG_NO_TERM || tid >= sh.n_terms) continue;
        __builtin_prefetch(&sh.term_offsets[tid]);
        if (sh.champ && k <= 1024u) {
          __builtin_prefetch(&sh.champ[(size_t)tid * slg::kChampions + slg::champ_index(k)]);
          if (c.maxscore_on) __builtin_prefetch(&sh.champ[(size_t)tid * slg::kChampions]);
        }
      }
  };
  for (uint32_t q = q_lo; q < std::min(q_hi, q_lo + 8u); q++) prefetch_query(q);
  for (uint32_t q = q_lo; q < q_hi; q++) {
    if (q + 8u < q_hi) prefetch_query(q + 8u);
    c.q_sq_begin[q] = (uint32_t)sqs.size();
    const uint32_t t0 = in.q_offsets[q], nt = in.q_offsets[q + 1] - t0;
    uint32_t fq = 0;  // doc filter of the query (0 none, id + 1)
    if (in.q_filter && in.q_filter[q] >= 0) {
      PLAN_REQUIRE((size_t)in.q_filter[q] < in.n_filters && in.filter_live[in.q_filter[q]],
                   "unknown filter id in query " + std::to_string(q));
      fq = (uint32_t)in.q_filter[q] + 1u;
      o.any_filter = true;
    }
    // score plan of the query (query/planner.rs:113-153): root over leaves, root over groups of leaves,
    // or (slg_score_plans::q_node_offsets) a tree given node by node, which is resolved here into one
    // of the first two when it has one or two levels, and into a canonical node table otherwise
    int plan_kind = pl.q_plan ? pl.q_plan[q] : SLG_PLAN_SUM;
    float tie = pl.q_tie ? pl.q_tie[q] : 0.0f;
    const uint32_t min_match = pl.q_min_match ? pl.q_min_match[q] : 0u;
    uint32_t n_leaves = pl.q_nleaves ? pl.q_nleaves[q] : 0;
    bool groups = pl.leaf_group != nullptr && pl.q_node_offsets == nullptr;
    const uint32_t *lgroup = groups ? pl.leaf_group + pl.q_leaf_offsets[q] : nullptr;
    uint32_t n_groups = groups ? pl.q_group_offsets[q + 1] - pl.q_group_offsets[q] : 0u;
    const int32_t *gplan = groups ? pl.group_plan + pl.q_group_offsets[q] : nullptr;
    const float *gtie = groups ? pl.group_tie + pl.q_group_offsets[q] : nullptr;
    uint32_t tree_lgroup[kMaxPlanNodes + 1];
    int32_t tree_gplan[kMaxPlanNodes + 1];
    float tree_gtie[kMaxPlanNodes + 1];
    uint32_t depth = 0, node_begin = 0;        // deep tree: levels of internal nodes, first canonical node
    uint32_t leaf_node[kMaxPlanNodes + 1];     // deep tree: canonical node every leaf hangs off
    if (pl.q_node_offsets) {
      const uint32_t n0 = pl.q_node_offsets[q], nn = pl.q_node_offsets[q + 1] - n0;
      const int32_t *kind = pl.node_kind + n0;
      const float *ntie = pl.node_tie + n0;
      const uint32_t *par = pl.node_parent + n0;
      uint32_t dep[2 * kMaxPlanNodes + 1];
      uint32_t deepest = 0;
      n_leaves = 0;
      for (uint32_t i = 0; i < nn; i++) {
        dep[i] = i == 0 ? 0u : dep[par[i]] + 1u;
        if (kind[i] == SLG_PLAN_LEAF) {
          n_leaves++;
          deepest = std::max(deepest, dep[i]);
        }
      }
      if (kind[0] == SLG_PLAN_LEAF) {  // the plan is one leaf: Sum of one leaf
        plan_kind = SLG_PLAN_SUM;
        tie = 0.0f;
      } else if (deepest <= 2) {  // root over leaves, or root over groups: the forms above
        plan_kind = kind[0];
        tie = kind[0] == SLG_PLAN_DISMAX ? ntie[0] : 0.0f;
        if (deepest == 2) {
          groups = true;
          n_groups = 0;
          uint32_t lf = 0;
          for (uint32_t i = 1; i < nn; i++) {
            if (dep[i] == 1) {  // a child of the root: a group (a bare leaf = a Sum group of one leaf)
              tree_gplan[n_groups] = kind[i] == SLG_PLAN_DISMAX ? SLG_PLAN_DISMAX : SLG_PLAN_SUM;
              tree_gtie[n_groups] = kind[i] == SLG_PLAN_DISMAX ? ntie[i] : 0.0f;
              n_groups++;
            }
            if (kind[i] == SLG_PLAN_LEAF) tree_lgroup[lf++] = n_groups - 1u;
          }
          lgroup = tree_lgroup;
          gplan = tree_gplan;
          gtie = tree_gtie;
        }
      } else {
        // deep tree: canonical node table — the internal nodes in pre-order, and under every leaf that
        // hangs above the deepest level a chain of one-child Sum nodes down to it
        depth = deepest;
        plan_kind = kind[0];
        tie = kind[0] == SLG_PLAN_DISMAX ? ntie[0] : 0.0f;
        node_begin = (uint32_t)o.nodes.size();
        uint32_t canon[2 * kMaxPlanNodes + 1];  // original internal node -> canonical index
        uint32_t lf = 0;
        for (uint32_t i = 0; i < nn; i++) {
          if (kind[i] != SLG_PLAN_LEAF) {
            canon[i] = (uint32_t)o.nodes.size() - node_begin;
            slg::PlanNode pn{};
            pn.parent = i == 0 ? 0u : canon[par[i]];
            pn.n_children = 0;
            pn.kind = kind[i] == SLG_PLAN_DISMAX ? 1u : 0u;
            pn.tie = kind[i] == SLG_PLAN_DISMAX ? ntie[i] : 0.0f;
            o.nodes.push_back(pn);
            if (i != 0) o.nodes[node_begin + canon[par[i]]].n_children++;
          } else {
            uint32_t above = canon[par[i]];
            o.nodes[node_begin + above].n_children++;
            for (uint32_t d = dep[i]; d < deepest; d++) {  // pad: Sum of one child
              slg::PlanNode pn{};
              pn.parent = above;
              pn.n_children = 1;
              pn.kind = 0u;
              pn.tie = 0.0f;
              above = (uint32_t)o.nodes.size() - node_begin;
              o.nodes.push_back(pn);
            }
            leaf_node[lf++] = above;
          }
        }
      }
    } else {
      PLAN_REQUIRE(plan_kind == SLG_PLAN_SUM || plan_kind == SLG_PLAN_DISMAX,
                   "unknown score plan in query " + std::to_string(q));
      // validate_tie_breaker (query/planner.rs:850-856); the threshold seed and the pruning bounds
      // also rely on it: with tie in [0, 1] a DisMax is >= each of its non-negative leaves
      PLAN_REQUIRE(tie >= 0.0f && tie <= 1.0f, "tie breaker outside [0, 1] in query " + std::to_string(q));
    }
    for (uint32_t i = 0; i < nt; i++) {
      const uint32_t lf = pl.q_leaf ? pl.q_leaf[t0 + i] : i;
      PLAN_REQUIRE(lf < 0x80000000u, "leaf index >= 2^31 in query " + std::to_string(q));
      n_leaves = std::max(n_leaves, lf + 1u);
    }
    // two-level plan: group of every leaf, leaves per group
    uint32_t leaves_in_group[kMaxPlanNodes + 1];  // (zeroed only for queries with groups: 1 KB per query otherwise)
    bool nested = false;
    if (groups) {
      std::memset(leaves_in_group, 0, sizeof(leaves_in_group));
      if (!pl.q_node_offsets)
        PLAN_REQUIRE(n_leaves == pl.q_nleaves[q], "a term names a leaf beyond q_nleaves in query " + std::to_string(q));
      for (uint32_t l = 0; l < n_leaves; l++) leaves_in_group[lgroup[l]]++;
      for (uint32_t g = 0; g < n_groups; g++)
        if (leaves_in_group[g] != 1 || gplan[g] == SLG_PLAN_DISMAX) nested = true;
    }
    if (k == 0) continue;  // wand.rs:413-416: k == 0 and no collector => no work
    for (uint32_t s = 0; s < n_segs; s++) {
      const SegView &sh = c.segs[s];
      slg::RoundQuery sq{};
      sq.q = q;
      sq.seg = s;
      sq.filter = fq;
      sq.term_begin = (uint32_t)terms.size();
      for (uint32_t i = 0; i < nt; i++) {
        const uint32_t tid = in.q_term_ids[(size_t)(t0 + i) * n_segs + s];
        if (tid == SLG_NO_TERM) continue;
        PLAN_REQUIRE(tid < sh.n_terms, "term id out of range in query " + std::to_string(q));
        const uint32_t df = (uint32_t)(sh.term_offsets[tid + 1] - sh.term_offsets[tid]);
        if (df == 0) continue;  // wand.rs:441 filter(postings.len() > 0)
        const uint64_t off = sh.term_offsets[tid] + (uint64_t)slg::kListPad * tid;  // padded layout (SegDev)
        const float w = in.q_weights[t0 + i];
        PLAN_REQUIRE(std::isfinite(w), "non-finite weight in query " + std::to_string(q));
        slg::TermRef tr{};
        tr.off = off;
        tr.df = df;
        tr.weight = w;
        tr.term = tid;
        tr.leaf = pl.q_leaf ? pl.q_leaf[t0 + i] : i;
        if (depth) {
          tr.gmeta = leaf_node[tr.leaf];  // the canonical node the leaf hangs off
        } else if (nested) {
          const uint32_t g = lgroup[tr.leaf];
          tr.gmeta = g | (leaves_in_group[g] << 8) | ((gplan[g] == SLG_PLAN_DISMAX ? 1u : 0u) << 16);
          tr.gtie = gtie[g];
        }
        terms.push_back(tr);
      }
      sq.n_terms = (uint32_t)terms.size() - sq.term_begin;
      if (sq.n_terms == 0) continue;
      slg::TermRef *first = terms.data() + sq.term_begin;
      {
        // Lists go to the device sorted by leaf (stable: a leaf's terms keep the term order in
        // which the reference adds them, wand.rs:488-497; a group's leaves are consecutive, so the
        // lists are sorted by group too).  plan 0 = the flat term-order sum, which is what Sum
        // gives when no leaf holds two terms.
        if (pl.q_leaf)  // (without leaves given, leaf = term position: already in order)
          std::stable_sort(first, first + sq.n_terms,
                           [](const slg::TermRef &a, const slg::TermRef &b) { return a.leaf < b.leaf; });
        bool shared = false;
        uint32_t present = 0;
        for (uint32_t i = 0; i < sq.n_terms; i++) {
          const bool fresh = i == 0 || first[i].leaf != first[i - 1].leaf;
          present += fresh ? 1u : 0u;
          shared = shared || !fresh;
        }
        sq.plan = plan_kind == SLG_PLAN_DISMAX ? 2u : ((shared || nested || depth) ? 1u : 0u);
        if (min_match > 1u) {  // leaves are counted in the plan kernel's leaf close: Sum of leaves, bits 8.. = the count asked for
          if (sq.plan == 0u) sq.plan = 1u;
          sq.plan |= min_match << 8;
        }
        sq.tie = tie;
        sq.max_init = present < n_leaves ? 0.0f : -INFINITY;
        sq.n_leaves = n_leaves;
        sq.n_groups = nested ? n_groups : 0u;
        sq.depth = depth;
        sq.node_begin = node_begin;
        if (sq.plan) o.any_plan = true;
        if (nested) o.any_nested = true;
        if (depth) o.any_deep = true;
      }
      // (minimum_should_match > 1: the champions behind the seed are single postings — docs the matcher may reject)
      sq.theta0 = min_match > 1u ? 0.0f : threshold_seed(sh, first, sq.n_terms, k, fq);
      // MaxScore: by default for batches with a query of >= 5 terms (plan_batch drops the
      // classification again when block skipping has nothing to gain);
      // slg_tuning.pruning = 1 / 0 forces it on / off.  Never with score plans in the batch (the
      // plan kernels have no classified path).
      uint32_t ess_mask = full_mask(sq.n_terms);
      if (in.strategy != SLG_STRATEGY_BM25 && sq.theta0 > 0.0f && !c.facts.plans_requested && sq.n_terms > 1 &&
          c.maxscore_on)
        ess_mask = classify_essential(sh, first, sq.n_terms, sq.theta0);
      sq.ess_mask = ess_mask;
      // the round planner works on the essential lists only
      uint64_t P = 0, P_all = 0;
      uint32_t longest = 0, longest_df = 0, longest_all = 0, longest_all_df = 0;
      for (uint32_t i = 0; i < sq.n_terms; i++) {
        const uint32_t df = first[i].df;
        P_all += df;
        if (df > longest_all_df) {
          longest_all_df = df;
          longest_all = i;
        }
        if (!((ess_mask >> i) & 1u)) continue;
        P += df;
        if (df > longest_df) {
          longest_df = df;
          longest = i;
        }
      }
      // block skipping pays where a 64-posting block of a non-essential list usually holds no
      // candidate doc: a block spans 64 * N / df docs, which hold 64 * P / df essential postings
      // on average; blocks are tested only below 2 (>= e^-2 = 13 % of them can be skipped).
      // Config 3's lists are all of similar density (>= 64 per block): no test, no cost.
      sq.skip_mask = 0;
      if (c.tn.block_max)
        for (uint32_t i = 0; i < sq.n_terms && i < 32; i++) {
          const uint64_t df = first[i].df;
          if (!((ess_mask >> i) & 1u) && 64ull * P < 2ull * df) {
            sq.skip_mask |= 1u << i;
            o.skip_est += (double)df * std::exp(-64.0 * (double)P / (double)df);  // blocks without a candidate doc
          }
        }
      o.n_noness += P_all - P;
      c.q_postings[q] += P_all;
      o.n_postings += P_all;
      o.n_ess += P;
      sq.longest = longest;
      o.max_terms = std::max(o.max_terms, sq.n_terms);
      sqs.push_back(sq);
      o.sq_postings.push_back(P);
      o.sq_postings_all.push_back(P_all);
      o.sq_longest_all.push_back(longest_all);
    }
  }
}

// ---- pass 2: rounds of about one register set of postings, slices of consecutive rounds ----------
// postings per round of a few-term sub-query.  Slot forms of the kernel (slg_score_uni.hpp, _uni3):
// every list is padded to a 64-lane slot (half a slot wasted per list on average).  A round that needs more than 8 slots is streamed in chunks at 2-3x
// the cost, so the target follows the sub-query's own mix of list lengths: the largest R (steps of
// 16) whose expected slots stay under 8 with 1.6 sigma to spare.  The longest list is cut at exact
// strides (its count is R * f); every other list's count c is roughly Poisson around R * f:
// ceil(c / 64) has mean c/64 + 1/2 and variance c/4096 + 1/12.
uint32_t uniform_round_target(const slg::RoundQuery &sq, const slg::TermRef *t, uint64_t P, const slg_tuning &tn) {
  const uint32_t n = sq.n_terms;
  const double Pd = (double)P;
  // sigmas to spare: blocked form 1.0 / 1.6 / 2.0 / 2.5 / 3.0 / 3.5 / 4.0 / 5.0 -> 0.0874 / 0.0791 / 0.0762 /
  // 0.0744 / 0.0728 / 0.0727 / 0.0734 / 0.0752 ms on config 2 (an over-full round costs 2-3 rounds)
  const double sigmas = tn.uniform_sigma_x100 ? tn.uniform_sigma_x100 / 100.0 : (tn.uniform_kernel >= 4 ? 3.2 : 1.6);
  uint32_t dflt;
  if (tn.uniform_kernel >= 4) {
    // blocked layout (slg_score_uni4.hpp): a list is padded to whole lanes of 8 postings, a round
    // has 64 lanes.  ceil(c / 8) has mean c/8 + 7/16 and variance c/64 + 1/12 (c roughly Poisson)
    dflt = (uint32_t)slg::kUniCap;
    if (!tn.uniform_round_target && n > 1) {
      // lanes a round of R postings is expected to need, plus `sigmas` standard deviations
      auto lanes_needed = [&](const uint32_t R) {
        double mu = 0.0, var = 0.0;
        for (uint32_t j = 0; j < n; j++) {
          const double c = (double)R * (double)t[j].df / Pd;
          if (j == sq.longest) {
            mu += std::ceil(c / 8.0);
          } else {
            mu += c / 8.0 + 7.0 / 16.0;
            var += c / 64.0 + 1.0 / 12.0;
          }
        }
        return mu + sigmas * std::sqrt(var);
      };
      // the largest R (steps of 8) that stays under 64.3 lanes.  With f = the longest list's share of
      // the postings: mu ~ R/8 + b, var = c R + d (b = 7/16 (n-1) + 1/2, c = (1-f)/64, d = (n-1)/12);
      // R/8 + b + s sqrt(c R + d) = 64.3 is a quadratic in y = sqrt(c R + d); the root is then
      // corrected against the exact count (the ceil) in steps of 8 — usually two evaluations (a scan
      // over all 53 candidates cost 70 ms of planning on config 4's 65 536 sub-queries)
      const double f = (double)t[sq.longest].df / Pd;
      const double b = 7.0 / 16.0 * (n - 1) + 0.5, c = (1.0 - f) / 64.0, d = (n - 1) / 12.0;
      double x = (64.3 - b) * 8.0;
      if (c > 1e-9) {
        const double qa = 1.0 / (8.0 * c), qc = b - 64.3 - d / (8.0 * c);
        const double y = (-sigmas + std::sqrt(sigmas * sigmas - 4.0 * qa * qc)) / (2.0 * qa);
        x = (y * y - d) / c;
      }
      uint32_t R = (uint32_t)std::min(std::max(x, 64.0), (double)slg::kUniCap) & ~7u;
      while (R > 64 && lanes_needed(R) > 64.3) R -= 8;
      while (R + 8 <= (uint32_t)slg::kUniCap && lanes_needed(R + 8) <= 64.3) R += 8;
      dflt = R;
    }
  } else {
    dflt = 64u * (slg::kUniSlots > (int)n ? slg::kUniSlots - n : 0u) + 64u;
    if (!tn.uniform_round_target && n > 1) {
      uint32_t best = 64;
      for (uint32_t R = 96; R <= (uint32_t)slg::kUniCap; R += 16) {
        double mu = 0.0, var = 0.0;
        for (uint32_t j = 0; j < n; j++) {
          const double c = (double)R * (double)t[j].df / Pd;
          if (j == sq.longest) {
            mu += std::ceil(c / 64.0);
          } else {
            mu += c / 64.0 + 0.5;
            var += c / 4096.0 + 1.0 / 12.0;
          }
        }
        // (round-2 kernel, 1.0 / 1.3 / 1.6 / 2.0 / 2.5 sigma: 0.1024 / 0.1011 / 0.1006 / 0.1019 / 0.1039 ms on
        //  config 2; fixed 384: 0.1042)
        if (mu + sigmas * std::sqrt(var) > 8.3) break;
        best = R;
      }
      dflt = best;
    }
  }
  return std::max<uint32_t>(48, std::min<uint32_t>(tn.uniform_round_target ? tn.uniform_round_target : dflt,
                                                   (uint32_t)slg::kUniCap));
}

// the many-term kernel's bitmap covers a window of kSpan docs: a round whose essential postings are
// spread over more is cut into chunks, each paying the round's fixed costs.  Sparse sub-queries get
// rounds that fit the window (postings per round <= 0.85 * kSpan * density of the essential lists)
// Threads a large batch's planning may use: the CPUs this process may run on (the cgroup's quota where there
// is one: a container shows every CPU of its host — the GPU boxes 256 for a quota of 16), divided among
// the plan_batch calls running at this moment (config 4's harness has four caller threads planning at
// once: 4 x 8 planner threads on 16 CPUs made one box plan a batch in 42 ms instead of 19), at most 8.
std::atomic<int> g_plans_running{0};
uint32_t cpu_budget() {
  static const uint32_t quota = [] {
    uint32_t n = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char a[32] = {0};
      unsigned long long period = 0;
      if (std::fscanf(f, "%31s %llu", a, &period) == 2 && period > 0 && std::strcmp(a, "max") != 0) {
        const unsigned long long q = std::strtoull(a, nullptr, 10);
        if (q > 0) n = std::min<uint32_t>(n, (uint32_t)std::max<unsigned long long>(1, (q + period - 1) / period));
      }
      std::fclose(f);
    }
    return n;
  }();
  return quota;
}
uint32_t planner_threads() {
  const int running = std::max(1, g_plans_running.load(std::memory_order_relaxed));
  return std::max<uint32_t>(1, std::min<uint32_t>(8, cpu_budget() / (uint32_t)running));
}

uint32_t multi_round_target(uint64_t P, uint32_t n_docs, const slg_tuning &tn) {
  uint32_t target = std::max<uint32_t>(64, std::min<uint32_t>(tn.multi_round_target, (uint32_t)slg::kMultiCap));
  const double dens = (double)P / (double)std::max<uint32_t>(1u, n_docs);
  const double fit = 0.85 * (double)slg::kSpan * dens;  // (0.65 / 0.75 / 0.85 / 0.95 / 1.0 measured on config 3:
                                                         //  7.88 / 7.54 / 7.35 / 7.76 / 8.20 ms; no rule: 8.69)
  if (fit < (double)target) target = (uint32_t)std::max(64.0, fit);
  return target;
}

void plan_rounds(const std::vector<SegView> &segs, const slg_tuning &tn, uint32_t k,
                 const std::vector<uint64_t> &sq_postings, const std::vector<uint64_t> &sq_postings_all, Plan &out) {
  auto &sqs = out.sqs;
  const uint32_t probe_target = std::max<uint32_t>((uint32_t)slg::kMultiCap, tn.probe_target);
  // rounds per slice: short slices pack the tail of the launch better (one wave per slice).  The
  // few-term kernel's slices are cheap to start (threshold seed + buffered top-k) as long as k is
  // small: every slice writes k candidates for the merge (measured: config 2 k=11 best at 4,
  // config 3 k=101 best at 8).
  const bool rps_pinned = tn.rounds_per_slice != 0;
  const uint32_t max_rps = std::max<uint32_t>(
      1, std::min<uint32_t>(rps_pinned ? tn.rounds_per_slice
                                       : (out.uniform && k <= 64 ? (uint32_t)slg::kUniRoundsPerSlice
                                                                 : (uint32_t)slg::kDefaultRoundsPerSlice),
                            (uint32_t)slg::kMaxRoundsPerSlice));
  // longest slices: 8 rounds on the few-term kernel (measured on config 2: the heaviest
  // sub-queries' 16-round slices were the tail of the launch), 16 on the many-term kernel and on the
  // 5..8-list form for large k (config 3, k = 101: 5.03 ms at 16, 5.11 at 8), 6 on that form for
  // k <= 64 (multi-field workload, k = 11: the 15-round slices of the dense sub-queries ran 125 us
  // of a 144-us launch; 0.165 ms at 15, 0.124 at 6, 0.135 at 4)
  const bool blocked8 = out.uniform && tn.uniform_kernel >= 4 && out.max_terms > (uint32_t)slg::kUniMaxLists;
  const uint32_t rps_cap = std::max<uint32_t>(
      max_rps, tn.max_rounds_per_slice ? tn.max_rounds_per_slice
                                       : (out.uniform && !blocked8 ? 8u
                                          : (blocked8 && k <= 64 ? 6u : (uint32_t)slg::kMaxRoundsPerSlice)));
  const uint32_t slices_per_sq = tn.slices_per_subquery;
  const bool slice_lists = !out.cand_mode;
  // the round targets are independent per sub-query: large batches (config 4: 65 536 sub-queries)
  // compute them on several threads; the offsets below are a serial prefix
  std::vector<uint32_t> targets(sqs.size());
  {
    auto fill = [&](size_t a, size_t b) {
      for (size_t i = a; i < b; i++) {
        const slg::RoundQuery &sq = sqs[i];
        const slg::TermRef *t = out.terms.data() + sq.term_begin;
        targets[i] = out.uniform ? uniform_round_target(sq, t, sq_postings[i], tn)
                                 : multi_round_target(sq_postings[i], segs[sq.seg].n_docs, tn);
      }
    };
    const size_t n_thr = sqs.size() >= 8192 ? planner_threads() : 1;
    if (n_thr <= 1) {
      fill(0, sqs.size());
    } else {
      std::vector<std::thread> pool;
      for (size_t th = 0; th < n_thr; th++)
        pool.emplace_back(fill, sqs.size() * th / n_thr, sqs.size() * (th + 1) / n_thr);
      for (auto &th : pool) th.join();
    }
  }
  for (size_t i = 0; i < sqs.size(); i++) {
    slg::RoundQuery &sq = sqs[i];
    const slg::TermRef *t = out.terms.data() + sq.term_begin;
    const uint32_t dfL = t[sq.longest].df;
    const uint32_t round_target = targets[i];
    // a round holds <= ~round_target postings of the essential lists (register slots) and
    // <= ~probe_target postings overall (non-essential lists are streamed per round), so
    // slices stay balanced whatever the mix
    uint64_t nr = (sq_postings[i] + round_target - 1) / round_target;
    nr = std::max<uint64_t>(nr, (sq_postings_all[i] + probe_target - 1) / probe_target);
    nr = std::max<uint64_t>(1, std::min<uint64_t>(nr, dfL));
    // sub-queries with many rounds get longer slices (fewer candidate lists for the merge,
    // whose time is set by the heaviest query); they are launched first (slice_order below)
    uint32_t want_rps = max_rps;
    if (!rps_pinned)
      want_rps = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(max_rps, (nr + slices_per_sq - 1) / slices_per_sq), rps_cap);
    // (the few-term kernel keeps a slice's cut points in one row of 64 words, 128 in the 5..8-list
    //  instance of the blocked form: (rps+1)*T <= 64 / 128)
    const uint32_t row = blocked8 && sq.n_terms > (uint32_t)slg::kUniMaxLists ? 128u : 64u;
    const uint32_t rps = out.multi ? std::max<uint32_t>(1, want_rps)
                                   : std::max<uint32_t>(1, std::min<uint32_t>(want_rps, row / sq.n_terms - 1));
    const uint64_t S = (nr + rps - 1) / rps;
    // (the per-slice candidate lists, n_slices * k entries indexed with 32 bits, exist only for
    //  k <= 256; larger k goes through the candidate array, one slot per posting)
    PLAN_REQUIRE(nr < 0x7FFFFFFFull && out.slice_sq.size() + S < 0x7FFFFFFFull &&
                     (!slice_lists ||
                      (out.slice_sq.size() + S) * (uint64_t)std::max<uint32_t>(k, 1) < 0xFFFFFFFFull),
                 "batch too large (rounds)");
    sq.n_rounds = (uint32_t)nr;
    sq.rounds_per_slice = rps;
    sq.slice_begin = (uint32_t)out.slice_sq.size();
    sq.n_slices = (uint32_t)S;
    PLAN_REQUIRE(out.n_bounds + (nr + 1) * sq.n_terms < 0xFFFFFFFFull, "batch too large (bounds)");
    sq.bounds_begin = (uint32_t)out.n_bounds;
    sq.rdoc_begin = (uint32_t)out.n_bnd;
    sq.bnd_begin = (uint32_t)out.n_bnd;
    out.n_bounds += (nr + 1) * sq.n_terms;
    out.n_bnd += nr + 1;
    out.n_rounds += nr;
    out.slice_sq.insert(out.slice_sq.end(), (size_t)S, (uint32_t)i);
    out.slice_seg.insert(out.slice_seg.end(), (size_t)S, sq.seg);
  }
}

// launch order: slices with the most rounds first (counting sort, stable), so the short ones fill
// the tail of the launch
void order_slices(const slg_tuning &tn, Plan &out) {
  out.slice_order.resize(out.slice_sq.size());
  std::vector<uint32_t> nrounds(out.slice_sq.size());
  uint32_t hist[slg::kMaxRoundsPerSlice + 2] = {0};
  for (const slg::RoundQuery &sq : out.sqs)
    for (uint32_t j = 0; j < sq.n_slices; j++) {
      const uint32_t r0 = j * sq.rounds_per_slice;
      const uint32_t n = std::min<uint32_t>(sq.rounds_per_slice, sq.n_rounds - r0);
      nrounds[sq.slice_begin + j] = n;
      hist[slg::kMaxRoundsPerSlice - n + 1]++;  // bucket 0 = most rounds
    }
  for (int i = 1; i <= slg::kMaxRoundsPerSlice + 1; i++) hist[i] += hist[i - 1];
  const bool lpt = tn.slice_order != 0;
  for (size_t sidx = 0; sidx < out.slice_sq.size(); sidx++)
    out.slice_order[lpt ? hist[slg::kMaxRoundsPerSlice - nrounds[sidx]]++ : sidx] = (uint32_t)sidx;
}

template <typename T>
size_t place(size_t &cursor, size_t count) {
  cursor = (cursor + 15) & ~(size_t)15;
  const size_t at = cursor;
  cursor += count * sizeof(T);
  return at;
}

}  // namespace

void Plan::layout() {
  size_t cur = 0;
  o_sq = place<slg::RoundQuery>(cur, sqs.size());
  o_terms = place<slg::TermRef>(cur, terms.size());
  o_slice = place<uint32_t>(cur, slice_sq.size());
  o_sseg = place<uint32_t>(cur, slice_seg.size());
  o_sord = place<uint32_t>(cur, slice_order.size());
  o_q = place<slg::QueryRef>(cur, qrefs.size());
  o_bc = place<uint32_t>(cur, bnd_coarse.size());
  o_nodes = place<slg::PlanNode>(cur, nodes.size());
  image_bytes = (cur + 15) & ~(size_t)15;
}

void Plan::pack(unsigned char *hb) const {
  if (!sqs.empty()) std::memcpy(hb + o_sq, sqs.data(), sqs.size() * sizeof(slg::RoundQuery));
  if (!terms.empty()) std::memcpy(hb + o_terms, terms.data(), terms.size() * sizeof(slg::TermRef));
  if (!slice_sq.empty()) std::memcpy(hb + o_slice, slice_sq.data(), slice_sq.size() * 4);
  if (!slice_seg.empty()) std::memcpy(hb + o_sseg, slice_seg.data(), slice_seg.size() * 4);
  if (!slice_order.empty()) std::memcpy(hb + o_sord, slice_order.data(), slice_order.size() * 4);
  if (!qrefs.empty()) std::memcpy(hb + o_q, qrefs.data(), qrefs.size() * sizeof(slg::QueryRef));
  if (!bnd_coarse.empty()) std::memcpy(hb + o_bc, bnd_coarse.data(), bnd_coarse.size() * 4);
  if (!nodes.empty()) std::memcpy(hb + o_nodes, nodes.data(), nodes.size() * sizeof(slg::PlanNode));
}

void plan_batch(const std::vector<SegView> &segs, const slg_tuning &tn, const BatchIn &in, Plan &out) {
  struct Running {
    Running() { g_plans_running.fetch_add(1, std::memory_order_relaxed); }
    ~Running() { g_plans_running.fetch_sub(1, std::memory_order_relaxed); }
  } running;
  const uint32_t nq = in.nq, k = in.k;
  const uint32_t n_segs = (uint32_t)segs.size();
  const BatchFacts facts = validate_batch(in, n_segs);
  out = Plan();
  out.q_postings.assign(nq, 0);
  std::vector<uint32_t> q_sq_begin(nq + 1, 0);
  // MaxScore classification: on request, or (auto) for batches with a query of more than 4 terms
  const bool maxscore_on =
      tn.pruning >= 0 ? tn.pruning != 0 : facts.max_nt > std::min<uint32_t>(tn.uniform_max_terms, slg::kUniMaxLists);
  Pass1Ctx ctx{segs, tn, in, facts, maxscore_on, q_sq_begin, out.q_postings};

  // Pass 1 is per query: large batches (config 4: 8192 queries x 8 segments = 65K sub-queries,
  // 15 ms on one thread, mostly cache misses in the champion tables) are planned by several
  // threads, each into its own vectors, stitched together in query order afterwards.
  std::vector<uint64_t> sq_postings, sq_postings_all;
  std::vector<uint32_t> sq_longest_all;
  double skip_est = 0.0;
  bool any_plan = false, any_filter = false;
  {
    const uint64_t work = (uint64_t)nq * n_segs;
    uint32_t n_thr = 1;
    if (work >= 8192)
      n_thr = (uint32_t)std::min<uint64_t>(planner_threads(), nq / 512);
    n_thr = std::max(1u, n_thr);
    std::vector<Pass1Out> parts(n_thr);
    auto lo_of = [&](uint32_t t) { return n_thr == 1 ? 0u : (uint32_t)((uint64_t)nq * t / n_thr); };
    if (n_thr == 1) {
      plan_queries(ctx, 0, nq, parts[0]);
    } else {
      std::vector<std::thread> pool;
      for (uint32_t t = 0; t < n_thr; t++)
        pool.emplace_back([&, t] {
          try {
            plan_queries(ctx, lo_of(t), t + 1 == n_thr ? nq : lo_of(t + 1), parts[t]);
          } catch (...) {
            parts[t].err = std::current_exception();
          }
        });
      for (auto &th : pool) th.join();
      for (auto &pt : parts)
        if (pt.err) std::rethrow_exception(pt.err);
    }
    for (uint32_t t = 0; t < n_thr; t++) {
      Pass1Out &pt = parts[t];
      const uint32_t sq_base = (uint32_t)out.sqs.size(), term_base = (uint32_t)out.terms.size();
      const uint32_t q_lo = lo_of(t), q_hi = t + 1 == n_thr ? nq : lo_of(t + 1);
      for (uint32_t q = q_lo; q < q_hi; q++) q_sq_begin[q] += sq_base;
      {
        const uint32_t node_base = (uint32_t)out.nodes.size();
        if (node_base)
          for (auto &sq : pt.sqs)
            if (sq.depth) sq.node_begin += node_base;
        out.nodes.insert(out.nodes.end(), pt.nodes.begin(), pt.nodes.end());
      }
      if (n_thr == 1) {
        out.sqs.swap(pt.sqs);
        out.terms.swap(pt.terms);
        sq_postings.swap(pt.sq_postings);
        sq_postings_all.swap(pt.sq_postings_all);
        sq_longest_all.swap(pt.sq_longest_all);
      } else {
        for (auto &sq : pt.sqs) sq.term_begin += term_base;
        out.sqs.insert(out.sqs.end(), pt.sqs.begin(), pt.sqs.end());
        out.terms.insert(out.terms.end(), pt.terms.begin(), pt.terms.end());
        sq_postings.insert(sq_postings.end(), pt.sq_postings.begin(), pt.sq_postings.end());
        sq_postings_all.insert(sq_postings_all.end(), pt.sq_postings_all.begin(), pt.sq_postings_all.end());
        sq_longest_all.insert(sq_longest_all.end(), pt.sq_longest_all.begin(), pt.sq_longest_all.end());
      }
      skip_est += pt.skip_est;
      out.n_postings += pt.n_postings;
      out.n_postings_essential += pt.n_ess;
      out.n_postings_nonessential += pt.n_noness;
      out.max_terms = std::max(out.max_terms, pt.max_terms);
      any_plan = any_plan || pt.any_plan;
      any_filter = any_filter || pt.any_filter;
      out.nested = out.nested || pt.any_nested;
      out.deep = out.deep || pt.any_deep;
    }
  }
  q_sq_begin[nq] = (uint32_t)out.sqs.size();

  // Classified lists only pay through block skipping (the many-term kernel loads a non-essential
  // list as it loads any other; what it saves is the blocks without a candidate doc).  A batch the
  // few-term kernel could take keeps its classification only if the expected skipped postings are
  // worth the slower kernel (config 3: lists of similar density, nothing to skip -> few-term kernel;
  // a stop word next to rare terms: 74 % skipped -> many-term kernel).  slg_tuning.pruning = 1 keeps
  // the classification whatever the estimate.
  if (tn.pruning < 0 && out.max_terms <= tn.uniform_max_terms && !any_plan &&
      skip_est < 0.15 * (double)out.n_postings) {
    for (size_t i = 0; i < out.sqs.size(); i++) {
      slg::RoundQuery &sq = out.sqs[i];
      sq.ess_mask = full_mask(sq.n_terms);
      sq.skip_mask = 0;
      sq.longest = sq_longest_all[i];
      sq_postings[i] = sq_postings_all[i];
    }
    out.n_postings_essential = out.n_postings;
    out.n_postings_nonessential = 0;
  }
  // which kernel: the few-term kernel (slg_score_uni4.hpp) takes batches of <= 8 lists per sub-query
  // without non-essential lists — flat sums, and (its plan instantiation, blocked form only) flat score
  // plans: Sum / DisMax over leaves of one or more terms, i.e. every multi-field query string
  // (api/reader.rs:2576-2586: `fields: None` = all text fields).  Two-level plans, more lists and
  // classified batches run on the many-term kernel
  const bool plans_fit = !any_plan || (tn.uniform_kernel >= 4 && !out.nested && !out.deep && tn.uniform_plans != 0);
  out.uniform = out.max_terms <= tn.uniform_max_terms && plans_fit;
  if (facts.min_match && !out.uniform)
    throw SlgError(SLG_ERR_UNSUPPORTED,
                   "minimum_should_match > 1 needs a flat score plan and at most 8 scored lists per segment in every query "
                   "of the batch (the few-term kernel's plan instantiation)");
  for (const slg::RoundQuery &sq : out.sqs)
    if (sq.ess_mask != full_mask(sq.n_terms)) {
      out.pruned = true;
      out.uniform = false;
    }
  out.multi = !out.uniform;
  out.plan_batch = any_plan;
  // large k: per-slice top-k lists would be mostly the slice itself; keep every doc above the
  // seed threshold instead (one candidate slot per posting) and select per query afterwards
  out.cand_mode = k > 256 && (k > 1024 || tn.cand_mode);

  plan_rounds(segs, tn, k, sq_postings, sq_postings_all, out);

  if (out.cand_mode)
    for (size_t i = 0; i < out.sqs.size(); i++) {
      out.sqs[i].cand_lo = (uint32_t)out.cand_total;
      out.sqs[i].cand_hi = (uint32_t)(out.cand_total >> 32);
      out.cand_total += sq_postings_all[i];
    }
  out.qrefs.resize(nq);
  for (uint32_t q = 0; q < nq; q++) {
    const uint32_t a = q_sq_begin[q], e = q_sq_begin[q + 1];
    if (a == e) {
      out.qrefs[q] = slg::QueryRef{0, 0};
    } else {
      out.qrefs[q].slice_begin = out.sqs[a].slice_begin;
      out.qrefs[q].slice_end = out.sqs[e - 1].slice_begin + out.sqs[e - 1].n_slices;
    }
  }
  order_slices(tn, out);
  // sub-query of every 32nd round boundary (partition_rounds_kernel walks from there)
  out.bnd_coarse.resize((size_t)((out.n_bnd + 31) / 32));
  {
    size_t i = 0;
    for (size_t c = 0; c < out.bnd_coarse.size(); c++) {
      const uint64_t bb = (uint64_t)c * 32;
      while (i + 1 < out.sqs.size() && out.sqs[i + 1].bnd_begin <= bb) i++;
      out.bnd_coarse[c] = (uint32_t)i;
    }
  }
  if (any_filter) {
    out.q_filter.assign(nq, 0u);
    for (uint32_t q = 0; q < nq; q++)
      if (in.q_filter[q] >= 0) out.q_filter[q] = (uint32_t)in.q_filter[q] + 1u;
  }
  out.layout();
}

}  // namespace slgplan
