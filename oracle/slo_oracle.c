/*
 * slo_oracle.c — CPU restatement of searchlite-core's BM25 / WAND / BMW top-k scorer,
 * the cross-segment merge and the vector rerank arithmetic.  See slo_oracle.h.
 *
 * TEST INFRASTRUCTURE ONLY (parity checker + CPU baseline).  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math   (see oracle/Makefile)
 * so every f32 operation is a separately rounded IEEE operation in the order the Rust
 * source performs it (rustc never contracts or re-associates float arithmetic).
 *
 * Citations are file:line relative to /root/reference/.
 */
#include "slo_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* scalar formulas                                                                       */
/* ------------------------------------------------------------------------------------ */

/* query/bm25.rs:1-6.  f32::ln is libm logf; f32::max(0.0) ignores NaN like fmaxf. */
float slo_bm25(float tf, float df, float doc_len, float avgdl, float docs, float k1, float b) {
  float idf = fmaxf(logf((docs - df + 0.5f) / (df + 0.5f)), 0.0f) + 1.0f;
  float norm_dl = (avgdl > 0.0f) ? (doc_len / avgdl) : 1.0f;
  float denom = tf + k1 * (1.0f - b + b * norm_dl);
  return idf * (tf * (k1 + 1.0f)) / fmaxf(denom, 1e-6f);
}

/* query/wand.rs:269-286 */
float slo_score_tf(float tf, float df, float doc_len, float avgdl, float docs, float k1, float b,
                   float weight) {
  float norm_len = (doc_len > 0.0f) ? doc_len : fmaxf(avgdl, tf);
  float base = slo_bm25(tf, df, norm_len, avgdl, docs, k1, b);
  return base * weight;
}

/* query/wand.rs:289-303 */
float slo_upper_bound_tf(float tf, float df, float doc_len, float avgdl, float docs, float k1,
                         float b, float weight) {
  if (tf <= 0.0f) return 0.0f;
  return slo_score_tf(tf, df, doc_len, avgdl, docs, k1, b, weight);
}

/* f32::total_cmp (core::f32): compare sign-magnitude bits as two's-complement keys. */
static inline int32_t total_key(float x) {
  int32_t bits;
  memcpy(&bits, &x, 4);
  bits ^= (int32_t)(((uint32_t)(bits >> 31)) >> 1);
  return bits;
}
int slo_total_cmp(float a, float b) {
  int32_t ka = total_key(a), kb = total_key(b);
  return (ka > kb) - (ka < kb);
}

/* query/wand.rs:77-84 ScoredTerm::doc_len (and TermState::doc_len :166-173) */
static inline float term_doc_len(const slo_term *t, uint32_t doc) {
  if (t->doc_lengths && doc < t->n_doc_lengths) {
    float v = t->doc_lengths[doc];
    if (v > 0.0f) return v;
  }
  return fmaxf(t->avgdl, 1.0f);
}

static inline int is_deleted(const uint8_t *deleted, uint32_t doc) {
  return deleted && ((deleted[doc >> 3] >> (doc & 7)) & 1);
}

/* ------------------------------------------------------------------------------------ */
/* std::collections::BinaryHeap restated (library/alloc/src/collections/binary_heap):    */
/* push = sift_up; pop = swap last into root, sift_down_to_bottom, sift_up;              */
/* From<Vec> = rebuild (sift_down from len/2-1 to 0).  Items are u64 payloads, ordered   */
/* by a caller comparator returning the Ord of a relative to b.                          */
/* ------------------------------------------------------------------------------------ */
typedef int (*heap_cmp_fn)(const void *ctx, uint64_t a, uint64_t b);
typedef struct {
  uint64_t *data;
  size_t len, cap;
  heap_cmp_fn cmp;
  const void *ctx;
} bheap;

static void bheap_init(bheap *h, heap_cmp_fn cmp, const void *ctx, size_t cap) {
  h->cap = cap < 8 ? 8 : cap;
  h->data = (uint64_t *)malloc(h->cap * sizeof(uint64_t));
  h->len = 0;
  h->cmp = cmp;
  h->ctx = ctx;
}
static void bheap_free(bheap *h) { free(h->data); }

static size_t bheap_sift_up(bheap *h, size_t start, size_t pos) {
  uint64_t elt = h->data[pos];
  while (pos > start) {
    size_t parent = (pos - 1) / 2;
    if (h->cmp(h->ctx, elt, h->data[parent]) <= 0) break;
    h->data[pos] = h->data[parent];
    pos = parent;
  }
  h->data[pos] = elt;
  return pos;
}
static void bheap_sift_down_range(bheap *h, size_t pos, size_t end) {
  uint64_t elt = h->data[pos];
  size_t child = 2 * pos + 1;
  size_t lim = end >= 2 ? end - 2 : 0;
  while (child <= lim && end >= 2) {
    child += (h->cmp(h->ctx, h->data[child], h->data[child + 1]) <= 0) ? 1 : 0;
    if (h->cmp(h->ctx, elt, h->data[child]) >= 0) {
      h->data[pos] = elt;
      return;
    }
    h->data[pos] = h->data[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (end >= 1 && child == end - 1 && h->cmp(h->ctx, elt, h->data[child]) < 0) {
    h->data[pos] = h->data[child];
    pos = child;
  }
  h->data[pos] = elt;
}
static void bheap_sift_down_to_bottom(bheap *h, size_t pos) {
  size_t end = h->len, start = pos;
  uint64_t elt = h->data[pos];
  size_t child = 2 * pos + 1;
  while (end >= 2 && child <= end - 2) {
    child += (h->cmp(h->ctx, h->data[child], h->data[child + 1]) <= 0) ? 1 : 0;
    h->data[pos] = h->data[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (end >= 1 && child == end - 1) {
    h->data[pos] = h->data[child];
    pos = child;
  }
  h->data[pos] = elt;
  bheap_sift_up(h, start, pos);
}
static void bheap_push(bheap *h, uint64_t item) {
  if (h->len == h->cap) {
    h->cap *= 2;
    h->data = (uint64_t *)realloc(h->data, h->cap * sizeof(uint64_t));
  }
  size_t old = h->len;
  h->data[h->len++] = item;
  bheap_sift_up(h, 0, old);
}
static int bheap_pop(bheap *h, uint64_t *out) {
  if (h->len == 0) return 0;
  uint64_t item = h->data[--h->len];
  if (h->len > 0) {
    uint64_t top = h->data[0];
    h->data[0] = item;
    item = top;
    bheap_sift_down_to_bottom(h, 0);
  }
  *out = item;
  return 1;
}
static void bheap_rebuild(bheap *h) {
  size_t n = h->len / 2;
  while (n > 0) {
    n--;
    bheap_sift_down_range(h, n, h->len);
  }
}

/* ------------------------------------------------------------------------------------ */
/* RankedDoc ordering and the bounded result heap                                        */
/* ------------------------------------------------------------------------------------ */

/* payload: score bits in the high word, doc id in the low word */
static inline uint64_t rd_pack(uint32_t doc, float score) {
  uint32_t sb;
  memcpy(&sb, &score, 4);
  return ((uint64_t)sb << 32) | doc;
}
static inline float rd_score(uint64_t p) {
  uint32_t sb = (uint32_t)(p >> 32);
  float s;
  memcpy(&s, &sb, 4);
  return s;
}
static inline uint32_t rd_doc(uint64_t p) { return (uint32_t)p; }

/* query/wand.rs:30-37 Ord for RankedDoc: score.total_cmp, then smaller doc_id is greater */
static int ranked_cmp(uint64_t a, uint64_t b) {
  int c = slo_total_cmp(rd_score(a), rd_score(b));
  if (c != 0) return c;
  uint32_t da = rd_doc(a), db = rd_doc(b);
  return (db > da) - (db < da); /* other.doc_id.cmp(&self.doc_id) */
}
/* BinaryHeap<Reverse<RankedDoc>>: Reverse flips the order */
static int ranked_rev_cmp(const void *ctx, uint64_t a, uint64_t b) {
  (void)ctx;
  return ranked_cmp(b, a);
}

/* query/wand.rs:905-916 push_top_k */
static void push_top_k(bheap *heap, uint64_t doc, uint32_t k) {
  if (heap->len < k) {
    bheap_push(heap, doc);
    return;
  }
  if (heap->len > 0) {
    uint64_t worst = heap->data[0];
    if (ranked_cmp(doc, worst) > 0) {
      uint64_t tmp;
      bheap_pop(heap, &tmp);
      bheap_push(heap, doc);
    }
  }
}

static int final_order_cmp(const void *pa, const void *pb) {
  /* query/wand.rs:920-924: b.score.total_cmp(&a.score).then(a.doc_id.cmp(&b.doc_id)) */
  uint64_t a = *(const uint64_t *)pa, b = *(const uint64_t *)pb;
  int c = slo_total_cmp(rd_score(b), rd_score(a));
  if (c != 0) return c;
  uint32_t da = rd_doc(a), db = rd_doc(b);
  return (da > db) - (da < db);
}

/* query/wand.rs:918-926 finalize_heap */
static int finalize_heap(bheap *heap, uint32_t *out_doc, float *out_score) {
  qsort(heap->data, heap->len, sizeof(uint64_t), final_order_cmp);
  for (size_t i = 0; i < heap->len; i++) {
    out_doc[i] = rd_doc(heap->data[i]);
    out_score[i] = rd_score(heap->data[i]);
  }
  return (int)heap->len;
}

/* ScoreExpr (query/planner.rs:113-153), restated as the recursive enum it is:
 *   Leaf(idx)                     -> leaves.get(idx).copied().unwrap_or(0.0)            (:124)
 *   Sum(children)                 -> children.iter().map(evaluate).sum::<f32>()          (:125-126);
 *                                    since Rust 1.83 the f32 Sum identity is -0.0 (toolchain pin
 *                                    1.92, rust-toolchain.toml:2)
 *   DisMax{children, tie_breaker} -> empty => 0.0; max starts at -inf, sum at 0.0; every child
 *                                    counts (a leaf without contributions evaluates to 0.0);
 *                                    max + tie * (sum - max)                             (:127-151)
 * A plan (ScorePlan, :156-165) = root expression + leaf_count. */
enum { EXPR_LEAF = 0, EXPR_SUM = 1, EXPR_DISMAX = 2 };
typedef struct expr {
  int kind;
  float tie;                   /* DisMax */
  uint32_t leaf;               /* Leaf */
  uint32_t n_children;         /* Sum / DisMax */
  const struct expr *children;
} expr;
typedef struct {
  const expr *root;
  uint32_t leaf_count;
} plan_t;

static float expr_evaluate(const expr *e, const float *leaves, uint32_t n_leaves) {
  switch (e->kind) {
    case EXPR_LEAF:
      return e->leaf < n_leaves ? leaves[e->leaf] : 0.0f;
    case EXPR_SUM: {
      float s = -0.0f;
      for (uint32_t i = 0; i < e->n_children; i++) s += expr_evaluate(&e->children[i], leaves, n_leaves);
      return s;
    }
    default: {
      if (e->n_children == 0) return 0.0f;
      float mx = -INFINITY, sum = 0.0f;
      for (uint32_t i = 0; i < e->n_children; i++) {
        float sc = expr_evaluate(&e->children[i], leaves, n_leaves);
        mx = fmaxf(mx, sc); /* f32::max: a NaN operand yields the other one, as fmaxf */
        sum += sc;
      }
      return mx + e->tie * (sum - mx);
    }
  }
}

static inline float plan_evaluate(const plan_t *p, const float *leaves) {
  return expr_evaluate(p->root, leaves, p->leaf_count);
}

/* Root over the leaves 0..leaf_count-1 (what a query string builds, planner.rs:354-360; DisMax of
 * leaves: dis_max of term queries).  nodes: caller storage for 1 + leaf_count expressions. */
static void plan_flat(plan_t *p, expr *nodes, int kind, float tie, uint32_t leaf_count) {
  for (uint32_t i = 0; i < leaf_count; i++) {
    expr lf = {EXPR_LEAF, 0.0f, i, 0, NULL};
    nodes[1 + i] = lf;
  }
  expr root = {kind == SLO_PLAN_DISMAX ? EXPR_DISMAX : EXPR_SUM, tie, 0, leaf_count, nodes + 1};
  nodes[0] = root;
  p->root = &nodes[0];
  p->leaf_count = leaf_count;
}

/* Two-level tree: the root combines groups, a group combines its (consecutive) leaves — the shapes
 * dis_max{queries} (planner.rs:470-487) and bool{should: [multi_match ...]} (:670-690) build.  A
 * Sum group of ONE leaf is the bare Leaf child the reference has there.
 * nodes: caller storage for 1 + n_groups + leaf_count expressions. */
static void plan_tree(plan_t *p, expr *nodes, int kind, float tie, uint32_t leaf_count, uint32_t n_groups,
                      const uint32_t *leaf_group, const int32_t *group_plan, const float *group_tie) {
  expr *groups = nodes + 1, *leafs = nodes + 1 + n_groups;
  for (uint32_t i = 0; i < leaf_count; i++) {
    expr lf = {EXPR_LEAF, 0.0f, i, 0, NULL};
    leafs[i] = lf;
  }
  uint32_t l = 0;
  for (uint32_t g = 0; g < n_groups; g++) {
    uint32_t first = l;
    while (l < leaf_count && leaf_group[l] == g) l++;
    if (l - first == 1 && group_plan[g] != SLO_PLAN_DISMAX) {
      groups[g] = leafs[first];
    } else {
      expr ge = {group_plan[g] == SLO_PLAN_DISMAX ? EXPR_DISMAX : EXPR_SUM, group_tie[g], 0, l - first, leafs + first};
      groups[g] = ge;
    }
  }
  expr root = {kind == SLO_PLAN_DISMAX ? EXPR_DISMAX : EXPR_SUM, tie, 0, n_groups, groups};
  nodes[0] = root;
  p->root = &nodes[0];
  p->leaf_count = leaf_count;
}

/* Any tree (ScoreExpr is recursive, planner.rs:113-153): nodes in PRE-ORDER, node 0 the root,
 * parent[i] < i; kind SLO_PLAN_SUM / SLO_PLAN_DISMAX / SLO_PLAN_LEAF; the i-th LEAF node in pre-order is
 * leaf i.  store: caller storage for 2 * n_nodes expressions (the nodes, then their child arrays). */
static void plan_nodes(plan_t *p, expr *store, uint32_t n_nodes, const int32_t *kind, const float *tie,
                       const uint32_t *parent) {
  uint32_t *cnt = (uint32_t *)calloc(n_nodes ? n_nodes : 1, sizeof(uint32_t));
  uint32_t *first = (uint32_t *)malloc((n_nodes ? n_nodes : 1) * sizeof(uint32_t));
  for (uint32_t i = 1; i < n_nodes; i++) cnt[parent[i]]++;
  uint32_t at = 0;
  for (uint32_t i = 0; i < n_nodes; i++) {
    first[i] = at;
    at += cnt[i];
  }
  expr *slots = store + n_nodes;
  uint32_t leaf = 0;
  for (uint32_t i = 0; i < n_nodes; i++) {
    expr e = {EXPR_SUM, 0.0f, 0, cnt[i], slots + first[i]};
    if (kind[i] == SLO_PLAN_LEAF) {
      e.kind = EXPR_LEAF;
      e.leaf = leaf++;
      e.n_children = 0;
      e.children = NULL;
    } else if (kind[i] == SLO_PLAN_DISMAX) {
      e.kind = EXPR_DISMAX;
      e.tie = tie[i];
    }
    store[i] = e;
  }
  /* children in pre-order = the reference's child order; a copied expr keeps pointing at its own
   * (stable) child slots, so the copy order does not matter */
  memset(cnt, 0, (n_nodes ? n_nodes : 1) * sizeof(uint32_t));
  for (uint32_t i = 1; i < n_nodes; i++) slots[first[parent[i]] + cnt[parent[i]]++] = store[i];
  free(cnt);
  free(first);
  p->root = &store[0];
  p->leaf_count = leaf;
}

/* ------------------------------------------------------------------------------------ */
/* brute force (ExecutionStrategy::Bm25)  query/wand.rs:459-566                          */
/* ------------------------------------------------------------------------------------ */

typedef struct {
  uint32_t *keys; /* doc ids, SLO_DOCID_END = empty */
  uint32_t mask;
  uint32_t used;
} docmap;

static void docmap_init(docmap *m, uint64_t expect) {
  uint64_t cap = 16;
  while (cap < expect * 2 + 2) cap <<= 1;
  m->keys = (uint32_t *)malloc(cap * sizeof(uint32_t));
  memset(m->keys, 0xFF, cap * sizeof(uint32_t));
  m->mask = (uint32_t)(cap - 1);
  m->used = 0;
}
static inline uint32_t docmap_slot(docmap *m, uint32_t doc, int *is_new) {
  uint32_t h = (doc * 2654435761u) & m->mask;
  for (;;) {
    uint32_t k = m->keys[h];
    if (k == doc) {
      *is_new = 0;
      return h;
    }
    if (k == SLO_DOCID_END) {
      m->keys[h] = doc;
      m->used++;
      *is_new = 1;
      return h;
    }
    h = (h + 1) & m->mask;
  }
}

static int brute_force(const slo_term *terms, uint32_t n_terms, uint32_t k, const plan_t *plan,
                       const uint8_t *deleted, uint32_t *out_doc, float *out_score,
                       slo_stats *stats) {
  const int use_plan = plan != NULL;
  uint64_t total = 0;
  for (uint32_t t = 0; t < n_terms; t++) total += terms[t].len;
  const uint32_t leaf_count = use_plan ? plan->leaf_count : 0;
  uint32_t width = use_plan ? (leaf_count ? leaf_count : 1) : 1;
  docmap map;
  docmap_init(&map, total);
  float *vals = (float *)calloc((size_t)(map.mask + 1) * width, sizeof(float));
  for (uint32_t t = 0; t < n_terms; t++) {
    const slo_term *term = &terms[t];
    float df = (float)term->len;
    if (stats) stats->postings_advanced += term->len;
    for (uint32_t i = 0; i < term->len; i++) {
      uint32_t doc = term->doc_ids[i];
      float score = slo_score_tf((float)term->tfs[i], df, term_doc_len(term, doc), term->avgdl,
                                 term->docs, term->k1, term->b, term->weight);
      int is_new;
      uint32_t slot = docmap_slot(&map, doc, &is_new);
      /* :488-497 buf[term.leaf] += score  |  :539 *entry.or_insert(0.0) += score */
      vals[(size_t)slot * width + (use_plan ? term->leaf : 0)] += score;
    }
  }
  if (stats) {
    stats->scored_docs += map.used;
    stats->candidates_examined += map.used;
  }
  bheap heap;
  bheap_init(&heap, ranked_rev_cmp, NULL, (size_t)k + 1);
  for (uint32_t s = 0; s <= map.mask; s++) {
    uint32_t doc = map.keys[s];
    if (doc == SLO_DOCID_END) continue;
    float score = use_plan ? plan_evaluate(plan, &vals[(size_t)s * width]) : vals[s];
    if (is_deleted(deleted, doc)) continue; /* accept */
    if (k > 0) push_top_k(&heap, rd_pack(doc, score), k);
  }
  int n = finalize_heap(&heap, out_doc, out_score);
  bheap_free(&heap);
  free(vals);
  free(map.keys);
  return n;
}

/* ------------------------------------------------------------------------------------ */
/* TermState  query/wand.rs:87-266                                                       */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  const slo_term *t;
  uint32_t idx;
  float df, ub, min_doc_len;
  uint32_t *block_max_doc_ids;
  float *block_max_tfs;
  uint32_t n_blocks;
  uint32_t block_size;
} term_state;

/* query/wand.rs:305-330 build_block_meta (the reuse branch :306-311 yields the same
 * arrays as rebuilding, index/postings.rs:101-111) */
static void build_block_meta(term_state *s) {
  uint32_t len = s->t->len, bs = s->block_size;
  uint32_t nb = (len + bs - 1) / bs;
  s->n_blocks = nb;
  s->block_max_doc_ids = (uint32_t *)malloc((nb ? nb : 1) * sizeof(uint32_t));
  s->block_max_tfs = (float *)malloc((nb ? nb : 1) * sizeof(float));
  uint32_t idx = 0, bi = 0;
  while (idx < len) {
    uint32_t end = idx + bs < len ? idx + bs : len;
    float tf_max = 0.0f;
    s->block_max_doc_ids[bi] = s->t->doc_ids[end - 1];
    for (uint32_t i = idx; i < end; i++) tf_max = fmaxf(tf_max, (float)s->t->tfs[i]);
    s->block_max_tfs[bi] = tf_max;
    bi++;
    idx = end;
  }
}

/* query/wand.rs:107-153 TermState::new */
static void term_state_new(term_state *s, const slo_term *t, uint32_t block_size,
                           const float *min_len_cached) {
  s->t = t;
  s->idx = 0;
  s->df = (float)t->len;
  s->block_size = block_size < 1 ? 1 : block_size;
  build_block_meta(s);
  if (t->doc_lengths) {
    float mn;
    if (min_len_cached) {
      mn = *min_len_cached;
    } else {
      mn = INFINITY; /* :112-116 fold(f32::INFINITY, f32::min) over positive entries */
      for (uint32_t i = 0; i < t->n_doc_lengths; i++) {
        float l = t->doc_lengths[i];
        if (l > 0.0f) mn = fminf(mn, l);
      }
    }
    s->min_doc_len = isfinite(mn) ? mn : fmaxf(t->avgdl, 1.0f);
  } else {
    s->min_doc_len = fmaxf(t->avgdl, 1.0f);
  }
  float max_tf = 0.0f; /* index/postings.rs:91-94 + :199-202: max over all tfs */
  for (uint32_t i = 0; i < s->n_blocks; i++) max_tf = fmaxf(max_tf, s->block_max_tfs[i]);
  s->ub = slo_upper_bound_tf(max_tf, s->df, s->min_doc_len, t->avgdl, t->docs, t->k1, t->b,
                             t->weight);
}
static void term_state_free(term_state *s) {
  free(s->block_max_doc_ids);
  free(s->block_max_tfs);
}
static inline int ts_is_done(const term_state *s) { return s->idx >= s->t->len; }
static inline uint32_t ts_doc_id(const term_state *s) {
  return s->idx < s->t->len ? s->t->doc_ids[s->idx] : SLO_DOCID_END;
}
static inline float ts_tf(const term_state *s) {
  return s->idx < s->t->len ? (float)s->t->tfs[s->idx] : 0.0f;
}
/* :184-195 */
static inline float ts_score_current(const term_state *s) {
  const slo_term *t = s->t;
  return slo_score_tf(ts_tf(s), s->df, term_doc_len(t, ts_doc_id(s)), t->avgdl, t->docs, t->k1,
                      t->b, t->weight);
}
/* :197-203 */
static inline uint32_t ts_advance(term_state *s) {
  if (ts_is_done(s)) return 0;
  s->idx += 1;
  return 1;
}
/* :205-232 galloping advance_to */
static uint32_t ts_advance_to(term_state *s, uint32_t target) {
  if (ts_is_done(s) || ts_doc_id(s) >= target) return 0;
  uint32_t len = s->t->len;
  uint32_t low = s->idx + 1;
  if (low >= len) {
    uint32_t delta = len - s->idx;
    s->idx = len;
    return delta;
  }
  uint64_t step = 1;
  while (low + step < len) {
    if (s->t->doc_ids[low + step] >= target) break;
    step <<= 1;
  }
  uint32_t upper = (low + step < len) ? (uint32_t)(low + step) : len;
  /* slice[low..upper].partition_point(|p| p.doc_id < target) */
  uint32_t lo = low, hi = upper;
  while (lo < hi) {
    uint32_t mid = lo + (hi - lo) / 2;
    if (s->t->doc_ids[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  uint32_t new_idx = lo < len ? lo : len;
  uint32_t delta = new_idx - s->idx;
  s->idx = new_idx;
  return delta;
}
/* :238-251 */
static inline float ts_block_upper_bound(const term_state *s) {
  uint32_t bi = s->idx / s->block_size;
  float tf = bi < s->n_blocks ? s->block_max_tfs[bi] : 0.0f;
  const slo_term *t = s->t;
  return slo_score_tf(tf, s->df, s->min_doc_len, t->avgdl, t->docs, t->k1, t->b, t->weight);
}
/* :257-265 */
static uint32_t ts_skip_to_block(term_state *s, uint32_t target) {
  uint32_t prev = s->idx;
  uint32_t lo = 0, hi = s->n_blocks;
  while (lo < hi) {
    uint32_t mid = lo + (hi - lo) / 2;
    if (s->block_max_doc_ids[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  uint64_t start = (uint64_t)lo * s->block_size;
  if (start > s->idx) s->idx = start < s->t->len ? (uint32_t)start : s->t->len;
  return s->idx - prev;
}

/* TermWrapper Ord (wand.rs:689-694): other.doc_id().cmp(&self.doc_id()) */
static int termq_cmp(const void *ctx, uint64_t a, uint64_t b) {
  const term_state *st = (const term_state *)ctx;
  uint32_t da = ts_doc_id(&st[a]), db = ts_doc_id(&st[b]);
  return (db > da) - (db < da);
}

/* ------------------------------------------------------------------------------------ */
/* wand_loop  query/wand.rs:659-903 (no collector, no score_adjust)                      */
/* ------------------------------------------------------------------------------------ */
static int wand_loop(term_state *st, uint32_t n_states, uint32_t k, int use_block_bounds,
                     const plan_t *plan, const uint8_t *deleted,
                     uint32_t *out_doc, float *out_score, slo_stats *stats) {
  const int use_plan = plan != NULL;
  const uint32_t leaf_count = use_plan ? plan->leaf_count : 0;
  int rank_hits = k > 0;
  bheap heap;
  bheap_init(&heap, ranked_rev_cmp, NULL, (size_t)k + 1);
  bheap queue;
  bheap_init(&queue, termq_cmp, st, n_states + 1);
  for (uint32_t i = 0; i < n_states; i++)
    if (!ts_is_done(&st[i])) queue.data[queue.len++] = i;
  bheap_rebuild(&queue); /* .collect() into BinaryHeap == From<Vec> == rebuild */

  float *leaf_scores = use_plan ? (float *)calloc(leaf_count ? leaf_count : 1, sizeof(float)) : NULL;
  uint8_t *touched_flags = use_plan ? (uint8_t *)calloc(leaf_count ? leaf_count : 1, 1) : NULL;
  uint32_t *touched = (uint32_t *)malloc((leaf_count ? leaf_count : 1) * sizeof(uint32_t));
  uint32_t n_touched = 0;
  uint32_t *pending = (uint32_t *)malloc((n_states ? n_states : 1) * sizeof(uint32_t));
  uint32_t n_pending = 0;

  for (;;) {
    if (queue.len == 0) break;
    if (ts_is_done(&st[queue.data[0]])) { /* :712-715 */
      uint64_t tmp;
      bheap_pop(&queue, &tmp);
      continue;
    }
    float heap_threshold = 0.0f; /* :717-721 */
    if (rank_hits && heap.len >= k) heap_threshold = heap.len ? rd_score(heap.data[0]) : 0.0f;
    float pivot_threshold = heap_threshold; /* no collector: :724-728 */

    int pivot_idx = -1;
    float acc = 0.0f;
    uint64_t w;
    while (bheap_pop(&queue, &w)) { /* :750-768 */
      const term_state *term = &st[w];
      float bound = use_block_bounds ? ts_block_upper_bound(term) : term->ub;
      pending[n_pending++] = (uint32_t)w;
      if (!isfinite(bound)) continue;
      acc += bound;
      if (acc >= pivot_threshold) {
        pivot_idx = (int)n_pending - 1;
        break;
      }
    }
    if (pivot_idx < 0) { /* :770-778 */
      for (uint32_t i = 0; i < n_pending; i++)
        if (!ts_is_done(&st[pending[i]])) bheap_push(&queue, pending[i]);
      n_pending = 0;
      break;
    }
    uint32_t pivot_doc = ts_doc_id(&st[pending[pivot_idx]]);
    uint32_t smallest_doc = ts_doc_id(&st[pending[0]]);

    if (pivot_doc == smallest_doc) {
      uint32_t doc_id = pivot_doc;
      while (queue.len > 0 && ts_doc_id(&st[queue.data[0]]) == doc_id) { /* :790-796 */
        uint64_t tmp;
        bheap_pop(&queue, &tmp);
        pending[n_pending++] = (uint32_t)tmp;
      }
      float score_sum = 0.0f;
      for (uint32_t i = 0; i < n_pending; i++) { /* :800-829 */
        term_state *term = &st[pending[i]];
        if (ts_doc_id(term) != doc_id) continue;
        float contribution = ts_score_current(term);
        score_sum += contribution;
        if (use_plan) {
          uint32_t leaf = term->t->leaf;
          if (!touched_flags[leaf]) {
            touched_flags[leaf] = 1;
            touched[n_touched++] = leaf;
          }
          leaf_scores[leaf] += contribution;
        }
        uint32_t moved = ts_advance(term);
        if (stats) stats->postings_advanced += moved;
      }
      if (stats) {
        stats->candidates_examined += 1;
        stats->scored_docs += 1;
      }
      float score = score_sum;
      if (use_plan) score = plan_evaluate(plan, leaf_scores); /* :834-839 */
      for (uint32_t i = 0; i < n_touched; i++) {                   /* :848-852 */
        leaf_scores[touched[i]] = 0.0f;
        touched_flags[touched[i]] = 0;
      }
      n_touched = 0;
      if (!is_deleted(deleted, doc_id)) { /* accept :858 */
        if (rank_hits && (heap.len < k || score > heap_threshold)) /* :862 */
          push_top_k(&heap, rd_pack(doc_id, score), k);
      }
    } else {
      for (int i = 0; i < pivot_idx; i++) { /* :883-891 .take(p_idx) */
        term_state *term = &st[pending[i]];
        if (use_block_bounds) {
          uint32_t moved = ts_skip_to_block(term, pivot_doc);
          if (stats) stats->postings_advanced += moved;
        }
        uint32_t moved = ts_advance_to(term, pivot_doc);
        if (stats) stats->postings_advanced += moved;
      }
    }
    for (uint32_t i = 0; i < n_pending; i++) /* :895-899 */
      if (!ts_is_done(&st[pending[i]])) bheap_push(&queue, pending[i]);
    n_pending = 0;
  }
  int n = finalize_heap(&heap, out_doc, out_score);
  bheap_free(&heap);
  bheap_free(&queue);
  free(leaf_scores);
  free(touched_flags);
  free(touched);
  free(pending);
  return n;
}

/* query/wand.rs:398-456 */
static int execute_with_plan(const slo_term *terms, uint32_t n_terms, uint32_t k, int strategy,
                             uint32_t block_size, const plan_t *plan, const uint8_t *deleted,
                             const float *min_len_cache, uint32_t *out_doc, float *out_score,
                             slo_stats *stats) {
  if (n_terms == 0 || k == 0) return 0; /* :413-416 (no collector) */
  if (strategy == SLO_BM25)
    return brute_force(terms, n_terms, k, plan, deleted, out_doc, out_score, stats);
  uint32_t bsize = block_size ? block_size : SLO_DEFAULT_BLOCK_SIZE;
  if (bsize < 1) bsize = 1;
  term_state *st = (term_state *)malloc((n_terms ? n_terms : 1) * sizeof(term_state));
  uint32_t ns = 0;
  for (uint32_t t = 0; t < n_terms; t++) {
    if (terms[t].len == 0) continue; /* :439-441 filter(postings.len() > 0) */
    term_state_new(&st[ns], &terms[t], bsize, min_len_cache ? &min_len_cache[t] : NULL);
    ns++;
  }
  int n = wand_loop(st, ns, k, strategy == SLO_BMW, plan, deleted, out_doc, out_score, stats);
  for (uint32_t i = 0; i < ns; i++) term_state_free(&st[i]);
  free(st);
  return n;
}

int slo_execute_top_k(const slo_term *terms, uint32_t n_terms, uint32_t k, int strategy,
                      uint32_t block_size, int use_plan, const uint8_t *deleted,
                      const float *min_len_cache, uint32_t *out_doc, float *out_score,
                      slo_stats *stats) {
  uint32_t leaf_count = 0;
  for (uint32_t t = 0; t < n_terms; t++)
    if (terms[t].leaf + 1 > leaf_count) leaf_count = terms[t].leaf + 1;
  expr *nodes = (expr *)malloc(((size_t)leaf_count + 1) * sizeof(expr));
  plan_t plan;
  plan_flat(&plan, nodes, SLO_PLAN_SUM, 0.0f, leaf_count);
  int n = execute_with_plan(terms, n_terms, k, strategy, block_size, use_plan ? &plan : NULL, deleted,
                            min_len_cache, out_doc, out_score, stats);
  free(nodes);
  return n;
}

int slo_execute_top_k_plan(const slo_term *terms, uint32_t n_terms, uint32_t k, int strategy,
                           uint32_t block_size, int plan_kind, float tie_breaker,
                           uint32_t leaf_count, const uint8_t *deleted, const float *min_len_cache,
                           uint32_t *out_doc, float *out_score, slo_stats *stats) {
  for (uint32_t t = 0; t < n_terms; t++)
    if (terms[t].leaf + 1 > leaf_count) leaf_count = terms[t].leaf + 1;
  expr *nodes = (expr *)malloc(((size_t)leaf_count + 1) * sizeof(expr));
  plan_t plan;
  plan_flat(&plan, nodes, plan_kind, tie_breaker, leaf_count);
  int n = execute_with_plan(terms, n_terms, k, strategy, block_size, &plan, deleted, min_len_cache,
                            out_doc, out_score, stats);
  free(nodes);
  return n;
}

int slo_execute_top_k_tree(const slo_term *terms, uint32_t n_terms, uint32_t k, int strategy,
                           uint32_t block_size, int plan_kind, float tie_breaker, uint32_t leaf_count,
                           uint32_t n_groups, const uint32_t *leaf_group, const int32_t *group_plan,
                           const float *group_tie, const uint8_t *deleted, const float *min_len_cache,
                           uint32_t *out_doc, float *out_score, slo_stats *stats) {
  expr *nodes = (expr *)malloc(((size_t)leaf_count + n_groups + 1) * sizeof(expr));
  plan_t plan;
  plan_tree(&plan, nodes, plan_kind, tie_breaker, leaf_count, n_groups, leaf_group, group_plan, group_tie);
  int n = execute_with_plan(terms, n_terms, k, strategy, block_size, &plan, deleted, min_len_cache,
                            out_doc, out_score, stats);
  free(nodes);
  return n;
}

int slo_execute_top_k_nodes(const slo_term *terms, uint32_t n_terms, uint32_t k, int strategy,
                            uint32_t block_size, uint32_t n_nodes, const int32_t *node_kind, const float *node_tie,
                            const uint32_t *node_parent, const uint8_t *deleted, const float *min_len_cache,
                            uint32_t *out_doc, float *out_score, slo_stats *stats) {
  expr *nodes = (expr *)malloc(((size_t)2 * n_nodes + 1) * sizeof(expr));
  plan_t plan;
  plan_nodes(&plan, nodes, n_nodes, node_kind, node_tie, node_parent);
  int n = execute_with_plan(terms, n_terms, k, strategy, block_size, &plan, deleted, min_len_cache,
                            out_doc, out_score, stats);
  free(nodes);
  return n;
}

/* ------------------------------------------------------------------------------------ */
/* batch driver: api/reader.rs:2670-2745 (segment loop), :2908-3128 (search_segment),    */
/* :2776-2778 + query/sort.rs:80-93 (merge order)                                        */
/* ------------------------------------------------------------------------------------ */
typedef struct {
  float score;
  uint32_t seg, doc;
} seg_hit;

static int seg_hit_cmp(const void *pa, const void *pb) {
  const seg_hit *a = (const seg_hit *)pa, *b = (const seg_hit *)pb;
  int c = slo_total_cmp(b->score, a->score); /* score desc */
  if (c != 0) return c;
  if (a->seg != b->seg) return (a->seg > b->seg) - (a->seg < b->seg);
  return (a->doc > b->doc) - (a->doc < b->doc);
}

typedef struct {
  const slo_segment *segs;
  uint32_t n_segs, nq;
  const uint32_t *q_offsets, *q_terms;
  const float *q_weights;
  uint32_t k;
  int strategy;
  uint32_t block_size;
  const float *min_len; /* [n_segs][max_fields] or NULL */
  uint32_t max_fields;
  uint32_t *out_doc, *out_seg;
  float *out_score;
  uint32_t *out_count;
  slo_stats *stats; /* per query or NULL */
  int tid, n_threads;
  const uint32_t *q_leaf;    /* per query term, or NULL: term i of a query is leaf i */
  const int32_t *q_plan;     /* per query SLO_PLAN_*, or NULL: SUM */
  const float *q_tie;        /* per query, or NULL */
  const uint32_t *q_nleaves; /* per query (leaves of the plan), or NULL: max leaf + 1 */
  /* two-level plans (NULL leaf_group: flat): CSR of per-leaf groups and per-group kind / tie */
  const uint32_t *q_leaf_offsets, *leaf_group, *q_group_offsets;
  const int32_t *group_plan;
  const float *group_tie;
  /* any tree (NULL q_node_offsets: the forms above): CSR of per-query node arrays in pre-order */
  const uint32_t *q_node_offsets, *node_parent;
  const int32_t *node_kind;
  const float *node_tie;
} batch_ctx;

static void run_query(const batch_ctx *c, uint32_t q, slo_term *terms, float *mins, uint32_t *tmp_doc,
                      float *tmp_score, seg_hit *hits) {
  uint32_t t0 = c->q_offsets[q], nt = c->q_offsets[q + 1] - t0;
  uint32_t n_hits = 0;
  uint32_t n_leaves = c->q_nleaves ? c->q_nleaves[q] : 0;
  for (uint32_t i = 0; i < nt; i++) {
    uint32_t lf = c->q_leaf ? c->q_leaf[t0 + i] : i;
    if (lf + 1 > n_leaves) n_leaves = lf + 1;
  }
  for (uint32_t s = 0; s < c->n_segs; s++) {
    const slo_segment *seg = &c->segs[s];
    uint32_t n = 0;
    for (uint32_t i = 0; i < nt; i++) {
      uint32_t tid = c->q_terms[(size_t)(t0 + i) * c->n_segs + s];
      if (tid == SLO_NO_TERM || tid >= seg->n_terms) continue; /* api/reader.rs:2989 */
      uint64_t off = seg->term_offsets[tid];
      uint32_t len = (uint32_t)(seg->term_offsets[tid + 1] - off);
      uint32_t field = seg->term_field ? seg->term_field[tid] : 0;
      slo_term *t = &terms[n];
      t->doc_ids = seg->doc_ids + off;
      t->tfs = seg->tfs + off;
      t->len = len;
      t->weight = c->q_weights[t0 + i];
      t->avgdl = seg->field_avgdl[field];
      t->docs = seg->docs;
      t->k1 = seg->k1;
      t->b = seg->b;
      t->leaf = c->q_leaf ? c->q_leaf[t0 + i] : i;
      /* field_lengths_for (api/reader.rs:3604-3621) always yields Some(vec of len doc_count) */
      t->doc_lengths = seg->field_doc_len[field];
      t->n_doc_lengths = seg->field_doc_len[field] ? seg->n_docs : 0;
      if (c->min_len) mins[n] = c->min_len[(size_t)s * c->max_fields + field];
      n++;
    }
    if (n == 0) continue; /* api/reader.rs:3003-3005 */
    int got;
    if (c->q_node_offsets) {
      uint32_t no = c->q_node_offsets[q];
      got = slo_execute_top_k_nodes(terms, n, c->k, c->strategy, c->block_size, c->q_node_offsets[q + 1] - no,
                                    c->node_kind + no, c->node_tie + no, c->node_parent + no, seg->deleted,
                                    c->min_len ? mins : NULL, tmp_doc, tmp_score, c->stats ? &c->stats[q] : NULL);
    } else if (c->leaf_group) {
      uint32_t lo = c->q_leaf_offsets[q], go = c->q_group_offsets[q];
      got = slo_execute_top_k_tree(terms, n, c->k, c->strategy, c->block_size,
                                   c->q_plan ? c->q_plan[q] : SLO_PLAN_SUM, c->q_tie ? c->q_tie[q] : 0.0f,
                                   c->q_leaf_offsets[q + 1] - lo, c->q_group_offsets[q + 1] - go,
                                   c->leaf_group + lo, c->group_plan + go, c->group_tie + go, seg->deleted,
                                   c->min_len ? mins : NULL, tmp_doc, tmp_score, c->stats ? &c->stats[q] : NULL);
    } else {
      got = slo_execute_top_k_plan(terms, n, c->k, c->strategy, c->block_size,
                                   c->q_plan ? c->q_plan[q] : SLO_PLAN_SUM,
                                   c->q_tie ? c->q_tie[q] : 0.0f, n_leaves, seg->deleted,
                                   c->min_len ? mins : NULL, tmp_doc, tmp_score,
                                   c->stats ? &c->stats[q] : NULL);
    }
    for (int i = 0; i < got; i++) {
      hits[n_hits].score = tmp_score[i];
      hits[n_hits].seg = s;
      hits[n_hits].doc = tmp_doc[i];
      n_hits++;
    }
  }
  qsort(hits, n_hits, sizeof(seg_hit), seg_hit_cmp);
  uint32_t keep = n_hits < c->k ? n_hits : c->k;
  for (uint32_t i = 0; i < keep; i++) {
    c->out_doc[(size_t)q * c->k + i] = hits[i].doc;
    c->out_seg[(size_t)q * c->k + i] = hits[i].seg;
    c->out_score[(size_t)q * c->k + i] = hits[i].score;
  }
  c->out_count[q] = keep;
}

static void *batch_worker(void *arg) {
  const batch_ctx *c = (const batch_ctx *)arg;
  uint32_t max_terms = 1;
  for (uint32_t q = 0; q < c->nq; q++) {
    uint32_t nt = c->q_offsets[q + 1] - c->q_offsets[q];
    if (nt > max_terms) max_terms = nt;
  }
  slo_term *terms = (slo_term *)malloc(max_terms * sizeof(slo_term));
  float *mins = (float *)malloc(max_terms * sizeof(float));
  uint32_t *tmp_doc = (uint32_t *)malloc((c->k ? c->k : 1) * sizeof(uint32_t));
  float *tmp_score = (float *)malloc((c->k ? c->k : 1) * sizeof(float));
  seg_hit *hits = (seg_hit *)malloc(((size_t)c->k * c->n_segs + 1) * sizeof(seg_hit));
  for (uint32_t q = (uint32_t)c->tid; q < c->nq; q += (uint32_t)c->n_threads)
    run_query(c, q, terms, mins, tmp_doc, tmp_score, hits);
  free(terms);
  free(mins);
  free(tmp_doc);
  free(tmp_score);
  free(hits);
  return NULL;
}

int slo_search_batch(const slo_segment *segs, uint32_t n_segs, uint32_t nq,
                     const uint32_t *q_offsets, const uint32_t *q_terms, const float *q_weights,
                     uint32_t k, int strategy, uint32_t block_size, int n_threads,
                     int cache_min_len, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                     uint32_t *out_count, slo_stats *stats_or_null) {
  return slo_search_batch_plan(segs, n_segs, nq, q_offsets, q_terms, q_weights, NULL, NULL, NULL, NULL,
                               k, strategy, block_size, n_threads, cache_min_len, out_doc, out_seg,
                               out_score, out_count, stats_or_null);
}

int slo_search_batch_plan(const slo_segment *segs, uint32_t n_segs, uint32_t nq,
                          const uint32_t *q_offsets, const uint32_t *q_terms, const float *q_weights,
                          const uint32_t *q_leaf, const int32_t *q_plan, const float *q_tie,
                          const uint32_t *q_nleaves, uint32_t k, int strategy, uint32_t block_size,
                          int n_threads, int cache_min_len, uint32_t *out_doc, uint32_t *out_seg,
                          float *out_score, uint32_t *out_count, slo_stats *stats_or_null) {
  return slo_search_batch_tree(segs, n_segs, nq, q_offsets, q_terms, q_weights, q_leaf, q_plan, q_tie, q_nleaves,
                               NULL, NULL, NULL, NULL, NULL, k, strategy, block_size, n_threads, cache_min_len,
                               out_doc, out_seg, out_score, out_count, stats_or_null);
}

static int search_batch_any(const slo_segment *segs, uint32_t n_segs, uint32_t nq,
                          const uint32_t *q_offsets, const uint32_t *q_terms, const float *q_weights,
                          const uint32_t *q_leaf, const int32_t *q_plan, const float *q_tie,
                          const uint32_t *q_nleaves, const uint32_t *q_leaf_offsets,
                          const uint32_t *leaf_group, const uint32_t *q_group_offsets,
                          const int32_t *group_plan, const float *group_tie, const uint32_t *q_node_offsets,
                          const int32_t *node_kind, const float *node_tie, const uint32_t *node_parent, uint32_t k,
                          int strategy, uint32_t block_size, int n_threads, int cache_min_len, uint32_t *out_doc,
                          uint32_t *out_seg, float *out_score, uint32_t *out_count,
                          slo_stats *stats_or_null);

int slo_search_batch_tree(const slo_segment *segs, uint32_t n_segs, uint32_t nq,
                          const uint32_t *q_offsets, const uint32_t *q_terms, const float *q_weights,
                          const uint32_t *q_leaf, const int32_t *q_plan, const float *q_tie,
                          const uint32_t *q_nleaves, const uint32_t *q_leaf_offsets,
                          const uint32_t *leaf_group, const uint32_t *q_group_offsets,
                          const int32_t *group_plan, const float *group_tie, uint32_t k, int strategy,
                          uint32_t block_size, int n_threads, int cache_min_len, uint32_t *out_doc,
                          uint32_t *out_seg, float *out_score, uint32_t *out_count,
                          slo_stats *stats_or_null) {
  return search_batch_any(segs, n_segs, nq, q_offsets, q_terms, q_weights, q_leaf, q_plan, q_tie, q_nleaves,
                          q_leaf_offsets, leaf_group, q_group_offsets, group_plan, group_tie, NULL, NULL, NULL, NULL, k,
                          strategy, block_size, n_threads, cache_min_len, out_doc, out_seg, out_score, out_count,
                          stats_or_null);
}

int slo_search_batch_nodes(const slo_segment *segs, uint32_t n_segs, uint32_t nq, const uint32_t *q_offsets,
                           const uint32_t *q_terms, const float *q_weights, const uint32_t *q_leaf,
                           const uint32_t *q_node_offsets, const int32_t *node_kind, const float *node_tie,
                           const uint32_t *node_parent, uint32_t k, int strategy, uint32_t block_size, int n_threads,
                           int cache_min_len, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                           uint32_t *out_count, slo_stats *stats_or_null) {
  if (!q_node_offsets || !node_kind || !node_tie || !node_parent) return -1;
  return search_batch_any(segs, n_segs, nq, q_offsets, q_terms, q_weights, q_leaf, NULL, NULL, NULL, NULL, NULL, NULL,
                          NULL, NULL, q_node_offsets, node_kind, node_tie, node_parent, k, strategy, block_size,
                          n_threads, cache_min_len, out_doc, out_seg, out_score, out_count, stats_or_null);
}

static int search_batch_any(const slo_segment *segs, uint32_t n_segs, uint32_t nq,
                          const uint32_t *q_offsets, const uint32_t *q_terms, const float *q_weights,
                          const uint32_t *q_leaf, const int32_t *q_plan, const float *q_tie,
                          const uint32_t *q_nleaves, const uint32_t *q_leaf_offsets,
                          const uint32_t *leaf_group, const uint32_t *q_group_offsets,
                          const int32_t *group_plan, const float *group_tie, const uint32_t *q_node_offsets,
                          const int32_t *node_kind, const float *node_tie, const uint32_t *node_parent, uint32_t k,
                          int strategy, uint32_t block_size, int n_threads, int cache_min_len, uint32_t *out_doc,
                          uint32_t *out_seg, float *out_score, uint32_t *out_count,
                          slo_stats *stats_or_null) {
  if (!segs || !q_offsets || !out_doc || !out_seg || !out_score || !out_count) return -1;
  if (n_threads < 1) n_threads = 1;
  if (nq == 0) return 0;
  for (uint32_t q = 0; q < nq; q++) out_count[q] = 0;
  if (stats_or_null) memset(stats_or_null, 0, nq * sizeof(slo_stats));
  uint32_t max_fields = 1;
  for (uint32_t s = 0; s < n_segs; s++)
    if (segs[s].n_fields > max_fields) max_fields = segs[s].n_fields;
  float *min_len = NULL;
  if (cache_min_len) {
    min_len = (float *)malloc((size_t)n_segs * max_fields * sizeof(float));
    for (uint32_t s = 0; s < n_segs; s++)
      for (uint32_t f = 0; f < segs[s].n_fields; f++) {
        float mn = INFINITY;
        const float *dl = segs[s].field_doc_len[f];
        if (dl)
          for (uint32_t i = 0; i < segs[s].n_docs; i++)
            if (dl[i] > 0.0f) mn = fminf(mn, dl[i]);
        min_len[(size_t)s * max_fields + f] = mn; /* non-finite => avgdl.max(1) in TermState::new */
      }
  }
  batch_ctx *ctxs = (batch_ctx *)malloc(n_threads * sizeof(batch_ctx));
  pthread_t *th = (pthread_t *)malloc(n_threads * sizeof(pthread_t));
  for (int t = 0; t < n_threads; t++) {
    batch_ctx c = {segs,      n_segs,   nq,        q_offsets, q_terms,   q_weights, k,
                   strategy,  block_size, min_len, max_fields, out_doc,  out_seg,   out_score,
                   out_count, stats_or_null, t,    n_threads, q_leaf, q_plan, q_tie, q_nleaves,
                   q_leaf_offsets, leaf_group, q_group_offsets, group_plan, group_tie,
                   q_node_offsets, node_parent, node_kind, node_tie};
    ctxs[t] = c;
  }
  if (n_threads == 1) {
    batch_worker(&ctxs[0]);
  } else {
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, batch_worker, &ctxs[t]);
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  }
  free(ctxs);
  free(th);
  free(min_len);
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* vectors / rerank                                                                      */
/* ------------------------------------------------------------------------------------ */

/* vectors/mod.rs:74-81 */
void slo_normalize_in_place(float *v, uint32_t dim) {
  float s = -0.0f;
  for (uint32_t i = 0; i < dim; i++) s += v[i] * v[i];
  float norm = sqrtf(s);
  if (norm > 0.0f)
    for (uint32_t i = 0; i < dim; i++) v[i] /= norm;
}

/* vectors/mod.rs:98-105 l2_distance, :107-120 metric_similarity */
float slo_metric_similarity(int metric, const float *a, const float *b, uint32_t dim) {
  if (metric == SLO_COSINE) {
    float dot = -0.0f; /* .sum::<f32>() left to right */
    for (uint32_t i = 0; i < dim; i++) dot += a[i] * b[i];
    return isnan(dot) ? 0.0f : dot;
  }
  float sum = 0.0f;
  for (uint32_t i = 0; i < dim; i++) {
    float d = a[i] - b[i];
    sum += d * d;
  }
  return -sqrtf(sum);
}

/* vectors/mod.rs:122-129 */
float slo_blend_scores(float bm25, float vector_score, float alpha, int higher_is_better) {
  float vec_component = higher_is_better ? vector_score : -vector_score;
  return alpha * bm25 + (1.0f - alpha) * vec_component;
}

/* api/reader.rs:217-223: Cosine => -1.0, L2 => f32::MIN */
float slo_missing_vector_score(int metric) { return metric == SLO_COSINE ? -1.0f : -3.40282347e+38f; }

typedef struct {
  float score, vec;
  uint32_t doc;
} rr_hit;
static int rr_cmp(const void *pa, const void *pb) {
  const rr_hit *a = (const rr_hit *)pa, *b = (const rr_hit *)pb;
  int c = slo_total_cmp(b->score, a->score);
  if (c != 0) return c;
  return (a->doc > b->doc) - (a->doc < b->doc);
}

int slo_rerank(int metric, uint32_t dim, const uint32_t *vec_offsets, uint32_t n_docs,
               const float *vec_values, const float *qvec, float alpha, const uint32_t *cand_doc,
               const float *cand_bm25, uint32_t n_cand, uint32_t k_out, uint32_t *out_doc,
               float *out_score, float *out_vec_score) {
  rr_hit *hits = (rr_hit *)malloc((n_cand ? n_cand : 1) * sizeof(rr_hit));
  for (uint32_t i = 0; i < n_cand; i++) {
    uint32_t doc = cand_doc[i];
    float vs;
    /* VectorStore::vector vectors/mod.rs:63-71 */
    if (doc < n_docs && vec_offsets[doc] != 0xFFFFFFFFu)
      vs = slo_metric_similarity(metric, qvec, vec_values + (size_t)vec_offsets[doc] * dim, dim);
    else
      vs = slo_missing_vector_score(metric);
    float blended; /* api/reader.rs:240-246, one clause => /1.0 */
    if (alpha >= 1.0f)
      blended = cand_bm25[i];
    else if (alpha <= 0.0f)
      blended = vs;
    else
      blended = slo_blend_scores(cand_bm25[i], vs, alpha, 1);
    float blended_sum = 0.0f;
    blended_sum += blended;
    hits[i].score = blended_sum / 1.0f;
    hits[i].vec = vs;
    hits[i].doc = doc;
  }
  qsort(hits, n_cand, sizeof(rr_hit), rr_cmp);
  uint32_t keep = n_cand < k_out ? n_cand : k_out;
  for (uint32_t i = 0; i < keep; i++) {
    out_doc[i] = hits[i].doc;
    out_score[i] = hits[i].score;
    if (out_vec_score) out_vec_score[i] = hits[i].vec;
  }
  free(hits);
  return (int)keep;
}


/* api/reader.rs:225-254 compute_hybrid_score with n_clauses vector clauses over one candidate set
 * (one vector field): per clause vec = boost * metric_similarity (api/reader.rs:2421), missing
 * vector => missing_vector_score for every clause; blended mean; vector_sum reported. */
int slo_rerank_multi(int metric, uint32_t dim, const uint32_t *vec_offsets, uint32_t n_docs,
                     const float *vec_values, uint32_t n_clauses, const float *qvecs, const float *alpha,
                     const float *boost, const uint32_t *cand_doc, const float *cand_bm25,
                     uint32_t n_cand, uint32_t k_out, uint32_t *out_doc, float *out_score,
                     float *out_vec_score) {
  rr_hit *hits = (rr_hit *)malloc((n_cand ? n_cand : 1) * sizeof(rr_hit));
  for (uint32_t i = 0; i < n_cand; i++) {
    uint32_t doc = cand_doc[i];
    int has = doc < n_docs && vec_offsets[doc] != 0xFFFFFFFFu;
    float blended_sum = 0.0f, vector_sum = 0.0f;
    for (uint32_t c = 0; c < n_clauses; c++) {
      float vs;
      if (has) {
        vs = slo_metric_similarity(metric, qvecs + (size_t)c * dim,
                                   vec_values + (size_t)vec_offsets[doc] * dim, dim);
        vs *= boost ? boost[c] : 1.0f;
        vector_sum += vs;
      } else {
        vs = slo_missing_vector_score(metric);
      }
      float blended;
      if (alpha[c] >= 1.0f)
        blended = cand_bm25[i];
      else if (alpha[c] <= 0.0f)
        blended = vs;
      else
        blended = slo_blend_scores(cand_bm25[i], vs, alpha[c], 1);
      blended_sum += blended;
    }
    float denom = (float)(n_clauses > 1 ? n_clauses : 1);
    hits[i].score = blended_sum / denom;
    hits[i].vec = has ? vector_sum : slo_missing_vector_score(metric);
    hits[i].doc = doc;
  }
  qsort(hits, n_cand, sizeof(rr_hit), rr_cmp);
  uint32_t keep = n_cand < k_out ? n_cand : k_out;
  for (uint32_t i = 0; i < keep; i++) {
    out_doc[i] = hits[i].doc;
    out_score[i] = hits[i].score;
    if (out_vec_score) out_vec_score[i] = hits[i].vec;
  }
  free(hits);
  return (int)keep;
}


/* ------------------------------------------------------------------------------------ */
/* BASELINE.md "Baseline A": the per-query cost IndexReader::search pays around the scorer —   */
/* modelled for context only (it is what a searchlite user experiences at scale).  Per query and */
/* segment, for every term: its posting list is varint-decoded from the serialized layout TWICE  */
/* (matcher doc lists, api/reader.rs:1732-1735; scorer postings, index/segment.rs:1328 ->        */
/* index/postings.rs:142-212, one byte at a time, util/varint.rs:31-48); per field the dense      */
/* doc-length vector is rebuilt with one keyed lookup per doc (field_lengths_for,                */
/* api/reader.rs:3604-3621: `fast_fields.i64_value(&key, doc)` hashes the column name each time); */
/* TermState::new rescans it for min_doc_len (wand.rs:111-125).  Favourable to the CPU where the */
/* reference does more (no Vec growth, FNV instead of SipHash, no RefCell / RwLock per byte).    */
/* Returns the seconds spent in the query phase (serialization of the lists is setup).           */
/* ------------------------------------------------------------------------------------ */
#include <time.h>

typedef struct {
  uint8_t *bytes;      /* concatenated serialized lists of one segment */
  uint64_t *off;       /* [n_terms] offset of a term's list, UINT64_MAX if not serialized */
} enc_seg;

static size_t put_var(uint8_t *o, uint32_t v) {
  size_t n = 0;
  while (v >= 0x80) {
    o[n++] = (uint8_t)((v & 0x7F) | 0x80);
    v >>= 7;
  }
  o[n++] = (uint8_t)v;
  return n;
}

static inline uint32_t get_var(const uint8_t **pp) { /* util/varint.rs:31-48, byte at a time */
  const uint8_t *p = *pp;
  uint32_t value = 0, shift = 0;
  for (;;) {
    uint8_t b = *p++;
    value |= (uint32_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) break;
    shift += 7;
  }
  *pp = p;
  return value;
}

typedef struct {
  batch_ctx base;
  const enc_seg *enc;
} faithful_ctx;

static uint64_t fnv1a(const char *s) {
  uint64_t h = 1469598103934665603ull;
  while (*s) {
    h ^= (uint8_t)*s++;
    h *= 1099511628211ull;
  }
  return h;
}

static void *faithful_worker(void *arg) {
  const faithful_ctx *fc = (const faithful_ctx *)arg;
  const batch_ctx *c = &fc->base;
  uint32_t max_terms = 1;
  for (uint32_t q = 0; q < c->nq; q++) {
    uint32_t nt = c->q_offsets[q + 1] - c->q_offsets[q];
    if (nt > max_terms) max_terms = nt;
  }
  slo_term *terms = (slo_term *)malloc(max_terms * sizeof(slo_term));
  uint32_t *tmp_doc = (uint32_t *)malloc((c->k ? c->k : 1) * sizeof(uint32_t));
  float *tmp_score = (float *)malloc((c->k ? c->k : 1) * sizeof(float));
  seg_hit *hits = (seg_hit *)malloc(((size_t)c->k * c->n_segs + 1) * sizeof(seg_hit));
  uint32_t **dbuf = (uint32_t **)calloc(max_terms * 3, sizeof(uint32_t *));
  volatile uint64_t sink = 0;
  for (uint32_t q = (uint32_t)c->tid; q < c->nq; q += (uint32_t)c->n_threads) {
    uint32_t t0 = c->q_offsets[q], nt = c->q_offsets[q + 1] - t0, n_hits = 0;
    for (uint32_t s = 0; s < c->n_segs; s++) {
      const slo_segment *seg = &c->segs[s];
      float *lens_by_field[64] = {0};
      uint32_t n = 0;
      for (uint32_t i = 0; i < nt; i++) {
        uint32_t tid = c->q_terms[(size_t)(t0 + i) * c->n_segs + s];
        if (tid == SLO_NO_TERM || tid >= seg->n_terms || fc->enc[s].off[tid] == UINT64_MAX) continue;
        uint32_t field = seg->term_field ? seg->term_field[tid] : 0;
        /* decode #1: matcher doc list */
        const uint8_t *p = fc->enc[s].bytes + fc->enc[s].off[tid];
        uint32_t df;
        memcpy(&df, p, 4);
        p += 4 + 1 + 4 + 4 + 4; /* doc_freq, flag, blocks, max_doc, max_tf (no block arrays kept here) */
        uint32_t *m_docs = (uint32_t *)malloc((df ? df : 1) * sizeof(uint32_t));
        {
          const uint8_t *r = p;
          for (uint32_t j = 0; j < df; j++) {
            m_docs[j] = get_var(&r);
            (void)get_var(&r);
          }
        }
        sink += m_docs[df ? df - 1 : 0];
        /* decode #2: scorer postings */
        uint32_t *docs = (uint32_t *)malloc((df ? df : 1) * sizeof(uint32_t));
        uint32_t *tfs = (uint32_t *)malloc((df ? df : 1) * sizeof(uint32_t));
        {
          const uint8_t *r = p;
          for (uint32_t j = 0; j < df; j++) {
            docs[j] = get_var(&r);
            tfs[j] = get_var(&r);
          }
        }
        dbuf[n * 3] = m_docs;
        dbuf[n * 3 + 1] = docs;
        dbuf[n * 3 + 2] = tfs;
        /* field_lengths_for: rebuilt once per field per (query, segment) */
        if (field < 64 && lens_by_field[field] == NULL && seg->field_doc_len[field]) {
          float *lens = (float *)malloc((seg->n_docs ? seg->n_docs : 1) * sizeof(float));
          const float *col = seg->field_doc_len[field];
          uint64_t h = 0;
          for (uint32_t d = 0; d < seg->n_docs; d++) {
            h += fnv1a("_len:body"); /* the keyed column lookup of every i64_value call */
            lens[d] = col[d];
          }
          sink += h;
          lens_by_field[field] = lens;
        }
        slo_term *t = &terms[n];
        t->doc_ids = docs;
        t->tfs = tfs;
        t->len = df;
        t->weight = c->q_weights[t0 + i];
        t->avgdl = seg->field_avgdl[field];
        t->docs = seg->docs;
        t->k1 = seg->k1;
        t->b = seg->b;
        t->leaf = i;
        t->doc_lengths = field < 64 ? lens_by_field[field] : NULL;
        t->n_doc_lengths = t->doc_lengths ? seg->n_docs : 0;
        n++;
      }
      if (n) {
        int got = slo_execute_top_k_plan(terms, n, c->k, c->strategy, c->block_size, SLO_PLAN_SUM, 0.0f, nt,
                                         seg->deleted, NULL /* min_doc_len rescanned */, tmp_doc, tmp_score, NULL);
        for (int i = 0; i < got; i++) {
          hits[n_hits].score = tmp_score[i];
          hits[n_hits].seg = s;
          hits[n_hits].doc = tmp_doc[i];
          n_hits++;
        }
      }
      for (uint32_t i = 0; i < n * 3; i++) free(dbuf[i]);
      for (int f = 0; f < 64; f++) free(lens_by_field[f]);
    }
    qsort(hits, n_hits, sizeof(seg_hit), seg_hit_cmp);
    uint32_t keep = n_hits < c->k ? n_hits : c->k;
    for (uint32_t i = 0; i < keep; i++) {
      c->out_doc[(size_t)q * c->k + i] = hits[i].doc;
      c->out_seg[(size_t)q * c->k + i] = hits[i].seg;
      c->out_score[(size_t)q * c->k + i] = hits[i].score;
    }
    c->out_count[q] = keep;
  }
  free(terms);
  free(tmp_doc);
  free(tmp_score);
  free(hits);
  free(dbuf);
  return NULL;
}

double slo_search_batch_faithful(const slo_segment *segs, uint32_t n_segs, uint32_t nq,
                                 const uint32_t *q_offsets, const uint32_t *q_terms, const float *q_weights,
                                 uint32_t k, int strategy, int n_threads, uint32_t *out_doc,
                                 uint32_t *out_seg, float *out_score, uint32_t *out_count) {
  if (n_threads < 1) n_threads = 1;
  /* setup (untimed): serialize every distinct query term's list as PostingsWriter::write_term
   * does (header without block arrays; absolute varint doc ids, varint tf) */
  enc_seg *enc = (enc_seg *)calloc(n_segs, sizeof(enc_seg));
  uint32_t total_terms = nq ? q_offsets[nq] : 0;
  for (uint32_t s = 0; s < n_segs; s++) {
    const slo_segment *seg = &segs[s];
    enc[s].off = (uint64_t *)malloc((seg->n_terms ? seg->n_terms : 1) * sizeof(uint64_t));
    for (uint32_t t = 0; t < seg->n_terms; t++) enc[s].off[t] = UINT64_MAX;
    size_t cap = 1 << 20, used = 0;
    enc[s].bytes = (uint8_t *)malloc(cap);
    for (uint32_t i = 0; i < total_terms; i++) {
      uint32_t tid = q_terms[(size_t)i * n_segs + s];
      if (tid == SLO_NO_TERM || tid >= seg->n_terms || enc[s].off[tid] != UINT64_MAX) continue;
      uint64_t a = seg->term_offsets[tid];
      uint32_t df = (uint32_t)(seg->term_offsets[tid + 1] - a);
      size_t need = 17 + (size_t)df * 10;
      if (used + need > cap) {
        while (used + need > cap) cap *= 2;
        enc[s].bytes = (uint8_t *)realloc(enc[s].bytes, cap);
      }
      enc[s].off[tid] = used;
      uint8_t *o = enc[s].bytes + used;
      memcpy(o, &df, 4);
      o[4] = 0;
      memset(o + 5, 0, 12);
      size_t n = 17;
      for (uint32_t j = 0; j < df; j++) {
        n += put_var(o + n, seg->doc_ids[a + j]);
        n += put_var(o + n, seg->tfs[a + j]);
      }
      used += n;
    }
  }
  faithful_ctx *ctxs = (faithful_ctx *)calloc(n_threads, sizeof(faithful_ctx));
  pthread_t *th = (pthread_t *)malloc(n_threads * sizeof(pthread_t));
  struct timespec ta, tb;
  clock_gettime(CLOCK_MONOTONIC, &ta);
  for (int t = 0; t < n_threads; t++) {
    batch_ctx *c = &ctxs[t].base;
    c->segs = segs;
    c->n_segs = n_segs;
    c->nq = nq;
    c->q_offsets = q_offsets;
    c->q_terms = q_terms;
    c->q_weights = q_weights;
    c->k = k;
    c->strategy = strategy;
    c->block_size = 0;
    c->out_doc = out_doc;
    c->out_seg = out_seg;
    c->out_score = out_score;
    c->out_count = out_count;
    c->tid = t;
    c->n_threads = n_threads;
    ctxs[t].enc = enc;
    pthread_create(&th[t], NULL, faithful_worker, &ctxs[t]);
  }
  for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  clock_gettime(CLOCK_MONOTONIC, &tb);
  for (uint32_t s = 0; s < n_segs; s++) {
    free(enc[s].bytes);
    free(enc[s].off);
  }
  free(enc);
  free(ctxs);
  free(th);
  return (double)(tb.tv_sec - ta.tv_sec) + 1e-9 * (double)(tb.tv_nsec - ta.tv_nsec);
}
