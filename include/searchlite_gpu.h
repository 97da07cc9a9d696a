/*
 * searchlite_gpu.h — C ABI of the MI355X (gfx950) batched BM25 top-k scorer + vector rerank.
 *
 * This is the drop-in boundary for ONE path of davidkelley/searchlite: the per-segment
 * top-k scorer that `IndexReader::search_segment` calls
 * (searchlite-core/src/api/reader.rs:3075-3099 ->
 *  execute_top_k_with_stats_and_mode_internal, searchlite-core/src/query/wand.rs:398-412),
 * the cross-segment merge that follows it (api/reader.rs:2776-2778, query/sort.rs:80-93),
 * and the rerank slot `gpu::rerank` (searchlite-core/src/gpu/rerank.rs:3) behind the
 * `gpu` cargo feature (searchlite-core/src/lib.rs:11-12, Cargo.toml:13).
 *
 * Conventions follow searchlite's own C FFI (searchlite-ffi/searchlite.h:12-18,
 * searchlite-ffi/src/lib.rs:24-43,58-91): opaque handles, NULL / negative int on error,
 * caller-owned output buffers, no exceptions or panics across the boundary, and a
 * thread-local last-error string.  Plain pointers and sizes only.
 *
 * Eligibility (the caller keeps every other request shape on searchlite's CPU scorer;
 * SURVEY.md section 8b): ScoreMode::Score, sort = _score desc, no collector/aggs, no
 * score_adjust/explain, no cursor, matcher = pure disjunction; filters as doc bitmaps
 * (slg_index_add_filter*: accept() = !is_deleted(doc) && filter(doc)); ScorePlan = any tree of Sum /
 * DisMax nodes up to SLG_MAX_PLAN_DEPTH levels above its leaves, each leaf the sum of one or more
 * scored terms (slg_batch_prepare_plans; the default is leaf i == query term i, summed);
 * k = limit + 1 up to 20 001; up to 32 scored terms per query and segment.
 */
#ifndef SEARCHLITE_GPU_H
#define SEARCHLITE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLG_ABI_VERSION 3u  /* 3: slg_tuning grew (updatable, uniform_plans, score_waves_per_simd); index updates (slg_index_update_deleted / _add_segment /
                               _remove_segment / _generation); request coalescer; slg_batch_device_candidates.
                               2: pool_cap_mb, uniform_kernel, uniform_sigma_x100, inline_cuts; shard groups; slg_batch_prepare_plans */
#define SLG_NO_TERM 0xFFFFFFFFu      /* term absent from a segment (api/reader.rs:2989) */
#define SLG_NO_VECTOR 0xFFFFFFFFu    /* vectors/mod.rs:65-67 (u32::MAX offset) */
#define SLG_MAX_QUERY_TERMS 32u      /* scored terms per query per segment */
#define SLG_MAX_K 20001u             /* k = min(max(candidate_size, limit), 20000) + 1 (api/reader.rs:2615-2619) */
#define SLG_MAX_RERANK_K 1024u       /* largest k_out of slg_rerank_*: the reference's own cap on a vector
                                        clause's k (MAX_VECTOR_K, api/reader.rs:136) */
#define SLG_MAX_VECTOR_CLAUSES 8u    /* MAX_VECTOR_CLAUSES, api/reader.rs:134 */
#define SLG_BLOCK_SIZE 128u          /* index/postings.rs:11 DEFAULT_BLOCK_SIZE */

/* return codes (searchlite-ffi/src/lib.rs returns NULL / -1..-5 / 0) */
enum {
  SLG_OK = 0,
  SLG_ERR_INVALID = -1,     /* bad argument / malformed arrays */
  SLG_ERR_DEVICE = -2,      /* HIP runtime error, no gfx950 device */
  SLG_ERR_OOM = -3,         /* host or device allocation failed */
  SLG_ERR_UNSUPPORTED = -4, /* request shape outside the eligibility predicate */
  SLG_ERR_INTERNAL = -5
};

/* api/types.rs:6-13 ExecutionStrategy.  All three return the exhaustive-exact top-k
 * (the reference's own parity standard: tests/pruning.rs:44-104 Bm25 == Wand == Bmw).
 * Scores are ordered by f32 total_cmp (-0.0 below +0.0), as RankedDoc::cmp does
 * (query/wand.rs:30-37). */
enum { SLG_STRATEGY_BM25 = 0, SLG_STRATEGY_WAND = 1, SLG_STRATEGY_BMW = 2 };

/* index/manifest.rs VectorMetric */
enum { SLG_METRIC_COSINE = 0, SLG_METRIC_L2 = 1 };

typedef struct slg_index slg_index; /* device-resident index (all segments of one shard) */
typedef struct slg_batch slg_batch; /* one prepared query batch */

/*
 * One searchlite segment, in exactly the decoded form `search_segment` hands the scorer
 * (api/reader.rs:2985-3000): per-term posting lists (PostingsReader, index/postings.rs:133-139)
 * as a CSR over terms, per-field dense doc-length vectors (field_lengths_for,
 * api/reader.rs:3604-3621), avg field length (index/segment.rs:848), live_docs, and the
 * index-wide BM25 parameters (IndexOptions, api/types.rs:16-26).
 * All host arrays are borrowed for the duration of slg_index_create only.
 */
typedef struct {
  uint32_t n_docs;               /* seg.meta.doc_count */
  uint32_t n_terms;              /* V: number of "field:term" keys */
  const uint64_t *term_offsets;  /* [V+1] postings CSR; df(t) = off[t+1]-off[t] */
  const uint32_t *doc_ids;       /* [P] strictly increasing within a term */
  const uint32_t *tfs;           /* [P] term_freq */
  const uint16_t *term_field;    /* [V] field id of each term, NULL => field 0 */
  uint32_t n_fields;
  const float *const *field_doc_len; /* [n_fields] each f32[n_docs] (0 => missing) or NULL */
  const float *field_avgdl;          /* [n_fields] seg.avg_field_length(field) */
  float docs;                        /* seg.live_docs() as f32 */
  float k1, b;                       /* options.bm25_k1 / bm25_b */
  const uint8_t *deleted;            /* bitmap, bit (d&7) of byte d>>3; NULL => none */
  /* optional vector field for slg_rerank_batch (vectors/mod.rs:10-17 VectorStore) */
  uint32_t vec_dim;                  /* 0 => no vectors */
  int32_t vec_metric;                /* SLG_METRIC_* */
  const uint32_t *vec_offsets;       /* [n_docs] row index or SLG_NO_VECTOR */
  const float *vec_values;           /* [vec_rows * vec_dim] row-major (cosine: pre-normalized,
                                        index/segment.rs:508-510) */
  uint32_t vec_rows;
} slg_segment_desc;

/* query/wand.rs:45-50 QueryStats, per query, with brute_force's accounting (wand.rs:472,500-503)
 * applied to the work the device did:
 *   scored_docs = candidates_examined = distinct docs that got a score.  SLG_STRATEGY_BM25: every
 *     doc holding a query term (tombstoned docs included: accept() runs at top-k insertion) — the
 *     count brute_force reports.  SLG_STRATEGY_WAND / _BMW: the same when the batch runs unclassified
 *     (slg_tuning.pruning: the default for queries of <= 8 terms unless block skipping pays); with the
 *     MaxScore classification kept, docs of the ESSENTIAL lists only (a doc found in non-essential
 *     lists alone is never scored);
 *   postings_advanced = postings of the query's lists minus those block skipping never loaded.
 * Under Wand / Bmw these are NOT the reference's wand_loop counters (wand.rs:826-835, 883-891 count
 * the pivot sequence of its cursors, which the device does not walk): a caller that derives
 * total_hits_estimate from them gets a different (larger, closer to the true match count) estimate
 * than searchlite's CPU path, whose own figure under Wand / Bmw already depends on what pruning
 * skipped.  Only the Bm25 numbers are comparable across the two paths. */
typedef struct {
  uint64_t scored_docs;
  uint64_t candidates_examined;
  uint64_t postings_advanced;
} slg_stats;

/* One query = the folded term list search_segment builds (api/reader.rs:2971-3000):
 * distinct terms, weight = summed boost, term i is ScorePlan leaf i. */
typedef struct {
  uint32_t n_terms;
  const uint32_t *term_ids; /* [n_terms * n_segs]: entry [i*n_segs + s] = id of term i in
                               segment s's dictionary, or SLG_NO_TERM */
  const float *weights;     /* [n_terms] */
} slg_query;

/* ---- lifecycle ------------------------------------------------------------------- */

uint32_t slg_abi_version(void);

/* Thread-local description of the last failure on this thread ("" if none), and its code
 * (SLG_OK if none): functions that return a handle report the reason here. */
const char *slg_last_error(void);
int slg_last_error_code(void);

/* Number of visible HIP devices, or negative error. */
int slg_device_count(void);

/*
 * Stage segments into HBM on `device` and precompute per-posting BM25 impacts
 * (query/bm25.rs:1-6 + query/wand.rs:269-286 with weight factored out).  Returns NULL on
 * error (see slg_last_error).  The handle may be shared by host threads: batch planning
 * takes no lock, launches are serialized, waits happen outside the lock.
 */
slg_index *slg_index_create(const slg_segment_desc *segs, uint32_t n_segs, int device);
/* Destroys the index.  Batches prepared on it that are still alive are detached first: their
 * device buffers are freed, every later call on them fails with SLG_ERR_INVALID, and
 * slg_batch_destroy on them stays valid (so either destruction order is safe). */
void slg_index_destroy(slg_index *index);

/*
 * Tuning knobs of the batch planner, fixed per index at creation (they never change results,
 * only how the work is cut).  slg_tuning_default() fills the defaults and then applies the
 * SLG_* environment overrides named below: it is the ONLY place the library reads the
 * environment.  slg_index_create(segs, n, dev) == slg_index_create_tuned(segs, n, dev, NULL),
 * NULL meaning slg_tuning_default().
 */
typedef struct {
  uint32_t struct_size;          /* sizeof(slg_tuning) */
  int32_t validate;              /* SLG_VALIDATE (1): check every posting at staging; 0: only the
                                    last doc id of each list (ids < n_docs is always enforced) */
  int32_t champions;             /* !SLG_NO_CHAMPIONS (1): per-term champion table = threshold seed */
  int32_t allow_any_arch;        /* SLG_ALLOW_ANY_ARCH (0) */
  int32_t pruning;               /* SLG_MAXSCORE (-1): MaxScore classification, strategies Wand/Bmw.  -1 auto =
                                    batches with a query of >= 5 terms are classified and keep it if block
                                    skipping is expected to leave >= 15 % of the postings unread (else the
                                    batch runs unclassified on the few-term kernel); 0 off; 1 on */
  uint32_t uniform_max_terms;    /* SLG_UNIFORM_MAX_TERMS (8): lists the few-term kernel takes, <= 8 */
  uint32_t uniform_round_target; /* SLG_UNIFORM_ROUND_TARGET (0 = auto): postings per round */
  uint32_t multi_round_target;   /* SLG_MULTI_ROUND_TARGET (448) */
  uint32_t probe_target;         /* SLG_PROBE_TARGET (2048): postings per round incl. probed lists */
  uint32_t rounds_per_slice;     /* SLG_ROUNDS_PER_SLICE (0 = auto) */
  uint32_t max_rounds_per_slice; /* SLG_MAX_ROUNDS_PER_SLICE (0 = auto: 8 few-term kernel, 16 many-term) */
  uint32_t slices_per_subquery;  /* SLG_SLICES_PER_SUBQUERY (16) */
  int32_t cand_mode;             /* !SLG_NO_CAND_MODE (1): 256 < k <= 1024 via candidates + select */
  int32_t slice_order;           /* !SLG_NO_SLICE_ORDER (1): longest slices launch first */
  int32_t block_max;             /* !SLG_NO_BLOCK_MAX (1): block skipping — 64-posting blocks of
                                    pruning-classified lists whose doc range holds no candidate doc
                                    are not loaded (query/wand.rs:205-265) */
  uint32_t pool_cap_mb;          /* SLG_POOL_CAP_MB (0 = auto): MiB of freed batch work buffers the
                                    index keeps for reuse; auto = a quarter of the HBM free after
                                    staging, within [1 GiB, 24 GiB].  The pool is drained whenever a
                                    device allocation of the library fails (slg_index_trim_pool) */
  uint32_t uniform_kernel;       /* SLG_UNIFORM_KERNEL (4): form of the few-term scoring kernel: 4 = blocked
                                    layout, <= 8 lists; 3 = one list per 64-lane slot, <= 8 lists; 2 = the
                                    round-2 kernel, <= 4 lists (both kept for A/B timing on one device) */
  uint32_t uniform_sigma_x100;   /* SLG_UNIFORM_SIGMA (0 = 160): the few-term planner keeps a round's
                                    expected lanes (slots) + this many hundredths of a sigma under 64.3 (8.3) */
  int32_t inline_cuts;           /* SLG_INLINE_CUTS (-1 = auto: on): the blocked few-term kernel cuts the lists at
                                    its slice's round boundaries itself instead of reading cut points that
                                    partition_rounds_kernel wrote for the whole batch; 0 off; 1 on */
  int32_t updatable;             /* !SLG_NOT_UPDATABLE (1): keep term frequencies and doc lengths resident
                                    (4 B per posting + 4 B per doc and field) so that slg_index_update_deleted
                                    can re-derive a segment's impacts on the device when live_docs changes;
                                    0: that call fails with SLG_ERR_UNSUPPORTED (add / remove segment still work) */
  int32_t uniform_plans;         /* !SLG_NO_UNIFORM_PLANS (1): batches with flat score plans (Sum / DisMax over leaves
                                    of one or more terms) and <= uniform_max_terms lists per sub-query run on the
                                    few-term kernel's plan instantiation; 0: on the many-term kernel, as two-level
                                    plans do (A/B timing) */
  uint32_t score_waves_per_simd; /* SLG_SCORE_WAVES (0): 0 = the few-term kernel launches one wave per slice, longest
                                    slices first (the hardware dispatcher hands out the work); n > 0 = persistent
                                    waves: n_CU x 4 x min(n, what registers and LDS allow) waves that pull slices
                                    from 64 work queues.  Measured slower on MI355X: kept for
                                    A/B timing */
} slg_tuning;
void slg_tuning_default(slg_tuning *out);
slg_index *slg_index_create_tuned(const slg_segment_desc *segs, uint32_t n_segs, int device,
                                  const slg_tuning *tuning_or_null);
int slg_index_get_tuning(const slg_index *index, slg_tuning *out);

/* Gives the freed batch work buffers the index keeps for reuse back to the runtime (the library
 * does the same by itself when one of its device allocations runs out of memory).  Call it before
 * another consumer of the device (a second index, RCCL, the application) needs the memory. */
int slg_index_trim_pool(slg_index *index, uint64_t *freed_bytes_or_null);

/* Bytes of HBM held by the index; total postings; segments. */
int slg_index_info(const slg_index *index, uint32_t *n_segs, uint64_t *n_postings,
                   uint64_t *device_bytes);

/* ---- index updates (the reference's commit: api/writer.rs:106-240) -------------------------------
 * searchlite opens a fresh IndexReader per request (searchlite-http/src/lib.rs:640-643 ->
 * index/mod.rs:98-100 -> api/reader.rs:1887-1913), so a staged index must outlive readers and follow
 * the manifest: a commit merges new tombstones into segments' deleted_docs (api/writer.rs:150-158:
 * the set only grows), appends at most one new segment (:160-192, at the END of manifest.segments, so
 * existing segment ordinals keep their meaning) and compaction replaces segments (index/mod.rs:102+).
 *
 * The index is a sequence of immutable states.  Every call below builds the next state (unchanged
 * segments are shared, not copied), publishes it atomically and bumps the generation.  A batch is
 * bound to the state that was current when it was PREPARED: batches prepared or in flight during an
 * update run to completion against their own state — same segments, same tombstones, same filters —
 * and the device memory of a retired state is released when its last batch is destroyed.  Updates are
 * serialised among themselves and never wait for running batches.
 *
 * The caller builds q_term_ids rows for the segment count it knows: it must not race
 * slg_index_add_segment / _remove_segment with building query arrays for the same index (the Rust
 * shim holds these calls under the index's write lock; slg_index_generation tells a reader whether
 * the staged index still matches its manifest snapshot). */

/* New tombstones for segment `seg`: `deleted` is the segment's COMPLETE bitmap (bit d&7 of byte d>>3;
 * a superset of the previous one — the reference never resurrects a doc), `live_docs` =
 * seg.live_docs() = doc_count - |deleted| (index/segment.rs:1365-1370).  live_docs is the `docs` of
 * every ScoredTerm (api/reader.rs:2985), so every idf — and with it every posting's impact, every
 * champion bound and threshold seed — changes: they are re-derived on the device from the resident
 * term frequencies and doc lengths by the same kernels as at creation, i.e. bit-identical to a fresh
 * slg_index_create on the updated descriptor.  Registered filters follow (their reject bitmaps of
 * this segment take the new tombstones).  Needs slg_tuning.updatable. */
int slg_index_update_deleted(slg_index *index, uint32_t seg, const uint8_t *deleted, float live_docs);
/* Stage one more segment; it takes the next ordinal (returned, >= 0; negative error code otherwise).
 * Filters registered before the call have no bitmap for it: a batch that names one of them fails with
 * SLG_ERR_INVALID until the filter is removed and registered again.  Extra vector fields
 * (slg_index_add_vector_field) hold no vectors for the new segment. */
int slg_index_add_segment(slg_index *index, const slg_segment_desc *seg);
/* Drop segment `seg` (compaction, index/mod.rs:102+); the ordinals above it move down by one. */
int slg_index_remove_segment(slg_index *index, uint32_t seg);
/* The HIP device the index lives on (>= 0), or a negative error code. */
int slg_index_device(const slg_index *index);
/* Number of updates applied since creation (0 for a fresh index). */
uint64_t slg_index_generation(const slg_index *index);

/* Use an external HIP stream (hipStream_t) for all work of this index, e.g. the current
 * PyTorch stream so RCCL collectives order after the kernels.  NULL is a valid handle (the
 * HIP null stream, which is what PyTorch's default stream is); SLG_OWN_STREAM restores the
 * index's own non-blocking stream. */
#define SLG_OWN_STREAM ((void *)(intptr_t)-1)
int slg_index_set_stream(slg_index *index, void *hip_stream);

/* ---- doc filters ----------------------------------------------------------
 * The reference's accept() is `!deleted && matcher && filter && cursor`
 * (api/reader.rs:3009-3036); for a pure disjunction the matcher is implied and a filter
 * (query/filters.rs: keyword equality / numeric range on fast fields) reduces to a doc bitmap.
 * A filter is registered once per index and referenced by id from any number of queries.
 * All return a filter id >= 0, or a negative error code. */

/* seg_bitmaps[s]: bit d of byte d/8 set = doc d of segment s passes; NULL = all pass. */
int slg_index_add_filter(slg_index *index, const uint8_t *const *seg_bitmaps);
/* Pre-pass on the device: doc d passes iff lo <= column[d] <= hi (one fast-field column per
 * segment, n_docs values; a NaN never passes). */
int slg_index_add_filter_range_i64(slg_index *index, const int64_t *const *seg_columns, int64_t lo,
                                   int64_t hi);
int slg_index_add_filter_range_f64(slg_index *index, const double *const *seg_columns, double lo,
                                   double hi);
/* A filter from posting lists that are already on the device: the docs that hold NONE of the given terms
 * (pass_if_absent != 0: the query-string matcher's not-terms, api/reader.rs:1499-1503 — a doc in any
 * not-term group never matches) or at least one of them (pass_if_absent == 0).  term_ids: n_terms rows of
 * one id per segment (SLG_NO_TERM: the segment does not have the term), as in slg_query.  and_bitmaps_or_null:
 * pass bitmaps as in slg_index_add_filter, AND-ed with the above (the request's own filter), or NULL.
 * Returns the filter id (>= 0) or an error code; use it like any other filter id. */
int slg_index_add_filter_terms(slg_index *index, const uint32_t *term_ids, uint32_t n_terms, int pass_if_absent,
                               const uint8_t *const *and_bitmaps_or_null);
/* Unregisters the filter.  Batches already prepared with it keep their bitmaps (they belong to the
 * batch's index state) and may still run; the id may be handed out again by a later add. */
int slg_index_remove_filter(slg_index *index, int filter_id);

/* ---- one-shot search (what a searchlite `gpu` shim calls) -------------------------- */

/*
 * Replaces, for a batch of eligible queries, the per-segment scorer call plus the
 * cross-segment sort: for every query, top-k by (score desc [f32 total_cmp],
 * segment_ord asc, doc_id asc) over all segments of the index.
 * Outputs are caller-owned host arrays of nq*k (out_count: nq); row q holds
 * out_count[q] <= k hits.  Blocks until the results are in the output arrays.
 */
int slg_search_batch(slg_index *index, const slg_query *queries, uint32_t nq, uint32_t k,
                     int strategy, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                     uint32_t *out_count, slg_stats *stats_or_null);

/* Same with a doc filter per query (q_filter[q] = filter id, < 0 none; NULL = none at all). */
int slg_search_batch_filtered(slg_index *index, const slg_query *queries, uint32_t nq,
                              const int32_t *q_filter, uint32_t k, int strategy, uint32_t *out_doc,
                              uint32_t *out_seg, float *out_score, uint32_t *out_count,
                              slg_stats *stats_or_null);

/* ---- prepared batches (device-resident inputs; used for steady-state serving) ------ */

/*
 * Queries in CSR form: q_offsets[nq+1] indexes q_weights and the rows of q_term_ids
 * ([total_terms * n_segs], same layout as slg_query.term_ids).  Plans the batch on the
 * host, uploads the descriptors and allocates all device work buffers.
 */
slg_batch *slg_batch_prepare(slg_index *index, uint32_t nq, const uint32_t *q_offsets,
                             const uint32_t *q_term_ids, const float *q_weights, uint32_t k,
                             int strategy);
/* Same, with a doc filter per query: q_filter[q] = filter id, or < 0 for none (q_filter may be
 * NULL).  Filtered queries get no threshold seed (the filter may reject the champions). */
slg_batch *slg_batch_prepare_filtered(slg_index *index, uint32_t nq, const uint32_t *q_offsets,
                                      const uint32_t *q_term_ids, const float *q_weights,
                                      const int32_t *q_filter, uint32_t k, int strategy);
/* Same with a score plan per query (query/planner.rs:113-153).  q_tie[q] must lie in
 * [0, 1] (validate_tie_breaker, query/planner.rs:850-856) and leaves must be < 2^31.  The reference adds
 * every scored term's contribution to a ScorePlan leaf (wand.rs:488-497 `buf[term.leaf] +=`) and
 * combines the leaves: a multi-field query string maps all fields of a word to one leaf and sums
 * the leaves; multi_match best_fields / dis_max take DisMax over the leaves.
 *   q_leaf[i]   leaf of query term i (same indexing as q_weights); NULL: term i of a query is
 *               leaf i (the plain disjunction)
 *   q_plan[q]   SLG_PLAN_SUM or SLG_PLAN_DISMAX over the leaves; NULL: SUM
 *   q_tie[q]    DisMax tie breaker (max + tie * (sum - max)); NULL: 0
 *   q_nleaves[q] leaves of the plan (>= max leaf + 1; leaves without a term count as 0.0 in a
 *               DisMax); NULL: max leaf + 1
 * Results are bit-identical to the reference's exhaustive scorer (per-leaf sums in term order,
 * leaves combined in leaf order). */
#define SLG_PLAN_SUM 0
#define SLG_PLAN_DISMAX 1
slg_batch *slg_batch_prepare_plan(slg_index *index, uint32_t nq, const uint32_t *q_offsets,
                                  const uint32_t *q_term_ids, const float *q_weights,
                                  const uint32_t *q_leaf, const int32_t *q_plan, const float *q_tie,
                                  const uint32_t *q_nleaves, const int32_t *q_filter, uint32_t k,
                                  int strategy);

/* Two-level score plans.  ScoreExpr::evaluate is recursive (query/planner.rs:122-153) and real
 * requests build two-level trees: `dis_max{queries}` = a DisMax of sub-scorers
 * (planner.rs:470-487), `bool{should:[multi_match ...]}` = a Sum of DisMax groups
 * (planner.rs:670-690).  Here the ROOT (q_plan / q_tie) combines GROUPS, a group (group_plan /
 * group_tie) combines LEAVES, a leaf sums the scored terms that name it (wand.rs:488-497).
 * Leaves are numbered in the plan's traversal order, so a group's leaves are consecutive:
 * leaf_group is non-decreasing within a query and names every group 0 .. n_groups-1.  A leaf that
 * hangs off the root directly is a SLG_PLAN_SUM group of one leaf (Sum of one child is the child,
 * bit for bit).  With leaf_group == NULL this is slg_batch_prepare_plan.  Every DisMax counts all
 * of its children, the ones without a posting for a doc as 0.0, as the reference does. */
typedef struct {
  const uint32_t *q_leaf;           /* [total terms] leaf of every query term; NULL: term i = leaf i */
  const int32_t *q_plan;            /* [nq] root: SLG_PLAN_SUM | SLG_PLAN_DISMAX; NULL: Sum */
  const float *q_tie;               /* [nq] root tie breaker in [0, 1]; NULL: 0 */
  const uint32_t *q_nleaves;        /* [nq] leaves of the plan; NULL: 1 + the largest leaf named */
  const uint32_t *q_leaf_offsets;   /* [nq + 1] into leaf_group (two-level plans only) */
  const uint32_t *leaf_group;       /* group of every leaf of every query; NULL: flat plans */
  const uint32_t *q_group_offsets;  /* [nq + 1] into group_plan / group_tie */
  const int32_t *group_plan;        /* SLG_PLAN_SUM | SLG_PLAN_DISMAX per group */
  const float *group_tie;           /* tie breaker per group, in [0, 1] */
  /* Trees of any shape (ScoreExpr is recursive, query/planner.rs:113-153), up to SLG_MAX_PLAN_DEPTH levels
   * of Sum / DisMax nodes above the leaves: per query a node array in PRE-ORDER (node 0 = the root,
   * node_parent[i] < i, node_parent[0] ignored), node_kind = SLG_PLAN_SUM | SLG_PLAN_DISMAX | SLG_PLAN_LEAF,
   * node_tie in [0, 1] for DisMax nodes; the i-th LEAF node of a query in pre-order is ScorePlan leaf i
   * (q_leaf names it per query term).  Every Sum / DisMax node has at least one child.  With
   * q_node_offsets != NULL the root / group arrays above are not read (q_leaf still is).  Trees of one
   * or two levels run exactly as the forms above; deeper ones on the many-term kernel's tree mode. */
  const uint32_t *q_node_offsets;   /* [nq + 1] into node_kind / node_tie / node_parent; NULL: the forms above */
  const int32_t *node_kind;
  const float *node_tie;
  const uint32_t *node_parent;
  /* minimum_should_match of a query string (api/reader.rs:1509-1517: a doc matches if at least that many of
   * the matcher's term groups hold it; a term group = a ScorePlan leaf, i.e. one query word over its
   * fields).  [nq] or NULL; 0 and 1 = any doc of any list.  Batches with a value > 1 are accepted for flat
   * plans (one level of Sum / DisMax over the leaves) of at most 8 scored lists per segment — what the
   * few-term kernel's plan instantiation runs; other shapes: SLG_ERR_UNSUPPORTED (CPU scorer). */
  const uint32_t *q_min_match;
} slg_score_plans;
#define SLG_PLAN_LEAF 2
#define SLG_MAX_PLAN_DEPTH 4u
slg_batch *slg_batch_prepare_plans(slg_index *index, uint32_t nq, const uint32_t *q_offsets,
                                   const uint32_t *q_term_ids, const float *q_weights,
                                   const slg_score_plans *plans_or_null, const int32_t *q_filter_or_null,
                                   uint32_t k, int strategy);
/* Enqueue the partition / score / merge kernels on the batch's stream (asynchronous). */
int slg_batch_run(slg_batch *batch);
/* Run this batch on its own HIP stream instead of the index stream, so several prepared
 * batches can be in flight at once (their partition / merge kernels then overlap the other
 * batches' scoring).  A batch owns all its work buffers; batches never share state.
 * SLG_OWN_STREAM returns the batch to the index stream. */
int slg_batch_set_stream(slg_batch *batch, void *hip_stream);
/* Wait for everything enqueued for this batch. */
int slg_batch_sync(slg_batch *batch);
/* Copy results to host arrays (nq*k, nq); waits for completion. */
int slg_batch_fetch(slg_batch *batch, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                    uint32_t *out_count, slg_stats *stats_or_null);
/* Device pointers of the result arrays (u32[nq*k], u32[nq*k], f32[nq*k], u32[nq]) for
 * device-side consumers (RCCL all-gather of per-shard top-k, rerank). */
int slg_batch_device_results(slg_batch *batch, void **d_doc, void **d_seg, void **d_score,
                             void **d_count);
/* The four result arrays live back to back in ONE device allocation, in the order
 * doc[nq*k] | seg[nq*k] | score[nq*k] | count[nq] (4-byte elements, so (3*k+1)*nq*4 bytes):
 * a multi-GPU caller exchanges the per-shard top-k with a single all-gather of this block. */
int slg_batch_device_result_block(slg_batch *batch, void **d_block, uint64_t *n_bytes);
/* Planning facts: total postings the batch scores, number of work slices, and the
 * algorithmic byte count 12*postings + 8*k*nq. */
int slg_batch_info(const slg_batch *batch, uint64_t *n_postings, uint32_t *n_slices,
                   uint64_t *algorithmic_bytes);
/* Block skipping (query/wand.rs:205-265), last run of the batch: postings of the batch's
 * pruning-classified (non-essential) lists, and how many of them were never loaded — their
 * 64-posting block, or their whole round, held no candidate doc.  Blocks are tested only in lists
 * much denser than the query's essential lists.  Both 0 when the batch has no classified list or
 * slg_tuning.block_max is off.  Waits for the batch. */
int slg_batch_skip_counts(slg_batch *batch, uint64_t *probed_postings, uint64_t *skipped_postings);
void slg_batch_destroy(slg_batch *batch);

/* ---- request coalescer ------------------------------------------------------------------------------
 * searchlite has no batch API: IndexReader::search takes one request (api/reader.rs:2539) and the HTTP
 * server gives every request its own blocking thread (searchlite-http/src/lib.rs:628-652).  The
 * coalescer turns concurrent single-query callers into batches: slg_coalescer_search blocks its caller
 * thread, the query joins the batch that is collecting (same k, strategy and segment count), the
 * coalescer's two dispatcher threads plan / run / fetch the batch on a HIP stream of its own while the next
 * batch already collects, and every caller returns with its own row — bit-identical to the same query in
 * slg_search_batch.  A batch closes when it holds max_batch queries, or max_wait_us after its first
 * query arrived; a query that finds the coalescer idle (nothing in flight) does not wait at all.
 * Thread-safe; out_doc / out_seg / out_score hold k entries, out_count one. */
typedef struct slg_coalescer slg_coalescer;
slg_coalescer *slg_coalescer_create(slg_index *index, uint32_t max_batch, uint32_t max_wait_us);
/* No caller may be inside slg_coalescer_search any more. */
void slg_coalescer_destroy(slg_coalescer *coalescer);
int slg_coalescer_search(slg_coalescer *coalescer, const slg_query *query, uint32_t k, int strategy,
                         uint32_t *out_doc, uint32_t *out_seg, float *out_score, uint32_t *out_count,
                         slg_stats *stats_or_null);
/* The same for a query with a flat score plan (slg_batch_prepare_plan: leaf[i] = leaf of query term i or
 * NULL, plan = SLG_PLAN_SUM | SLG_PLAN_DISMAX over the leaves, tie in [0, 1], n_leaves = leaves of the
 * plan or 0 = 1 + the largest leaf named) and / or a registered doc filter (filter_id, < 0: none): what
 * an unmodified request over the default fields is (api/reader.rs:2576-2586).  Rows with and without
 * plans or filters share batches.  Two-level plans go through slg_batch_prepare_plans directly. */
int slg_coalescer_search_plan(slg_coalescer *coalescer, const slg_query *query, const uint32_t *leaf, int plan,
                              float tie, uint32_t n_leaves, int32_t filter_id, uint32_t k, int strategy,
                              uint32_t *out_doc, uint32_t *out_seg, float *out_score, uint32_t *out_count,
                              slg_stats *stats_or_null);
/* The two halves of slg_coalescer_search_plan, for callers that keep SEVERAL requests in flight per thread
 * (an async server task, a client that pipelines): slg_coalescer_submit puts the query into the collecting
 * batch and returns at once with a ticket; slg_coalescer_wait blocks until that batch's results are in,
 * copies the ticket's row out and gives the row back (every ticket must be waited for exactly once, by any
 * thread; the ticket is cleared).  slg_coalescer_poll: 1 if slg_coalescer_wait would not block, else 0.
 * The query's arrays may be reused as soon as submit returns.  A thread-per-request caller pays a sleep and
 * a wake-up per query — what bounds slg_coalescer_search on a host with few cores; with D tickets per
 * thread a thread sleeps at most once per D queries. */
typedef struct slg_ticket {
  void *batch;   /* opaque; NULL once waited for */
  uint32_t row, k, kind;
} slg_ticket;
int slg_coalescer_submit(slg_coalescer *coalescer, const slg_query *query, const uint32_t *leaf, int plan, float tie,
                         uint32_t n_leaves, int32_t filter_id, uint32_t k, int strategy, int want_stats,
                         slg_ticket *ticket);
int slg_coalescer_poll(const slg_coalescer *coalescer, const slg_ticket *ticket);
int slg_coalescer_wait(slg_coalescer *coalescer, slg_ticket *ticket, uint32_t *out_doc, uint32_t *out_seg,
                       float *out_score, uint32_t *out_count, slg_stats *stats_or_null);
/* Thread-local text of the last failure of slg_coalescer_search on this thread. */
const char *slg_coalescer_last_error(void);
/* Mean time (ms) a batch's leader spent collecting / in slg_batch_prepare / in set_stream + run / in
 * fetch + destroy, over the batches run so far. */
int slg_coalescer_phase_ms(const slg_coalescer *coalescer, double *collect, double *prepare, double *run,
                           double *fetch);
/* Batches run and queries served so far (their ratio = the mean batch size reached). */
int slg_coalescer_stats(const slg_coalescer *coalescer, uint64_t *n_batches, uint64_t *n_queries);

/* ---- index sharding over RCCL ------------------------------------------------------
 * The reference scores every segment independently and merges by (score desc, segment_ord asc,
 * doc asc) (api/reader.rs:2670-2778, query/sort.rs:80-93); with one shard of segments per GPU the
 * merge spans ranks.  A shard group ties this rank's index to an RCCL communicator: one process
 * (or host thread) per GPU creates its index over ITS segments and joins the group; rank 0 makes
 * the 128-byte id with slg_shard_unique_id and hands it to the other ranks out of band (the way
 * ncclGetUniqueId / ncclCommInitRank are used).  segs_per_rank = the largest shard's segment count:
 * a hit's segment ordinal in the merged result is rank * segs_per_rank + its local ordinal.
 * librccl is bound at run time; without it these calls fail with SLG_ERR_UNSUPPORTED. */
#define SLG_SHARD_UNIQUE_ID_BYTES 128u
typedef struct slg_shard_group slg_shard_group;
int slg_shard_unique_id(void *out, size_t out_bytes);
/* Collective: every rank of the group calls it (it returns when all have). */
slg_shard_group *slg_shard_group_create(slg_index *index, int rank, int world, const void *unique_id,
                                        uint32_t segs_per_rank);
void slg_shard_group_destroy(slg_shard_group *group);
/* slg_batch_run on this rank's segments, ONE ncclAllGather of the contiguous result blocks
 * ((3k+1) * Q * 4 bytes per rank) on the batch's stream, merge of the world's rows on the device.
 * Every rank prepares the SAME queries (same order, same k) and issues its sharded runs in the same
 * order.  out_* (host, [nq*k] / [nq]) receive the merged top-k on every rank; pass NULL for all four
 * to leave the result on the device (slg_batch_sharded_device_results) without waiting. */
int slg_batch_run_sharded(slg_batch *batch, slg_shard_group *group, uint32_t *out_doc, uint32_t *out_seg,
                          float *out_score, uint32_t *out_count);
/* Collectives on one communicator must be issued in the same order on every rank.  The group issues
 * them on a stream of its own, one at a time: slg_batch_run_sharded takes its turn in CALL order — so
 * every rank must call it for its batches in the same order, which one issuing thread per rank
 * guarantees (several batches may still be in flight, each on its own stream).  With several caller
 * threads per rank use the _seq form: `seq` numbers the sharded runs of the group 0, 1, 2, ... without
 * gaps, the SAME number for the same query batch on every rank; run `seq` issues its all-gather when
 * runs 0 .. seq-1 of this rank have issued theirs (a call may block until then; a failed run still
 * passes its turn on).  Do not mix the two forms on one group. */
int slg_batch_run_sharded_seq(slg_batch *batch, slg_shard_group *group, uint64_t seq, uint32_t *out_doc,
                              uint32_t *out_seg, float *out_score, uint32_t *out_count);
/* Passes the turn of run `seq` on without a collective: for a rank that could not prepare the batch of
 * that number (the runs behind it would wait for ever).  The other ranks must skip the same number —
 * an all-gather that one rank never joins does not complete. */
int slg_shard_group_skip_seq(slg_shard_group *group, uint64_t seq);
/* With slg_profile_enable on: device time (ms, summed) of the sharded runs FETCHED since the last call —
 * this rank's kernels, the all-gather (incl. waiting for the slowest rank), the merge — and their number.
 * Resets the sums. */
int slg_shard_group_stats(slg_shard_group *group, double *ms_kernels, double *ms_gather, double *ms_merge,
                          uint64_t *n_runs);
int slg_batch_sharded_device_results(slg_batch *batch, void **d_doc, void **d_seg, void **d_score,
                                     void **d_count);
/* Waits for the batch's sharded run and copies the merged top-k to host arrays (what
 * slg_batch_run_sharded does itself when given output arrays): lets a caller keep several sharded
 * batches in flight, each on its own stream, and collect them later. */
int slg_batch_fetch_sharded(slg_batch *batch, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                            uint32_t *out_count);

/*
 * Merge per-shard results gathered from several indexes/GPUs (device arrays, as produced
 * by slg_batch_device_results and concatenated shard-major: [n_shards][nq*k]) into the
 * global top-k by (score desc, shard_ord asc, segment asc, doc asc).  out_seg receives
 * shard_ord * seg_stride + seg.  All pointers are device pointers on `index`'s device.
 * k up to SLG_MAX_K, as the scorer (an index-sharded request with limit = 20 000 merges too).
 */
int slg_merge_shards_device(slg_index *index, uint32_t n_shards, uint32_t nq, uint32_t k,
                            const uint32_t *d_doc, const uint32_t *d_seg, const float *d_score,
                            const uint32_t *d_count, uint32_t seg_stride, uint32_t *d_out_doc,
                            uint32_t *d_out_seg, float *d_out_score, uint32_t *d_out_count);

/* ---- profiling hooks (bench.py roofline) -------------------------------------------- */

/* When enabled, every slg_batch_run brackets its scoring kernel with HIP events on the
 * launch stream. */
int slg_profile_enable(slg_index *index, int on);
/* Sum of scoring-kernel durations and number of launches since the last reset; waits
 * for the recorded events.  Resets the accumulators. */
int slg_profile_read(slg_index *index, uint32_t *n_launches, float *total_ms);

/* ---- rerank (fills gpu::rerank, gpu/rerank.rs:3) ------------------------------------- */

/*
 * For each query: candidates (cand_doc, cand_seg, cand_bm25)[cand_count[q] <= max_cand] ->
 * vector similarity against qvecs[q] (vectors/mod.rs:107-120: cosine = dot of
 * pre-normalized vectors, NaN -> 0; L2 = -sqrt(sum d^2)), blended as
 * compute_hybrid_score does for one clause (api/reader.rs:225-254; missing vector =>
 * -1.0 / f32::MIN, api/reader.rs:217-223), then top-k_out by (blended desc, seg asc, doc asc).
 * Host arrays in and out; blocks.
 */
int slg_rerank_batch(slg_index *index, uint32_t nq, const float *qvecs, const float *alpha,
                     const uint32_t *cand_doc, const uint32_t *cand_seg, const float *cand_bm25,
                     const uint32_t *cand_count, uint32_t max_cand, uint32_t k_out,
                     uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                     float *out_vec_score, uint32_t *out_count);

/* Same, all pointers device-resident and asynchronous on the index stream (chains after
 * slg_batch_run without a host round trip). */
int slg_rerank_batch_device(slg_index *index, uint32_t nq, const float *d_qvecs,
                            const float *d_alpha, const uint32_t *d_cand_doc,
                            const uint32_t *d_cand_seg, const float *d_cand_bm25,
                            const uint32_t *d_cand_count, uint32_t max_cand, uint32_t k_out,
                            uint32_t *d_out_doc, uint32_t *d_out_seg, float *d_out_score,
                            float *d_out_vec_score, uint32_t *d_out_count);

/* Rerank a batch's OWN device results (k candidates per query, as slg_batch_run left them) with
 * n_clauses vector clauses, asynchronously on the batch's stream: BM25 top-k -> rerank chains without a
 * host round trip and without leaving the batch's stream, so several such pipelines can be in flight
 * (slg_batch_set_stream).  Arguments as slg_rerank_multi_batch_device; n_clauses == 1 with d_boost ==
 * NULL takes the single-clause kernel. */
int slg_batch_rerank_device(slg_batch *batch, uint32_t n_clauses, const float *d_qvecs, const float *d_alpha,
                            const float *d_boost, uint32_t k_out, uint32_t *d_out_doc, uint32_t *d_out_seg,
                            float *d_out_score, float *d_out_vec_score, uint32_t *d_out_count);

/*
 * Hybrid rerank with several vector clauses over one candidate set (api/reader.rs:225-254;
 * n_clauses <= SLG_MAX_VECTOR_CLAUSES): qvecs[q][c][dim], alpha[q][c], boost[q][c] (NULL: 1.0; the
 * clause's similarity is multiplied by it, api/reader.rs:2421).  Blended score = mean over the
 * clauses of blend(alpha_c, bm25, vec_c); a candidate without a vector counts as -1.0 / f32::MIN in
 * every clause (:217-223); out_vec_score = sum of the clause similarities (:236-238).  All clauses
 * use the index's one vector field.  The [candidates x clauses] cosine products run on the f32
 * matrix cores (v_mfma_f32_16x16x4_f32).  n_clauses * (dim + 4 + max_cand) + 2 * max_cand floats
 * must fit the kernel's LDS budget (36 Ki floats), else SLG_ERR_UNSUPPORTED.
 */
/*
 * Vector fields beyond the one in slg_segment_desc (vectors/mod.rs:10-17: one VectorStore per
 * vector field).  Field 0 is the store of the segment descriptors; every call stages one more
 * field — one descriptor per segment of the index, vec_dim 0 where the segment has no vectors in
 * it — and returns its id (>= 1), or a negative error code.
 */
typedef struct {
  uint32_t vec_dim;            /* 0 => this segment has no vectors in the field */
  int32_t vec_metric;          /* SLG_METRIC_* */
  const uint32_t *vec_offsets; /* [n_docs] row index or SLG_NO_VECTOR */
  const float *vec_values;     /* [vec_rows * vec_dim] row-major */
  uint32_t vec_rows;
} slg_vector_field_desc;
int slg_index_add_vector_field(slg_index *index, const slg_vector_field_desc *per_segment, uint32_t n_segs);

/*
 * Hybrid rerank whose clauses name different vector fields (api/reader.rs:225-254: every clause has
 * its own field, metric and dimension).  clause_field[c] = field id of clause c (host array);
 * qvecs[q] = the clause vectors of query q one after another (sum of the clause dimensions
 * floats); alpha / boost [nq][n_clauses] as in slg_rerank_multi_batch.  A candidate without a
 * vector in clause c's field takes that clause's missing-vector score (-1.0 / f32::MIN by the
 * clause's metric); out_vec_score sums the clauses that found one.
 */
int slg_rerank_fields_batch(slg_index *index, uint32_t nq, uint32_t n_clauses, const uint32_t *clause_field,
                            const float *qvecs, const float *alpha, const float *boost,
                            const uint32_t *cand_doc, const uint32_t *cand_seg, const float *cand_bm25,
                            const uint32_t *cand_count, uint32_t max_cand, uint32_t k_out, uint32_t *out_doc,
                            uint32_t *out_seg, float *out_score, float *out_vec_score, uint32_t *out_count);
int slg_rerank_fields_batch_device(slg_index *index, uint32_t nq, uint32_t n_clauses,
                                   const uint32_t *clause_field, const float *d_qvecs, const float *d_alpha,
                                   const float *d_boost, const uint32_t *d_cand_doc,
                                   const uint32_t *d_cand_seg, const float *d_cand_bm25,
                                   const uint32_t *d_cand_count, uint32_t max_cand, uint32_t k_out,
                                   uint32_t *d_out_doc, uint32_t *d_out_seg, float *d_out_score,
                                   float *d_out_vec_score, uint32_t *d_out_count);

int slg_rerank_multi_batch(slg_index *index, uint32_t nq, uint32_t n_clauses, const float *qvecs,
                           const float *alpha, const float *boost, const uint32_t *cand_doc,
                           const uint32_t *cand_seg, const float *cand_bm25, const uint32_t *cand_count,
                           uint32_t max_cand, uint32_t k_out, uint32_t *out_doc, uint32_t *out_seg,
                           float *out_score, float *out_vec_score, uint32_t *out_count);
int slg_rerank_multi_batch_device(slg_index *index, uint32_t nq, uint32_t n_clauses, const float *d_qvecs,
                                  const float *d_alpha, const float *d_boost, const uint32_t *d_cand_doc,
                                  const uint32_t *d_cand_seg, const float *d_cand_bm25,
                                  const uint32_t *d_cand_count, uint32_t max_cand, uint32_t k_out,
                                  uint32_t *d_out_doc, uint32_t *d_out_seg, float *d_out_score,
                                  float *d_out_vec_score, uint32_t *d_out_count);

#ifdef __cplusplus
}
#endif
#endif /* SEARCHLITE_GPU_H */
