// slg_api.hip — host side of the C ABI declared in include/searchlite_gpu.h.
//
// Mirrors, for the GPU-eligible request shape, what IndexReader::search_segment does
// before and after the scorer call (searchlite-core/src/api/reader.rs:2971-3000 build the
// ScoredTerm list; :3075-3099 call the scorer; :2776-2778 merge across segments), but for
// a whole batch of queries at once.  The product path has NO CPU fallback: every entry
// point either runs the HIP kernels or fails with an error code.
#include "../../include/searchlite_gpu.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "slg_kernels.hpp"
#include "slg_plan.hpp"
#include "slg_rerank.hpp"
#include "slg_score.hpp"
#include "slg_score_uni4.hpp"
#include "slg_score_multi.hpp"

namespace {

thread_local std::string g_last_error;
thread_local int g_last_code = SLG_OK;

using slgplan::SlgError;  // {code, message}; also what the host planner throws

#define SLG_HIP(expr)                                                                     \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      int _code = (_e == hipErrorOutOfMemory) ? SLG_ERR_OOM : SLG_ERR_DEVICE;             \
      throw SlgError(_code, std::string(#expr) + ": " + hipGetErrorString(_e));           \
    }                                                                                     \
  } while (0)

#define SLG_REQUIRE(cond, msg)                              \
  do {                                                      \
    if (!(cond)) throw SlgError(SLG_ERR_INVALID, (msg));    \
  } while (0)

template <typename F>
int guarded(F &&f) {
  try {
    g_last_error.clear();
    g_last_code = SLG_OK;
    f();
    return SLG_OK;
  } catch (const SlgError &e) {
    g_last_error = e.what();
    g_last_code = e.code;
  } catch (const std::bad_alloc &) {
    g_last_error = "host allocation failed";
    g_last_code = SLG_ERR_OOM;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    g_last_code = SLG_ERR_INTERNAL;
  } catch (...) {
    g_last_error = "unknown error";
    g_last_code = SLG_ERR_INTERNAL;
  }
  return g_last_code;
}

// (the environment is read in slg_tuning_default() only)
uint32_t env_u32(const char *name, uint32_t dflt) {
  const char *v = std::getenv(name);
  if (!v || !*v) return dflt;
  return (uint32_t)std::strtoul(v, nullptr, 10);
}
int32_t env_i32(const char *name, int32_t dflt) {
  const char *v = std::getenv(name);
  if (!v || !*v) return dflt;
  return (int32_t)std::strtol(v, nullptr, 10);
}

// Freed batch buffers are kept for the next batch: hipMalloc / hipFree cost ~100 us each and
// hipFree synchronizes the device, which would serialize host threads that serve batches
// concurrently.  Size classes: powers of two from 4 KiB to 1 MiB, above that eight steps per
// octave (<= 12.5 % over-allocation).  The pool is bounded (slg_tuning.pool_cap_mb) and is the
// first thing given back when the device runs out of memory: every allocation of the library that
// fails with hipErrorOutOfMemory drains it and tries once more, so parked blocks of size classes
// nobody asks for any more can never starve a new batch, a second index or the application.
struct BufPool {
  std::mutex mu;
  std::multimap<size_t, void *> free_;
  size_t pooled = 0;
  // (config 4 on one GPU holds ~1 GB of work buffers per 8192-query batch and keeps three batches
  //  alive: with a 4 GB cap every batch ended in hipFree + hipMalloc, which synchronise the device)
  size_t cap = 24ull << 30;
  // pinned staging images of slg_batch_prepare* (descriptor uploads), owned by the index: taken
  // for one prepare call, handed back afterwards, released with the index (a thread_local image
  // would outlive its thread's usefulness and leak when caller threads come and go)
  std::vector<std::pair<void *, size_t>> images;
  static size_t size_class(size_t n) {  // powers of two up to 1 MiB, then eighths of an octave
    size_t c = 4096;
    while (c < n && c < (1u << 20)) c <<= 1;
    if (c >= n) return c;
    while ((c << 1) < n) c <<= 1;  // c <= n < 2c
    const size_t step = c >> 3;
    return c + ((n - c + step - 1) / step) * step;
  }
  // a free block of class cls, or the next larger one within 25 % (its size goes back in *got)
  void *get(size_t cls, size_t *got) {
    std::lock_guard<std::mutex> lk(mu);
    auto it = free_.lower_bound(cls);
    if (it == free_.end() || it->first > cls + cls / 4) return nullptr;
    void *p = it->second;
    *got = it->first;
    pooled -= it->first;
    free_.erase(it);
    return p;
  }
  bool put(void *p, size_t cls) {
    std::lock_guard<std::mutex> lk(mu);
    if (pooled + cls > cap) return false;
    free_.emplace(cls, p);
    pooled += cls;
    return true;
  }
  // give every parked block back to the runtime (largest first); returns the bytes freed
  size_t drain() {
    std::multimap<size_t, void *> take;
    {
      std::lock_guard<std::mutex> lk(mu);
      take.swap(free_);
      pooled = 0;
    }
    size_t freed = 0;
    for (auto it = take.rbegin(); it != take.rend(); ++it) {
      (void)hipFree(it->second);
      freed += it->first;
    }
    return freed;
  }
  // a pinned host image of at least n bytes (hipHostMallocPortable: usable from any device)
  void *take_image(size_t n, size_t *got) {
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < images.size(); i++)
        if (images[i].second >= n) {
          void *p = images[i].first;
          *got = images[i].second;
          images[i] = images.back();
          images.pop_back();
          return p;
        }
      if (images.size() >= 16) {  // only too-small ones are parked: drop one
        (void)hipHostFree(images.back().first);
        images.pop_back();
      }
    }
    const size_t want = std::max<size_t>((n * 5) / 4, 1u << 20);
    void *p = nullptr;
    if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess) return nullptr;
    *got = want;
    return p;
  }
  void give_image(void *p, size_t bytes) {
    std::lock_guard<std::mutex> lk(mu);
    images.emplace_back(p, bytes);
  }
  ~BufPool() {
    for (auto &kv : free_) (void)hipFree(kv.second);
    for (auto &im : images) (void)hipHostFree(im.first);
  }
};

// hipMalloc that drains `pool` (may be null) and tries once more when the device is out of memory
static void *device_alloc(size_t n, BufPool *pool) {
  void *p = nullptr;
  hipError_t e = hipMalloc(&p, n);
  if (e == hipErrorOutOfMemory && pool && pool->drain() > 0) {
    (void)hipGetLastError();
    e = hipMalloc(&p, n);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw SlgError(e == hipErrorOutOfMemory ? SLG_ERR_OOM : SLG_ERR_DEVICE,
                   std::string("hipMalloc(") + std::to_string(n) + "): " + hipGetErrorString(e));
  }
  return p;
}

struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
  BufPool *pool = nullptr;  // set: bytes is a size class and the block goes back to the pool
  // relief: a pool to drain if the device is out of memory (the block itself is not pooled)
  void alloc(size_t n, BufPool *relief = nullptr) {
    release();
    if (n == 0) n = 16;
    p = device_alloc(n, relief);
    bytes = n;
  }
  void alloc_pooled(BufPool *pl, size_t n) {
    release();
    size_t cls = BufPool::size_class(n ? n : 16);
    p = pl->get(cls, &cls);
    if (!p) p = device_alloc(cls, pl);
    bytes = cls;
    pool = pl;
  }
  void release() {
    if (p && !(pool && pool->put(p, bytes))) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    pool = nullptr;
  }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes), pool(o.pool) {
    o.p = nullptr;
    o.bytes = 0;
    o.pool = nullptr;
  }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p;
      bytes = o.bytes;
      pool = o.pool;
      o.p = nullptr;
      o.bytes = 0;
      o.pool = nullptr;
    }
    return *this;
  }
  template <typename T>
  T *as() const {
    return static_cast<T *>(p);
  }
};

// What a staged segment keeps for as long as any version of it lives: the posting arrays in the
// padded device layout, and (slg_tuning.updatable) everything stage_impacts_kernel needs to derive the
// impacts again when live_docs changes (slg_index_update_deleted).  Immutable after staging.
struct PostingStore {
  uint32_t n_docs = 0, n_terms = 0, n_fields = 0;
  uint64_t n_postings = 0;
  uint64_t null_idx = 0;               // SegDev::null_idx
  std::vector<uint64_t> term_offsets;  // as given (unpadded); device position = + kListPad * term
  std::vector<float> avgdl;            // [n_fields]
  float k1 = 0.0f, b = 0.0f;
  bool has_term_field = false;
  bool updatable = false;
  DevBuf d_docs;                                        // padded doc ids
  DevBuf d_offs, d_tfs, d_tfield, d_avgdl, d_lenptrs;   // updatable only (else freed after staging)
  std::vector<DevBuf> d_lens;                           // updatable only: per-field doc lengths
  // vectors (field 0)
  uint32_t vec_dim = 0, vec_rows = 0;
  int32_t vec_metric = 0;
  DevBuf d_vec_offsets, d_vec_values;
  size_t device_bytes() const {
    size_t n = d_docs.bytes + d_offs.bytes + d_tfs.bytes + d_tfield.bytes + d_avgdl.bytes + d_lenptrs.bytes +
               d_vec_offsets.bytes + d_vec_values.bytes;
    for (auto &l : d_lens) n += l.bytes;
    return n;
  }
};

// One VERSION of a staged segment: what depends on live_docs and the tombstones (impacts, champion
// bounds, the bitmap).  Versions of one segment share its PostingStore.
struct SegHost {
  std::shared_ptr<PostingStore> store;
  uint32_t n_docs = 0, n_terms = 0;
  uint64_t n_postings = 0;
  uint64_t null_idx = 0;
  float docs = 0.0f;         // live_docs this version's idf values were computed with
  std::vector<float> champ;  // host mirror of d_champ [V * kChampions] (query planning)
  DevBuf d_imps, d_deleted, d_champ;
  size_t device_bytes() const { return d_imps.bytes + d_deleted.bytes + d_champ.bytes; }
};

// a vector field beyond the one in the segment descriptors (slg_index_add_vector_field): per segment
// shared stores (null: the segment has no vectors in the field)
struct VecSegStore {
  DevBuf offsets, values;
  uint32_t dim = 0;
};
struct VecFieldHost {
  uint32_t dim = 0;
  int32_t metric = 0;
  std::vector<std::shared_ptr<VecSegStore>> per_seg;
  DevBuf d_vsegs;  // slg::VecSegDev[n_segs] of the state this object belongs to
};

// a registered doc filter: per segment a reject bitmap (deleted | ~filter); null = the filter predates
// the segment (slg_index_add_segment) and cannot be used until it is registered again
struct FilterData {
  std::vector<std::shared_ptr<DevBuf>> per_seg;
  bool complete() const {
    for (auto &b : per_seg)
      if (!b) return false;
    return true;
  }
};

// One immutable state of the index (see "index updates" in searchlite_gpu.h).  Batches hold the state
// they were prepared on; the index holds the current one.
struct IndexState {
  uint64_t generation = 0;
  int device = 0;
  std::vector<std::shared_ptr<SegHost>> segs;
  DevBuf d_segs;   // slg::SegDev[n_segs]
  DevBuf d_vsegs;  // slg::VecSegDev[n_segs]
  std::vector<std::shared_ptr<VecFieldHost>> vfields;  // field id f >= 1 is vfields[f - 1]
  std::vector<std::shared_ptr<FilterData>> filters;    // slot = filter id; null = free
  std::vector<const uint32_t *> reject_host;           // flattened [filter * n_segs + seg] device pointers
  DevBuf d_reject_table;                               // the same table on the device
  ~IndexState() {
    // kernels of already-destroyed batches, or rerank calls on the index stream, may still read the
    // tables: retiring a state is rare (one per update), so wait for the device once
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

}  // namespace

struct slg_index {
  // work buffers of finished batches.  Shared with the batches (a batch that outlives the index
  // still returns its buffers somewhere valid); declared first: destroyed last
  BufPool pool;
  slg_tuning tune{};
  std::vector<slg_batch *> live;  // batches prepared on this index and not yet destroyed (under mu)
  int device = 0;
  uint32_t n_cu = 256;  // compute units of the device (persistent launches fill its wave slots)
  hipStream_t own_stream = nullptr;
  // descriptor uploads of slg_batch_prepare*: non-blocking streams picked by caller thread.  A plain
  // hipMemcpy runs on the legacy default stream and waits for whatever the application has queued
  // there (config 4: the previous batch's all-gather, shard merge and D2H) — planning would then
  // serialise with the GPU work it is supposed to overlap
  static constexpr int kUploadStreams = 8;
  hipStream_t upload_streams[kUploadStreams] = {};
  hipStream_t stream = nullptr;
  std::shared_ptr<const IndexState> state;  // the current state (under mu)
  std::atomic<uint64_t> generation{0};      // = state->generation, readable without the lock
  std::mutex mu;
  std::mutex update_mu;  // serialises slg_index_update_* / add_filter / add_vector_field (taken before mu)
  // profiling of the scoring kernel
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  // set by a scoring wave that had to give up on a round (chunk-loop guard): checked at fetch
  DevBuf d_error_flag;
  std::shared_ptr<const IndexState> snapshot() {
    std::lock_guard<std::mutex> lk(mu);
    return state;
  }
};

struct slg_batch {
  slg_index *idx = nullptr;
  std::shared_ptr<const IndexState> snap;  // the index state the batch was prepared on
  uint32_t nq = 0, k = 0;
  int strategy = 0;
  uint32_t n_sq = 0, n_slices = 0, n_terms = 0, n_boundaries = 0, max_terms = 0;
  bool uniform = false;  // every sub-query fits the one-list-per-slot kernel
  bool plan_batch = false;  // some sub-query has a score plan (multi kernel only)
  bool nested = false;      // some sub-query has a two-level plan (groups of leaves)
  bool deep = false;        // some sub-query has a score tree of more than two levels
  const slg::PlanNode *d_nodes = nullptr;
  bool pruned = false;      // some sub-query has non-essential lists (MaxScore)
  bool multi = false;    // many-term form of it (slg_score_multi.hpp); else the packed kernel
  uint64_t n_postings = 0, n_postings_essential = 0, n_rounds = 0;
  std::vector<uint64_t> q_postings;  // per query (stats.postings_advanced)
  bool launched = false;             // slg_batch_run was called at least once
  bool own_stream_set = false;       // slg_batch_set_stream: run on `stream` instead of the index's
  hipStream_t stream = nullptr;
  DevBuf d_desc;                     // packed descriptors
  const slg::RoundQuery *d_sq = nullptr;
  const slg::TermRef *d_terms = nullptr;
  const uint32_t *d_slice_sq = nullptr;
  const uint32_t *d_slice_seg = nullptr;
  const uint32_t *d_slice_order = nullptr;
  const slg::QueryRef *d_queries = nullptr;
  const uint32_t *d_bnd_coarse = nullptr;
  DevBuf d_bounds, d_rdoc, d_slice_tk, d_slice_doc, d_q_scored, d_slice_desc;
  DevBuf d_work_ctr;       // work counter of the persistent scoring waves (zeroed by partition_rounds_kernel)
  DevBuf d_q_filter;       // [nq] 0 = none, f + 1 (select_topk_kernel); empty when unfiltered
  bool cand_mode = false;  // uniform kernel, k > 256: candidates + select_topk_kernel
  DevBuf d_cand, d_slice_cbeg, d_slice_ccnt;
  DevBuf d_out;  // doc | seg | score | count, contiguous
  uint32_t *d_out_doc = nullptr, *d_out_seg = nullptr, *d_out_count = nullptr;
  float *d_out_score = nullptr;
  DevBuf d_stamps;  // SLG_STAMPS diagnostic builds
  DevBuf d_blk_skip;  // block skipping: postings of non-essential lists that were never loaded (u64)
  // index-sharded runs (slg_batch_run_sharded): the gathered result blocks of all ranks and the merged
  // top-k doc | seg | score | count
  DevBuf d_gather, d_merged;
  slg_shard_group *shard_group = nullptr;  // the group of the last sharded run (timing goes there)
  hipEvent_t ev_shard[4] = {nullptr, nullptr, nullptr, nullptr};  // start | local kernels done | gathered | merged
  bool shard_timed = false;
  uint64_t n_postings_nonessential = 0;  // postings of the pruning-classified (non-essential) lists
};

namespace {

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    SLG_HIP(hipGetDevice(&prev));
    if (prev != dev) SLG_HIP(hipSetDevice(dev));
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

int kregs_for(uint32_t k) {
  if (k <= 64) return 1;
  if (k <= 128) return 2;
  if (k <= 256) return 4;
  if (k <= 512) return 8;
  return 16;
}

}  // namespace
namespace slg {
// defined in slg_score_inst.hip, one translation unit per KREGS
template <int KREGS>
void launch_score_kregs(const RoundScoreParams &sp, int kind, hipStream_t st);
template <> void launch_score_kregs<1>(const RoundScoreParams &, int, hipStream_t);
template <> void launch_score_kregs<2>(const RoundScoreParams &, int, hipStream_t);
template <> void launch_score_kregs<4>(const RoundScoreParams &, int, hipStream_t);
template <> void launch_score_kregs<8>(const RoundScoreParams &, int, hipStream_t);
template <> void launch_score_kregs<16>(const RoundScoreParams &, int, hipStream_t);
}  // namespace slg
namespace {

// kind: 1 few-term kernel (slg_score_uni3.hpp; 5: its 5..8-list form), 2 many-term kernel (slg_score_multi.hpp),
// 3 many-term kernel with pruning-classified lists, 4 the round-2 few-term kernel (slg_score_uni.hpp)
// 6 / 7: the few-term kernel in its blocked form (slg_score_uni4.hpp), <= 4 / 5..8 lists; 8 / 9: the same
// with score plans (flat Sum / DisMax over leaves)
int uniform_kind(uint32_t form, uint32_t max_terms, bool plans) {
  const bool few = max_terms <= (uint32_t)slg::kUniMaxLists;
  if (form == 2) return 4;
  if (form == 3) return few ? 1 : 5;
  if (plans) return few ? 8 : 9;
  return few ? 6 : 7;
}
void launch_score(const slg::RoundScoreParams &sp, int kind, hipStream_t st) {
#ifdef SLG_STAMPS  // diagnostic build: only the k <= 64 variant is compiled
  if (kregs_for(sp.k) != 1) throw SlgError(SLG_ERR_UNSUPPORTED, "stamps build supports k <= 64 only");
  slg::launch_score_kregs<1>(sp, kind, st);
  SLG_HIP(hipGetLastError());
  return;
#else
  switch (kregs_for(sp.k)) {
    case 1: slg::launch_score_kregs<1>(sp, kind, st); break;
    case 2: slg::launch_score_kregs<2>(sp, kind, st); break;
    case 4: slg::launch_score_kregs<4>(sp, kind, st); break;
    case 8: slg::launch_score_kregs<8>(sp, kind, st); break;
    default: slg::launch_score_kregs<16>(sp, kind, st); break;
  }
  SLG_HIP(hipGetLastError());
#endif
}

template <int KREGS>
void launch_merge_t(const slg::MergeParams &mp, hipStream_t st) {
  const uint32_t blocks = (mp.nq + slg::kWavesPerBlock - 1) / slg::kWavesPerBlock;
  hipLaunchKernelGGL((slg::merge_topk_kernel<KREGS>), dim3(blocks), dim3(256), 0, st, mp);
}
void launch_merge(const slg::MergeParams &mp, hipStream_t st) {
  switch (kregs_for(mp.k)) {
    case 1: launch_merge_t<1>(mp, st); break;
    case 2: launch_merge_t<2>(mp, st); break;
    case 4: launch_merge_t<4>(mp, st); break;
    case 8: launch_merge_t<8>(mp, st); break;
    default: launch_merge_t<16>(mp, st); break;
  }
  SLG_HIP(hipGetLastError());
}

template <int KREGS>
void launch_shard_merge_t(const slg::ShardMergeParams &mp, hipStream_t st) {
  const uint32_t blocks = (mp.nq + slg::kWavesPerBlock - 1) / slg::kWavesPerBlock;
  hipLaunchKernelGGL((slg::merge_shards_kernel<KREGS>), dim3(blocks), dim3(256), 0, st, mp);
}
void launch_shard_merge(const slg::ShardMergeParams &mp, hipStream_t st) {
  if (mp.k > 1024u) {  // beyond the register top-k: rank every entry by binary searches
    hipLaunchKernelGGL(slg::merge_shards_large_kernel, dim3(mp.nq), dim3(256), 0, st, mp);
    SLG_HIP(hipGetLastError());
    return;
  }
  switch (kregs_for(mp.k)) {
    case 1: launch_shard_merge_t<1>(mp, st); break;
    case 2: launch_shard_merge_t<2>(mp, st); break;
    case 4: launch_shard_merge_t<4>(mp, st); break;
    case 8: launch_shard_merge_t<8>(mp, st); break;
    default: launch_shard_merge_t<16>(mp, st); break;
  }
  SLG_HIP(hipGetLastError());
}

void validate_segment(const slg_segment_desc &d, uint32_t si, bool deep) {
  const std::string pfx = "segment " + std::to_string(si) + ": ";
  SLG_REQUIRE(d.term_offsets != nullptr, pfx + "term_offsets is NULL");
  SLG_REQUIRE(d.n_fields >= 1 && d.field_avgdl && d.field_doc_len, pfx + "field arrays missing");
  SLG_REQUIRE(d.term_offsets[0] == 0, pfx + "term_offsets[0] != 0");
  const uint64_t P = d.term_offsets[d.n_terms];
  SLG_REQUIRE(P == 0 || (d.doc_ids && d.tfs), pfx + "doc_ids/tfs missing");
  for (uint32_t t = 0; t < d.n_terms; t++) {
    const uint64_t a = d.term_offsets[t], b = d.term_offsets[t + 1];
    SLG_REQUIRE(b >= a, pfx + "term_offsets not monotone");
    SLG_REQUIRE(b - a <= 0xFFFFFFFEull, pfx + "posting list too long");
    if (d.term_field) SLG_REQUIRE(d.term_field[t] < d.n_fields, pfx + "term_field out of range");
    // The kernels index per-doc bitmaps with the raw doc id and the planner interpolates on
    // doc / n_docs, so ids must be < n_docs: checked on every posting, or (validate == 0, where
    // the caller vouches for increasing ids) on the last = largest posting of each list.
    if (deep) {
      for (uint64_t i = a; i < b; i++) {
        SLG_REQUIRE(d.doc_ids[i] < d.n_docs,
                    pfx + "doc id >= n_docs in term " + std::to_string(t));
        SLG_REQUIRE(i == a || d.doc_ids[i] > d.doc_ids[i - 1],
                    pfx + "doc ids not strictly increasing in term " + std::to_string(t));
      }
    } else if (b > a) {
      SLG_REQUIRE(d.doc_ids[b - 1] < d.n_docs, pfx + "doc id >= n_docs in term " + std::to_string(t));
    }
  }
  if (d.vec_dim) {
    SLG_REQUIRE(d.vec_offsets && (d.vec_values || d.vec_rows == 0), pfx + "vector arrays missing");
    SLG_REQUIRE(d.vec_metric == SLG_METRIC_COSINE || d.vec_metric == SLG_METRIC_L2,
                pfx + "bad vec_metric");
    if (deep)
      for (uint32_t i = 0; i < d.n_docs; i++)
        SLG_REQUIRE(d.vec_offsets[i] == SLG_NO_VECTOR || d.vec_offsets[i] < d.vec_rows,
                    pfx + "vec_offsets out of range");
  }
}

// Waiting for a stream is the blocking hipStreamSynchronize.  (Polling hipStreamQuery first — to spare a
// small batch the sleep / wake-up of the blocking wait — was built and measured: with 8 caller threads
// polling, config 2's host-inclusive rate fell from 11M to 2.2M queries/s and fetch + destroy grew from
// 0.28 to 1.36 ms per batch: the query takes a runtime lock the other threads' launches and copies need.)
static inline hipError_t wait_stream(hipStream_t st) { return hipStreamSynchronize(st); }

static inline hipStream_t batch_stream(const slg_batch *b) {
  return b->own_stream_set ? b->stream : b->idx->stream;
}

// The version of a staged segment for (deleted bitmap, live_docs): impacts (stage_impacts_kernel:
// query/bm25.rs:1-6 with idf from the host's logf), champion bounds, host mirror.  At creation the
// kernel also scatters the uploaded doc ids into the padded layout (docs_in != nullptr); for an
// update it reads them back from there.  Temporaries of a non-updatable store are passed in `tmp`.
struct StageTemps {
  const uint32_t *docs_in = nullptr;   // [P] unpadded doc ids as uploaded (creation only)
};
void derive_version(slg_index *ix, const std::shared_ptr<PostingStore> &ps, SegHost &sh, const uint8_t *deleted,
                    float docs, const StageTemps &tmp) {
  hipStream_t st = ix->stream;
  sh.store = ps;
  sh.n_docs = ps->n_docs;
  sh.n_terms = ps->n_terms;
  sh.n_postings = ps->n_postings;
  sh.null_idx = ps->null_idx;
  sh.docs = docs;
  const uint64_t P = ps->n_postings;
  const uint64_t P_pad = P + (uint64_t)slg::kListPad * (uint64_t)ps->n_terms + (uint64_t)slg::kNullRun;
  sh.d_imps.alloc(P_pad * 4, &ix->pool);
  SLG_HIP(hipMemsetAsync(sh.d_imps.p, 0, P_pad * 4, st));
  if (deleted) {
    const size_t words = ((size_t)ps->n_docs + 31) / 32;
    std::vector<uint32_t> w(words ? words : 1, 0u);
    std::memcpy(w.data(), deleted, ((size_t)ps->n_docs + 7) / 8);
    sh.d_deleted.alloc(w.size() * 4, &ix->pool);
    SLG_HIP(hipMemcpy(sh.d_deleted.p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  }
  if (P == 0) return;
  // idf per term: query/bm25.rs:2 with df = postings.len() as f32 (wand.rs:108,471)
  std::vector<float> idf(ps->n_terms);
  for (uint32_t t = 0; t < ps->n_terms; t++) {
    const float df = (float)(uint32_t)(ps->term_offsets[t + 1] - ps->term_offsets[t]);
    idf[t] = fmaxf(logf((docs - df + 0.5f) / (df + 0.5f)), 0.0f) + 1.0f;
  }
  DevBuf d_idf;
  d_idf.alloc((size_t)ps->n_terms * 4, &ix->pool);
  SLG_HIP(hipMemcpyAsync(d_idf.p, idf.data(), (size_t)ps->n_terms * 4, hipMemcpyHostToDevice, st));
  slg::StageParams sp{};
  sp.n_postings = P;
  sp.n_terms = ps->n_terms;
  sp.n_docs = ps->n_docs;
  sp.term_offsets = ps->d_offs.as<uint64_t>();
  sp.docs = tmp.docs_in;
  sp.docs_out = ps->d_docs.as<uint32_t>();
  sp.tfs = ps->d_tfs.as<uint32_t>();
  sp.term_idf = d_idf.as<float>();
  sp.term_field = ps->has_term_field ? ps->d_tfield.as<uint16_t>() : nullptr;
  sp.field_doc_len = ps->d_lenptrs.as<const float *>();
  sp.field_avgdl = ps->d_avgdl.as<float>();
  sp.k1 = ps->k1;
  sp.b = ps->b;
  sp.imps = sh.d_imps.as<float>();
  const uint64_t want = (P + 255) / 256;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>(want, 256ull * 32);
  hipLaunchKernelGGL(slg::stage_impacts_kernel, dim3(blocks), dim3(256), 0, st, sp);
  SLG_HIP(hipGetLastError());
  if (ix->tune.champions) {
    sh.d_champ.alloc((size_t)ps->n_terms * slg::kChampions * 4, &ix->pool);
    slg::ChampParams cp{};
    cp.term_offsets = ps->d_offs.as<uint64_t>();
    cp.imps = sh.d_imps.as<float>();
    cp.docs = ps->d_docs.as<uint32_t>();
    cp.deleted = sh.d_deleted.as<uint32_t>();
    cp.champ = sh.d_champ.as<float>();
    cp.n_terms = ps->n_terms;
    const uint32_t cblocks = std::min<uint32_t>((ps->n_terms + 3) / 4, 256u * 16);
    hipLaunchKernelGGL(slg::stage_champions_kernel, dim3(cblocks ? cblocks : 1), dim3(256), 0, st, cp);
    SLG_HIP(hipGetLastError());
  }
  SLG_HIP(hipStreamSynchronize(st));  // d_idf dies here
  if (sh.d_champ.p) {
    sh.champ.resize((size_t)ps->n_terms * slg::kChampions);
    SLG_HIP(hipMemcpy(sh.champ.data(), sh.d_champ.p, sh.champ.size() * 4, hipMemcpyDeviceToHost));
  }
}

std::shared_ptr<SegHost> stage_segment(slg_index *ix, const slg_segment_desc &d) {
  hipStream_t st = ix->stream;
  auto ps = std::make_shared<PostingStore>();
  ps->n_docs = d.n_docs;
  ps->n_terms = d.n_terms;
  ps->n_fields = d.n_fields;
  ps->term_offsets.assign(d.term_offsets, d.term_offsets + d.n_terms + 1);
  ps->avgdl.assign(d.field_avgdl, d.field_avgdl + d.n_fields);
  ps->k1 = d.k1;
  ps->b = d.b;
  ps->has_term_field = d.term_field != nullptr;
  ps->updatable = ix->tune.updatable != 0;
  const uint64_t P = ps->term_offsets[d.n_terms];
  ps->n_postings = P;

  // padded layout (SegDev): every list is followed by kListPad sentinel entries, + a run of kNullRun
  // at the end (null_idx): the scoring kernels load whole 64-lane slots starting at any posting
  const uint64_t P_pad = P + (uint64_t)slg::kListPad * (uint64_t)d.n_terms + (uint64_t)slg::kNullRun;
  ps->null_idx = P + (uint64_t)slg::kListPad * d.n_terms;
  ps->d_docs.alloc(P_pad * 4, &ix->pool);
  SLG_HIP(hipMemsetAsync(ps->d_docs.p, 0xFF, P_pad * 4, st));
  DevBuf d_docs_in;  // the uploaded (unpadded) doc ids: staging only
  if (P > 0) {
    d_docs_in.alloc(P * 4, &ix->pool);
    ps->d_tfs.alloc(P * 4, &ix->pool);
    SLG_HIP(hipMemcpyAsync(d_docs_in.p, d.doc_ids, P * 4, hipMemcpyHostToDevice, st));
    SLG_HIP(hipMemcpyAsync(ps->d_tfs.p, d.tfs, P * 4, hipMemcpyHostToDevice, st));
    ps->d_offs.alloc(((size_t)d.n_terms + 1) * 8, &ix->pool);
    SLG_HIP(hipMemcpyAsync(ps->d_offs.p, ps->term_offsets.data(), ((size_t)d.n_terms + 1) * 8,
                           hipMemcpyHostToDevice, st));
    if (d.term_field) {
      ps->d_tfield.alloc((size_t)d.n_terms * 2, &ix->pool);
      SLG_HIP(hipMemcpyAsync(ps->d_tfield.p, d.term_field, (size_t)d.n_terms * 2, hipMemcpyHostToDevice, st));
    }
    ps->d_avgdl.alloc((size_t)d.n_fields * 4, &ix->pool);
    SLG_HIP(hipMemcpyAsync(ps->d_avgdl.p, d.field_avgdl, (size_t)d.n_fields * 4, hipMemcpyHostToDevice, st));
    ps->d_lens.resize(d.n_fields);
    std::vector<const float *> lenptrs(d.n_fields, nullptr);
    for (uint32_t f = 0; f < d.n_fields; f++) {
      if (d.field_doc_len[f] && d.n_docs) {
        ps->d_lens[f].alloc((size_t)d.n_docs * 4, &ix->pool);
        SLG_HIP(hipMemcpyAsync(ps->d_lens[f].p, d.field_doc_len[f], (size_t)d.n_docs * 4, hipMemcpyHostToDevice, st));
        lenptrs[f] = ps->d_lens[f].as<float>();
      }
    }
    ps->d_lenptrs.alloc((size_t)d.n_fields * sizeof(float *), &ix->pool);
    SLG_HIP(hipMemcpy(ps->d_lenptrs.p, lenptrs.data(), (size_t)d.n_fields * sizeof(float *), hipMemcpyHostToDevice));
  }
  auto sh = std::make_shared<SegHost>();
  StageTemps tmp;
  tmp.docs_in = d_docs_in.as<uint32_t>();
  derive_version(ix, ps, *sh, d.deleted, d.docs, tmp);  // (synchronises the stream when P > 0)
  if (!ps->updatable) {  // what only an update would read again
    ps->d_tfs.release();
    ps->d_offs.release();
    ps->d_tfield.release();
    ps->d_avgdl.release();
    ps->d_lenptrs.release();
    ps->d_lens.clear();
  }
  if (d.vec_dim) {
    ps->vec_dim = d.vec_dim;
    ps->vec_rows = d.vec_rows;
    ps->vec_metric = d.vec_metric;
    ps->d_vec_offsets.alloc((size_t)d.n_docs * 4, &ix->pool);
    if (d.n_docs)
      SLG_HIP(hipMemcpy(ps->d_vec_offsets.p, d.vec_offsets, (size_t)d.n_docs * 4, hipMemcpyHostToDevice));
    const size_t vb = (size_t)d.vec_rows * d.vec_dim * 4;
    ps->d_vec_values.alloc(vb, &ix->pool);
    if (vb) SLG_HIP(hipMemcpy(ps->d_vec_values.p, d.vec_values, vb, hipMemcpyHostToDevice));
  }
  SLG_HIP(hipStreamSynchronize(st));
  return sh;
}

// the device tables of a state (segment descriptors, vector stores, reject-bitmap pointers): small,
// rebuilt for every state
void finish_state(slg_index *ix, IndexState &s) {
  const size_t n_segs = s.segs.size();
  s.device = ix->device;
  std::vector<slg::SegDev> sd(n_segs);
  std::vector<slg::VecSegDev> vd(n_segs);
  for (size_t i = 0; i < n_segs; i++) {
    const SegHost &sh = *s.segs[i];
    sd[i].docs = sh.store->d_docs.as<uint32_t>();
    sd[i].imps = sh.d_imps.as<float>();
    sd[i].deleted = sh.d_deleted.as<uint32_t>();
    sd[i].champ = sh.d_champ.as<float>();
    sd[i].n_docs = sh.n_docs;
    sd[i].pad = 0;
    sd[i].null_idx = sh.null_idx;
    vd[i].offsets = sh.store->d_vec_offsets.as<uint32_t>();
    vd[i].values = sh.store->d_vec_values.as<float>();
    vd[i].n_docs = sh.n_docs;
    vd[i].dim = sh.store->vec_dim;
    vd[i].metric = sh.store->vec_metric;
    vd[i].pad = 0;
  }
  s.d_segs.alloc(std::max<size_t>(n_segs, 1) * sizeof(slg::SegDev), &ix->pool);
  s.d_vsegs.alloc(std::max<size_t>(n_segs, 1) * sizeof(slg::VecSegDev), &ix->pool);
  if (n_segs) {
    SLG_HIP(hipMemcpy(s.d_segs.p, sd.data(), n_segs * sizeof(slg::SegDev), hipMemcpyHostToDevice));
    SLG_HIP(hipMemcpy(s.d_vsegs.p, vd.data(), n_segs * sizeof(slg::VecSegDev), hipMemcpyHostToDevice));
  }
  for (auto &vfp : s.vfields) {  // (the field objects of a new state are fresh copies: see copy_state)
    VecFieldHost &vf = *vfp;
    std::vector<slg::VecSegDev> fv(n_segs);
    for (size_t i = 0; i < n_segs; i++) {
      fv[i] = slg::VecSegDev{nullptr, nullptr, s.segs[i]->n_docs, 0u, vf.metric, 0u};
      if (i < vf.per_seg.size() && vf.per_seg[i]) {
        fv[i].offsets = vf.per_seg[i]->offsets.as<uint32_t>();
        fv[i].values = vf.per_seg[i]->values.as<float>();
        fv[i].dim = vf.per_seg[i]->dim;
      }
    }
    vf.d_vsegs.alloc(std::max<size_t>(n_segs, 1) * sizeof(slg::VecSegDev), &ix->pool);
    if (n_segs) SLG_HIP(hipMemcpy(vf.d_vsegs.p, fv.data(), n_segs * sizeof(slg::VecSegDev), hipMemcpyHostToDevice));
  }
  s.reject_host.assign(s.filters.size() * n_segs, nullptr);
  for (size_t f = 0; f < s.filters.size(); f++)
    if (s.filters[f])
      for (size_t i = 0; i < n_segs && i < s.filters[f]->per_seg.size(); i++)
        if (s.filters[f]->per_seg[i]) s.reject_host[f * n_segs + i] = s.filters[f]->per_seg[i]->as<uint32_t>();
  s.d_reject_table.alloc(std::max<size_t>(s.reject_host.size(), 1) * sizeof(void *), &ix->pool);
  if (!s.reject_host.empty())
    SLG_HIP(hipMemcpy(s.d_reject_table.p, s.reject_host.data(), s.reject_host.size() * sizeof(void *),
                      hipMemcpyHostToDevice));
}

// the next state: the current one's segments / fields / filters (shared), the device tables not yet built
std::unique_ptr<IndexState> copy_state(const IndexState &cur) {
  auto n = std::make_unique<IndexState>();
  n->generation = cur.generation + 1;
  n->device = cur.device;
  n->segs = cur.segs;
  n->filters = cur.filters;
  for (auto &vf : cur.vfields) {  // the per-state table d_vsegs is rebuilt: own object, shared stores
    auto c = std::make_shared<VecFieldHost>();
    c->dim = vf->dim;
    c->metric = vf->metric;
    c->per_seg = vf->per_seg;
    n->vfields.push_back(std::move(c));
  }
  return n;
}

void publish(slg_index *ix, std::unique_ptr<IndexState> ns) {
  std::shared_ptr<const IndexState> keep;  // (the old state may die here: outside the lock)
  std::shared_ptr<const IndexState> fresh(std::move(ns));
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    keep.swap(ix->state);
    ix->state = fresh;
    ix->generation.store(fresh->generation, std::memory_order_release);
  }
}

size_t state_device_bytes(const IndexState &s) {
  size_t n = s.d_segs.bytes + s.d_vsegs.bytes + s.d_reject_table.bytes;
  for (auto &sh : s.segs) n += sh->device_bytes() + sh->store->device_bytes();
  for (auto &f : s.filters)
    if (f)
      for (auto &b : f->per_seg)
        if (b) n += b->bytes;
  for (auto &vf : s.vfields) {
    n += vf->d_vsegs.bytes;
    for (auto &v : vf->per_seg)
      if (v) n += v->offsets.bytes + v->values.bytes;
  }
  return n;
}

}  // namespace

extern "C" {

uint32_t slg_abi_version(void) { return SLG_ABI_VERSION; }

const char *slg_last_error(void) { return g_last_error.c_str(); }
int slg_last_error_code(void) { return g_last_code; }

int slg_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return SLG_ERR_DEVICE;
  }
  return n;
}

void slg_tuning_default(slg_tuning *t) {
  if (!t) return;
  std::memset(t, 0, sizeof(*t));
  t->struct_size = (uint32_t)sizeof(slg_tuning);
  t->validate = env_i32("SLG_VALIDATE", 1) != 0;
  t->champions = env_i32("SLG_NO_CHAMPIONS", 0) == 0;
  t->allow_any_arch = env_i32("SLG_ALLOW_ANY_ARCH", 0) != 0;
  t->pruning = env_i32("SLG_MAXSCORE", -1);
  t->uniform_max_terms = env_u32("SLG_UNIFORM_MAX_TERMS", 8);
  t->uniform_round_target = env_u32("SLG_UNIFORM_ROUND_TARGET", 0);
  t->multi_round_target = env_u32("SLG_MULTI_ROUND_TARGET", slg::kMultiTarget);
  t->probe_target = env_u32("SLG_PROBE_TARGET", 2048);
  t->rounds_per_slice = env_u32("SLG_ROUNDS_PER_SLICE", 0);
  t->max_rounds_per_slice = env_u32("SLG_MAX_ROUNDS_PER_SLICE", 0);
  t->slices_per_subquery = env_u32("SLG_SLICES_PER_SUBQUERY", 16);
  t->cand_mode = env_i32("SLG_NO_CAND_MODE", 0) == 0;
  t->slice_order = env_i32("SLG_NO_SLICE_ORDER", 0) == 0;
  t->block_max = env_i32("SLG_NO_BLOCK_MAX", 0) == 0;
  t->pool_cap_mb = env_u32("SLG_POOL_CAP_MB", 0);
  t->uniform_kernel = env_u32("SLG_UNIFORM_KERNEL", 4);
  t->uniform_sigma_x100 = env_u32("SLG_UNIFORM_SIGMA", 0);
  t->inline_cuts = env_i32("SLG_INLINE_CUTS", -1);
  t->updatable = env_i32("SLG_NOT_UPDATABLE", 0) == 0;
  t->uniform_plans = env_i32("SLG_NO_UNIFORM_PLANS", 0) == 0;
  t->score_waves_per_simd = env_u32("SLG_SCORE_WAVES", 0);
}

slg_index *slg_index_create(const slg_segment_desc *segs, uint32_t n_segs, int device) {
  return slg_index_create_tuned(segs, n_segs, device, nullptr);
}

slg_index *slg_index_create_tuned(const slg_segment_desc *segs, uint32_t n_segs, int device,
                                  const slg_tuning *tuning) {
  slg_index *ix = nullptr;
  int rc = guarded([&] {
    SLG_REQUIRE(segs != nullptr && n_segs >= 1, "segs is NULL or n_segs == 0");
    slg_tuning tune;
    if (tuning) {
      SLG_REQUIRE(tuning->struct_size == sizeof(slg_tuning), "slg_tuning.struct_size mismatch");
      tune = *tuning;
    } else {
      slg_tuning_default(&tune);
    }
    // (a zero-initialised struct from a C or Rust caller must not plan for one kernel and launch another)
    SLG_REQUIRE(tune.uniform_kernel >= 2 && tune.uniform_kernel <= 4, "slg_tuning.uniform_kernel must be 2, 3 or 4");
#ifndef SLG_LEGACY_KERNELS
    if (tune.uniform_kernel != 4)
      throw SlgError(SLG_ERR_UNSUPPORTED, "slg_tuning.uniform_kernel 2 / 3 need a library built with -DSLG_LEGACY_KERNELS");
#endif
    tune.uniform_max_terms = std::min<uint32_t>(
        tune.uniform_max_terms, tune.uniform_kernel == 2 ? slg::kUniMaxLists : slg::kU3MaxLists);
    tune.max_rounds_per_slice = std::min<uint32_t>(tune.max_rounds_per_slice, slg::kMaxRoundsPerSlice);
    tune.slices_per_subquery = std::max<uint32_t>(1, tune.slices_per_subquery);
    for (uint32_t s = 0; s < n_segs; s++) validate_segment(segs[s], s, tune.validate != 0);
    int ndev = 0;
    SLG_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
      throw SlgError(SLG_ERR_DEVICE, "no such HIP device " + std::to_string(device));
    ix = new slg_index();
    ix->tune = tune;
    ix->device = device;
    DeviceGuard g(device);
    hipDeviceProp_t prop;
    SLG_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !tune.allow_any_arch)
      throw SlgError(SLG_ERR_DEVICE,
                     std::string("device is ") + prop.gcnArchName + ", this library targets gfx950");
    ix->n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    SLG_HIP(hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking));
    ix->stream = ix->own_stream;
    for (auto &us : ix->upload_streams) SLG_HIP(hipStreamCreateWithFlags(&us, hipStreamNonBlocking));
    auto st0 = std::make_unique<IndexState>();
    st0->generation = 0;
    st0->device = device;
    for (uint32_t s = 0; s < n_segs; s++) st0->segs.push_back(stage_segment(ix, segs[s]));
    {  // bound of the work-buffer pool (BufPool): a quarter of what staging left free
      size_t cap = (size_t)tune.pool_cap_mb << 20;
      if (tune.pool_cap_mb == 0) {
        size_t free_b = 0, total_b = 0;
        SLG_HIP(hipMemGetInfo(&free_b, &total_b));
        cap = std::min<size_t>(24ull << 30, std::max<size_t>(1ull << 30, free_b / 4));
        ix->tune.pool_cap_mb = (uint32_t)(cap >> 20);
      }
      ix->pool.cap = cap;
    }
    ix->d_error_flag.alloc(16, &ix->pool);
    SLG_HIP(hipMemset(ix->d_error_flag.p, 0, 16));
    finish_state(ix, *st0);
    publish(ix, std::move(st0));
  });
  if (rc != SLG_OK) {
    const std::string keep = g_last_error;
    if (ix) slg_index_destroy(ix);
    g_last_error = keep;
    g_last_code = rc;
    return nullptr;
  }
  return ix;
}

namespace {
// free everything a batch holds on the device; with `to_pool` false the blocks go straight back
// to the runtime (the index and its pool are going away)
void release_batch_buffers(slg_batch *b, bool to_pool) {
  DevBuf *bufs[] = {&b->d_desc, &b->d_bounds, &b->d_rdoc, &b->d_slice_desc, &b->d_slice_tk, &b->d_slice_doc,
                    &b->d_q_scored, &b->d_work_ctr, &b->d_q_filter, &b->d_cand, &b->d_slice_cbeg, &b->d_slice_ccnt,
                    &b->d_out, &b->d_stamps, &b->d_blk_skip, &b->d_gather, &b->d_merged};
  for (DevBuf *d : bufs) {
    if (!to_pool) d->pool = nullptr;
    d->release();
  }
}
}  // namespace

void slg_index_destroy(slg_index *ix) {
  if (!ix) return;
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(ix->device);
  (void)hipDeviceSynchronize();  // batches may run on streams of their own
  {
    // batches that outlive the index are detached: buffers freed, handle stays valid for
    // slg_batch_destroy, every other call on it fails with SLG_ERR_INVALID
    std::lock_guard<std::mutex> lk(ix->mu);
    for (slg_batch *b : ix->live) {
      release_batch_buffers(b, false);
      b->snap.reset();
      b->idx = nullptr;
    }
    ix->live.clear();
  }
  for (auto &pr : ix->prof_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  {
    std::shared_ptr<const IndexState> last;
    std::lock_guard<std::mutex> lk(ix->mu);
    last.swap(ix->state);
  }  // (the state's device memory goes here unless a detached batch handle still holds a snapshot)
  ix->d_error_flag.release();
  if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
  for (auto us : ix->upload_streams)
    if (us) (void)hipStreamDestroy(us);
  delete ix;
  if (prev >= 0) (void)hipSetDevice(prev);
}

int slg_index_get_tuning(const slg_index *ix, slg_tuning *out) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr && out != nullptr, "index or out is NULL");
    *out = ix->tune;
  });
}

int slg_index_trim_pool(slg_index *ix, uint64_t *freed_bytes) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    DeviceGuard g(ix->device);
    const size_t freed = ix->pool.drain();
    if (freed_bytes) *freed_bytes = freed;
  });
}

int slg_index_info(const slg_index *ix, uint32_t *n_segs, uint64_t *n_postings,
                   uint64_t *device_bytes) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    const auto st = const_cast<slg_index *>(ix)->snapshot();
    uint64_t P = 0;
    for (auto &s : st->segs) P += s->n_postings;
    if (n_segs) *n_segs = (uint32_t)st->segs.size();
    if (n_postings) *n_postings = P;
    if (device_bytes) *device_bytes = state_device_bytes(*st);
  });
}

int slg_index_set_stream(slg_index *ix, void *hip_stream) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    SLG_HIP(hipStreamSynchronize(ix->stream));
    ix->stream = hip_stream == SLG_OWN_STREAM ? ix->own_stream : (hipStream_t)hip_stream;
  });
}


// ---- doc filters -------------------------------------------------------------
namespace {
int add_filter_impl(slg_index *ix, const uint8_t *const *seg_bitmaps, const void *const *seg_columns,
                    int column_kind, long long lo_i, long long hi_i, double lo_f, double hi_f,
                    const uint32_t *term_ids = nullptr, uint32_t n_terms = 0, int pass_if_absent = 0) {
  int id = -1;
  int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    std::lock_guard<std::mutex> ulk(ix->update_mu);
    const auto cur = ix->snapshot();
    DeviceGuard g(ix->device);
    hipStream_t st = ix->stream;
    const size_t n_segs = cur->segs.size();
    auto fd = std::make_shared<FilterData>();
    fd->per_seg.resize(n_segs);
    std::vector<DevBuf> tmp(n_segs);  // uploaded pass bitmaps / columns (freed on return)
    std::vector<DevBuf> marked(n_segs);  // docs that hold one of the terms (slg_index_add_filter_terms)
    SLG_REQUIRE(n_terms == 0 || term_ids != nullptr, "term_ids is NULL");
    for (size_t s = 0; s < n_segs; s++) {
      const SegHost &sh = *cur->segs[s];
      const size_t words = ((size_t)sh.n_docs + 31) / 32;
      fd->per_seg[s] = std::make_shared<DevBuf>();
      fd->per_seg[s]->alloc((words ? words : 1) * 4, &ix->pool);
      slg::FilterBuildParams fp{};
      fp.deleted = sh.d_deleted.as<uint32_t>();
      fp.n_docs = sh.n_docs;
      fp.reject = fd->per_seg[s]->as<uint32_t>();
      fp.column_kind = 0;
      if (column_kind) {
        SLG_REQUIRE(seg_columns && seg_columns[s], "filter column of a segment is NULL");
        tmp[s].alloc((size_t)std::max<uint32_t>(sh.n_docs, 1) * 8, &ix->pool);
        SLG_HIP(hipMemcpyAsync(tmp[s].p, seg_columns[s], (size_t)sh.n_docs * 8, hipMemcpyHostToDevice, st));
        fp.column = tmp[s].p;
        fp.column_kind = column_kind;
        fp.lo_i = lo_i;
        fp.hi_i = hi_i;
        fp.lo_f = lo_f;
        fp.hi_f = hi_f;
      } else if (seg_bitmaps && seg_bitmaps[s]) {
        std::vector<uint32_t> w(words ? words : 1, 0u);
        std::memcpy(w.data(), seg_bitmaps[s], ((size_t)sh.n_docs + 7) / 8);
        tmp[s].alloc(w.size() * 4, &ix->pool);
        SLG_HIP(hipMemcpy(tmp[s].p, w.data(), w.size() * 4, hipMemcpyHostToDevice));
        fp.pass = tmp[s].as<uint32_t>();
      }
      if (term_ids) {
        // the docs of the given posting lists, marked on the device (the lists are resident: nothing is
        // uploaded but a bitmap's worth of zeros); a caller's bitmap, if any, is AND-ed as a second pass set
        const PostingStore &ps = *sh.store;
        marked[s].alloc((words ? words : 1) * 4, &ix->pool);
        SLG_HIP(hipMemsetAsync(marked[s].p, 0, (words ? words : 1) * 4, st));
        for (uint32_t t = 0; t < n_terms; t++) {
          const uint32_t id = term_ids[(size_t)t * n_segs + s];
          if (id == SLG_NO_TERM) continue;
          SLG_REQUIRE(id < ps.n_terms, "term id out of range");
          const uint64_t a = ps.term_offsets[id], b = ps.term_offsets[(size_t)id + 1];
          if (b == a || sh.n_docs == 0) continue;
          slg::PostingMarkParams mp{};
          mp.docs = ps.d_docs.as<uint32_t>() + a + (uint64_t)slg::kListPad * id;
          mp.df = (uint32_t)(b - a);
          mp.n_docs = sh.n_docs;
          mp.bitmap = marked[s].as<uint32_t>();
          hipLaunchKernelGGL(slg::posting_mark_kernel, dim3((mp.df + 255) / 256), dim3(256), 0, st, mp);
          SLG_HIP(hipGetLastError());
        }
        fp.pass2 = fp.pass;  // (the caller's bitmap, or nullptr)
        fp.pass = marked[s].as<uint32_t>();
        fp.invert_pass = pass_if_absent ? 1 : 0;
      }
      if (sh.n_docs) {
        hipLaunchKernelGGL(slg::filter_build_kernel, dim3((sh.n_docs + 255) / 256), dim3(256), 0, st, fp);
        SLG_HIP(hipGetLastError());
      }
    }
    SLG_HIP(hipStreamSynchronize(st));
    // the next state: the filter takes the lowest free id (ids of removed filters are reused, so the
    // pointer table stays as small as the number of filters alive at once).  Batches prepared on
    // earlier states keep their own table: nobody waits for the device here
    auto ns = copy_state(*cur);
    ns->generation = cur->generation;  // (filters do not change what a manifest snapshot sees)
    size_t slot = 0;
    while (slot < ns->filters.size() && ns->filters[slot]) slot++;
    if (slot == ns->filters.size()) ns->filters.emplace_back();
    ns->filters[slot] = std::move(fd);
    finish_state(ix, *ns);
    publish(ix, std::move(ns));
    id = (int)slot;
  });
  return rc == SLG_OK ? id : rc;
}
}  // namespace

int slg_index_add_filter(slg_index *ix, const uint8_t *const *seg_bitmaps) {
  return add_filter_impl(ix, seg_bitmaps, nullptr, 0, 0, 0, 0.0, 0.0);
}
int slg_index_add_filter_range_i64(slg_index *ix, const int64_t *const *seg_columns, int64_t lo, int64_t hi) {
  return add_filter_impl(ix, nullptr, reinterpret_cast<const void *const *>(seg_columns), 1, lo, hi, 0.0, 0.0);
}
int slg_index_add_filter_range_f64(slg_index *ix, const double *const *seg_columns, double lo, double hi) {
  return add_filter_impl(ix, nullptr, reinterpret_cast<const void *const *>(seg_columns), 2, 0, 0, lo, hi);
}
int slg_index_add_filter_terms(slg_index *ix, const uint32_t *term_ids, uint32_t n_terms, int pass_if_absent,
                               const uint8_t *const *and_bitmaps_or_null) {
  if (!term_ids && n_terms) {
    return guarded([&] { SLG_REQUIRE(false, "term_ids is NULL"); });
  }
  static const uint32_t none = SLG_NO_TERM;
  return add_filter_impl(ix, and_bitmaps_or_null, nullptr, 0, 0, 0, 0.0, 0.0, term_ids ? term_ids : &none, n_terms,
                         pass_if_absent);
}
int slg_index_remove_filter(slg_index *ix, int filter_id) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    std::lock_guard<std::mutex> ulk(ix->update_mu);
    const auto cur = ix->snapshot();
    DeviceGuard g(ix->device);
    SLG_REQUIRE(filter_id >= 0 && (size_t)filter_id < cur->filters.size() && cur->filters[filter_id],
                "unknown filter id");
    // batches prepared with the filter hold the state that owns its bitmaps and its table row: they
    // may still run.  The slot is free for the next add
    auto ns = copy_state(*cur);
    ns->generation = cur->generation;
    ns->filters[filter_id].reset();
    while (!ns->filters.empty() && !ns->filters.back()) ns->filters.pop_back();
    finish_state(ix, *ns);
    publish(ix, std::move(ns));
  });
}

// ---- index updates (api/writer.rs:106-240) ---------------------------------------------------
int slg_index_update_deleted(slg_index *ix, uint32_t seg, const uint8_t *deleted, float live_docs) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    std::lock_guard<std::mutex> ulk(ix->update_mu);
    const auto cur = ix->snapshot();
    SLG_REQUIRE(seg < cur->segs.size(), "no such segment");
    SLG_REQUIRE(live_docs >= 0.0f && live_docs <= (float)cur->segs[seg]->n_docs, "live_docs outside [0, n_docs]");
    const std::shared_ptr<PostingStore> &ps = cur->segs[seg]->store;
    if (!ps->updatable)
      throw SlgError(SLG_ERR_UNSUPPORTED, "index was created with slg_tuning.updatable = 0");
    DeviceGuard g(ix->device);
    auto sh = std::make_shared<SegHost>();
    derive_version(ix, ps, *sh, deleted, live_docs, StageTemps{});
    auto ns = copy_state(*cur);
    ns->segs[seg] = sh;
    // registered filters: reject = deleted | ~filter, so this segment's bitmaps take the new tombstones
    const size_t words = std::max<size_t>(((size_t)ps->n_docs + 31) / 32, 1);
    for (auto &f : ns->filters) {
      if (!f || seg >= f->per_seg.size() || !f->per_seg[seg]) continue;
      auto nf = std::make_shared<FilterData>(*f);
      auto nb = std::make_shared<DevBuf>();
      nb->alloc(words * 4, &ix->pool);
      slg::BitmapOrParams bp{};
      bp.a = f->per_seg[seg]->as<uint32_t>();
      bp.b = sh->d_deleted.as<uint32_t>();
      bp.out = nb->as<uint32_t>();
      bp.n_words = (uint32_t)words;
      hipLaunchKernelGGL(slg::bitmap_or_kernel, dim3((uint32_t)((words + 255) / 256)), dim3(256), 0, ix->stream, bp);
      SLG_HIP(hipGetLastError());
      nf->per_seg[seg] = std::move(nb);
      f = std::move(nf);
    }
    SLG_HIP(hipStreamSynchronize(ix->stream));
    finish_state(ix, *ns);
    publish(ix, std::move(ns));
  });
}

int slg_index_add_segment(slg_index *ix, const slg_segment_desc *seg) {
  int ord = -1;
  const int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr && seg != nullptr, "index or segment descriptor is NULL");
    validate_segment(*seg, 0, ix->tune.validate != 0);
    std::lock_guard<std::mutex> ulk(ix->update_mu);
    const auto cur = ix->snapshot();
    DeviceGuard g(ix->device);
    auto sh = stage_segment(ix, *seg);
    auto ns = copy_state(*cur);
    ns->segs.push_back(std::move(sh));
    for (auto &f : ns->filters)  // no bitmap for the new segment: unusable until registered again
      if (f) {
        auto nf = std::make_shared<FilterData>(*f);
        nf->per_seg.resize(ns->segs.size());
        f = std::move(nf);
      }
    for (auto &vf : ns->vfields) vf->per_seg.resize(ns->segs.size());
    finish_state(ix, *ns);
    ord = (int)ns->segs.size() - 1;
    publish(ix, std::move(ns));
  });
  return rc == SLG_OK ? ord : rc;
}

int slg_index_remove_segment(slg_index *ix, uint32_t seg) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    std::lock_guard<std::mutex> ulk(ix->update_mu);
    const auto cur = ix->snapshot();
    SLG_REQUIRE(seg < cur->segs.size(), "no such segment");
    SLG_REQUIRE(cur->segs.size() > 1, "the last segment of an index cannot be removed (add the replacement first)");
    DeviceGuard g(ix->device);
    auto ns = copy_state(*cur);
    ns->segs.erase(ns->segs.begin() + seg);
    for (auto &f : ns->filters)
      if (f) {
        auto nf = std::make_shared<FilterData>(*f);
        if (seg < nf->per_seg.size()) nf->per_seg.erase(nf->per_seg.begin() + seg);
        f = std::move(nf);
      }
    for (auto &vf : ns->vfields)
      if (seg < vf->per_seg.size()) vf->per_seg.erase(vf->per_seg.begin() + seg);
    finish_state(ix, *ns);
    publish(ix, std::move(ns));
  });
}

int slg_index_device(const slg_index *ix) {
  if (!ix) {
    g_last_error = "index is NULL";
    g_last_code = SLG_ERR_INVALID;
    return SLG_ERR_INVALID;
  }
  return ix->device;
}

uint64_t slg_index_generation(const slg_index *ix) {
  return ix ? ix->generation.load(std::memory_order_acquire) : 0;  // (no lock: callers poll it per request)
}

slg_batch *slg_batch_prepare(slg_index *ix, uint32_t nq, const uint32_t *q_offsets,
                             const uint32_t *q_term_ids, const float *q_weights, uint32_t k,
                             int strategy) {
  return slg_batch_prepare_filtered(ix, nq, q_offsets, q_term_ids, q_weights, nullptr, k, strategy);
}

slg_batch *slg_batch_prepare_filtered(slg_index *ix, uint32_t nq, const uint32_t *q_offsets,
                                      const uint32_t *q_term_ids, const float *q_weights,
                                      const int32_t *q_filter, uint32_t k, int strategy) {
  return slg_batch_prepare_plan(ix, nq, q_offsets, q_term_ids, q_weights, nullptr, nullptr, nullptr,
                                nullptr, q_filter, k, strategy);
}

slg_batch *slg_batch_prepare_plan(slg_index *ix, uint32_t nq, const uint32_t *q_offsets,
                                  const uint32_t *q_term_ids, const float *q_weights,
                                  const uint32_t *q_leaf, const int32_t *q_plan, const float *q_tie,
                                  const uint32_t *q_nleaves, const int32_t *q_filter, uint32_t k,
                                  int strategy) {
  slg_score_plans pl{};
  pl.q_leaf = q_leaf;
  pl.q_plan = q_plan;
  pl.q_tie = q_tie;
  pl.q_nleaves = q_nleaves;
  return slg_batch_prepare_plans(ix, nq, q_offsets, q_term_ids, q_weights, &pl, q_filter, k, strategy);
}

slg_batch *slg_batch_prepare_plans(slg_index *ix, uint32_t nq, const uint32_t *q_offsets,
                                   const uint32_t *q_term_ids, const float *q_weights,
                                   const slg_score_plans *plans, const int32_t *q_filter, uint32_t k,
                                   int strategy) {
  slg_batch *b = nullptr;
  int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    // Planning (slg_plan.cpp: a pure host function) reads only the immutable state the batch binds
    // to, so host threads may prepare batches for one index concurrently, also while an update builds
    // the next state; the index mutex is held just to take the snapshot.
    const std::shared_ptr<const IndexState> snap = ix->snapshot();
    std::vector<char> filter_live(snap->filters.size());
    for (size_t f = 0; f < snap->filters.size(); f++)
      filter_live[f] = snap->filters[f] && snap->filters[f]->complete() &&
                       snap->filters[f]->per_seg.size() == snap->segs.size();
    std::vector<slgplan::SegView> views(snap->segs.size());
    for (size_t s = 0; s < snap->segs.size(); s++) {
      const SegHost &sh = *snap->segs[s];
      views[s].n_docs = sh.n_docs;
      views[s].n_terms = sh.n_terms;
      views[s].term_offsets = sh.store->term_offsets.data();
      views[s].champ = sh.champ.empty() ? nullptr : sh.champ.data();
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
    in.filter_live = filter_live.data();
    in.n_filters = filter_live.size();
    slgplan::Plan plan;
    slgplan::plan_batch(views, ix->tune, in, plan);

    DeviceGuard g(ix->device);
    b = new slg_batch();
    b->idx = ix;
    b->snap = snap;
    b->nq = nq;
    b->k = k;
    b->strategy = strategy;
    b->q_postings.swap(plan.q_postings);
    b->n_postings = plan.n_postings;
    b->n_postings_essential = plan.n_postings_essential;
    b->n_postings_nonessential = plan.n_postings_nonessential;
    b->n_rounds = plan.n_rounds;
    b->max_terms = plan.max_terms;
    b->uniform = plan.uniform;
    b->multi = plan.multi;
    b->plan_batch = plan.plan_batch;
    b->nested = plan.nested;
    b->deep = plan.deep;
    b->pruned = plan.pruned;
    b->cand_mode = plan.cand_mode;
    b->n_sq = (uint32_t)plan.sqs.size();
    b->n_terms = (uint32_t)plan.terms.size();
    b->n_slices = (uint32_t)plan.slice_sq.size();
    b->n_boundaries = (uint32_t)plan.n_bnd;

    // ---- the descriptor image: one H2D copy the caller waits for, outside any lock, on an upload
    // stream of its own.  (Measured against an image copied asynchronously on the batch's stream in
    // front of the kernels: that variant served 5.2-7.4M queries/s from 4-8 caller threads where
    // this one serves 8.0-8.8M — the stream-ordered copy delays each batch's first kernel.)  The
    // staging image is pinned, so the copy is one DMA at PCIe speed (26 MB: 0.6 ms; from pageable
    // memory 1-6 ms), and comes from the index's free list (a fresh 26 MB vector per config-4
    // batch spent half of its 5 ms in page faults); it goes back when prepare returns.
    const size_t total = plan.image_bytes;
    struct ImageLease {
      BufPool *pool;
      void *p = nullptr;
      size_t bytes = 0;
      ~ImageLease() {
        if (p) pool->give_image(p, bytes);
      }
    } lease{&ix->pool};
    lease.p = ix->pool.take_image(total ? total : 16, &lease.bytes);
    if (!lease.p) throw SlgError(SLG_ERR_OOM, "pinned staging image: hipHostMalloc failed");
    plan.pack(static_cast<unsigned char *>(lease.p));
    b->d_desc.alloc_pooled(&ix->pool, total);
    {
      hipStream_t us = ix->upload_streams[std::hash<std::thread::id>()(std::this_thread::get_id()) %
                                          slg_index::kUploadStreams];
      SLG_HIP(hipMemcpyAsync(b->d_desc.p, lease.p, total, hipMemcpyHostToDevice, us));
      SLG_HIP(wait_stream(us));
    }
    unsigned char *db = b->d_desc.as<unsigned char>();
    b->d_sq = reinterpret_cast<const slg::RoundQuery *>(db + plan.o_sq);
    b->d_terms = reinterpret_cast<const slg::TermRef *>(db + plan.o_terms);
    b->d_slice_sq = reinterpret_cast<const uint32_t *>(db + plan.o_slice);
    b->d_slice_seg = reinterpret_cast<const uint32_t *>(db + plan.o_sseg);
    b->d_slice_order = reinterpret_cast<const uint32_t *>(db + plan.o_sord);
    b->d_queries = reinterpret_cast<const slg::QueryRef *>(db + plan.o_q);
    b->d_bnd_coarse = reinterpret_cast<const uint32_t *>(db + plan.o_bc);
    b->d_nodes = reinterpret_cast<const slg::PlanNode *>(db + plan.o_nodes);
    b->d_bounds.alloc_pooled(&ix->pool, (size_t)plan.n_bounds * 4);
    b->d_rdoc.alloc_pooled(&ix->pool, (size_t)plan.n_bnd * 4);
    b->d_slice_desc.alloc_pooled(&ix->pool, (size_t)b->n_slices * sizeof(slg::SliceDesc));
    if (b->cand_mode) {
      b->d_cand.alloc_pooled(&ix->pool, (size_t)(plan.cand_total + 1) * 8);
      b->d_slice_cbeg.alloc_pooled(&ix->pool, (size_t)b->n_slices * 8);
      b->d_slice_ccnt.alloc_pooled(&ix->pool, (size_t)b->n_slices * 4);
    } else {
      b->d_slice_tk.alloc_pooled(&ix->pool, (size_t)b->n_slices * k * 4);
      b->d_slice_doc.alloc_pooled(&ix->pool, (size_t)b->n_slices * k * 4);
    }
    b->d_q_scored.alloc_pooled(&ix->pool, (size_t)nq * 4);
    b->d_work_ctr.alloc_pooled(&ix->pool, (size_t)slg::kWorkQueues * slg::kWorkCtrStride * 4);
    // (from the pool like every per-batch buffer: a raw hipMalloc / hipFree per batch synchronises
    // the device and cost config 4's two-in-flight pipeline 60 %)
    if (b->pruned && !b->uniform && ix->tune.block_max) b->d_blk_skip.alloc_pooled(&ix->pool, ((size_t)nq + 1) * 8);
    if (!plan.q_filter.empty()) {
      b->d_q_filter.alloc_pooled(&ix->pool, (size_t)nq * 4);
      SLG_HIP(hipMemcpy(b->d_q_filter.p, plan.q_filter.data(), (size_t)nq * 4, hipMemcpyHostToDevice));
    }
    b->d_out.alloc_pooled(&ix->pool, ((size_t)nq * k * 3 + nq + 1) * 4);  // (+ the error word: MergeParams::out_flag)
    b->d_out_doc = b->d_out.as<uint32_t>();
    b->d_out_seg = b->d_out_doc + (size_t)nq * k;
    b->d_out_score = reinterpret_cast<float *>(b->d_out_seg + (size_t)nq * k);
    b->d_out_count = b->d_out_seg + (size_t)nq * k * 2;
    {
      std::lock_guard<std::mutex> lk(ix->mu);
      ix->live.push_back(b);
    }
  });
  if (rc != SLG_OK) {
    const std::string keep = g_last_error;
    const int keep_code = g_last_code;
    delete b;
    g_last_error = keep;
    g_last_code = keep_code;
    return nullptr;
  }
  return b;
}

#define SLG_REQUIRE_LIVE(b) \
  SLG_REQUIRE((b) != nullptr && (b)->idx != nullptr, "batch is NULL or its index was destroyed")

int slg_batch_run(slg_batch *b) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    slg_index *ix = b->idx;
    const IndexState &S = *b->snap;  // the state the batch was prepared on (not the index's current one)
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    hipStream_t st = batch_stream(b);
    b->launched = true;
    if (b->nq == 0) return;
    if (b->n_slices == 0) SLG_HIP(hipMemsetAsync(b->d_q_scored.p, 0, (size_t)b->nq * 4, st));
    if (b->n_slices > 0) {
      slg::RoundPartParams pp{};
      pp.sq = b->d_sq;
      pp.terms = b->d_terms;
      pp.n_sq = b->n_sq;
      pp.bnd_coarse = b->d_bnd_coarse;
      pp.segs = S.d_segs.as<slg::SegDev>();
      pp.bounds = b->d_bounds.as<uint32_t>();
      pp.rdoc = b->d_rdoc.as<uint32_t>();
      pp.q_scored = b->d_q_scored.as<uint32_t>();
      const bool skipping = b->d_blk_skip.p != nullptr;
      pp.skip_counts = skipping ? b->d_blk_skip.as<unsigned long long>() : nullptr;
      pp.slice_sq = b->d_slice_sq;
      pp.slice_order = b->d_slice_order;
      pp.slice_desc = b->d_slice_desc.as<slg::SliceDesc>();
      pp.nq = b->nq;
      // the blocked few-term kernel can cut its slices itself (slg_tuning.inline_cuts)
      const bool inline_cuts = b->uniform && ix->tune.uniform_kernel >= 4 &&
                               (ix->tune.inline_cuts >= 0 ? ix->tune.inline_cuts != 0 : true);
      pp.n_boundaries = inline_cuts ? 0u : b->n_boundaries;
      pp.n_slices = b->n_slices;
      pp.tpb_shift = b->max_terms <= 4 ? 2u : 3u;
      // few-term kernel: one wave per slice, or (slg_tuning.score_waves_per_simd) persistent waves that
      // pull slices from the batch's work queues (slg_score_uni4.hpp)
      const int score_kind =
          b->uniform ? uniform_kind(ix->tune.uniform_kernel, b->max_terms, b->plan_batch) : (b->pruned ? 3 : 2);
      uint32_t n_waves = b->n_slices;
      const bool persistent = (score_kind == 6 || score_kind == 7) && ix->tune.score_waves_per_simd != 0;
      if (persistent)
        n_waves = slg::u4_launch_blocks(kregs_for(b->k), (score_kind & 1) ? 8 : 4, score_kind >= 8, b->n_slices, ix->n_cu,
                                        ix->tune.score_waves_per_simd) *
                  (uint32_t)slg::kU4WavesPerBlock;
      pp.work_ctr = persistent ? b->d_work_ctr.as<uint32_t>() : nullptr;
      pp.n_waves = n_waves;
      const uint64_t pthreads = std::max<uint64_t>(
          std::max<uint64_t>((uint64_t)pp.n_boundaries << pp.tpb_shift, (uint64_t)b->nq + 1), b->n_slices);
      hipLaunchKernelGGL(slg::partition_rounds_kernel, dim3((uint32_t)((pthreads + 255) / 256)),
                         dim3(256), 0, st, pp);
      SLG_HIP(hipGetLastError());

      slg::RoundScoreParams sp{};
      sp.sq = b->d_sq;
      sp.terms = b->d_terms;
      sp.slice_sq = b->d_slice_sq;
      sp.slice_order = b->d_slice_order;
      sp.slice_desc = b->d_slice_desc.as<slg::SliceDesc>();
      sp.reject_table = S.d_reject_table.as<const uint32_t *>();
      sp.n_segs = (uint32_t)S.segs.size();
      sp.plan_batch = b->plan_batch ? (b->deep ? 4u : (b->nested ? 2u : 1u)) : 0u;
      sp.plan_nodes = b->d_nodes;
      sp.cand = b->d_cand.as<uint2>();
      sp.slice_cbeg = b->d_slice_cbeg.as<uint64_t>();
      sp.slice_ccnt = b->d_slice_ccnt.as<uint32_t>();
      sp.segs = S.d_segs.as<slg::SegDev>();
      sp.bounds = inline_cuts ? nullptr : b->d_bounds.as<uint32_t>();
      sp.rdoc = inline_cuts ? nullptr : b->d_rdoc.as<uint32_t>();
      sp.slice_tk = b->d_slice_tk.as<int32_t>();
      sp.slice_doc = b->d_slice_doc.as<uint32_t>();
      sp.q_scored = b->d_q_scored.as<uint32_t>();
      sp.n_slices = b->n_slices;
      sp.k = b->k;
      sp.block_skip = skipping ? 1u : 0u;
      sp.skip_counts = pp.skip_counts;
      sp.stamps = nullptr;
      sp.error_flag = ix->d_error_flag.as<uint32_t>();
      sp.work_ctr = pp.work_ctr;
      sp.n_waves = n_waves;
#ifdef SLG_STAMPS
      b->d_stamps.alloc((size_t)b->n_slices * 96);
      sp.stamps = b->d_stamps.as<unsigned long long>();
#endif
      std::pair<hipEvent_t, hipEvent_t> *ev = nullptr;
      if (ix->profile) {
        if (ix->prof_used == ix->prof_events.size()) {
          hipEvent_t a, c;
          SLG_HIP(hipEventCreate(&a));
          SLG_HIP(hipEventCreate(&c));
          ix->prof_events.emplace_back(a, c);
        }
        ev = &ix->prof_events[ix->prof_used++];
        SLG_HIP(hipEventRecord(ev->first, st));
      }
      launch_score(sp, score_kind, st);
      if (ev) SLG_HIP(hipEventRecord(ev->second, st));
    }
    if (b->k > 0 && b->cand_mode && b->n_slices > 0) {
      slg::SelectParams sp{};
      sp.queries = b->d_queries;
      sp.slice_seg = b->d_slice_seg;
      sp.slice_cbeg = b->d_slice_cbeg.as<uint64_t>();
      sp.slice_ccnt = b->d_slice_ccnt.as<uint32_t>();
      sp.cand = b->d_cand.as<uint2>();
      sp.segs = S.d_segs.as<slg::SegDev>();
      sp.q_filter = b->d_q_filter.as<uint32_t>();
      sp.reject_table = S.d_reject_table.as<const uint32_t *>();
      sp.n_segs = (uint32_t)S.segs.size();
      sp.out_doc = b->d_out_doc;
      sp.out_seg = b->d_out_seg;
      sp.out_score = b->d_out_score;
      sp.out_count = b->d_out_count;
      sp.nq = b->nq;
      sp.k = b->k;
      sp.error_flag = ix->d_error_flag.as<uint32_t>();
      sp.out_flag = b->d_out_count + b->nq;
      hipLaunchKernelGGL(slg::select_topk_kernel, dim3(b->nq), dim3(slg::kSelectThreads), 0, st, sp);
      SLG_HIP(hipGetLastError());
    } else if (b->k > 0) {
      slg::MergeParams mp{};
      mp.queries = b->d_queries;
      mp.slice_seg = b->d_slice_seg;
      mp.slice_tk = b->d_slice_tk.as<int32_t>();
      mp.slice_doc = b->d_slice_doc.as<uint32_t>();
      mp.out_doc = b->d_out_doc;
      mp.out_seg = b->d_out_seg;
      mp.out_score = b->d_out_score;
      mp.out_count = b->d_out_count;
      mp.nq = b->nq;
      mp.k = b->k;
      mp.error_flag = ix->d_error_flag.as<uint32_t>();
      mp.out_flag = b->d_out_count + b->nq;
      launch_merge(mp, st);
    } else {
      SLG_HIP(hipMemsetAsync(b->d_out_count, 0, ((size_t)b->nq + 1) * 4, st));  // (k = 0: nothing was scored)
    }
  });
}

int slg_batch_sync(slg_batch *b) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    DeviceGuard g(b->idx->device);
    SLG_HIP(hipStreamSynchronize(batch_stream(b)));
  });
}

// result blocks up to this size are fetched into pageable memory (see slg_batch_fetch)
static constexpr size_t kPageableFetchBytes = 256u << 10;

int slg_batch_fetch(slg_batch *b, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                    uint32_t *out_count, slg_stats *stats) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    SLG_REQUIRE(b->nq == 0 || (out_count != nullptr), "out_count is NULL");
    SLG_REQUIRE(b->nq == 0 || b->k == 0 || (out_doc && out_seg && out_score), "output array is NULL");
    slg_index *ix = b->idx;
    DeviceGuard g(ix->device);
    hipStream_t st;
    {  // (not held while waiting: other host threads keep launching their batches)
      std::lock_guard<std::mutex> lk(ix->mu);
      st = batch_stream(b);
    }
    const size_t n = (size_t)b->nq * b->k;
    std::vector<uint32_t> scored;
    std::vector<unsigned long long> skipped;
    if (b->nq) {
      // the results are one contiguous block doc | seg | score | count | error word: ONE D2H copy.
      // Large blocks (config 4: 10 MB) go into a PINNED staging image of the index's pool: a pageable
      // destination makes the runtime stage the copy itself, chunk by chunk behind a lock that every
      // caller thread's copies share.  Small ones (config 2: 46 KB) are copied straight into a pageable
      // image: for them the runtime's own path is the faster one (measured on one box, 8 caller threads,
      // 20-step regions: 12.1-12.5M against 10.4-11.0M queries/s with the pinned image and a separate
      // 4-byte copy of the error word).
      const size_t words = 3 * n + b->nq + 1;
      const size_t extra = stats ? (size_t)b->nq + ((b->d_blk_skip.p && b->launched) ? 2 * ((size_t)b->nq + 1) : 0) : 0;
      struct ImageLease {
        BufPool *pool;
        void *p = nullptr;
        size_t bytes = 0;
        ~ImageLease() {
          if (p) pool->give_image(p, bytes);
        }
      } lease{&ix->pool};
      std::vector<uint32_t> pageable;
      uint32_t *blk = nullptr;
      if ((words + extra) * 4 <= kPageableFetchBytes) {
        pageable.resize(words + extra);
        blk = pageable.data();
      } else {
        lease.p = ix->pool.take_image((words + extra) * 4, &lease.bytes);
        if (!lease.p) throw SlgError(SLG_ERR_OOM, "pinned staging image: hipHostMalloc failed");
        blk = static_cast<uint32_t *>(lease.p);
      }
      SLG_HIP(hipMemcpyAsync(blk, b->d_out.p, words * 4, hipMemcpyDeviceToHost, st));
      uint32_t *flagw = blk + words - 1;  // the index's error word as the batch's last kernel saw it
      uint32_t *const sblk = blk + words;
      if (stats) {
        SLG_HIP(hipMemcpyAsync(sblk, b->d_q_scored.p, (size_t)b->nq * 4, hipMemcpyDeviceToHost, st));
        if (b->d_blk_skip.p && b->launched)
          SLG_HIP(hipMemcpyAsync(sblk + b->nq, b->d_blk_skip.p, ((size_t)b->nq + 1) * 8, hipMemcpyDeviceToHost, st));
      }
      SLG_HIP(wait_stream(st));
      if (*flagw != 0u)
        throw SlgError(SLG_ERR_INTERNAL, "a scoring wave gave up on a round (chunk-loop guard): results are incomplete");
      if (n) {
        std::memcpy(out_doc, blk, n * 4);
        std::memcpy(out_seg, blk + n, n * 4);
        std::memcpy(out_score, blk + 2 * n, n * 4);
      }
      std::memcpy(out_count, blk + 3 * n, (size_t)b->nq * 4);
      if (stats) {
        scored.assign(sblk, sblk + b->nq);
        if (b->d_blk_skip.p && b->launched) {
          skipped.resize((size_t)b->nq + 1);
          std::memcpy(skipped.data(), sblk + b->nq, skipped.size() * 8);
        }
      }
    }
    if (stats)
      for (uint32_t q = 0; q < b->nq; q++) {
        // brute-force accounting: wand.rs:472 (postings_advanced += len), :500-503; with block
        // skipping, the postings that were never loaded are not counted as advanced over
        stats[q].postings_advanced = b->q_postings[q] - (skipped.empty() ? 0ull : skipped[q + 1]);
        stats[q].scored_docs = scored[q];
        stats[q].candidates_examined = scored[q];
      }
  });
}

int slg_batch_device_results(slg_batch *b, void **d_doc, void **d_seg, void **d_score,
                             void **d_count) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    if (d_doc) *d_doc = b->d_out_doc;
    if (d_seg) *d_seg = b->d_out_seg;
    if (d_score) *d_score = b->d_out_score;
    if (d_count) *d_count = b->d_out_count;
  });
}

int slg_batch_device_result_block(slg_batch *b, void **d_block, uint64_t *n_bytes) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    if (d_block) *d_block = b->d_out.p;
    if (n_bytes) *n_bytes = ((uint64_t)b->nq * b->k * 3 + b->nq) * 4;
  });
}

int slg_batch_info(const slg_batch *b, uint64_t *n_postings, uint32_t *n_slices,
                   uint64_t *algorithmic_bytes) {
  return guarded([&] {
    SLG_REQUIRE(b != nullptr, "batch is NULL");
    if (n_postings) *n_postings = b->n_postings;
    if (n_slices) *n_slices = b->n_slices;
    if (algorithmic_bytes) *algorithmic_bytes = 12ull * b->n_postings + 8ull * b->k * b->nq;
  });
}

int slg_batch_skip_counts(slg_batch *b, uint64_t *probed_postings, uint64_t *skipped_postings) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    DeviceGuard g(b->idx->device);
    unsigned long long c = 0ull;
    if (b->d_blk_skip.p) {
      SLG_HIP(hipStreamSynchronize(batch_stream(b)));
      SLG_HIP(hipMemcpy(&c, b->d_blk_skip.p, 8, hipMemcpyDeviceToHost));
    }
    if (probed_postings) *probed_postings = b->d_blk_skip.p ? b->n_postings_nonessential : 0ull;
    if (skipped_postings) *skipped_postings = c;
  });
}

#ifdef SLG_STAMPS
int slg_debug_read_stamps(slg_batch *b, unsigned long long *out, uint32_t n_slices) {
  return guarded([&] {
    SLG_HIP(hipStreamSynchronize(batch_stream(b)));
    SLG_HIP(hipMemcpy(out, b->d_stamps.p, (size_t)n_slices * 96, hipMemcpyDeviceToHost));
  });
}
#endif

int slg_batch_set_stream(slg_batch *b, void *hip_stream) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    slg_index *ix = b->idx;
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (b->launched) SLG_HIP(hipStreamSynchronize(batch_stream(b)));  // queued work finishes first
    b->own_stream_set = hip_stream != SLG_OWN_STREAM;
    b->stream = b->own_stream_set ? (hipStream_t)hip_stream : nullptr;
  });
}

void slg_batch_destroy(slg_batch *b) {
  if (!b) return;
  slg_index *ix = b->idx;
  if (!ix) {  // detached by slg_index_destroy: nothing left on the device
    delete b;
    return;
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(ix->device);
  hipStream_t st;
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    st = batch_stream(b);
    auto it = std::find(ix->live.begin(), ix->live.end(), b);
    if (it != ix->live.end()) {
      *it = ix->live.back();
      ix->live.pop_back();
    }
  }
  (void)wait_stream(st);
  for (hipEvent_t e : b->ev_shard)
    if (e) (void)hipEventDestroy(e);
  delete b;
  if (prev >= 0) (void)hipSetDevice(prev);
}

int slg_search_batch(slg_index *ix, const slg_query *queries, uint32_t nq, uint32_t k,
                     int strategy, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                     uint32_t *out_count, slg_stats *stats) {
  return slg_search_batch_filtered(ix, queries, nq, nullptr, k, strategy, out_doc, out_seg, out_score,
                                   out_count, stats);
}

int slg_search_batch_filtered(slg_index *ix, const slg_query *queries, uint32_t nq,
                              const int32_t *q_filter, uint32_t k, int strategy, uint32_t *out_doc,
                              uint32_t *out_seg, float *out_score, uint32_t *out_count,
                              slg_stats *stats) {
  slg_batch *b = nullptr;
  int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    SLG_REQUIRE(nq == 0 || queries != nullptr, "queries is NULL");
  });
  if (rc != SLG_OK) return rc;
  std::vector<uint32_t> offs(nq + 1, 0), tids;
  std::vector<float> ws;
  rc = guarded([&] {
    const size_t n_segs = ix->snapshot()->segs.size();
    for (uint32_t q = 0; q < nq; q++) {
      const slg_query &qq = queries[q];
      SLG_REQUIRE(qq.n_terms == 0 || (qq.term_ids && qq.weights), "query arrays are NULL");
      offs[q + 1] = offs[q] + qq.n_terms;
      tids.insert(tids.end(), qq.term_ids, qq.term_ids + (size_t)qq.n_terms * n_segs);
      ws.insert(ws.end(), qq.weights, qq.weights + qq.n_terms);
    }
  });
  if (rc != SLG_OK) return rc;
  b = slg_batch_prepare_filtered(ix, nq, offs.data(), tids.data(), ws.data(), q_filter, k, strategy);
  if (!b) return g_last_code;  // slg_batch_prepare set the thread-local error and its code
  rc = slg_batch_run(b);
  if (rc == SLG_OK) rc = slg_batch_fetch(b, out_doc, out_seg, out_score, out_count, stats);
  const std::string keep = g_last_error;
  slg_batch_destroy(b);
  g_last_error = keep;
  g_last_code = rc;
  return rc;
}

int slg_merge_shards_device(slg_index *ix, uint32_t n_shards, uint32_t nq, uint32_t k,
                            const uint32_t *d_doc, const uint32_t *d_seg, const float *d_score,
                            const uint32_t *d_count, uint32_t seg_stride, uint32_t *d_out_doc,
                            uint32_t *d_out_seg, float *d_out_score, uint32_t *d_out_count) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    if (k > SLG_MAX_K) throw SlgError(SLG_ERR_UNSUPPORTED, "k > SLG_MAX_K");
    if (nq == 0) return;
    SLG_REQUIRE(n_shards >= 1, "n_shards == 0");
    SLG_REQUIRE(d_count && d_out_count, "count arrays are NULL");
    SLG_REQUIRE(k == 0 || (d_doc && d_seg && d_score && d_out_doc && d_out_seg && d_out_score),
                "device arrays are NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (k == 0) {
      SLG_HIP(hipMemsetAsync(d_out_count, 0, (size_t)nq * 4, ix->stream));
      return;
    }
    slg::ShardMergeParams mp{};
    mp.doc = d_doc;
    mp.seg = d_seg;
    mp.score = d_score;
    mp.count = d_count;
    mp.out_doc = d_out_doc;
    mp.out_seg = d_out_seg;
    mp.out_score = d_out_score;
    mp.out_count = d_out_count;
    mp.n_shards = n_shards;
    mp.nq = nq;
    mp.k = k;
    mp.seg_stride = seg_stride;
    mp.arr_stride = (uint64_t)nq * k;
    mp.cnt_stride = nq;
    launch_shard_merge(mp, ix->stream);
  });
}

// ---- index sharding over RCCL (api/reader.rs:2670-2778 with segment = shard) --------
// One process (or host thread) per GPU holds the segments of its shard; every rank scores the same
// query batch, ONE ncclAllGather exchanges the contiguous per-rank result blocks
// doc | seg | score | count ((3k+1) * Q * 4 bytes) over xGMI, and every rank merges the world's rows
// by (score desc, segment_ord asc, doc asc), segment_ord = rank * segs_per_rank + local segment
// (query/sort.rs:80-93).  RCCL is bound at run time (dlopen): a single-GPU user of the library does
// not need it, and inside a PyTorch process the librccl torch has loaded is the one used.
extern "C++" {
namespace {
typedef int (*nccl_get_uid_fn)(void *);
struct NcclUid {  // ncclUniqueId (rccl.h): passed to ncclCommInitRank BY VALUE
  char internal[128];
};
typedef int (*nccl_destroy_fn)(void *);
typedef int (*nccl_allgather_fn)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_fn)(int);
struct RcclApi {
  void *handle = nullptr;
  nccl_get_uid_fn get_uid = nullptr;
  int (*init_rank)(void **, int, NcclUid, int) = nullptr;
  nccl_destroy_fn destroy = nullptr;
  nccl_allgather_fn allgather = nullptr;
  nccl_errstr_fn errstr = nullptr;
  std::string error;
};
RcclApi &rccl() {
  static RcclApi api = [] {
    RcclApi a;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)  // a copy that is already resident (PyTorch's) first
      if (!a.handle) a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : {"librccl.so.1", "librccl.so"})
      if (!a.handle) a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!a.handle) {
      const char *e = dlerror();  // (one call: dlerror() clears the state it reports)
      a.error = std::string("librccl not found: ") + (e ? e : "");
      return a;
    }
    a.get_uid = (nccl_get_uid_fn)dlsym(a.handle, "ncclGetUniqueId");
    a.init_rank = (int (*)(void **, int, NcclUid, int))dlsym(a.handle, "ncclCommInitRank");
    a.destroy = (nccl_destroy_fn)dlsym(a.handle, "ncclCommDestroy");
    a.allgather = (nccl_allgather_fn)dlsym(a.handle, "ncclAllGather");
    a.errstr = (nccl_errstr_fn)dlsym(a.handle, "ncclGetErrorString");
    if (!a.get_uid || !a.init_rank || !a.destroy || !a.allgather || !a.errstr) a.error = "librccl lacks a required symbol";
    return a;
  }();
  if (!api.error.empty()) throw SlgError(SLG_ERR_UNSUPPORTED, api.error);
  return api;
}
void nccl_check(int rc, const char *what) {
  if (rc != 0) throw SlgError(SLG_ERR_DEVICE, std::string(what) + ": " + rccl().errstr(rc));
}
constexpr int kNcclInt32 = 2;  // ncclDataType_t ncclInt32 (rccl.h)
}  // namespace
}  // extern "C++"

struct slg_shard_group {
  slg_index *idx = nullptr;
  int rank = 0, world = 1;
  uint32_t segs_per_rank = 1;
  void *comm = nullptr;  // ncclComm_t
  // Collectives on one communicator must be issued in the same order on every rank.  The group owns
  // the stream they run on and hands out turns: sharded run number `seq` issues its all-gather when
  // the runs 0 .. seq-1 have issued theirs, whatever host thread or batch stream it comes from (the
  // batch's stream and the collective stream are tied together with events).
  hipStream_t coll_stream = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t next_seq = 0;   // the run whose collective may be issued next
  uint64_t auto_seq = 0;   // tickets of slg_batch_run_sharded (call order)
  // device time of the sharded runs fetched so far (slg_index profiling on): local kernels, all-gather
  // (incl. waiting for the slowest rank), merge; ms sums and the number of runs
  double ms_kernels = 0.0, ms_gather = 0.0, ms_merge = 0.0;
  uint64_t n_timed = 0;
};

int slg_shard_unique_id(void *out, size_t out_bytes) {
  return guarded([&] {
    SLG_REQUIRE(out != nullptr && out_bytes >= SLG_SHARD_UNIQUE_ID_BYTES, "unique id buffer is NULL or too small");
    NcclUid id;
    nccl_check(rccl().get_uid(&id), "ncclGetUniqueId");
    std::memcpy(out, id.internal, sizeof(id.internal));
  });
}

slg_shard_group *slg_shard_group_create(slg_index *ix, int rank, int world, const void *unique_id,
                                        uint32_t segs_per_rank) {
  slg_shard_group *g = nullptr;
  const int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr && unique_id != nullptr, "index or unique id is NULL");
    SLG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank outside [0, world)");
    SLG_REQUIRE(segs_per_rank >= ix->snapshot()->segs.size(), "segs_per_rank is smaller than this shard's segment count");
    DeviceGuard dg(ix->device);
    NcclUid id;
    std::memcpy(id.internal, unique_id, sizeof(id.internal));
    g = new slg_shard_group();
    g->idx = ix;
    g->rank = rank;
    g->world = world;
    g->segs_per_rank = segs_per_rank;
    nccl_check(rccl().init_rank(&g->comm, world, id, rank), "ncclCommInitRank");
    SLG_HIP(hipStreamCreateWithFlags(&g->coll_stream, hipStreamNonBlocking));
  });
  if (rc != SLG_OK) {
    const std::string keep = g_last_error;
    delete g;
    g_last_error = keep;
    g_last_code = rc;
    return nullptr;
  }
  return g;
}

void slg_shard_group_destroy(slg_shard_group *g) {
  if (!g) return;
  if (g->comm) {
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(g->idx->device);
    (void)hipDeviceSynchronize();
    try {
      (void)rccl().destroy(g->comm);
    } catch (...) {
    }
    if (g->coll_stream) (void)hipStreamDestroy(g->coll_stream);
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  delete g;
}

namespace {
int run_sharded_impl(slg_batch *b, slg_shard_group *g, bool have_seq, uint64_t seq, uint32_t *out_doc,
                     uint32_t *out_seg, float *out_score, uint32_t *out_count) {
  int rc = guarded([&] {
    SLG_REQUIRE_LIVE(b);
    SLG_REQUIRE(g != nullptr && g->idx == b->idx, "shard group is NULL or belongs to another index");
    SLG_REQUIRE(g->segs_per_rank >= b->snap->segs.size(), "the shard grew beyond the group's segs_per_rank");
  });
  if (rc != SLG_OK) return rc;
  if (!have_seq) {  // call order = the order on every rank, if one thread issues the runs
    std::lock_guard<std::mutex> lk(g->mu);
    seq = g->auto_seq++;
  }
  slg_index *ix = b->idx;
  const bool timed = ix->profile;
  rc = guarded([&] {
    DeviceGuard dg(ix->device);
    for (int i = 0; i < 4; i++)
      if (!b->ev_shard[i]) SLG_HIP(hipEventCreateWithFlags(&b->ev_shard[i], timed ? hipEventDefault : hipEventDisableTiming));
    if (timed) {
      std::lock_guard<std::mutex> lk(ix->mu);
      SLG_HIP(hipEventRecord(b->ev_shard[0], batch_stream(b)));
    }
  });
  if (rc == SLG_OK) rc = slg_batch_run(b);  // this rank's segments: partition + score + merge, on the batch's stream
  // From here on the turn MUST be passed on, error or not: the runs behind this one wait for it.
  int rc2 = guarded([&] {
    const size_t n = (size_t)b->nq * b->k, blk = 3 * n + b->nq;  // words of one rank's block
    DeviceGuard dg(ix->device);
    hipStream_t st;
    {
      std::lock_guard<std::mutex> lk(ix->mu);
      st = batch_stream(b);
      if (rc == SLG_OK && b->nq) {
        if (!b->d_gather.p) b->d_gather.alloc_pooled(&ix->pool, (size_t)g->world * blk * 4);
        if (!b->d_merged.p) b->d_merged.alloc_pooled(&ix->pool, blk * 4);
        SLG_HIP(hipEventRecord(b->ev_shard[1], st));  // the local result block is complete
      }
    }
    {
      // my turn: ONE collective, every rank's contiguous block in rank order, on the group's stream
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv.wait(lk, [&] { return g->next_seq == seq; });
      struct PassOn {
        slg_shard_group *g;
        std::unique_lock<std::mutex> &lk;
        ~PassOn() {
          g->next_seq++;
          lk.unlock();
          g->cv.notify_all();
        }
      } pass{g, lk};
      if (rc != SLG_OK || b->nq == 0) return;
      SLG_HIP(hipStreamWaitEvent(g->coll_stream, b->ev_shard[1], 0));
      nccl_check(rccl().allgather(b->d_out.p, b->d_gather.p, blk, kNcclInt32, g->comm, g->coll_stream), "ncclAllGather");
      SLG_HIP(hipEventRecord(b->ev_shard[2], g->coll_stream));
    }
    {
      std::lock_guard<std::mutex> lk(ix->mu);
      SLG_HIP(hipStreamWaitEvent(st, b->ev_shard[2], 0));
      uint32_t *m = b->d_merged.as<uint32_t>();
      const uint32_t *gb = b->d_gather.as<uint32_t>();
      if (b->k == 0) {
        SLG_HIP(hipMemsetAsync(m, 0, blk * 4, st));
      } else {
        slg::ShardMergeParams mp{};
        mp.doc = gb;
        mp.seg = gb + n;
        mp.score = reinterpret_cast<const float *>(gb + 2 * n);
        mp.count = gb + 3 * n;
        mp.out_doc = m;
        mp.out_seg = m + n;
        mp.out_score = reinterpret_cast<float *>(m + 2 * n);
        mp.out_count = m + 3 * n;
        mp.n_shards = (uint32_t)g->world;
        mp.nq = b->nq;
        mp.k = b->k;
        mp.seg_stride = g->segs_per_rank;
        mp.arr_stride = blk;
        mp.cnt_stride = blk;
        launch_shard_merge(mp, st);
      }
      if (timed) SLG_HIP(hipEventRecord(b->ev_shard[3], st));
      b->shard_group = g;
      b->shard_timed = timed;
    }
  });
  if (rc == SLG_OK) rc = rc2;
  // merged top-k to the caller's host arrays (else: slg_batch_fetch_sharded / _device_results later)
  if (rc == SLG_OK && out_count) rc = slg_batch_fetch_sharded(b, out_doc, out_seg, out_score, out_count);
  return rc;
}
}  // namespace

int slg_batch_run_sharded(slg_batch *b, slg_shard_group *g, uint32_t *out_doc, uint32_t *out_seg,
                          float *out_score, uint32_t *out_count) {
  return run_sharded_impl(b, g, false, 0, out_doc, out_seg, out_score, out_count);
}

int slg_batch_run_sharded_seq(slg_batch *b, slg_shard_group *g, uint64_t seq, uint32_t *out_doc, uint32_t *out_seg,
                              float *out_score, uint32_t *out_count) {
  return run_sharded_impl(b, g, true, seq, out_doc, out_seg, out_score, out_count);
}

int slg_shard_group_skip_seq(slg_shard_group *g, uint64_t seq) {
  return guarded([&] {
    SLG_REQUIRE(g != nullptr, "shard group is NULL");
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return g->next_seq == seq; });
    g->next_seq++;
    lk.unlock();
    g->cv.notify_all();
  });
}

int slg_shard_group_stats(slg_shard_group *g, double *ms_kernels, double *ms_gather, double *ms_merge, uint64_t *n_runs) {
  return guarded([&] {
    SLG_REQUIRE(g != nullptr, "shard group is NULL");
    std::lock_guard<std::mutex> lk(g->mu);
    if (ms_kernels) *ms_kernels = g->ms_kernels;
    if (ms_gather) *ms_gather = g->ms_gather;
    if (ms_merge) *ms_merge = g->ms_merge;
    if (n_runs) *n_runs = g->n_timed;
    g->ms_kernels = g->ms_gather = g->ms_merge = 0.0;
    g->n_timed = 0;
  });
}

int slg_batch_fetch_sharded(slg_batch *b, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                            uint32_t *out_count) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    SLG_REQUIRE(b->nq == 0 || out_count != nullptr, "out_count is NULL");
    SLG_REQUIRE(b->nq == 0 || b->k == 0 || (out_doc && out_seg && out_score), "output array is NULL");
    if (b->nq == 0) return;
    SLG_REQUIRE(b->d_merged.p != nullptr, "slg_batch_run_sharded has not run on this batch");
    slg_index *ix = b->idx;
    DeviceGuard dg(ix->device);
    hipStream_t st;
    {
      std::lock_guard<std::mutex> lk(ix->mu);
      st = batch_stream(b);
    }
    const size_t n = (size_t)b->nq * b->k, blk = 3 * n + b->nq;
    struct ImageLease {  // pinned staging (see slg_batch_fetch)
      BufPool *pool;
      void *p = nullptr;
      size_t bytes = 0;
      ~ImageLease() {
        if (p) pool->give_image(p, bytes);
      }
    } lease{&ix->pool};
    lease.p = ix->pool.take_image(blk * 4, &lease.bytes);
    if (!lease.p) throw SlgError(SLG_ERR_OOM, "pinned staging image: hipHostMalloc failed");
    uint32_t *h = static_cast<uint32_t *>(lease.p);
    SLG_HIP(hipMemcpyAsync(h, b->d_merged.p, blk * 4, hipMemcpyDeviceToHost, st));
    SLG_HIP(wait_stream(st));
    if (b->shard_timed && b->shard_group) {  // device time of this run's three phases
      float k_ms = 0.0f, g_ms = 0.0f, m_ms = 0.0f;
      if (hipEventElapsedTime(&k_ms, b->ev_shard[0], b->ev_shard[1]) == hipSuccess &&
          hipEventElapsedTime(&g_ms, b->ev_shard[1], b->ev_shard[2]) == hipSuccess &&
          hipEventElapsedTime(&m_ms, b->ev_shard[2], b->ev_shard[3]) == hipSuccess) {
        std::lock_guard<std::mutex> lk(b->shard_group->mu);
        b->shard_group->ms_kernels += k_ms;
        b->shard_group->ms_gather += g_ms;
        b->shard_group->ms_merge += m_ms;
        b->shard_group->n_timed++;
      }
      b->shard_timed = false;
    }
    if (n) {
      std::memcpy(out_doc, h, n * 4);
      std::memcpy(out_seg, h + n, n * 4);
      std::memcpy(out_score, h + 2 * n, n * 4);
    }
    std::memcpy(out_count, h + 3 * n, (size_t)b->nq * 4);
  });
}

int slg_batch_sharded_device_results(slg_batch *b, void **d_doc, void **d_seg, void **d_score, void **d_count) {
  return guarded([&] {
    SLG_REQUIRE_LIVE(b);
    SLG_REQUIRE(b->d_merged.p != nullptr || b->nq == 0, "slg_batch_run_sharded has not run on this batch");
    const size_t n = (size_t)b->nq * b->k;
    uint32_t *m = b->d_merged.as<uint32_t>();
    if (d_doc) *d_doc = m;
    if (d_seg) *d_seg = m ? m + n : nullptr;
    if (d_score) *d_score = m ? m + 2 * n : nullptr;
    if (d_count) *d_count = m ? m + 3 * n : nullptr;
  });
}

int slg_profile_enable(slg_index *ix, int on) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->profile = on != 0;
  });
}

int slg_profile_read(slg_index *ix, uint32_t *n_launches, float *total_ms) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    float sum = 0.0f;
    for (size_t i = 0; i < ix->prof_used; i++) {
      SLG_HIP(hipEventSynchronize(ix->prof_events[i].second));
      float ms = 0.0f;
      SLG_HIP(hipEventElapsedTime(&ms, ix->prof_events[i].first, ix->prof_events[i].second));
      sum += ms;
    }
    if (n_launches) *n_launches = (uint32_t)ix->prof_used;
    if (total_ms) *total_ms = sum;
    ix->prof_used = 0;
  });
}

namespace {
// slg_batch_rerank_device runs the rerank kernels on the batch's stream: the launch sites below take
// the stream from here when it is set (this thread only, for the duration of that call)
thread_local bool g_rerank_stream_set = false;
thread_local hipStream_t g_rerank_stream = nullptr;
inline hipStream_t rerank_stream(slg_index *ix) { return g_rerank_stream_set ? g_rerank_stream : ix->stream; }
}  // namespace

int slg_rerank_batch_device(slg_index *ix, uint32_t nq, const float *d_qvecs, const float *d_alpha,
                            const uint32_t *d_cand_doc, const uint32_t *d_cand_seg,
                            const float *d_cand_bm25, const uint32_t *d_cand_count,
                            uint32_t max_cand, uint32_t k_out, uint32_t *d_out_doc,
                            uint32_t *d_out_seg, float *d_out_score, float *d_out_vec_score,
                            uint32_t *d_out_count) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    if (k_out > SLG_MAX_RERANK_K) throw SlgError(SLG_ERR_UNSUPPORTED, "k_out > SLG_MAX_RERANK_K");
    if (nq == 0) return;
    SLG_REQUIRE(d_qvecs && d_alpha && d_cand_count && d_out_count, "device arrays are NULL");
    SLG_REQUIRE(max_cand == 0 || (d_cand_doc && d_cand_seg && d_cand_bm25), "candidate arrays are NULL");
    SLG_REQUIRE(k_out == 0 || (d_out_doc && d_out_seg && d_out_score), "output arrays are NULL");
    const auto S = ix->snapshot();  // (a retired state waits for the device before its tables go)
    uint32_t dim = 0;
    for (auto &s : S->segs)
      if (s->store->vec_dim) {
        SLG_REQUIRE(dim == 0 || dim == s->store->vec_dim, "segments disagree on vec_dim");
        dim = s->store->vec_dim;
      }
    if (dim == 0) throw SlgError(SLG_ERR_UNSUPPORTED, "index has no vector field");
    if (max_cand > slg::kRerankMaxCand)
      throw SlgError(SLG_ERR_UNSUPPORTED, "max_cand > " + std::to_string(slg::kRerankMaxCand));
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    slg::RerankParams rp{};
    rp.vsegs = S->d_vsegs.as<slg::VecSegDev>();
    rp.n_segs = (uint32_t)S->segs.size();
    rp.dim = dim;
    rp.qvecs = d_qvecs;
    rp.alpha = d_alpha;
    rp.cand_doc = d_cand_doc;
    rp.cand_seg = d_cand_seg;
    rp.cand_bm25 = d_cand_bm25;
    rp.cand_count = d_cand_count;
    rp.max_cand = max_cand;
    rp.k_out = k_out;
    rp.out_doc = d_out_doc;
    rp.out_seg = d_out_seg;
    rp.out_score = d_out_score;
    rp.out_vec = d_out_vec_score;
    rp.out_count = d_out_count;
    rp.nq = nq;
    SLG_HIP(slg::launch_rerank(rp, kregs_for(k_out ? k_out : 1), rerank_stream(ix)));
  });
}

int slg_rerank_multi_batch_device(slg_index *ix, uint32_t nq, uint32_t n_clauses, const float *d_qvecs,
                                  const float *d_alpha, const float *d_boost, const uint32_t *d_cand_doc,
                                  const uint32_t *d_cand_seg, const float *d_cand_bm25,
                                  const uint32_t *d_cand_count, uint32_t max_cand, uint32_t k_out,
                                  uint32_t *d_out_doc, uint32_t *d_out_seg, float *d_out_score,
                                  float *d_out_vec_score, uint32_t *d_out_count) {
  if (n_clauses == 1 && d_boost == nullptr)  // one clause: the GEMV-shaped VALU kernel
    return slg_rerank_batch_device(ix, nq, d_qvecs, d_alpha, d_cand_doc, d_cand_seg, d_cand_bm25,
                                   d_cand_count, max_cand, k_out, d_out_doc, d_out_seg, d_out_score,
                                   d_out_vec_score, d_out_count);
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    if (n_clauses < 1 || n_clauses > SLG_MAX_VECTOR_CLAUSES)
      throw SlgError(SLG_ERR_UNSUPPORTED, "n_clauses outside 1..SLG_MAX_VECTOR_CLAUSES");
    if (k_out > SLG_MAX_RERANK_K) throw SlgError(SLG_ERR_UNSUPPORTED, "k_out > SLG_MAX_RERANK_K");
    if (nq == 0) return;
    SLG_REQUIRE(d_qvecs && d_alpha && d_cand_count && d_out_count, "device arrays are NULL");
    SLG_REQUIRE(max_cand == 0 || (d_cand_doc && d_cand_seg && d_cand_bm25), "candidate arrays are NULL");
    SLG_REQUIRE(k_out == 0 || (d_out_doc && d_out_seg && d_out_score), "output arrays are NULL");
    const auto S = ix->snapshot();
    uint32_t dim = 0;
    int32_t metric = -1;
    for (auto &s : S->segs) {
      if (!s->store->vec_dim) continue;
      SLG_REQUIRE(dim == 0 || dim == s->store->vec_dim, "segments disagree on vec_dim");
      SLG_REQUIRE(metric < 0 || metric == s->store->vec_metric, "segments disagree on the vector metric");
      dim = s->store->vec_dim;
      metric = s->store->vec_metric;
    }
    if (dim == 0) throw SlgError(SLG_ERR_UNSUPPORTED, "index has no vector field");
    for (auto &s : S->segs)
      if (!s->store->vec_dim) throw SlgError(SLG_ERR_UNSUPPORTED, "multi-clause rerank needs the vector field in every segment");
    if (slg::rerank_multi_lds_floats(n_clauses, dim, max_cand) > slg::kRerankMultiLdsFloats)
      throw SlgError(SLG_ERR_UNSUPPORTED, "n_clauses * (dim + max_cand) exceeds the LDS budget of the multi-clause rerank");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    slg::RerankMultiParams mp{};
    slg::RerankParams &rp = mp.base;
    rp.vsegs = S->d_vsegs.as<slg::VecSegDev>();
    rp.n_segs = (uint32_t)S->segs.size();
    rp.dim = dim;
    rp.qvecs = d_qvecs;
    rp.alpha = d_alpha;
    rp.cand_doc = d_cand_doc;
    rp.cand_seg = d_cand_seg;
    rp.cand_bm25 = d_cand_bm25;
    rp.cand_count = d_cand_count;
    rp.max_cand = max_cand;
    rp.k_out = k_out;
    rp.out_doc = d_out_doc;
    rp.out_seg = d_out_seg;
    rp.out_score = d_out_score;
    rp.out_vec = d_out_vec_score;
    rp.out_count = d_out_count;
    rp.nq = nq;
    mp.boost = d_boost;
    mp.n_clauses = n_clauses;
    mp.q_stride = dim + 4;
    SLG_HIP(slg::launch_rerank_multi(mp, kregs_for(k_out ? k_out : 1), rerank_stream(ix)));
  });
}

int slg_batch_rerank_device(slg_batch *b, uint32_t n_clauses, const float *d_qvecs, const float *d_alpha,
                            const float *d_boost, uint32_t k_out, uint32_t *d_out_doc, uint32_t *d_out_seg,
                            float *d_out_score, float *d_out_vec_score, uint32_t *d_out_count) {
  int rc = guarded([&] { SLG_REQUIRE_LIVE(b); });
  if (rc != SLG_OK) return rc;
  slg_index *ix = b->idx;
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    g_rerank_stream = batch_stream(b);
  }
  g_rerank_stream_set = true;
  rc = slg_rerank_multi_batch_device(ix, b->nq, n_clauses, d_qvecs, d_alpha, d_boost, b->d_out_doc, b->d_out_seg,
                                     b->d_out_score, b->d_out_count, b->k, k_out, d_out_doc, d_out_seg, d_out_score,
                                     d_out_vec_score, d_out_count);
  g_rerank_stream_set = false;
  return rc;
}

int slg_index_add_vector_field(slg_index *ix, const slg_vector_field_desc *per_segment, uint32_t n_segs) {
  int id = 0;
  const int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr && per_segment != nullptr, "index or descriptors are NULL");
    std::lock_guard<std::mutex> ulk(ix->update_mu);
    const auto cur = ix->snapshot();
    SLG_REQUIRE(n_segs == cur->segs.size(), "one descriptor per segment of the index is required");
    auto vf = std::make_shared<VecFieldHost>();
    for (uint32_t s = 0; s < n_segs; s++) {
      const slg_vector_field_desc &d = per_segment[s];
      if (!d.vec_dim) continue;
      SLG_REQUIRE(d.vec_offsets != nullptr && (d.vec_rows == 0 || d.vec_values != nullptr), "vector arrays are NULL");
      SLG_REQUIRE(d.vec_metric == SLG_METRIC_COSINE || d.vec_metric == SLG_METRIC_L2, "unknown vector metric");
      SLG_REQUIRE(vf->dim == 0 || (vf->dim == d.vec_dim && vf->metric == d.vec_metric),
                  "segments disagree on the field's dimension or metric");
      vf->dim = d.vec_dim;
      vf->metric = d.vec_metric;
      const uint32_t nd = cur->segs[s]->n_docs;
      for (uint32_t i = 0; i < nd; i++)
        SLG_REQUIRE(d.vec_offsets[i] == SLG_NO_VECTOR || d.vec_offsets[i] < d.vec_rows, "vector offset past vec_rows");
    }
    SLG_REQUIRE(vf->dim != 0, "no segment has vectors in this field");
    DeviceGuard g(ix->device);
    vf->per_seg.resize(n_segs);
    for (uint32_t s = 0; s < n_segs; s++) {
      const slg_vector_field_desc &d = per_segment[s];
      if (!d.vec_dim) continue;
      auto vs = std::make_shared<VecSegStore>();
      const size_t ob = (size_t)cur->segs[s]->n_docs * 4, vb = (size_t)d.vec_rows * d.vec_dim * 4;
      vs->offsets.alloc(ob, &ix->pool);
      if (ob) SLG_HIP(hipMemcpy(vs->offsets.p, d.vec_offsets, ob, hipMemcpyHostToDevice));
      vs->values.alloc(vb, &ix->pool);
      if (vb) SLG_HIP(hipMemcpy(vs->values.p, d.vec_values, vb, hipMemcpyHostToDevice));
      vs->dim = d.vec_dim;
      vf->per_seg[s] = std::move(vs);
    }
    auto ns = copy_state(*cur);
    ns->generation = cur->generation;
    ns->vfields.push_back(std::move(vf));
    finish_state(ix, *ns);
    id = (int)ns->vfields.size();
    publish(ix, std::move(ns));
  });
  return rc == SLG_OK ? id : rc;
}

namespace {
// dimension / metric / device stores of vector field f (0: the field of the segment descriptors)
void field_facts(const IndexState &S, uint32_t f, uint32_t *dim, int32_t *metric, const slg::VecSegDev **vsegs) {
  if (f == 0) {
    uint32_t d = 0;
    int32_t m = -1;
    for (auto &s : S.segs) {
      if (!s->store->vec_dim) continue;
      SLG_REQUIRE(d == 0 || d == s->store->vec_dim, "segments disagree on vec_dim");
      SLG_REQUIRE(m < 0 || m == s->store->vec_metric, "segments disagree on the vector metric");
      d = s->store->vec_dim;
      m = s->store->vec_metric;
    }
    if (d == 0) throw SlgError(SLG_ERR_UNSUPPORTED, "index has no vector field 0");
    *dim = d;
    *metric = m;
    *vsegs = S.d_vsegs.as<slg::VecSegDev>();
    return;
  }
  SLG_REQUIRE(f <= S.vfields.size(), "unknown vector field id");
  const VecFieldHost &vf = *S.vfields[f - 1];
  *dim = vf.dim;
  *metric = vf.metric;
  *vsegs = vf.d_vsegs.as<slg::VecSegDev>();
}
}  // namespace

int slg_rerank_fields_batch_device(slg_index *ix, uint32_t nq, uint32_t n_clauses, const uint32_t *clause_field,
                                   const float *d_qvecs, const float *d_alpha, const float *d_boost,
                                   const uint32_t *d_cand_doc, const uint32_t *d_cand_seg,
                                   const float *d_cand_bm25, const uint32_t *d_cand_count, uint32_t max_cand,
                                   uint32_t k_out, uint32_t *d_out_doc, uint32_t *d_out_seg, float *d_out_score,
                                   float *d_out_vec_score, uint32_t *d_out_count) {
  return guarded([&] {
    SLG_REQUIRE(ix != nullptr && clause_field != nullptr, "index or clause_field is NULL");
    if (n_clauses < 1 || n_clauses > SLG_MAX_VECTOR_CLAUSES)
      throw SlgError(SLG_ERR_UNSUPPORTED, "n_clauses outside 1..SLG_MAX_VECTOR_CLAUSES");
    if (k_out > SLG_MAX_RERANK_K) throw SlgError(SLG_ERR_UNSUPPORTED, "k_out > SLG_MAX_RERANK_K");
    if (nq == 0) return;
    SLG_REQUIRE(d_qvecs && d_alpha && d_cand_count && d_out_count, "device arrays are NULL");
    SLG_REQUIRE(max_cand == 0 || (d_cand_doc && d_cand_seg && d_cand_bm25), "candidate arrays are NULL");
    SLG_REQUIRE(k_out == 0 || (d_out_doc && d_out_seg && d_out_score), "output arrays are NULL");
    const auto S = ix->snapshot();
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    slg::RerankFieldsParams fp{};
    uint32_t qf = 0;
    for (uint32_t c = 0; c < n_clauses; c++) {
      field_facts(*S, clause_field[c], &fp.cdim[c], &fp.cmetric[c], &fp.cvsegs[c]);
      fp.coff[c] = qf;
      qf += fp.cdim[c];
    }
    if (slg::rerank_fields_lds_floats(n_clauses, qf, max_cand) > slg::kRerankMultiLdsFloats)
      throw SlgError(SLG_ERR_UNSUPPORTED, "clause vectors + n_clauses * max_cand exceed the LDS budget of the rerank");
    slg::RerankParams &rp = fp.base;
    rp.vsegs = nullptr;
    rp.n_segs = (uint32_t)S->segs.size();
    rp.dim = 0;
    rp.qvecs = d_qvecs;
    rp.alpha = d_alpha;
    rp.cand_doc = d_cand_doc;
    rp.cand_seg = d_cand_seg;
    rp.cand_bm25 = d_cand_bm25;
    rp.cand_count = d_cand_count;
    rp.max_cand = max_cand;
    rp.k_out = k_out;
    rp.out_doc = d_out_doc;
    rp.out_seg = d_out_seg;
    rp.out_score = d_out_score;
    rp.out_vec = d_out_vec_score;
    rp.out_count = d_out_count;
    rp.nq = nq;
    fp.boost = d_boost;
    fp.n_clauses = n_clauses;
    fp.q_floats = qf;
    SLG_HIP(slg::launch_rerank_fields(fp, kregs_for(k_out ? k_out : 1), ix->stream));
  });
}

int slg_rerank_fields_batch(slg_index *ix, uint32_t nq, uint32_t n_clauses, const uint32_t *clause_field,
                            const float *qvecs, const float *alpha, const float *boost,
                            const uint32_t *cand_doc, const uint32_t *cand_seg, const float *cand_bm25,
                            const uint32_t *cand_count, uint32_t max_cand, uint32_t k_out, uint32_t *out_doc,
                            uint32_t *out_seg, float *out_score, float *out_vec_score, uint32_t *out_count) {
  uint32_t qf = 0;
  int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr && clause_field != nullptr, "index or clause_field is NULL");
    if (n_clauses < 1 || n_clauses > SLG_MAX_VECTOR_CLAUSES)
      throw SlgError(SLG_ERR_UNSUPPORTED, "n_clauses outside 1..SLG_MAX_VECTOR_CLAUSES");
    if (nq == 0) return;
    SLG_REQUIRE(qvecs && alpha && cand_count && out_count, "host arrays are NULL");
    SLG_REQUIRE(max_cand == 0 || (cand_doc && cand_seg && cand_bm25), "candidate arrays are NULL");
    SLG_REQUIRE(k_out == 0 || (out_doc && out_seg && out_score), "output arrays are NULL");
    const auto S = ix->snapshot();
    for (uint32_t c = 0; c < n_clauses; c++) {
      uint32_t d;
      int32_t m;
      const slg::VecSegDev *v;
      field_facts(*S, clause_field[c], &d, &m, &v);
      qf += d;
    }
  });
  if (rc != SLG_OK || nq == 0) return rc;
  DevBuf dq, da, db, dcd, dcs, dcb, dcc, dod, dos, dosc, dov, doc_;
  const size_t nc = (size_t)nq * max_cand, no = (size_t)nq * k_out, nqc = (size_t)nq * n_clauses;
  rc = guarded([&] {
    DeviceGuard g(ix->device);
    hipStream_t st = ix->stream;
    dq.alloc_pooled(&ix->pool, (size_t)nq * qf * 4);
    da.alloc_pooled(&ix->pool, nqc * 4);
    if (boost) db.alloc_pooled(&ix->pool, nqc * 4);
    dcd.alloc_pooled(&ix->pool, nc * 4);
    dcs.alloc_pooled(&ix->pool, nc * 4);
    dcb.alloc_pooled(&ix->pool, nc * 4);
    dcc.alloc_pooled(&ix->pool, (size_t)nq * 4);
    dod.alloc_pooled(&ix->pool, no * 4);
    dos.alloc_pooled(&ix->pool, no * 4);
    dosc.alloc_pooled(&ix->pool, no * 4);
    dov.alloc_pooled(&ix->pool, no * 4);
    doc_.alloc_pooled(&ix->pool, (size_t)nq * 4);
    SLG_HIP(hipMemcpyAsync(dq.p, qvecs, (size_t)nq * qf * 4, hipMemcpyHostToDevice, st));
    SLG_HIP(hipMemcpyAsync(da.p, alpha, nqc * 4, hipMemcpyHostToDevice, st));
    if (boost) SLG_HIP(hipMemcpyAsync(db.p, boost, nqc * 4, hipMemcpyHostToDevice, st));
    if (nc) {
      SLG_HIP(hipMemcpyAsync(dcd.p, cand_doc, nc * 4, hipMemcpyHostToDevice, st));
      SLG_HIP(hipMemcpyAsync(dcs.p, cand_seg, nc * 4, hipMemcpyHostToDevice, st));
      SLG_HIP(hipMemcpyAsync(dcb.p, cand_bm25, nc * 4, hipMemcpyHostToDevice, st));
    }
    SLG_HIP(hipMemcpyAsync(dcc.p, cand_count, (size_t)nq * 4, hipMemcpyHostToDevice, st));
  });
  if (rc != SLG_OK) return rc;
  rc = slg_rerank_fields_batch_device(ix, nq, n_clauses, clause_field, dq.as<float>(), da.as<float>(),
                                      boost ? db.as<float>() : nullptr, dcd.as<uint32_t>(), dcs.as<uint32_t>(),
                                      dcb.as<float>(), dcc.as<uint32_t>(), max_cand, k_out, dod.as<uint32_t>(),
                                      dos.as<uint32_t>(), dosc.as<float>(), dov.as<float>(), doc_.as<uint32_t>());
  if (rc != SLG_OK) return rc;
  return guarded([&] {
    DeviceGuard g(ix->device);
    hipStream_t st = ix->stream;
    if (no) {
      SLG_HIP(hipMemcpyAsync(out_doc, dod.p, no * 4, hipMemcpyDeviceToHost, st));
      SLG_HIP(hipMemcpyAsync(out_seg, dos.p, no * 4, hipMemcpyDeviceToHost, st));
      SLG_HIP(hipMemcpyAsync(out_score, dosc.p, no * 4, hipMemcpyDeviceToHost, st));
      if (out_vec_score) SLG_HIP(hipMemcpyAsync(out_vec_score, dov.p, no * 4, hipMemcpyDeviceToHost, st));
    }
    SLG_HIP(hipMemcpyAsync(out_count, doc_.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    SLG_HIP(hipStreamSynchronize(st));
  });
}

int slg_rerank_multi_batch(slg_index *ix, uint32_t nq, uint32_t n_clauses, const float *qvecs,
                           const float *alpha, const float *boost, const uint32_t *cand_doc,
                           const uint32_t *cand_seg, const float *cand_bm25, const uint32_t *cand_count,
                           uint32_t max_cand, uint32_t k_out, uint32_t *out_doc, uint32_t *out_seg,
                           float *out_score, float *out_vec_score, uint32_t *out_count) {
  int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    if (n_clauses < 1 || n_clauses > SLG_MAX_VECTOR_CLAUSES)
      throw SlgError(SLG_ERR_UNSUPPORTED, "n_clauses outside 1..SLG_MAX_VECTOR_CLAUSES");
    if (nq == 0) return;
    SLG_REQUIRE(qvecs && alpha && cand_count && out_count, "host arrays are NULL");
    SLG_REQUIRE(max_cand == 0 || (cand_doc && cand_seg && cand_bm25), "candidate arrays are NULL");
    SLG_REQUIRE(k_out == 0 || (out_doc && out_seg && out_score), "output arrays are NULL");
  });
  if (rc != SLG_OK || nq == 0) return rc;
  uint32_t dim = 0;
  for (auto &s : ix->snapshot()->segs)
    if (s->store->vec_dim) dim = s->store->vec_dim;
  if (dim == 0) {
    g_last_error = "index has no vector field";
    g_last_code = SLG_ERR_UNSUPPORTED;
    return SLG_ERR_UNSUPPORTED;
  }
  DevBuf dq, da, db, dcd, dcs, dcb, dcc, dod, dos, dosc, dov, doc_;
  const size_t nc = (size_t)nq * max_cand, no = (size_t)nq * k_out, nqc = (size_t)nq * n_clauses;
  rc = guarded([&] {
    DeviceGuard g(ix->device);
    hipStream_t st = ix->stream;
    dq.alloc_pooled(&ix->pool, nqc * dim * 4);
    da.alloc_pooled(&ix->pool, nqc * 4);
    if (boost) db.alloc_pooled(&ix->pool, nqc * 4);
    dcd.alloc_pooled(&ix->pool, nc * 4);
    dcs.alloc_pooled(&ix->pool, nc * 4);
    dcb.alloc_pooled(&ix->pool, nc * 4);
    dcc.alloc_pooled(&ix->pool, (size_t)nq * 4);
    dod.alloc_pooled(&ix->pool, no * 4);
    dos.alloc_pooled(&ix->pool, no * 4);
    dosc.alloc_pooled(&ix->pool, no * 4);
    dov.alloc_pooled(&ix->pool, no * 4);
    doc_.alloc_pooled(&ix->pool, (size_t)nq * 4);
    SLG_HIP(hipMemcpyAsync(dq.p, qvecs, nqc * dim * 4, hipMemcpyHostToDevice, st));
    SLG_HIP(hipMemcpyAsync(da.p, alpha, nqc * 4, hipMemcpyHostToDevice, st));
    if (boost) SLG_HIP(hipMemcpyAsync(db.p, boost, nqc * 4, hipMemcpyHostToDevice, st));
    if (nc) {
      SLG_HIP(hipMemcpyAsync(dcd.p, cand_doc, nc * 4, hipMemcpyHostToDevice, st));
      SLG_HIP(hipMemcpyAsync(dcs.p, cand_seg, nc * 4, hipMemcpyHostToDevice, st));
      SLG_HIP(hipMemcpyAsync(dcb.p, cand_bm25, nc * 4, hipMemcpyHostToDevice, st));
    }
    SLG_HIP(hipMemcpyAsync(dcc.p, cand_count, (size_t)nq * 4, hipMemcpyHostToDevice, st));
  });
  if (rc != SLG_OK) return rc;
  rc = slg_rerank_multi_batch_device(ix, nq, n_clauses, dq.as<float>(), da.as<float>(),
                                     boost ? db.as<float>() : nullptr, dcd.as<uint32_t>(), dcs.as<uint32_t>(),
                                     dcb.as<float>(), dcc.as<uint32_t>(), max_cand, k_out, dod.as<uint32_t>(),
                                     dos.as<uint32_t>(), dosc.as<float>(), dov.as<float>(), doc_.as<uint32_t>());
  if (rc != SLG_OK) return rc;
  return guarded([&] {
    DeviceGuard g(ix->device);
    hipStream_t st = ix->stream;
    if (no) {
      SLG_HIP(hipMemcpyAsync(out_doc, dod.p, no * 4, hipMemcpyDeviceToHost, st));
      SLG_HIP(hipMemcpyAsync(out_seg, dos.p, no * 4, hipMemcpyDeviceToHost, st));
      SLG_HIP(hipMemcpyAsync(out_score, dosc.p, no * 4, hipMemcpyDeviceToHost, st));
      if (out_vec_score) SLG_HIP(hipMemcpyAsync(out_vec_score, dov.p, no * 4, hipMemcpyDeviceToHost, st));
    }
    SLG_HIP(hipMemcpyAsync(out_count, doc_.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    SLG_HIP(hipStreamSynchronize(st));
  });
}

int slg_rerank_batch(slg_index *ix, uint32_t nq, const float *qvecs, const float *alpha,
                     const uint32_t *cand_doc, const uint32_t *cand_seg, const float *cand_bm25,
                     const uint32_t *cand_count, uint32_t max_cand, uint32_t k_out,
                     uint32_t *out_doc, uint32_t *out_seg, float *out_score, float *out_vec_score,
                     uint32_t *out_count) {
  int rc = guarded([&] {
    SLG_REQUIRE(ix != nullptr, "index is NULL");
    if (nq == 0) return;
    SLG_REQUIRE(qvecs && alpha && cand_count && out_count, "host arrays are NULL");
    SLG_REQUIRE(max_cand == 0 || (cand_doc && cand_seg && cand_bm25), "candidate arrays are NULL");
    SLG_REQUIRE(k_out == 0 || (out_doc && out_seg && out_score), "output arrays are NULL");
  });
  if (rc != SLG_OK || nq == 0) return rc;
  uint32_t dim = 0;
  for (auto &s : ix->snapshot()->segs)
    if (s->store->vec_dim) dim = s->store->vec_dim;
  if (dim == 0) {
    g_last_error = "index has no vector field";
    return SLG_ERR_UNSUPPORTED;
  }
  DevBuf dq, da, dcd, dcs, dcb, dcc, dod, dos, dosc, dov, doc_;
  const size_t nc = (size_t)nq * max_cand, no = (size_t)nq * k_out;
  rc = guarded([&] {
    DeviceGuard g(ix->device);
    hipStream_t st = ix->stream;
    dq.alloc_pooled(&ix->pool, (size_t)nq * dim * 4);
    da.alloc_pooled(&ix->pool, (size_t)nq * 4);
    dcd.alloc_pooled(&ix->pool, nc * 4);
    dcs.alloc_pooled(&ix->pool, nc * 4);
    dcb.alloc_pooled(&ix->pool, nc * 4);
    dcc.alloc_pooled(&ix->pool, (size_t)nq * 4);
    dod.alloc_pooled(&ix->pool, no * 4);
    dos.alloc_pooled(&ix->pool, no * 4);
    dosc.alloc_pooled(&ix->pool, no * 4);
    dov.alloc_pooled(&ix->pool, no * 4);
    doc_.alloc_pooled(&ix->pool, (size_t)nq * 4);
    SLG_HIP(hipMemcpyAsync(dq.p, qvecs, (size_t)nq * dim * 4, hipMemcpyHostToDevice, st));
    SLG_HIP(hipMemcpyAsync(da.p, alpha, (size_t)nq * 4, hipMemcpyHostToDevice, st));
    if (nc) {
      SLG_HIP(hipMemcpyAsync(dcd.p, cand_doc, nc * 4, hipMemcpyHostToDevice, st));
      SLG_HIP(hipMemcpyAsync(dcs.p, cand_seg, nc * 4, hipMemcpyHostToDevice, st));
      SLG_HIP(hipMemcpyAsync(dcb.p, cand_bm25, nc * 4, hipMemcpyHostToDevice, st));
    }
    SLG_HIP(hipMemcpyAsync(dcc.p, cand_count, (size_t)nq * 4, hipMemcpyHostToDevice, st));
  });
  if (rc != SLG_OK) return rc;
  rc = slg_rerank_batch_device(ix, nq, dq.as<float>(), da.as<float>(), dcd.as<uint32_t>(),
                               dcs.as<uint32_t>(), dcb.as<float>(), dcc.as<uint32_t>(), max_cand,
                               k_out, dod.as<uint32_t>(), dos.as<uint32_t>(), dosc.as<float>(),
                               dov.as<float>(), doc_.as<uint32_t>());
  if (rc != SLG_OK) return rc;
  return guarded([&] {
    DeviceGuard g(ix->device);
    hipStream_t st = ix->stream;
    if (no) {
      SLG_HIP(hipMemcpyAsync(out_doc, dod.p, no * 4, hipMemcpyDeviceToHost, st));
      SLG_HIP(hipMemcpyAsync(out_seg, dos.p, no * 4, hipMemcpyDeviceToHost, st));
      SLG_HIP(hipMemcpyAsync(out_score, dosc.p, no * 4, hipMemcpyDeviceToHost, st));
      if (out_vec_score)
        SLG_HIP(hipMemcpyAsync(out_vec_score, dov.p, no * 4, hipMemcpyDeviceToHost, st));
    }
    SLG_HIP(hipMemcpyAsync(out_count, doc_.p, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
    SLG_HIP(hipStreamSynchronize(st));
  });
}

}  // extern "C"
