// slg_coalesce.hip — request coalescer behind the C ABI (include/searchlite_gpu.h: slg_coalescer_*).
//
// searchlite has no batch API: IndexReader::search takes ONE request (api/reader.rs:2539) and the HTTP
// server runs every request on its own blocking thread (searchlite-http/src/lib.rs:628-652,
// spawn_blocking -> index.reader() -> reader.search(&req)).  A `gpu` shim that forwards one query per
// call pays a whole launch chain per query (~80 us: ~12K queries/s per stream) where the device
// scores 1024 queries in the same time.  The coalescer is what sits between the two: every caller
// thread blocks in slg_coalescer_search with ITS query; concurrent callers are collected into one
// slg_batch_prepare / run / fetch and each gets its own row back.
//
// Mechanism.  A caller takes a ROW of the batch that is collecting: ONE fetch_add on the batch's ticket
// counter, no lock (a mutex per request made 256 callers queue for 0.4 ms per batch), writes its query
// into the row's fixed-size slot and sleeps (a futex on the batch's `done` word) until the batch's results
// are in — or, with slg_coalescer_submit / _wait, goes on to submit more and collects its rows later.  A few
// threads of the coalescer own every HIP call: a SUBMITTER closes the collecting batch — when it is
// full (max_batch), max_wait_us after its first row, or at once if nothing else is in flight (a lone
// request never waits) —, swaps a fresh batch in for the callers that keep arriving, plans and launches
// the closed one on a stream of its own; a COLLECTOR fetches a launched batch and wakes
// its callers.  (Letting one of a batch's callers do the planning, launching and fetching — leader /
// followers — was built first: with a dozen leaders inside the HIP runtime at once, prepare took 0.24 ms
// and fetch 0.41 ms for 23-query batches whose kernels take ~40 us; polling the stream instead of the
// blocking wait made it worse.)  Batches of different kinds — (k, strategy, segment count) — collect
// side by side.  Batch objects are recycled per kind, so a caller that still holds the pointer of a
// batch that has been closed and reopened meanwhile joins a batch of its own kind or bounces.
//
// Host code on top of the public ABI (no kernels here; HIP only for the leaders' streams); part of
// libsearchlite_gpu.so.
#include <hip/hip_runtime.h>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <climits>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/searchlite_gpu.h"

namespace {

constexpr uint32_t kClosed = 0x80000000u;  // ticket counter: top bit = the batch takes no more rows
constexpr int kMaxKinds = 8;               // (k, strategy, segment count) combinations collecting at once
using Clock = std::chrono::steady_clock;

// Callers sleep on the batch's `done` word itself (futex): waking a batch's callers is one system call
// and none of them takes a lock on the way out.  (A condition variable was built first: notify_all
// hands every woken caller the batch mutex in turn — 200 callers per batch queued for it.)
inline void futex_wait(std::atomic<uint32_t> *w, uint32_t seen) {
  (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
}
inline void futex_wake_all(std::atomic<uint32_t> *w) {
  (void)syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

inline void cpu_relax() {
#if defined(__x86_64__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}

// One collecting / running batch (see the header comment).  Fixed-size slots: SLG_MAX_QUERY_TERMS x
// n_segs term ids per row (a megabyte for an 8-segment index: objects are recycled, per kind).
struct CoBatch {
  uint32_t k = 0;
  int strategy = 0;
  uint32_t n_segs = 0, slot_terms = 0, cap = 0;
  std::atomic<uint32_t> tickets{kClosed};  // rows handed out; >= kClosed: closed
  uint32_t nq = 0;                         // rows of the closed batch (set by the submitter)
  std::atomic<uint32_t> ready{0};          // rows written
  std::atomic<uint32_t> leaving{0};        // callers that have copied their result out
  Clock::time_point first_seen{};          // when the submitter first saw a row in it
  bool seen = false;
  std::vector<uint32_t> slot_ids, slot_nt;
  std::vector<float> slot_w;
  // per-row score plan and doc filter (slg_coalescer_search_plan); plain rows: leaf i = term i, Sum, no filter
  std::vector<uint32_t> slot_leaf, slot_nleaves;
  std::vector<int32_t> slot_plan, slot_filter;
  std::vector<float> slot_tie;
  std::atomic<bool> want_stats{false}, any_plan{false}, any_filter{false};
  // the closed batch as CSR (built by the submitter), and its results
  std::vector<uint32_t> offs, term_ids, leaves;
  std::vector<float> weights;
  std::vector<uint32_t> doc, seg, count;
  std::vector<float> score;
  std::vector<slg_stats> stats;
  int rc = SLG_OK;
  std::string error;
  std::atomic<uint32_t> done{0};  // 1: results are in (callers sleep on this word: futex)
  int kind = 0;
};

struct Kind {
  std::atomic<bool> used{false};
  uint32_t k = 0, n_segs = 0;
  int strategy = 0;
  std::atomic<CoBatch *> cur{nullptr};  // the batch that is collecting (never null once the kind is used)
  std::mutex mu;                        // spare list
  std::vector<CoBatch *> spare;
  std::vector<CoBatch *> all;           // every batch object of the kind (destroy)
};

struct Flight {  // a launched batch on its way to the collector
  CoBatch *b;
  slg_batch *sb;
  void *stream;
  Clock::time_point launched;
};

}  // namespace

struct slg_coalescer {
  slg_index *index = nullptr;
  uint32_t max_batch = 1024, max_wait_us = 50;
  int device = 0;
  Kind kinds[kMaxKinds];
  std::mutex mu;  // kind registration, stream free list
  std::vector<void *> free_streams;
  std::atomic<uint32_t> in_flight{0};     // batches launched and not yet fetched
  std::atomic<uint32_t> rows_waiting{0};  // rows in collecting batches
  std::atomic<bool> stop{false};
  std::vector<std::thread> submitters, collectors;
  std::mutex close_mu;  // one submitter at a time looks for a batch to close (planning and launching run outside it)
  std::mutex wake_mu;   // the submitters sleep here while no row is waiting
  std::condition_variable wake_cv;
  std::mutex q_mu;     // submitter -> collector
  std::condition_variable q_cv;
  std::deque<Flight> flights;
  // the index's segment count, cached per generation (slg_index_info takes the index mutex)
  std::atomic<uint64_t> seen_generation{~0ull};
  std::atomic<uint32_t> seen_n_segs{0};
  // accounting (slg_coalescer_stats / _phase_ms)
  std::atomic<uint64_t> n_batches{0}, n_queries{0};
  std::atomic<uint64_t> ns_collect{0}, ns_prepare{0}, ns_run{0}, ns_fetch{0};
};

namespace {

thread_local std::string g_co_error;

uint64_t nanos(Clock::time_point a, Clock::time_point z) {
  return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(z - a).count();
}

void *take_stream(slg_coalescer *c) {
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->free_streams.empty()) {
      void *s = c->free_streams.back();
      c->free_streams.pop_back();
      return s;
    }
  }
  hipStream_t s = nullptr;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
  return (void *)s;
}

void give_stream(slg_coalescer *c, void *s) {
  if (!s) return;
  std::lock_guard<std::mutex> lk(c->mu);
  c->free_streams.push_back(s);
}

// a closed, reset batch object of the kind (recycled or new)
CoBatch *fresh_batch(slg_coalescer *c, Kind &kd) {
  CoBatch *b = nullptr;
  {
    std::lock_guard<std::mutex> lk(kd.mu);
    if (!kd.spare.empty()) {
      b = kd.spare.back();
      kd.spare.pop_back();
    } else {
      b = new CoBatch();
      kd.all.push_back(b);
    }
  }
  b->k = kd.k;
  b->strategy = kd.strategy;
  b->n_segs = kd.n_segs;
  b->kind = (int)(&kd - c->kinds);
  if (b->cap != c->max_batch || b->slot_terms != SLG_MAX_QUERY_TERMS * kd.n_segs) {
    b->cap = c->max_batch;
    b->slot_terms = SLG_MAX_QUERY_TERMS * kd.n_segs;
    b->slot_ids.resize((size_t)b->cap * b->slot_terms);
    b->slot_w.resize((size_t)b->cap * SLG_MAX_QUERY_TERMS);
    b->slot_nt.resize(b->cap);
    b->slot_leaf.resize((size_t)b->cap * SLG_MAX_QUERY_TERMS);
    b->slot_nleaves.resize(b->cap);
    b->slot_plan.resize(b->cap);
    b->slot_filter.resize(b->cap);
    b->slot_tie.resize(b->cap);
  }
  b->nq = 0;
  b->ready.store(0);
  b->leaving.store(0);
  b->seen = false;
  b->want_stats.store(false);
  b->any_plan.store(false);
  b->any_filter.store(false);
  b->done.store(0);
  b->rc = SLG_OK;
  b->error.clear();
  return b;
}

Kind *find_kind(slg_coalescer *c, uint32_t k, int strategy, uint32_t n_segs) {
  for (int i = 0; i < kMaxKinds; i++) {
    Kind &kd = c->kinds[i];
    if (kd.used.load(std::memory_order_acquire) && kd.k == k && kd.strategy == strategy && kd.n_segs == n_segs) return &kd;
  }
  std::lock_guard<std::mutex> lk(c->mu);
  for (int i = 0; i < kMaxKinds; i++) {
    Kind &kd = c->kinds[i];
    if (kd.used.load() && kd.k == k && kd.strategy == strategy && kd.n_segs == n_segs) return &kd;
  }
  for (int i = 0; i < kMaxKinds; i++) {
    Kind &kd = c->kinds[i];
    if (!kd.used.load()) {
      kd.k = k;
      kd.strategy = strategy;
      kd.n_segs = n_segs;
      CoBatch *nb = fresh_batch(c, kd);
      nb->tickets.store(0, std::memory_order_release);  // open
      kd.cur.store(nb, std::memory_order_release);
      kd.used.store(true, std::memory_order_release);
      return &kd;
    }
  }
  return nullptr;
}

void publish(CoBatch &b, int rc, const char *err) {
  b.rc = rc;
  if (err) b.error = err;
  b.done.store(1, std::memory_order_release);
  futex_wake_all(&b.done);
}

// the submitter's part for one closed batch: CSR, plan, launch
void launch_batch(slg_coalescer *c, CoBatch &b) {
  while (b.ready.load(std::memory_order_acquire) < b.nq) cpu_relax();  // (writers are a few stores behind their ticket)
  const uint32_t nq = b.nq, ns = b.n_segs;
  b.offs.resize((size_t)nq + 1);
  b.offs[0] = 0;
  for (uint32_t q = 0; q < nq; q++) b.offs[q + 1] = b.offs[q] + b.slot_nt[q];
  b.term_ids.resize((size_t)b.offs[nq] * ns);
  b.weights.resize(b.offs[nq]);
  const bool plans = b.any_plan.load(), filters = b.any_filter.load();
  if (plans) b.leaves.resize(b.offs[nq]);
  for (uint32_t q = 0; q < nq; q++) {
    const uint32_t nt = b.slot_nt[q];
    if (!nt) continue;
    std::memcpy(b.term_ids.data() + (size_t)b.offs[q] * ns, b.slot_ids.data() + (size_t)q * b.slot_terms, (size_t)nt * ns * 4);
    std::memcpy(b.weights.data() + b.offs[q], b.slot_w.data() + (size_t)q * SLG_MAX_QUERY_TERMS, (size_t)nt * 4);
    if (plans) std::memcpy(b.leaves.data() + b.offs[q], b.slot_leaf.data() + (size_t)q * SLG_MAX_QUERY_TERMS, (size_t)nt * 4);
  }
  const size_t n = (size_t)nq * b.k;
  b.doc.resize(n ? n : 1);
  b.seg.resize(n ? n : 1);
  b.score.resize(n ? n : 1);
  b.count.resize(nq);
  if (b.want_stats.load()) b.stats.assign(nq, slg_stats{});
  void *stream = take_stream(c);
  const auto t0 = Clock::now();
  slg_batch *sb = slg_batch_prepare_plan(c->index, nq, b.offs.data(), b.term_ids.data(), b.weights.data(),
                                         plans ? b.leaves.data() : nullptr, plans ? b.slot_plan.data() : nullptr,
                                         plans ? b.slot_tie.data() : nullptr, plans ? b.slot_nleaves.data() : nullptr,
                                         filters ? b.slot_filter.data() : nullptr, b.k, b.strategy);
  const auto t1 = Clock::now();
  int rc = sb ? SLG_OK : slg_last_error_code();
  if (sb && stream) rc = slg_batch_set_stream(sb, stream);
  if (sb && rc == SLG_OK) rc = slg_batch_run(sb);
  const auto t2 = Clock::now();
  c->ns_prepare += nanos(t0, t1);
  c->ns_run += nanos(t1, t2);
  if (rc != SLG_OK) {
    const std::string err = slg_last_error();
    if (sb) slg_batch_destroy(sb);
    give_stream(c, stream);
    c->n_batches.fetch_add(1);
    c->in_flight.fetch_sub(1);
    publish(b, rc, err.c_str());
    return;
  }
  {
    std::lock_guard<std::mutex> lk(c->q_mu);
    c->flights.push_back(Flight{&b, sb, stream, t2});
  }
  c->q_cv.notify_one();
}

// Several submitters (SLG_COALESCER_THREADS, 2): planning a batch is ~0.2 us of host time per query, one
// thread's worth of it caps the coalescer near 3.4M queries/s.  Closing is serialised (close_mu); the
// closed batch is planned and launched outside the lock while another submitter closes the next one.
void submitter_main(slg_coalescer *c) {
  (void)hipSetDevice(c->device);
  while (!c->stop.load(std::memory_order_acquire)) {
    if (c->rows_waiting.load(std::memory_order_acquire) == 0) {  // nothing collecting: sleep until a first row arrives
      std::unique_lock<std::mutex> lk(c->wake_mu);
      c->wake_cv.wait_for(lk, std::chrono::milliseconds(2), [&] {
        return c->rows_waiting.load(std::memory_order_acquire) != 0 || c->stop.load(std::memory_order_acquire);
      });
      continue;
    }
    CoBatch *closed = nullptr;
    {
      std::unique_lock<std::mutex> cl(c->close_mu, std::try_to_lock);
      if (cl.owns_lock()) {
        const auto now = Clock::now();
        for (int i = 0; i < kMaxKinds && !closed; i++) {
          Kind &kd = c->kinds[i];
          if (!kd.used.load(std::memory_order_acquire)) continue;
          CoBatch *b = kd.cur.load(std::memory_order_acquire);
          const uint32_t t = b->tickets.load(std::memory_order_acquire);
          if (t == 0u || (t & kClosed)) continue;
          if (!b->seen) {
            b->seen = true;
            b->first_seen = now;
          }
          // close: full, or max_wait_us after its first row — at once if nothing else is in flight (an idle
          // device: waiting would only add latency; under load the batches in flight give it time to fill).
          // A batch counts as in flight from here on (while it is being planned too).
          const bool full = t >= c->max_batch;
          const bool idle = c->in_flight.load(std::memory_order_acquire) == 0;
          const bool aged = nanos(b->first_seen, now) >= (uint64_t)c->max_wait_us * 1000ull;
          if (!(full || idle || aged)) continue;
          CoBatch *nb = fresh_batch(c, kd);
          nb->tickets.store(0, std::memory_order_release);  // open
          kd.cur.store(nb, std::memory_order_release);      // arriving callers go there from now on
          const uint32_t had = b->tickets.fetch_or(kClosed, std::memory_order_acq_rel);
          b->nq = had < c->max_batch ? had : c->max_batch;
          c->rows_waiting.fetch_sub(b->nq, std::memory_order_acq_rel);
          c->in_flight.fetch_add(1);
          c->ns_collect += nanos(b->first_seen, Clock::now());
          closed = b;
        }
      }
    }
    if (closed)
      launch_batch(c, *closed);
    else
      cpu_relax();
  }
}

void collector_main(slg_coalescer *c) {
  (void)hipSetDevice(c->device);
  for (;;) {
    Flight f{};
    {
      std::unique_lock<std::mutex> lk(c->q_mu);
      c->q_cv.wait(lk, [&] { return !c->flights.empty() || c->stop.load(std::memory_order_acquire); });
      if (c->flights.empty()) return;  // stop, and nothing left to collect
      f = c->flights.front();
      c->flights.pop_front();
    }
    CoBatch &b = *f.b;
    const auto t0 = Clock::now();
    int rc = slg_batch_fetch(f.sb, b.doc.data(), b.seg.data(), b.score.data(), b.count.data(),
                             b.want_stats.load() ? b.stats.data() : nullptr);
    const std::string err = rc != SLG_OK ? slg_last_error() : "";
    slg_batch_destroy(f.sb);
    give_stream(c, f.stream);
    c->ns_fetch += nanos(t0, Clock::now());
    c->in_flight.fetch_sub(1);
    c->n_batches.fetch_add(1);
    c->n_queries.fetch_add(b.nq);
    publish(b, rc, rc != SLG_OK ? err.c_str() : nullptr);
  }
}

}  // namespace

extern "C" {

slg_coalescer *slg_coalescer_create(slg_index *index, uint32_t max_batch, uint32_t max_wait_us) {
  if (!index) return nullptr;
  const int dev = slg_index_device(index);
  if (dev < 0) return nullptr;
  auto *c = new slg_coalescer();
  c->index = index;
  c->device = dev;
  c->max_batch = max_batch ? (max_batch < kClosed / 2 ? max_batch : 1024u) : 1024u;
  c->max_wait_us = max_wait_us;
  uint32_t n_thr = 2;
  if (const char *e = getenv("SLG_COALESCER_THREADS")) n_thr = (uint32_t)strtoul(e, nullptr, 10);
  n_thr = n_thr < 1u ? 1u : (n_thr > 8u ? 8u : n_thr);
  for (uint32_t i = 0; i < n_thr; i++) c->submitters.emplace_back(submitter_main, c);
  for (uint32_t i = 0; i < n_thr; i++) c->collectors.emplace_back(collector_main, c);
  return c;
}

// (no caller may be inside slg_coalescer_search)
void slg_coalescer_destroy(slg_coalescer *c) {
  if (!c) return;
  c->stop.store(true, std::memory_order_release);
  c->wake_cv.notify_all();
  for (auto &t : c->submitters) t.join();
  {
    std::lock_guard<std::mutex> lk(c->q_mu);
  }
  c->q_cv.notify_all();
  for (auto &t : c->collectors) t.join();
  for (void *s : c->free_streams) (void)hipStreamDestroy((hipStream_t)s);
  for (Kind &kd : c->kinds)
    for (CoBatch *b : kd.all) delete b;
  delete c;
}

const char *slg_coalescer_last_error(void) { return g_co_error.c_str(); }

int slg_coalescer_phase_ms(const slg_coalescer *c, double *collect, double *prepare, double *run, double *fetch) {
  if (!c) return SLG_ERR_INVALID;
  const double n = (double)(c->n_batches.load() ? c->n_batches.load() : 1);
  if (collect) *collect = (double)c->ns_collect.load() / n * 1e-6;
  if (prepare) *prepare = (double)c->ns_prepare.load() / n * 1e-6;
  if (run) *run = (double)c->ns_run.load() / n * 1e-6;
  if (fetch) *fetch = (double)c->ns_fetch.load() / n * 1e-6;
  return SLG_OK;
}

int slg_coalescer_stats(const slg_coalescer *c, uint64_t *n_batches, uint64_t *n_queries) {
  if (!c) return SLG_ERR_INVALID;
  if (n_batches) *n_batches = c->n_batches.load();
  if (n_queries) *n_queries = c->n_queries.load();
  return SLG_OK;
}

int slg_coalescer_search(slg_coalescer *c, const slg_query *query, uint32_t k, int strategy, uint32_t *out_doc,
                         uint32_t *out_seg, float *out_score, uint32_t *out_count, slg_stats *stats_or_null) {
  return slg_coalescer_search_plan(c, query, nullptr, SLG_PLAN_SUM, 0.0f, 0, -1, k, strategy, out_doc, out_seg,
                                   out_score, out_count, stats_or_null);
}

int slg_coalescer_search_plan(slg_coalescer *c, const slg_query *query, const uint32_t *leaf, int plan, float tie,
                              uint32_t n_leaves, int32_t filter_id, uint32_t k, int strategy, uint32_t *out_doc,
                              uint32_t *out_seg, float *out_score, uint32_t *out_count, slg_stats *stats_or_null) {
  if (!out_count || (k && (!out_doc || !out_seg || !out_score))) {
    g_co_error = "output array is NULL";
    return SLG_ERR_INVALID;
  }
  slg_ticket t;
  const int rc = slg_coalescer_submit(c, query, leaf, plan, tie, n_leaves, filter_id, k, strategy,
                                      stats_or_null ? 1 : 0, &t);
  if (rc != SLG_OK) return rc;
  return slg_coalescer_wait(c, &t, out_doc, out_seg, out_score, out_count, stats_or_null);
}

int slg_coalescer_submit(slg_coalescer *c, const slg_query *query, const uint32_t *leaf, int plan, float tie,
                         uint32_t n_leaves, int32_t filter_id, uint32_t k, int strategy, int want_stats,
                         slg_ticket *ticket) {
  g_co_error.clear();
  if (!c || !query || !ticket || (query->n_terms && (!query->term_ids || !query->weights))) {
    g_co_error = "coalescer, query or ticket is NULL";
    return SLG_ERR_INVALID;
  }
  if (query->n_terms > SLG_MAX_QUERY_TERMS) {
    g_co_error = "query has more than SLG_MAX_QUERY_TERMS terms";
    return SLG_ERR_UNSUPPORTED;
  }
  // the index's segment count (the width of the query's term-id rows), cached per index generation
  const uint64_t gen = slg_index_generation(c->index);
  uint32_t n_segs = c->seen_n_segs.load(std::memory_order_acquire);
  if (c->seen_generation.load(std::memory_order_acquire) != gen) {
    if (slg_index_info(c->index, &n_segs, nullptr, nullptr) != SLG_OK) {
      g_co_error = slg_last_error();
      return SLG_ERR_INVALID;
    }
    c->seen_n_segs.store(n_segs, std::memory_order_release);
    c->seen_generation.store(gen, std::memory_order_release);
  }
  Kind *kd = find_kind(c, k, strategy, n_segs);
  if (!kd) {
    g_co_error = "too many (k, strategy) combinations collecting at once";
    return SLG_ERR_UNSUPPORTED;
  }
  // ---- a row: one fetch_add on the collecting batch's ticket counter ----
  CoBatch *b = nullptr;
  uint32_t row = 0;
  for (;;) {
    b = kd->cur.load(std::memory_order_acquire);
    const uint32_t t = b->tickets.fetch_add(1, std::memory_order_acq_rel);
    if (t < c->max_batch) {  // (closed: t >= kClosed; full: t >= max_batch)
      row = t;
      break;
    }
    // closed or full: the submitter swaps the next batch in within microseconds
    if (c->stop.load(std::memory_order_acquire)) {
      g_co_error = "coalescer is being destroyed";
      return SLG_ERR_INVALID;
    }
    while (kd->cur.load(std::memory_order_acquire) == b && !c->stop.load(std::memory_order_relaxed)) cpu_relax();
  }
  // ---- my row ----
  b->slot_nt[row] = query->n_terms;
  if (query->n_terms) {
    std::memcpy(b->slot_ids.data() + (size_t)row * b->slot_terms, query->term_ids, (size_t)query->n_terms * n_segs * 4);
    std::memcpy(b->slot_w.data() + (size_t)row * SLG_MAX_QUERY_TERMS, query->weights, (size_t)query->n_terms * 4);
  }
  // (every row carries its plan: a batch may mix plain rows with planned ones)
  const bool planned = leaf != nullptr || plan != SLG_PLAN_SUM || n_leaves != 0;
  for (uint32_t i = 0; i < query->n_terms; i++) b->slot_leaf[(size_t)row * SLG_MAX_QUERY_TERMS + i] = leaf ? leaf[i] : i;
  b->slot_plan[row] = plan;
  b->slot_tie[row] = tie;
  b->slot_nleaves[row] = n_leaves ? n_leaves : (leaf ? 0u : query->n_terms);
  if (leaf && !n_leaves) {  // 1 + the largest leaf named
    uint32_t mx = 0;
    for (uint32_t i = 0; i < query->n_terms; i++) mx = leaf[i] + 1u > mx ? leaf[i] + 1u : mx;
    b->slot_nleaves[row] = mx;
  }
  b->slot_filter[row] = filter_id;
  if (planned) b->any_plan.store(true);
  if (filter_id >= 0) b->any_filter.store(true);
  if (want_stats) b->want_stats.store(true);
  b->ready.fetch_add(1, std::memory_order_release);
  if (c->rows_waiting.fetch_add(1, std::memory_order_acq_rel) == 0) {  // the first waiting row wakes the submitter
    {
      std::lock_guard<std::mutex> lk(c->wake_mu);
    }
    c->wake_cv.notify_one();
  }
  ticket->batch = b;
  ticket->row = row;
  ticket->k = k;
  ticket->kind = (uint32_t)(kd - c->kinds);
  return SLG_OK;
}

int slg_coalescer_poll(const slg_coalescer *c, const slg_ticket *ticket) {
  if (!c || !ticket || !ticket->batch) return SLG_ERR_INVALID;
  return static_cast<const CoBatch *>(ticket->batch)->done.load(std::memory_order_acquire) != 0u ? 1 : 0;
}

int slg_coalescer_wait(slg_coalescer *c, slg_ticket *ticket, uint32_t *out_doc, uint32_t *out_seg, float *out_score,
                       uint32_t *out_count, slg_stats *stats_or_null) {
  g_co_error.clear();
  if (!c || !ticket || !ticket->batch || ticket->kind >= (uint32_t)kMaxKinds) {
    g_co_error = "coalescer or ticket is NULL (a ticket is good for one wait)";
    return SLG_ERR_INVALID;
  }
  CoBatch *b = static_cast<CoBatch *>(ticket->batch);
  const uint32_t row = ticket->row, k = ticket->k;
  Kind *kd = &c->kinds[ticket->kind];
  // ---- wait (asleep: hundreds of callers spinning would take the cores the two dispatcher threads and
  //      the callers that are being woken need) ----
  while (b->done.load(std::memory_order_acquire) == 0u) futex_wait(&b->done, 0u);
  const int rc = b->rc;
  if (rc != SLG_OK) {
    g_co_error = b->error;
  } else if (!out_count || (k && (!out_doc || !out_seg || !out_score))) {
    g_co_error = "output array is NULL";
  } else {
    *out_count = b->count[row];
    if (k) {
      std::memcpy(out_doc, b->doc.data() + (size_t)row * k, (size_t)k * 4);
      std::memcpy(out_seg, b->seg.data() + (size_t)row * k, (size_t)k * 4);
      std::memcpy(out_score, b->score.data() + (size_t)row * k, (size_t)k * 4);
    }
    if (stats_or_null && !b->stats.empty()) *stats_or_null = b->stats[row];
  }
  const bool args_ok = rc != SLG_OK || (out_count && (!k || (out_doc && out_seg && out_score)));
  // the last caller to leave hands the batch object back to its kind (b->nq is final: done was seen)
  ticket->batch = nullptr;
  if (b->leaving.fetch_add(1) + 1 == b->nq) {
    std::lock_guard<std::mutex> lk(kd->mu);
    kd->spare.push_back(b);
  }
  return args_ok ? rc : SLG_ERR_INVALID;
}

}  // extern "C"
