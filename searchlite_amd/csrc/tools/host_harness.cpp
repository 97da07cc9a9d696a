// host_harness.cpp — bench harness: the caller threads of bench.py's host-inclusive leg, native.
//
// searchlite serves one request per OS thread (searchlite-http/src/lib.rs:640-643, spawn_blocking);
// its host language is Rust, i.e. threads without an interpreter lock.  bench.py's Python threads
// share the GIL, which made the host-inclusive rate a measurement of Python's thread hand-offs
// (a 20-step region is 2.5 steps per thread).  This file is the same loop as a C++ caller of the
// C ABI: persistent threads, each keeping TWO batches going on two HIP streams of its own —
// slg_batch_prepare (plan + H2D) -> slg_batch_set_stream -> slg_batch_run, then slg_batch_fetch
// (D2H into host arrays) + slg_batch_destroy of the batch launched before.  Harness only: built as
// lib/libslg_harness.so, linked against libsearchlite_gpu.so, never part of the product library.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/searchlite_gpu.h"

namespace {

struct QuerySet {
  const uint32_t *offs, *terms;
  const float *w;
};

struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<int64_t> todo;  // step numbers; -1 = flush, -2 = quit
  hipStream_t streams[2] = {nullptr, nullptr};
};

struct Harness {
  slg_index *ix = nullptr;
  // index-sharded mode (config 4): every batch runs slg_batch_run_sharded_seq(seq = step number) and is
  // collected with slg_batch_fetch_sharded; the step numbers of all slh_run calls must then be 0, 1, 2, ...
  slg_shard_group *group = nullptr;
  int device = 0;
  uint32_t nq = 0, k = 0;
  int strategy = 0;
  std::vector<QuerySet> sets;
  std::vector<Worker *> workers;
  std::mutex done_mu;
  std::condition_variable done_cv;
  int64_t done = 0;
  std::string error;
  // time the caller threads spent inside the C ABI, summed over all batches (ns) and their count
  std::atomic<uint64_t> ns_prepare{0}, ns_run{0}, ns_fetch{0}, n_timed{0};
  // results of the first batch of every query set (bench.py compares them with the other legs)
  std::vector<std::vector<uint32_t>> first_doc, first_seg, first_count;
  std::vector<std::vector<float>> first_score;
  std::vector<char> have_first;
};

struct Pending {
  slg_batch *b = nullptr;
  int64_t step = -1;
};

void collect(Harness *h, Pending &p, std::vector<uint32_t> &doc, std::vector<uint32_t> &seg,
             std::vector<float> &score, std::vector<uint32_t> &count) {
  if (!p.b) return;
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = h->group ? slg_batch_fetch_sharded(p.b, doc.data(), seg.data(), score.data(), count.data())
                          : slg_batch_fetch(p.b, doc.data(), seg.data(), score.data(), count.data(), nullptr);
  slg_batch_destroy(p.b);
  h->ns_fetch += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  const size_t set = (size_t)(p.step % (int64_t)h->sets.size());
  {
    std::lock_guard<std::mutex> lk(h->done_mu);
    if (rc != SLG_OK && h->error.empty()) h->error = std::string("slg_batch_fetch: ") + slg_last_error();
    if (rc == SLG_OK && !h->have_first[set]) {
      h->first_doc[set] = doc;
      h->first_seg[set] = seg;
      h->first_score[set] = score;
      h->first_count[set] = count;
      h->have_first[set] = 1;
    }
    h->done++;
  }
  h->done_cv.notify_all();
  p.b = nullptr;
}

void worker_main(Harness *h, Worker *w) {
  (void)hipSetDevice(h->device);
  const size_t n = (size_t)h->nq * h->k;
  std::vector<uint32_t> doc(n ? n : 1), seg(n ? n : 1), count(h->nq ? h->nq : 1);
  std::vector<float> score(n ? n : 1);
  Pending prev;
  uint64_t lap = 0;
  for (;;) {
    int64_t step;
    {
      std::unique_lock<std::mutex> lk(w->mu);
      w->cv.wait(lk, [&] { return !w->todo.empty(); });
      step = w->todo.front();
      w->todo.pop_front();
    }
    if (step == -2) {
      collect(h, prev, doc, seg, score, count);
      return;
    }
    if (step == -1) {
      collect(h, prev, doc, seg, score, count);
      continue;
    }
    const QuerySet &qs = h->sets[(size_t)(step % (int64_t)h->sets.size())];
    const auto t0 = std::chrono::steady_clock::now();
    slg_batch *b = slg_batch_prepare(h->ix, h->nq, qs.offs, qs.terms, qs.w, h->k, h->strategy);
    const auto t1 = std::chrono::steady_clock::now();
    int rc = b ? SLG_OK : slg_last_error_code();
    if (b) rc = slg_batch_set_stream(b, (void *)w->streams[lap++ & 1u]);
    if (b && rc == SLG_OK)
      rc = h->group ? slg_batch_run_sharded_seq(b, h->group, (uint64_t)step, nullptr, nullptr, nullptr, nullptr)
                    : slg_batch_run(b);
    const auto t2 = std::chrono::steady_clock::now();
    h->ns_prepare += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
    h->ns_run += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t2 - t1).count();
    h->n_timed++;
    if (rc != SLG_OK) {
      std::lock_guard<std::mutex> lk(h->done_mu);
      if (h->error.empty()) h->error = std::string("prepare/run: ") + slg_last_error();
    }
    if (!b && h->group) (void)slg_shard_group_skip_seq(h->group, (uint64_t)step);  // (no batch: the turn still moves on)
    collect(h, prev, doc, seg, score, count);  // the batch launched one lap ago
    if (b && rc == SLG_OK) {
      prev.b = b;
      prev.step = step;
    } else {
      if (b) slg_batch_destroy(b);
      std::lock_guard<std::mutex> lk(h->done_mu);
      h->done++;
      h->done_cv.notify_all();
    }
  }
}

}  // namespace

extern "C" {

// q_offsets / q_terms / q_weights: n_sets pointers to query arrays that stay valid for the harness' life
void *slh_create(slg_index *ix, int device, int n_threads, int n_sets, const uint32_t *const *q_offsets,
                 const uint32_t *const *q_terms, const float *const *q_weights, uint32_t nq, uint32_t k,
                 int strategy) {
  auto *h = new Harness();
  h->ix = ix;
  h->device = device;
  h->nq = nq;
  h->k = k;
  h->strategy = strategy;
  for (int s = 0; s < n_sets; s++) h->sets.push_back(QuerySet{q_offsets[s], q_terms[s], q_weights[s]});
  h->first_doc.resize(n_sets);
  h->first_seg.resize(n_sets);
  h->first_score.resize(n_sets);
  h->first_count.resize(n_sets);
  h->have_first.assign(n_sets, 0);
  (void)hipSetDevice(device);
  for (int t = 0; t < n_threads; t++) {
    auto *w = new Worker();
    for (auto &st : w->streams) (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    h->workers.push_back(w);
  }
  for (auto *w : h->workers) w->th = std::thread(worker_main, h, w);
  return h;
}

// index-sharded mode: the shard group every batch runs in (NULL: back to plain runs).  Call it before
// the first slh_run; from then on the step numbers must be 0, 1, 2, ... on every rank.
void slh_set_group(void *hp, slg_shard_group *group) { static_cast<Harness *>(hp)->group = group; }

// the most recent result of a worker thread is not kept; for parity checks bench.py reads the FIRST
// result of every query set (slh_first_result)

// Runs steps [first, first + n_steps), round-robin over the threads, and returns when every one of
// them has been fetched.  0 on success, -1 on error (slh_error).
int slh_run(void *hp, int64_t first, int64_t n_steps) {
  auto *h = static_cast<Harness *>(hp);
  int64_t target;
  {
    std::lock_guard<std::mutex> lk(h->done_mu);
    target = h->done + n_steps;
  }
  const size_t nt = h->workers.size();
  for (int64_t i = first; i < first + n_steps; i++) {
    Worker *w = h->workers[(size_t)(i % (int64_t)nt)];
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->todo.push_back(i);
    }
    w->cv.notify_one();
  }
  for (auto *w : h->workers) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->todo.push_back(-1);
    }
    w->cv.notify_one();
  }
  std::unique_lock<std::mutex> lk(h->done_mu);
  h->done_cv.wait(lk, [&] { return h->done >= target; });
  return h->error.empty() ? 0 : -1;
}

// mean time per batch a caller thread spent in prepare / set_stream + run / fetch + destroy (ms), batches timed
void slh_stats(void *hp, double *out4) {
  auto *h = static_cast<Harness *>(hp);
  const double n = (double)std::max<uint64_t>(1, h->n_timed.load());
  out4[0] = (double)h->ns_prepare.load() / n * 1e-6;
  out4[1] = (double)h->ns_run.load() / n * 1e-6;
  out4[2] = (double)h->ns_fetch.load() / n * 1e-6;
  out4[3] = (double)h->n_timed.load();
}

// forget what slh_stats has accumulated (bench.py: after the warm-up, so the means cover the timed regions only)
void slh_reset_stats(void *hp) {
  auto *h = static_cast<Harness *>(hp);
  h->ns_prepare = 0;
  h->ns_run = 0;
  h->ns_fetch = 0;
  h->n_timed = 0;
}

// ---- request coalescer: n_threads caller threads, each with ONE query at a time -------------------
// What a server built on searchlite does (searchlite-http/src/lib.rs:628-652: a blocking thread per
// request): every thread takes the next query of the set (round-robin over its nq queries), calls
// slg_coalescer_search and compares its row with the expected one (exp_*: the same query set through
// the batch API; NULL: no check).  Returns the seconds from the common start until total_queries have
// been answered, or a negative value on error; *mismatches = rows that differ, *n_batches = batches the
// coalescer ran.
double slh_coalesce_bench(slg_index *ix, int device, int n_threads, int64_t total_queries, const uint32_t *offs,
                          const uint32_t *terms, const float *w, uint32_t nq, uint32_t n_segs, uint32_t k,
                          int strategy, uint32_t max_batch, uint32_t max_wait_us, const uint32_t *exp_doc,
                          const float *exp_score, const uint32_t *exp_count, int64_t *mismatches,
                          uint64_t *n_batches, double *phase_ms4, int depth) {
  slg_coalescer *co = slg_coalescer_create(ix, max_batch, max_wait_us);
  if (!co) return -1.0;
  std::atomic<int64_t> next{0}, bad{0}, failed{0};
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; t++)
    pool.emplace_back([&] {
      (void)hipSetDevice(device);
      std::vector<uint32_t> doc(k ? k : 1), seg(k ? k : 1);
      std::vector<float> score(k ? k : 1);
      uint32_t count = 0;
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      auto check = [&](const uint32_t q) {
        if (!exp_count) return;
        bool same = count == exp_count[q];
        for (uint32_t j = 0; same && j < count; j++)
          same = doc[j] == exp_doc[(size_t)q * k + j] && std::memcmp(&score[j], &exp_score[(size_t)q * k + j], 4) == 0;
        if (!same) bad.fetch_add(1);
      };
      auto query_of = [&](const uint32_t q) {
        slg_query qq;
        qq.n_terms = offs[q + 1] - offs[q];
        qq.term_ids = terms + (size_t)offs[q] * n_segs;
        qq.weights = w + offs[q];
        return qq;
      };
      if (depth <= 1) {  // one blocking call per query (a thread per request)
        for (;;) {
          const int64_t i = next.fetch_add(1);
          if (i >= total_queries) break;
          const uint32_t q = (uint32_t)(i % (int64_t)nq);
          const slg_query qq = query_of(q);
          if (slg_coalescer_search(co, &qq, k, strategy, doc.data(), seg.data(), score.data(), &count, nullptr) != SLG_OK) {
            failed.fetch_add(1);
            continue;
          }
          check(q);
        }
        return;
      }
      // `depth` requests in flight per thread (slg_coalescer_submit / _wait): a ring of tickets, the oldest
      // is waited for and its place taken by a new query
      std::vector<slg_ticket> ring((size_t)depth);
      std::vector<uint32_t> ring_q((size_t)depth);
      int64_t head = 0, tail = 0;  // tickets [tail, head) are in flight
      bool more = true;
      while (more || tail < head) {
        while (more && head - tail < depth) {
          const int64_t i = next.fetch_add(1);
          if (i >= total_queries) {
            more = false;
            break;
          }
          const uint32_t q = (uint32_t)(i % (int64_t)nq);
          const slg_query qq = query_of(q);
          if (slg_coalescer_submit(co, &qq, nullptr, SLG_PLAN_SUM, 0.0f, 0, -1, k, strategy, 0,
                                   &ring[(size_t)(head % depth)]) != SLG_OK) {
            failed.fetch_add(1);
            continue;
          }
          ring_q[(size_t)(head % depth)] = q;
          head++;
        }
        if (tail < head) {
          if (slg_coalescer_wait(co, &ring[(size_t)(tail % depth)], doc.data(), seg.data(), score.data(), &count,
                                 nullptr) != SLG_OK)
            failed.fetch_add(1);
          else
            check(ring_q[(size_t)(tail % depth)]);
          tail++;
        }
      }
    });
  while (ready.load() < n_threads) std::this_thread::yield();
  const auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto &th : pool) th.join();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (mismatches) *mismatches = bad.load();
  if (n_batches) (void)slg_coalescer_stats(co, n_batches, nullptr);
  if (phase_ms4) (void)slg_coalescer_phase_ms(co, phase_ms4, phase_ms4 + 1, phase_ms4 + 2, phase_ms4 + 3);
  slg_coalescer_destroy(co);
  return failed.load() ? -2.0 : secs;
}

const char *slh_error(void *hp) { return static_cast<Harness *>(hp)->error.c_str(); }

// the results of the first batch of query set `set`; returns 1 if one was run
int slh_first_result(void *hp, int set, uint32_t *doc, uint32_t *seg, float *score, uint32_t *count) {
  auto *h = static_cast<Harness *>(hp);
  std::lock_guard<std::mutex> lk(h->done_mu);
  if (set < 0 || (size_t)set >= h->sets.size() || !h->have_first[set]) return 0;
  const size_t n = (size_t)h->nq * h->k;
  std::memcpy(doc, h->first_doc[set].data(), n * 4);
  std::memcpy(seg, h->first_seg[set].data(), n * 4);
  std::memcpy(score, h->first_score[set].data(), n * 4);
  std::memcpy(count, h->first_count[set].data(), (size_t)h->nq * 4);
  return 1;
}

void slh_destroy(void *hp) {
  auto *h = static_cast<Harness *>(hp);
  for (auto *w : h->workers) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->todo.push_back(-2);
    }
    w->cv.notify_one();
  }
  for (auto *w : h->workers) {
    w->th.join();
    (void)hipSetDevice(h->device);
    for (auto st : w->streams)
      if (st) (void)hipStreamDestroy(st);
    delete w;
  }
  delete h;
}

}  // extern "C"
