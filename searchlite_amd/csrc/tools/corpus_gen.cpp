// corpus_gen.cpp — deterministic synthetic Zipf corpus generator (harness tool, host only).
//
// The reference ships no corpus generator; this is the spec the
// benchmarks use.  One text field; document d has L_d = len_min + (h(seed,d) mod
// (len_span+1)) tokens; token j of document d is drawn from Zipf(s) over `vocab` term
// ranks by a counter-based hash (seed, d, j) through a Vose alias table, so any document
// can be regenerated independently (two passes, no token buffer).  The output is exactly
// what searchlite's segment writer would hold for such documents
// (searchlite-core/src/index/segment.rs:666-698): per-term postings (doc id ascending, tf =
// occurrences), doc_len = token count, avgdl = total tokens / docs (:848).
//
// C ABI, two phases so the caller owns the big arrays:
//   slc_zipf_count -> term_offsets[vocab+1], doc_len[n_docs]; returns total postings
//   slc_zipf_fill  -> doc_ids[P], tfs[P]
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

struct Alias {
  std::vector<uint32_t> thresh;  // accept own bucket if lo32 < thresh
  std::vector<uint32_t> alias;
};

// Vose alias table over p_r ∝ r^-s, r = 1..V (term id = r-1).  Deterministic.
void build_alias(uint32_t V, double s, Alias &a) {
  std::vector<double> p(V);
  double sum = 0.0;
  for (uint32_t i = 0; i < V; i++) {
    p[i] = (s == 1.0) ? 1.0 / (double)(i + 1) : std::pow((double)(i + 1), -s);
    sum += p[i];
  }
  a.thresh.assign(V, 0xFFFFFFFFu);
  a.alias.resize(V);
  std::vector<uint32_t> small, large;
  small.reserve(V);
  large.reserve(V);
  for (uint32_t i = 0; i < V; i++) {
    p[i] = p[i] / sum * (double)V;
    a.alias[i] = i;
    (p[i] < 1.0 ? small : large).push_back(i);
  }
  size_t si = 0, li = 0;
  while (si < small.size() && li < large.size()) {
    const uint32_t l = small[si++], g = large[li];
    double t = p[l] * 4294967296.0;
    a.thresh[l] = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    a.alias[l] = g;
    p[g] = (p[g] + p[l]) - 1.0;
    if (p[g] < 1.0) {
      small.push_back(g);
      li++;
    }
  }
}

struct Gen {
  uint32_t n_docs, vocab, len_min, len_span;
  uint64_t seed;
  Alias alias;
  inline uint32_t doc_length(uint32_t d) const {
    return len_min + (uint32_t)(splitmix64(seed ^ (0xD0C1E57ull + (uint64_t)d * 0x9E3779B97F4A7C15ull)) %
                                ((uint64_t)len_span + 1));
  }
  inline uint32_t token(uint32_t d, uint32_t j, uint64_t dkey) const {
    (void)d;
    const uint64_t u = splitmix64(dkey + j);
    const uint32_t hi = (uint32_t)(u >> 32), lo = (uint32_t)u;
    const uint32_t bucket = (uint32_t)(((uint64_t)hi * vocab) >> 32);
    return lo < alias.thresh[bucket] ? bucket : alias.alias[bucket];
  }
  // sorted (term, tf) runs of document d into buf; returns number of distinct terms
  inline uint32_t doc_terms(uint32_t d, std::vector<uint32_t> &tok, uint32_t *len_out) const {
    const uint32_t L = doc_length(d);
    *len_out = L;
    tok.resize(L);
    const uint64_t dkey = splitmix64(seed * 0x2545F4914F6CDD1Dull + d);
    for (uint32_t j = 0; j < L; j++) tok[j] = token(d, j, dkey);
    std::sort(tok.begin(), tok.end());
    return L;
  }
};

template <typename F>
void parallel_ranges(uint32_t n, int n_threads, F &&f) {
  if (n_threads < 1) n_threads = 1;
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++) {
    const uint32_t a = (uint32_t)((uint64_t)n * t / n_threads);
    const uint32_t b = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
    th.emplace_back([=, &f] { f(t, a, b); });
  }
  for (auto &x : th) x.join();
}

struct Handle {
  Gen g;
  int n_threads;
  std::vector<std::vector<uint32_t>> counts;  // [thread][term] postings contributed
};

}  // namespace

extern "C" {

// Phase 1.  Returns an opaque handle (free with slc_free) or NULL.
void *slc_zipf_count(uint32_t n_docs, uint32_t vocab, double s, uint32_t len_min,
                     uint32_t len_span, uint64_t seed, int n_threads, uint64_t *term_offsets,
                     float *doc_len, uint64_t *n_postings, double *avgdl) {
  if (!term_offsets || !doc_len || !n_postings || vocab == 0) return nullptr;
  if (n_threads < 1) n_threads = 1;
  Handle *h = new Handle();
  h->g.n_docs = n_docs;
  h->g.vocab = vocab;
  h->g.len_min = len_min;
  h->g.len_span = len_span;
  h->g.seed = seed;
  h->n_threads = n_threads;
  build_alias(vocab, s, h->g.alias);
  h->counts.assign(n_threads, std::vector<uint32_t>(vocab, 0));
  std::vector<uint64_t> tot_len(n_threads, 0);
  parallel_ranges(n_docs, n_threads, [&](int t, uint32_t a, uint32_t b) {
    std::vector<uint32_t> tok;
    uint32_t *cnt = h->counts[t].data();
    uint64_t tl = 0;
    for (uint32_t d = a; d < b; d++) {
      uint32_t L;
      h->g.doc_terms(d, tok, &L);
      doc_len[d] = (float)L;
      tl += L;
      for (uint32_t i = 0; i < L; i++)
        if (i == 0 || tok[i] != tok[i - 1]) cnt[tok[i]]++;
    }
    tot_len[t] = tl;
  });
  uint64_t acc = 0;
  for (uint32_t v = 0; v < vocab; v++) {
    term_offsets[v] = acc;
    for (int t = 0; t < n_threads; t++) acc += h->counts[t][v];
  }
  term_offsets[vocab] = acc;
  *n_postings = acc;
  uint64_t total = 0;
  for (auto x : tot_len) total += x;
  // index/segment.rs:946-957: sum as f32 / total_docs as f32 (done by the caller in f32)
  if (avgdl) *avgdl = n_docs ? (double)total / (double)n_docs : 0.0;
  return h;
}

// Phase 2.  doc_ids/tfs have term_offsets[vocab] entries.
int slc_zipf_fill(void *handle, const uint64_t *term_offsets, uint32_t *doc_ids, uint32_t *tfs) {
  Handle *h = static_cast<Handle *>(handle);
  if (!h || !term_offsets || !doc_ids || !tfs) return -1;
  const uint32_t V = h->g.vocab;
  // per-thread write cursors: thread t writes after threads < t within each term
  std::vector<std::vector<uint64_t>> cur(h->n_threads, std::vector<uint64_t>(V));
  for (uint32_t v = 0; v < V; v++) {
    uint64_t at = term_offsets[v];
    for (int t = 0; t < h->n_threads; t++) {
      cur[t][v] = at;
      at += h->counts[t][v];
    }
  }
  parallel_ranges(h->g.n_docs, h->n_threads, [&](int t, uint32_t a, uint32_t b) {
    std::vector<uint32_t> tok;
    uint64_t *c = cur[t].data();
    for (uint32_t d = a; d < b; d++) {
      uint32_t L;
      h->g.doc_terms(d, tok, &L);
      uint32_t i = 0;
      while (i < L) {
        uint32_t j = i + 1;
        while (j < L && tok[j] == tok[i]) j++;
        const uint64_t at = c[tok[i]]++;
        doc_ids[at] = d;
        tfs[at] = j - i;
        i = j;
      }
    }
  });
  return 0;
}

void slc_free(void *handle) { delete static_cast<Handle *>(handle); }

}  // extern "C"
