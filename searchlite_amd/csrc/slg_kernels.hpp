// slg_kernels.hpp — hand-written HIP kernels for gfx950 (CDNA4, wave64).
//
// Path restated (reference file:line relative to searchlite-core/src/):
//   stage_impacts   : query/bm25.rs:1-6 + query/wand.rs:77-84,269-286 evaluated once per
//                     posting at staging time with weight factored out
//   score_slices    : query/wand.rs:459-566 (exhaustive accumulate per doc, term order ==
//                     ScorePlan leaf order planner.rs:122-135) + push_top_k :905-916
//   merge_topk      : query/wand.rs:918-926 + api/reader.rs:2776-2778 (query/sort.rs:80-93)
//
// Design (DESIGN.md has the long form): the unit of work is a *slice* = one query's
// posting lists restricted to one doc-id range.  ONE WAVE owns one slice: it streams the
// T sorted lists in rounds of <= CAP postings bounded by a common doc id, accumulates
// per-doc sums in a private open-addressing hash table in LDS (term order is program
// order inside a wave, so sums are bit-identical to the reference's leaf-order sum and no
// workgroup barrier is ever needed), and keeps a wave-wide sorted top-k in registers.
// Everything is integer/f32 VALU + LDS work bounded by HBM streaming; no MFMA here.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slg {

constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;
constexpr uint32_t kDocEnd = 0xFFFFFFFFu;
constexpr int32_t kSentinelTk = INT32_MIN;
constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr uint32_t kMaxTerms = 32;

// ---- device-side descriptors (built on the host per batch) ---------------------------
struct SegDev {
  const uint32_t *docs;     // [P] doc ids
  const float *imps;        // [P] precomputed bm25 (weight == 1) per posting
  const uint32_t *deleted;  // bitmap words or nullptr
  uint32_t n_docs;
  uint32_t pad;
};

struct TermRef {  // one scored term of one sub-query
  uint64_t off;   // posting offset inside the segment arrays
  uint32_t df;    // list length
  float weight;
};

struct SubQuery {  // (query, segment) pair with >= 1 non-empty term
  uint32_t q, seg;
  uint32_t term_begin, n_terms;
  uint32_t slice_begin, n_slices;
  uint32_t bounds_begin;  // into bounds[], layout [slice j][term t]
  uint32_t longest;       // index of the longest list (splitter source)
};

struct QueryRef {
  uint32_t slice_begin, slice_end;  // all slices of all sub-queries of this query
};

// ---- small helpers --------------------------------------------------------------------
__device__ __forceinline__ int32_t total_key(float x) {
  // f32::total_cmp key: sign-magnitude bits -> two's complement order
  int32_t b = __float_as_int(x);
  return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1);
}
__device__ __forceinline__ float key_to_float(int32_t k) {
  return __int_as_float(k ^ (int32_t)(((uint32_t)(k >> 31)) >> 1));
}
__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint32_t u = __shfl_xor(v, o, 64);
    v = u < v ? u : v;
  }
  return v;
}
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t lane) {
  uint32_t x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t u = __shfl_up(x, o, 64);
    if (lane >= (uint32_t)o) x += u;
  }
  return x - v;
}
// lane l receives lane l-1's value (lane 0 keeps its own): one DPP move, no LDS traffic
__device__ __forceinline__ int32_t wave_shr1(int32_t v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// compiler-only ordering point for wave-synchronous LDS traffic (hardware keeps a wave's
// DS instructions in order; this stops the compiler from moving them across phases)
__device__ __forceinline__ void wave_fence() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }

// (score key, seg, doc) ordering: larger tk first, then smaller seg, then smaller doc
// (query/wand.rs:30-37, query/sort.rs:80-93)
template <bool HAS_SEG>
__device__ __forceinline__ bool better(int32_t tka, uint32_t sega, uint32_t doca, int32_t tkb,
                                       uint32_t segb, uint32_t docb) {
  if (tka != tkb) return tka > tkb;
  if (HAS_SEG && sega != segb) return sega < segb;
  return doca < docb;
}

// ---- wave-wide sorted top-k held in registers -------------------------------------------
// Position p = lane*KREGS + r (best first).  Capacity 64*KREGS >= k.  All methods are
// wave-uniform in control flow; candidates are passed as uniform (SGPR) values.
template <int KREGS, bool HAS_SEG>
struct WaveTopK {
  int32_t tk[KREGS];
  uint32_t doc[KREGS];
  uint32_t seg[HAS_SEG ? KREGS : 1];
  int32_t th_tk;  // threshold = entry at position k-1 (uniform)
  uint32_t th_seg, th_doc;
  uint32_t count;  // real entries held, capped at k (uniform)

  __device__ __forceinline__ void init() {
#pragma unroll
    for (int r = 0; r < KREGS; r++) {
      tk[r] = kSentinelTk;
      doc[r] = 0xFFFFFFFFu;
      if (HAS_SEG) seg[r] = 0xFFFFFFFFu;
    }
    if (!HAS_SEG) seg[0] = 0;
    th_tk = kSentinelTk;
    th_seg = 0xFFFFFFFFu;
    th_doc = 0xFFFFFFFFu;
    count = 0;
  }
  __device__ __forceinline__ bool passes(int32_t ctk, uint32_t cseg, uint32_t cdoc) const {
    return better<HAS_SEG>(ctk, cseg, cdoc, th_tk, th_seg, th_doc);
  }
  // insert a uniform candidate known to pass the threshold
  __device__ __forceinline__ void insert(int32_t ctk, uint32_t cseg, uint32_t cdoc, uint32_t k,
                                         uint32_t lane) {
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < KREGS; r++)
      cnt += better<HAS_SEG>(tk[r], HAS_SEG ? seg[r] : 0u, doc[r], ctk, cseg, cdoc) ? 1u : 0u;
    uint64_t full = __ballot(cnt == (uint32_t)KREGS);
    uint32_t pos_lane = (uint32_t)__popcll(full);  // fully-better lanes form a prefix
    uint32_t pos_r = pos_lane < 64 ? rl(cnt, pos_lane) : 0u;
    // value arriving from the previous lane's last register
    int32_t up_tk = wave_shr1(tk[KREGS - 1]);
    uint32_t up_doc = (uint32_t)wave_shr1((int32_t)doc[KREGS - 1]);
    uint32_t up_seg = HAS_SEG ? (uint32_t)wave_shr1((int32_t)seg[KREGS - 1]) : 0u;
#pragma unroll
    for (int r = KREGS - 1; r >= 0; r--) {
      bool shift = lane > pos_lane || (lane == pos_lane && (uint32_t)r > pos_r);
      bool here = lane == pos_lane && (uint32_t)r == pos_r;
      int32_t s_tk = r == 0 ? up_tk : tk[r > 0 ? r - 1 : 0];
      uint32_t s_doc = r == 0 ? up_doc : doc[r > 0 ? r - 1 : 0];
      uint32_t s_seg = HAS_SEG ? (r == 0 ? up_seg : seg[r > 0 ? r - 1 : 0]) : 0u;
      tk[r] = here ? ctk : (shift ? s_tk : tk[r]);
      doc[r] = here ? cdoc : (shift ? s_doc : doc[r]);
      if (HAS_SEG) seg[r] = here ? cseg : (shift ? s_seg : seg[r]);
    }
    if (count < k) count++;
    // refresh threshold = entry at position k-1
    uint32_t tl = (k - 1) / KREGS, tr = (k - 1) % KREGS;
    int32_t v_tk = tk[0];
    uint32_t v_doc = doc[0], v_seg = HAS_SEG ? seg[0] : 0u;
#pragma unroll
    for (int r = 1; r < KREGS; r++) {
      bool sel = tr == (uint32_t)r;
      v_tk = sel ? tk[r] : v_tk;
      v_doc = sel ? doc[r] : v_doc;
      if (HAS_SEG) v_seg = sel ? seg[r] : v_seg;
    }
    th_tk = (int32_t)rl((uint32_t)v_tk, tl);
    th_doc = rl(v_doc, tl);
    th_seg = HAS_SEG ? rl(v_seg, tl) : 0u;
  }
};

// ---- staging: per-posting impact ---------------------------------------------------------
// impact = bm25(tf, df, doc_len, avgdl, docs, k1, b) exactly as score_tf computes `base`
// (query/wand.rs:279-285 -> query/bm25.rs:1-6); idf is computed on the host with libm
// logf (bm25.rs:2, f32::ln) and passed per term.
struct StageParams {
  uint64_t n_postings;
  uint32_t n_terms;
  uint32_t n_docs;
  const uint64_t *term_offsets;  // [V+1]
  const uint32_t *docs;          // [P]
  const uint32_t *tfs;           // [P]
  const float *term_idf;         // [V]
  const uint16_t *term_field;    // [V] or nullptr
  const float *const *field_doc_len;  // [F] device pointers (or nullptr entries)
  const float *field_avgdl;           // [F]
  float k1, b;
  float *imps;  // out [P]
};

__global__ void __launch_bounds__(256) stage_impacts_kernel(StageParams p) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < p.n_postings; i += stride) {
    // term of posting i: last t with term_offsets[t] <= i
    uint32_t lo = 0, hi = p.n_terms;  // invariant: off[lo] <= i < off[hi]
    while (hi - lo > 1) {
      uint32_t mid = lo + ((hi - lo) >> 1);
      if (p.term_offsets[mid] <= i)
        lo = mid;
      else
        hi = mid;
    }
    uint32_t t = lo;
    float df = (float)(uint32_t)(p.term_offsets[t + 1] - p.term_offsets[t]);
    (void)df;
    uint32_t f = p.term_field ? p.term_field[t] : 0;
    float avgdl = p.field_avgdl[f];
    const float *lens = p.field_doc_len[f];
    uint32_t doc = p.docs[i];
    float tf = (float)p.tfs[i];
    // ScoredTerm::doc_len  query/wand.rs:77-84
    float dl = fmaxf(avgdl, 1.0f);
    if (lens && doc < p.n_docs) {
      float v = lens[doc];
      if (v > 0.0f) dl = v;
    }
    // score_tf  query/wand.rs:279-283
    float norm_len = dl > 0.0f ? dl : fmaxf(avgdl, tf);
    // bm25  query/bm25.rs:2-5 (idf precomputed)
    float idf = p.term_idf[t];
    float norm_dl = avgdl > 0.0f ? norm_len / avgdl : 1.0f;
    float denom = tf + p.k1 * (1.0f - p.b + p.b * norm_dl);
    p.imps[i] = idf * (tf * (p.k1 + 1.0f)) / fmaxf(denom, 1e-6f);
  }
}

// ---- partition: doc-range slice boundaries per sub-query ----------------------------------
// Slice j >= 1 of a sub-query starts at the doc id found at position j*stride of its
// longest list; the other lists are cut by lower_bound on that doc id.  Only a
// load-balancing decision: score_slices is correct for any slice sizes.
struct PartParams {
  const SubQuery *sq;
  const TermRef *terms;
  const uint32_t *slice_sq;
  const SegDev *segs;
  uint32_t *bounds;
  uint32_t n_slices;
};

__global__ void __launch_bounds__(256) partition_kernel(PartParams p) {
  uint32_t lane = threadIdx.x & 63;
  uint32_t slice = rfl(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (slice >= p.n_slices) return;
  uint32_t sqi = p.slice_sq[slice];
  SubQuery s = p.sq[sqi];
  uint32_t j = slice - s.slice_begin;
  if (lane >= s.n_terms) return;
  const uint32_t *docs = p.segs[s.seg].docs;
  TermRef me = p.terms[s.term_begin + lane];
  uint32_t out;
  if (j == 0) {
    out = 0;
  } else {
    TermRef L = p.terms[s.term_begin + s.longest];
    uint32_t stride = (L.df + s.n_slices - 1) / s.n_slices;
    uint64_t posL = (uint64_t)j * stride;
    if (posL >= L.df) {
      out = me.df;  // empty tail slice
    } else if (lane == s.longest) {
      out = (uint32_t)posL;
    } else {
      uint32_t target = docs[L.off + posL];
      const uint32_t *d = docs + me.off;
      uint32_t lo = 0, hi = me.df;  // first index with d[idx] >= target
      while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (d[mid] < target)
          lo = mid + 1;
        else
          hi = mid;
      }
      out = lo;
    }
  }
  p.bounds[s.bounds_begin + j * s.n_terms + lane] = out;
}

// ---- the hot kernel: score one slice per wave ----------------------------------------------
struct ScoreParams {
  const SubQuery *sq;
  const TermRef *terms;
  const uint32_t *slice_sq;
  const SegDev *segs;
  const uint32_t *bounds;
  int32_t *slice_tk;    // [n_slices * k] candidate score keys (kSentinelTk = empty)
  uint32_t *slice_doc;  // [n_slices * k]
  uint32_t *q_scored;   // [nq] distinct docs scored (QueryStats.scored_docs), may be null
  uint32_t n_slices;
  uint32_t k;
};

template <int NSLOT>
struct ScoreCfg {
  static constexpr int kCap = NSLOT * 64;  // postings per round
  static constexpr int kLogSlots = 31 - __builtin_clz((unsigned)(2 * kCap - 1)) + 1;
  static constexpr int kSlots = 1 << kLogSlots;   // hash slots (load <= 0.5)
  static constexpr int kLogBuckets = kLogSlots - 2;  // buckets of 4 keys (one ds_read_b128)
  static constexpr int kBuckets = 1 << kLogBuckets;
  static constexpr int kWaveLds = kSlots * 8;  // keys u32[kSlots] then vals f32[kSlots]
};

// One round's worth of postings held in registers (v = jj*64 + lane over the concatenated
// per-list chunks) plus the per-list plan that produced it.
template <int NSLOT>
struct Round {
  uint32_t doc[NSLOT];
  float imp[NSLOT];
  uint32_t et[NSLOT];   // list index of each element
  uint32_t chunk;       // lane t: postings of list t loaded this round
  uint32_t lastdoc;     // lane t: doc id that bounds the round if list t does not finish
  uint32_t total;       // uniform: sum of chunks
  uint32_t tfirst[NSLOT], tlast[NSLOT];  // uniform per slot: first / last list present
};

template <int KREGS, int NSLOT>
__global__ void __launch_bounds__(256) score_slices_kernel(ScoreParams p) {
  using Cfg = ScoreCfg<NSLOT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wib = threadIdx.x >> 6;
  const uint32_t slice = rfl(blockIdx.x * kWavesPerBlock + wib);
  if (slice >= p.n_slices) return;  // no workgroup barriers anywhere: waves are independent

  uint32_t *keys = reinterpret_cast<uint32_t *>(smem + (size_t)wib * Cfg::kWaveLds);
  uint32_t *vals = keys + Cfg::kSlots;
  uint4 *keys4 = reinterpret_cast<uint4 *>(keys);

  const uint32_t sqi = rfl(p.slice_sq[slice]);
  const SubQuery s = p.sq[sqi];
  const uint32_t T = rfl(s.n_terms);
  const uint32_t j = slice - rfl(s.slice_begin);
  const SegDev sd = p.segs[s.seg];
  const uint32_t *__restrict__ gdocs = sd.docs;
  const float *__restrict__ gimps = sd.imps;
  const uint32_t *__restrict__ gdel = sd.deleted;
  const uint32_t k = p.k;

  // per-lane list state (lane t < T owns list t)
  uint64_t my_off = 0;
  uint32_t my_cur = 0, my_end = 0;
  float my_w = 0.0f;
  if (lane < T) {
    TermRef tr = p.terms[s.term_begin + lane];
    my_off = tr.off;
    my_w = tr.weight;
    const uint32_t *bb = p.bounds + s.bounds_begin;
    my_cur = j == 0 ? 0u : bb[j * T + lane];
    my_end = (j + 1 == s.n_slices) ? tr.df : bb[(j + 1) * T + lane];
    if (my_end < my_cur) my_end = my_cur;
  }

  // clear the hash keys once; every round's owners restore kEmptyKey behind themselves
  for (uint32_t i = lane; i < (uint32_t)(Cfg::kSlots / 4); i += 64)
    keys4[i] = make_uint4(kEmptyKey, kEmptyKey, kEmptyKey, kEmptyKey);
  wave_fence();

  WaveTopK<KREGS, false> top;
  top.init();
  uint32_t n_scored = 0;

  // ---- plan a round at the current cursors and issue its loads.  Returns false if the
  //      slice is exhausted.  All per-list bookkeeping is a scalar loop over t < T. ----
  auto plan = [&](Round<NSLOT> &r) -> bool {
    const uint32_t rem = my_end - my_cur;
    uint32_t R = 0;
    for (uint32_t t = 0; t < T; t++) R += rl(rem, t);
    r.total = 0;
    if (R == 0) return false;
    uint32_t chunk;
    if (R <= (uint32_t)Cfg::kCap) {
      chunk = rem;
    } else {
      const float share = (float)(Cfg::kCap - 2 * (int)T) * ((float)rem / (float)R);
      uint32_t c = (uint32_t)share;
      c = c < 1u ? 1u : c;
      chunk = rem == 0 ? 0u : (c < rem ? c : rem);
    }
    uint32_t start = 0, run = 0;
    for (uint32_t t = 0; t < T; t++) {
      start = lane == t ? run : start;
      run += rl(chunk, t);
    }
    const uint32_t total = run;
    r.total = total;
    r.chunk = chunk;
    // a list that does not finish in this round bounds the round by its last loaded doc
    const uint64_t abs0 = my_off + my_cur;  // absolute index of the list's first loaded posting
    r.lastdoc = kDocEnd;
    if (chunk < rem) r.lastdoc = gdocs[abs0 + chunk - 1];
    const uint64_t rel = abs0 - start;  // absolute index of v = 0 for this list (mod 2^64)
    const uint32_t rel_lo = (uint32_t)rel, rel_hi = (uint32_t)(rel >> 32);
#pragma unroll
    for (int jj = 0; jj < NSLOT; jj++) {
      r.doc[jj] = kDocEnd;
      r.imp[jj] = 0.0f;
      r.et[jj] = 0;
      r.tfirst[jj] = 0;
      r.tlast[jj] = 0;
      const uint32_t v0 = jj * 64;
      if (v0 < total) {  // uniform
        const uint32_t vl = (total - v0) >= 64 ? v0 + 63 : total - 1;
        uint32_t tf = 0, tl = 0;
        for (uint32_t t = 1; t < T; t++) {
          const uint32_t st = rl(start, t);
          tf += v0 >= st ? 1u : 0u;
          tl += vl >= st ? 1u : 0u;
        }
        r.tfirst[jj] = tf;
        r.tlast[jj] = tl;
        const uint32_t v = v0 + lane;
        uint64_t a;
        if (tf == tl) {  // the whole slot lies in one list: uniform base + lane
          r.et[jj] = tf;
          a = (((uint64_t)rl(rel_hi, tf) << 32) | rl(rel_lo, tf)) + v;
        } else {
          uint32_t t = tf;
          for (uint32_t tt = tf + 1; tt <= tl; tt++) t += (v >= rl(start, tt)) ? 1u : 0u;
          r.et[jj] = t;
          a = (((uint64_t)__shfl(rel_hi, t, 64) << 32) | __shfl(rel_lo, t, 64)) + v;
        }
        if (v < total) {
          r.doc[jj] = gdocs[a];
          r.imp[jj] = gimps[a];
        }
      }
    }
    return true;
  };

  // ---- after the loads of `r` landed: fix the round's doc bound, mark the postings that
  //      belong to it (returned as a per-lane bit mask) and advance the cursors. ----
  auto finalize = [&](const Round<NSLOT> &r) -> uint32_t {
    uint32_t bound = kDocEnd;
    for (uint32_t t = 0; t < T; t++) {
      const uint32_t ld = rl(r.lastdoc, t);
      bound = ld < bound ? ld : bound;
    }
    uint32_t inmask = 0, consumed = 0;
#pragma unroll
    for (int jj = 0; jj < NSLOT; jj++) {
      if ((uint32_t)(jj * 64) < r.total) {
        const bool in_round = r.doc[jj] <= bound;  // out-of-range lanes hold kDocEnd... see below
        const bool valid = (uint32_t)(jj * 64) + lane < r.total && in_round;
        inmask |= valid ? (1u << jj) : 0u;
        for (uint32_t tc = r.tfirst[jj]; tc <= r.tlast[jj]; tc++) {
          const uint64_t am = __ballot(valid && r.et[jj] == tc);
          consumed += lane == tc ? (uint32_t)__popcll(am) : 0u;
        }
      }
    }
    my_cur += consumed;
    return inmask;
  };

  // ---- accumulate the round into the hash table (one list at a time: term order ==
  //      ScorePlan leaf order), read the finished sums back, feed the top-k. ----
  auto process = [&](const Round<NSLOT> &r, const uint32_t inmask) {
    uint32_t own_slot[NSLOT];
#pragma unroll
    for (int jj = 0; jj < NSLOT; jj++) {
      own_slot[jj] = 0xFFFFFFFFu;
      if ((uint32_t)(jj * 64) < r.total) {  // uniform
        const bool in_round = (inmask >> jj) & 1u;
        const uint32_t doc = r.doc[jj];
        const uint32_t b0 = (doc * 0x9E3779B1u) >> (32 - Cfg::kLogBuckets);
        uint32_t own_h = 0xFFFFFFFFu;
        for (uint32_t tc = r.tfirst[jj]; tc <= r.tlast[jj]; tc++) {
          bool pend = in_round && r.et[jj] == tc;
          if (__ballot(pend) == 0) continue;
          const float w = __int_as_float((int)rl((uint32_t)__float_as_int(my_w), tc));
          // score_tf: base * weight (query/wand.rs:285); 0.0 + x is or_insert(0.0) += x
          const float x = 0.0f + r.imp[jj] * w;
          uint32_t b = b0;
          do {
            uint4 kk = make_uint4(0u, 0u, 0u, 0u);
            if (pend) kk = keys4[b];
            const uint32_t sub_hit = kk.x == doc ? 0u : kk.y == doc ? 1u : kk.z == doc ? 2u
                                                                  : kk.w == doc ? 3u : 4u;
            const uint32_t sub_emp = kk.x == kEmptyKey ? 0u : kk.y == kEmptyKey ? 1u
                                   : kk.z == kEmptyKey ? 2u : kk.w == kEmptyKey ? 3u : 4u;
            const bool hit = pend && sub_hit < 4u;
            const bool can = pend && !hit && sub_emp < 4u;
            const uint32_t slot = b * 4 + (hit ? sub_hit : sub_emp);
            if (hit)  // doc already present from an earlier list: add in term order
              vals[slot] = __float_as_uint(__uint_as_float(vals[slot]) + x);
            uint32_t old = 0u;
            if (can) old = atomicCAS(&keys[slot], kEmptyKey, doc);
            const bool won = can && old == kEmptyKey;
            if (won) vals[slot] = __float_as_uint(x);
            own_h = won ? slot : own_h;
            // bucket full of other docs: next bucket.  A lost CAS re-reads the same bucket.
            b = (pend && !hit && !can) ? ((b + 1) & (Cfg::kBuckets - 1)) : b;
            pend = pend && !(hit || won);
            wave_fence();
          } while (__ballot(pend) != 0);
        }
        own_slot[jj] = own_h;
      }
    }
    wave_fence();
#pragma unroll
    for (int jj = 0; jj < NSLOT; jj++) {
      if ((uint32_t)(jj * 64) < r.total) {
        const bool own = own_slot[jj] != 0xFFFFFFFFu;
        int32_t ctk = kSentinelTk;
        if (own) {
          ctk = total_key(__uint_as_float(vals[own_slot[jj]]));
          keys[own_slot[jj]] = kEmptyKey;
        }
        const uint64_t om = __ballot(own);
        n_scored += (uint32_t)__popcll(om);
        uint64_t m = __ballot(own && top.passes(ctk, 0u, r.doc[jj]));
        while (m) {
          const uint32_t l = (uint32_t)__builtin_ctzll(m);
          m &= m - 1;
          const int32_t c_tk = (int32_t)rl((uint32_t)ctk, l);
          const uint32_t c_doc = rl(r.doc[jj], l);
          if (!top.passes(c_tk, 0u, c_doc)) continue;
          if (gdel && ((gdel[c_doc >> 5] >> (c_doc & 31)) & 1u)) continue;  // accept()
          top.insert(c_tk, 0u, c_doc, k, lane);
        }
      }
    }
    wave_fence();
  };

  // ---- software pipeline: the loads of round r+1 are in flight while round r is hashed ----
  Round<NSLOT> ra, rb;
  bool more = plan(ra);
  while (more) {
    const uint32_t ma = finalize(ra);
    more = plan(rb);
    process(ra, ma);
    if (!more) break;
    const uint32_t mb = finalize(rb);
    more = plan(ra);
    process(rb, mb);
  }

  // ---- write this slice's candidates (sorted best-first; sentinel-padded) ----
  int32_t *otk = p.slice_tk + (size_t)slice * k;
  uint32_t *odoc = p.slice_doc + (size_t)slice * k;
#pragma unroll
  for (int r = 0; r < KREGS; r++) {
    const uint32_t pos = lane * KREGS + r;
    if (pos < k) {
      otk[pos] = top.tk[r];
      odoc[pos] = top.doc[r];
    }
  }
  if (p.q_scored && lane == 0 && n_scored) atomicAdd(&p.q_scored[s.q], n_scored);
}

// ---- merge: per query, all slice candidate lists -> final top-k ---------------------------
struct MergeParams {
  const QueryRef *queries;
  const SubQuery *sq;
  const uint32_t *slice_sq;
  const int32_t *slice_tk;
  const uint32_t *slice_doc;
  uint32_t *out_doc;
  uint32_t *out_seg;
  float *out_score;
  uint32_t *out_count;
  uint32_t nq;
  uint32_t k;
};

template <int KREGS>
__global__ void __launch_bounds__(256) merge_topk_kernel(MergeParams p) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t q = rfl(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (q >= p.nq) return;
  const uint32_t k = p.k;
  const QueryRef qr = p.queries[q];
  WaveTopK<KREGS, true> top;
  top.init();
  for (uint32_t sl = qr.slice_begin; sl < qr.slice_end; sl++) {
    const uint32_t seg = rfl(p.sq[p.slice_sq[sl]].seg);
    const int32_t *itk = p.slice_tk + (size_t)sl * k;
    const uint32_t *idoc = p.slice_doc + (size_t)sl * k;
    for (uint32_t base = 0; base < k; base += 64) {
      const uint32_t i = base + lane;
      int32_t ctk = kSentinelTk;
      uint32_t cdoc = 0xFFFFFFFFu;
      if (i < k) {
        ctk = itk[i];
        cdoc = idoc[i];
      }
      const bool valid = !(ctk == kSentinelTk && cdoc == 0xFFFFFFFFu);
      uint64_t m = __ballot(valid && top.passes(ctk, seg, cdoc));
      if (m == 0) break;  // lists are sorted: nothing further in this slice can pass
      while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        const int32_t c_tk = (int32_t)rl((uint32_t)ctk, l);
        const uint32_t c_doc = rl(cdoc, l);
        if (!top.passes(c_tk, seg, c_doc)) continue;
        top.insert(c_tk, seg, c_doc, k, lane);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KREGS; r++) {
    const uint32_t pos = lane * KREGS + r;
    if (pos < k) {
      const bool real = pos < top.count;
      p.out_doc[(size_t)q * k + pos] = real ? top.doc[r] : 0u;
      p.out_seg[(size_t)q * k + pos] = real ? top.seg[r] : 0u;
      p.out_score[(size_t)q * k + pos] = real ? key_to_float(top.tk[r]) : 0.0f;
    }
  }
  if (lane == 0) p.out_count[q] = top.count;
}

// ---- merge of per-shard results gathered over RCCL (api/reader.rs:2776-2778 across shards) --
struct ShardMergeParams {
  const uint32_t *doc;    // [n_shards][nq*k]
  const uint32_t *seg;
  const float *score;
  const uint32_t *count;  // [n_shards][nq]
  uint32_t *out_doc, *out_seg;
  float *out_score;
  uint32_t *out_count;
  uint32_t n_shards, nq, k, seg_stride;
};

template <int KREGS>
__global__ void __launch_bounds__(256) merge_shards_kernel(ShardMergeParams p) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t q = rfl(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (q >= p.nq) return;
  const uint32_t k = p.k;
  WaveTopK<KREGS, true> top;
  top.init();
  for (uint32_t sh = 0; sh < p.n_shards; sh++) {
    const size_t row = ((size_t)sh * p.nq + q) * k;
    const uint32_t cnt = rfl(p.count[(size_t)sh * p.nq + q]);
    for (uint32_t base = 0; base < cnt; base += 64) {
      const uint32_t i = base + lane;
      int32_t ctk = kSentinelTk;
      uint32_t cdoc = 0xFFFFFFFFu, cseg = 0xFFFFFFFFu;
      if (i < cnt) {
        ctk = total_key(p.score[row + i]);
        cdoc = p.doc[row + i];
        cseg = sh * p.seg_stride + p.seg[row + i];
      }
      uint64_t m = __ballot(i < cnt && top.passes(ctk, cseg, cdoc));
      if (m == 0) break;
      while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        const int32_t c_tk = (int32_t)rl((uint32_t)ctk, l);
        const uint32_t c_doc = rl(cdoc, l), c_seg = rl(cseg, l);
        if (!top.passes(c_tk, c_seg, c_doc)) continue;
        top.insert(c_tk, c_seg, c_doc, k, lane);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KREGS; r++) {
    const uint32_t pos = lane * KREGS + r;
    if (pos < k) {
      const bool real = pos < top.count;
      p.out_doc[(size_t)q * k + pos] = real ? top.doc[r] : 0u;
      p.out_seg[(size_t)q * k + pos] = real ? top.seg[r] : 0u;
      p.out_score[(size_t)q * k + pos] = real ? key_to_float(top.tk[r]) : 0.0f;
    }
  }
  if (lane == 0) p.out_count[q] = top.count;
}

}  // namespace slg
