// slg_kernels.hpp — hand-written HIP kernels for gfx950 (CDNA4, wave64).
//
// Path restated (reference file:line relative to searchlite-core/src/):
//   stage_impacts   : query/bm25.rs:1-6 + query/wand.rs:77-84,269-286 evaluated once per
//                     posting at staging time with weight factored out
//   score_rounds    : (slg_score.hpp) query/wand.rs:459-566 + push_top_k :905-916
//   merge_topk      : query/wand.rs:918-926 + api/reader.rs:2776-2778 (query/sort.rs:80-93)
//
// This header: shared descriptors, wave-level helpers, the register top-k, the staging
// kernel and the merge kernels.  The hot scoring kernel lives in slg_score.hpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_desc.hpp"

namespace slg {

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
  int32_t th_tk;  // threshold = entry at position k-1 (uniform), never below the floor
  uint32_t th_seg, th_doc;
  uint32_t count;  // real entries held, capped at k (uniform)
  int32_t floor_tk;  // exact lower bound of the final k-th score known up front (or sentinel)

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
    floor_tk = kSentinelTk;
  }
  // At least k docs are known to score >= f: nothing below f can reach the final top-k.
  // Candidates equal to f still pass (ties are broken by doc id later).
  __device__ __forceinline__ void set_floor(float f) {
    floor_tk = total_key(f);
    th_tk = floor_tk;
    th_seg = 0xFFFFFFFFu;
    th_doc = 0xFFFFFFFFu;
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
    if (th_tk < floor_tk) {  // fewer than k entries so far: the up-front bound still rules
      th_tk = floor_tk;
      th_doc = 0xFFFFFFFFu;
      th_seg = 0xFFFFFFFFu;
    }
  }
};

// ---- buffered wave top-k (one segment) ------------------------------------------------------
// push_top_k (query/wand.rs:905-916) without a per-candidate sorted insert: a candidate that
// beats the current threshold is appended to a per-wave LDS buffer (one ds_write for all passing
// lanes of a slot); only when the buffer is full, and once at the end, are the entries ranked
// (every lane counts the entries better than its own) and the k best kept.  The threshold is
// the k-th best after a ranking, the up-front floor before, so it is always a valid lower bound
// of the final k-th score: the surviving set is exactly the top-k under (score desc, doc asc).
//
// Entries are single 64-bit keys: (order-preserving score bits << 32) | ~doc, so "better" is
// one unsigned 64-bit compare.
__device__ __forceinline__ uint32_t ordered_score(float x) {
  const int32_t b = __float_as_int(x);  // == total_key(x) ^ 0x80000000
  return (uint32_t)b ^ ((uint32_t)(b >> 31) | 0x80000000u);
}
__device__ __forceinline__ uint64_t cand_key(float score, uint32_t doc) {
  return ((uint64_t)ordered_score(score) << 32) | (uint32_t)~doc;
}

template <int KREGS>
struct BufTopK {
  static constexpr uint32_t kEntries = 64u * (KREGS + 1);  // >= k + 64 for k <= 64 * KREGS: after a
                                                          // ranking a whole slot of candidates fits
  static constexpr int E = KREGS + 1;                      // entries per lane while ranking
  uint64_t *buf;   // LDS [kEntries]
  uint32_t count;  // uniform: entries held
  uint64_t th;     // uniform: a candidate passes iff key > th

  __device__ __forceinline__ void init(uint64_t *lds) {
    buf = lds;
    count = 0;
    th = 0;  // any real doc passes (~doc > 0)
  }
  // At least k docs score >= f.  Candidates equal to f still pass (doc-id tie break later).
  __device__ __forceinline__ void set_floor(float f) {
    th = ((uint64_t)ordered_score(f) << 32) - 1ull;
  }
  __device__ __forceinline__ bool passes(uint64_t key) const { return key > th; }

  // drop deleted docs (accept(), query/wand.rs:905), rank the rest, keep the k best sorted at
  // buf[0..count), refresh the threshold.  Deleted docs are filtered here, before anything is
  // ranked, so they never influence the threshold.
  __device__ __forceinline__ void compact(uint32_t k, uint32_t lane, const uint32_t *deleted) {
    uint64_t e[E];
    uint32_t rank[E];
#pragma unroll
    for (int i = 0; i < E; i++) {
      const uint32_t idx = lane + 64u * i;
      e[i] = idx < count ? buf[idx] : 0ull;
      rank[i] = 0;
    }
    if (deleted) {
#pragma unroll
      for (int i = 0; i < E; i++) {
        const uint32_t doc = ~(uint32_t)e[i];
        if (e[i] != 0ull && ((deleted[doc >> 5] >> (doc & 31)) & 1u)) e[i] = 0ull;
      }
    }
    uint32_t nvalid = 0;
#pragma unroll
    for (int i = 0; i < E; i++) {
      if (64u * i < count) {  // uniform
        nvalid += (uint32_t)__popcll(__ballot(e[i] != 0ull));
        const uint32_t n_i = count - 64u * i < 64u ? count - 64u * i : 64u;
        for (uint32_t l = 0; l < n_i; l++) {
          const uint64_t c = ((uint64_t)rl((uint32_t)(e[i] >> 32), l) << 32) | rl((uint32_t)e[i], l);
#pragma unroll
          for (int j = 0; j < E; j++) rank[j] += c > e[j] ? 1u : 0u;
        }
      }
    }
    wave_fence();
#pragma unroll
    for (int i = 0; i < E; i++)
      if (e[i] != 0ull && rank[i] < k) buf[rank[i]] = e[i];
    wave_fence();
    count = nvalid < k ? nvalid : k;
    if (nvalid >= k) {
      const uint64_t kth = buf[k - 1];
      const uint64_t u = ((uint64_t)rfl((uint32_t)(kth >> 32)) << 32) | rfl((uint32_t)kth);
      th = u > th ? u : th;
    }
  }

  // split-key forms (the hot path keeps 32-bit halves: hi = ordered score, lo = ~doc)
  __device__ __forceinline__ bool passes(uint32_t hi, uint32_t lo) const {
    const uint32_t th_hi = (uint32_t)(th >> 32), th_lo = (uint32_t)th;
    return hi > th_hi || (hi == th_hi && lo > th_lo);
  }
  // append the candidates of the lanes with `pass` set (pass implies key > th); the caller
  // has checked that they fit (count + popcount <= kEntries)
  __device__ __forceinline__ void append(bool pass, uint32_t hi, uint32_t lo, uint32_t lane) {
    const uint64_t m = __ballot(pass);
    if (m == 0ull) return;
    const uint32_t dest =
        count + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    uint32_t *b32 = reinterpret_cast<uint32_t *>(buf);
    if (pass) {
      b32[2 * dest] = lo;
      b32[2 * dest + 1] = hi;
    }
    count += (uint32_t)__popcll(m);
  }
  // same, ranking first when the buffer would overflow (<= 64 candidates always fit after it)
  __device__ __forceinline__ void append_checked(bool pass, uint32_t hi, uint32_t lo, uint32_t k,
                                                 uint32_t lane, const uint32_t *deleted) {
    if (count + (uint32_t)__popcll(__ballot(pass)) > kEntries) {
      compact(k, lane, deleted);
      pass = pass && passes(hi, lo);
    }
    append(pass, hi, lo, lane);
  }

  // final candidates of this wave: k entries (int32 total_key, doc), sentinel padded (best
  // first when they were ranked; the merge does not rely on the order)
  __device__ __forceinline__ void write_out(int32_t *otk, uint32_t *odoc, uint32_t k, uint32_t lane,
                                            const uint32_t *deleted) {
    if (count > k || (deleted && count)) compact(k, lane, deleted);
    wave_fence();
#pragma unroll
    for (int r = 0; r < KREGS; r++) {
      const uint32_t pos = lane + 64u * r;
      if (pos < k) {
        const uint64_t e = pos < count ? buf[pos] : 0ull;
        otk[pos] = pos < count ? (int32_t)((uint32_t)(e >> 32) ^ 0x80000000u) : kSentinelTk;
        odoc[pos] = ~(uint32_t)e;
      }
    }
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
  const uint64_t *term_offsets;  // [V+1] (unpadded: positions in docs / tfs)
  const uint32_t *docs;          // [P] as uploaded; nullptr: docs_out already holds them (re-derivation)
  const uint32_t *tfs;           // [P]
  const float *term_idf;         // [V]
  const uint16_t *term_field;    // [V] or nullptr
  const float *const *field_doc_len;  // [F] device pointers (or nullptr entries)
  const float *field_avgdl;           // [F]
  float k1, b;
  uint32_t *docs_out;  // out, padded layout: posting i of term t -> i + kListPad * t
  float *imps;         // out, padded layout
};

static __global__ void __launch_bounds__(256) stage_impacts_kernel(StageParams p) {
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
    const uint64_t at = i + (uint64_t)kListPad * t;
    // creation: the uploaded doc id, scattered into the padded layout below; re-derivation after a
    // tombstone update (slg_index_update_deleted): the doc id is read back from there
    uint32_t doc = p.docs ? p.docs[i] : p.docs_out[at];
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
    if (p.docs) p.docs_out[at] = doc;
    p.imps[at] = idf * (tf * (p.k1 + 1.0f)) / fmaxf(denom, 1e-6f);
  }
}

// ---- staging: per-term champion impacts ------------------------------------------------------
// champ[t][r], r = 0..63 (descending): a value v such that at least r+1 postings of term t have
// impact >= v (0 where the list is shorter).  Lane l scans postings l, l+64, ... and keeps its 16
// largest; the 64 lane maxima, sorted, give ranks 1..64.  champ[t][64..67] bound ranks 128, 256,
// 512, 1024: every lane holds j values >= its own j-th largest, so all lanes together hold 64*j
// values >= the minimum over lanes of the j-th largest (j = 2, 4, 8, 16).  Not the exact order
// statistics, but valid lower bounds, which is all the threshold seed needs: a doc's total
// score is >= any one of its (non-negative) per-term contributions.
struct ChampParams {
  const uint64_t *term_offsets;  // [V+1] (unpadded)
  const float *imps;             // padded layout (SegDev)
  const uint32_t *docs;          // padded layout
  const uint32_t *deleted;       // bitmap words or nullptr: deleted docs never count (accept())
  float *champ;                  // [V * kChampions]
  uint32_t n_terms;
};

static __global__ void __launch_bounds__(256) stage_champions_kernel(ChampParams p) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * kWavesPerBlock;
  for (uint32_t t = wave; t < p.n_terms; t += n_waves) {
    const uint64_t a = p.term_offsets[t] + (uint64_t)kListPad * t;
    const uint64_t b = a + (p.term_offsets[t + 1] - p.term_offsets[t]);
    float m[16];
#pragma unroll
    for (int r = 0; r < 16; r++) m[r] = 0.0f;
    for (uint64_t i0 = a; i0 < b; i0 += 64) {
      const uint64_t i = i0 + lane;
      float x = i < b ? p.imps[i] : 0.0f;
      if (p.deleted && i < b) {
        const uint32_t d = p.docs[i];
        if ((p.deleted[d >> 5] >> (d & 31)) & 1u) x = 0.0f;
      }
      if (__ballot(x > m[15]) == 0ull) continue;  // nobody improves: the common case
#pragma unroll
      for (int r = 15; r >= 1; r--) m[r] = x > m[r - 1] ? m[r - 1] : (x > m[r] ? x : m[r]);
      m[0] = x > m[0] ? x : m[0];
    }
    // bounds for ranks 128 .. 1024
    float lo[4] = {m[1], m[3], m[7], m[15]};
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) lo[j] = fminf(lo[j], __shfl_xor(lo[j], o, 64));
    }
    // bitonic sort of the 64 lane maxima, descending (lane 0 = largest)
    float v = m[0];
#pragma unroll
    for (int size = 2; size <= 64; size <<= 1) {
#pragma unroll
      for (int d = size >> 1; d > 0; d >>= 1) {
        const float o = __shfl_xor(v, d, 64);
        const bool up = ((lane & size) == 0) == ((lane & d) == 0);  // keep the larger one
        v = up ? fmaxf(v, o) : fminf(v, o);
      }
    }
    float *row = p.champ + (size_t)t * kChampions;
    row[lane] = v;
    if (lane < 4) row[kChampSorted + lane] = lane == 0 ? lo[0] : lane == 1 ? lo[1] : lane == 2 ? lo[2] : lo[3];
  }
}

// ---- merge: per query, all slice candidate lists -> final top-k ---------------------------
struct MergeParams {
  const QueryRef *queries;
  const uint32_t *slice_seg;  // [n_slices] segment ordinal of each slice
  const int32_t *slice_tk;
  const uint32_t *slice_doc;
  uint32_t *out_doc;
  uint32_t *out_seg;
  float *out_score;
  uint32_t *out_count;
  uint32_t nq;
  uint32_t k;
  // the index's error word (a scoring wave that gave up on a round sets it) is copied behind the result
  // block, so that slg_batch_fetch reads it with the results: ONE device-to-host copy per batch
  const uint32_t *error_flag;
  uint32_t *out_flag;
};

// One wave per query.  The query's slices occupy a contiguous range of the candidate arrays
// ([slice_begin*k, slice_end*k)); it is streamed 64 entries at a time (independent coalesced
// loads), filtered against the running threshold and inserted into the register top-k.
template <int KREGS>
__global__ void __launch_bounds__(256) merge_topk_kernel(MergeParams p) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t q = rfl(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (q >= p.nq) return;
  if (q == 0 && lane == 0 && p.out_flag) *p.out_flag = *p.error_flag;
  const uint32_t k = p.k;
  const QueryRef qr = p.queries[q];
  WaveTopK<KREGS, true> top;
  top.init();
  // candidate arrays are indexed slice*k + i; the host guarantees n_slices*k < 2^32
  const uint32_t f0 = qr.slice_begin * k, f1 = qr.slice_end * k;
  // Several groups of 64 entries (8; 4 in the wide instantiations) are loaded at once and then inserted one after the other: the loop is bound
  // by the latency of its loads (a group's three loads, then a ballot that depends on them: 4.2 us per
  // group with one group in flight — config 3's 110 groups per query made this kernel 0.74 ms, 0.48 of
  // it without a single insert)
  constexpr int G = KREGS <= 4 ? 8 : 4;
  for (uint32_t base = f0; base < f1; base += 64 * G) {
    int32_t gtk[G];
    uint32_t gdoc[G], gseg[G];
#pragma unroll
    for (int u = 0; u < G; u++) {
      const uint32_t f = base + 64u * u + lane;
      gtk[u] = kSentinelTk;
      gdoc[u] = 0xFFFFFFFFu;
      gseg[u] = 0xFFFFFFFFu;
      if (f < f1) {
        gtk[u] = p.slice_tk[f];
        gdoc[u] = p.slice_doc[f];
        gseg[u] = p.slice_seg[f / k];
      }
    }
#pragma unroll
    for (int u = 0; u < G; u++) {
      const int32_t ctk = gtk[u];
      const uint32_t cdoc = gdoc[u], cseg = gseg[u];
      const bool valid = !(ctk == kSentinelTk && cdoc == 0xFFFFFFFFu);
      uint64_t m = __ballot(valid && top.passes(ctk, cseg, cdoc));
      while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        top.insert((int32_t)rl((uint32_t)ctk, l), rl(cseg, l), rl(cdoc, l), k, lane);
        m &= m - 1;
        m &= __ballot(top.passes(ctk, cseg, cdoc));
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
  const uint32_t *doc;    // shard sh's rows start at doc + sh * arr_stride ([nq*k] each)
  const uint32_t *seg;
  const float *score;
  const uint32_t *count;  // shard sh's counts start at count + sh * cnt_stride ([nq])
  uint32_t *out_doc, *out_seg;
  float *out_score;
  uint32_t *out_count;
  uint32_t n_shards, nq, k, seg_stride;
  // elements between two shards' arrays: nq*k / nq for separate shard-major arrays; (3k+1)*nq for
  // both when the shards' contiguous result blocks doc|seg|score|count lie one after another, as an
  // all-gather delivers them
  uint64_t arr_stride, cnt_stride;
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
    const size_t row = (size_t)sh * p.arr_stride + (size_t)q * k;
    const uint32_t cnt = rfl(p.count[(size_t)sh * p.cnt_stride + q]);
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


// ---- the same merge for k beyond the register top-k (k up to 20 001, api/reader.rs:2615-2619) --
// Every shard's row is already sorted by (score desc, segment asc, doc asc) and keys are unique,
// so the global rank of an entry = its own index + the number of better entries in every other
// shard's row (a binary search each); entries ranked below k write themselves to out[rank].
// One workgroup per query.
static __global__ void __launch_bounds__(256) merge_shards_large_kernel(ShardMergeParams p) {
  const uint32_t q = blockIdx.x;
  if (q >= p.nq) return;
  const uint32_t k = p.k;
  uint32_t total = 0;
  for (uint32_t sh = 0; sh < p.n_shards; sh++) {
    const uint32_t c = p.count[(size_t)sh * p.cnt_stride + q];
    total += c < k ? c : k;
  }
  const uint32_t nout = total < k ? total : k;
  for (uint32_t e = threadIdx.x; e < p.n_shards * k; e += blockDim.x) {
    const uint32_t sh = e / k, i = e % k;
    const uint32_t cnt_s = p.count[(size_t)sh * p.cnt_stride + q] < k ? p.count[(size_t)sh * p.cnt_stride + q] : k;
    if (i >= cnt_s) continue;
    const size_t row = (size_t)sh * p.arr_stride + (size_t)q * k;
    const int32_t tk = total_key(p.score[row + i]);
    const uint32_t doc = p.doc[row + i], seg = sh * p.seg_stride + p.seg[row + i];
    uint32_t rank = i;
    for (uint32_t t = 0; t < p.n_shards; t++) {
      if (t == sh) continue;
      const size_t rt = (size_t)t * p.arr_stride + (size_t)q * k;
      uint32_t lo = 0, hi = p.count[(size_t)t * p.cnt_stride + q] < k ? p.count[(size_t)t * p.cnt_stride + q] : k;
      while (lo < hi) {  // first entry of shard t that is NOT better than mine
        const uint32_t mid = (lo + hi) >> 1;
        if (better<true>(total_key(p.score[rt + mid]), t * p.seg_stride + p.seg[rt + mid], p.doc[rt + mid], tk, seg, doc))
          lo = mid + 1;
        else
          hi = mid;
      }
      rank += lo;
    }
    if (rank < k) {
      p.out_doc[(size_t)q * k + rank] = doc;
      p.out_seg[(size_t)q * k + rank] = seg;
      p.out_score[(size_t)q * k + rank] = p.score[row + i];
    }
  }
  for (uint32_t i = nout + threadIdx.x; i < k; i += blockDim.x) {
    p.out_doc[(size_t)q * k + i] = 0u;
    p.out_seg[(size_t)q * k + i] = 0u;
    p.out_score[(size_t)q * k + i] = 0.0f;
  }
  if (threadIdx.x == 0) p.out_count[q] = nout;
}

// ---- doc filters (accept = !deleted && filter, api/reader.rs:3009-3018) -----------
// A filter is kept per segment as a REJECT bitmap (deleted | ~filter, bit d of word d/32) so the
// scoring kernels use it exactly like the tombstone bitmap.
struct FilterBuildParams {
  const uint32_t *deleted;  // or nullptr
  const uint32_t *pass;     // uploaded pass bitmap (words), or nullptr when built from a column
  const void *column;       // i64 / f64 column [n_docs], or nullptr
  double lo_f, hi_f;
  long long lo_i, hi_i;
  int column_kind;  // 0 none, 1 i64, 2 f64
  uint32_t n_docs;
  uint32_t *reject;  // out [ceil(n_docs/32)]
  int invert_pass;        // the pass bitmap marks the docs to REJECT (docs that hold a not-term)
  const uint32_t *pass2;  // a second pass bitmap, AND-ed (the request's own filter), or nullptr
};

// marks the docs of one posting list in a bitmap (slg_index_add_filter_terms: the matcher's not-terms)
struct PostingMarkParams {
  const uint32_t *docs;  // the list's first posting (padded layout: only [0, df) are read)
  uint32_t df;
  uint32_t n_docs;
  uint32_t *bitmap;      // [ceil(n_docs/32)] zeroed before the first list
};
static __global__ void __launch_bounds__(256) posting_mark_kernel(PostingMarkParams p) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.df) return;
  const uint32_t d = p.docs[i];
  if (d < p.n_docs) atomicOr(&p.bitmap[d >> 5], 1u << (d & 31u));
}

static __global__ void __launch_bounds__(256) filter_build_kernel(FilterBuildParams p) {
  const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;  // one doc per lane
  const uint32_t lane = threadIdx.x & 63;
  bool pass = false;
  if (d < p.n_docs) {
    if (p.column_kind == 1) {
      const long long v = static_cast<const long long *>(p.column)[d];
      pass = v >= p.lo_i && v <= p.hi_i;
    } else if (p.column_kind == 2) {
      const double v = static_cast<const double *>(p.column)[d];
      pass = v >= p.lo_f && v <= p.hi_f;  // NaN never passes (query/filters.rs numeric range)
    } else {
      pass = p.pass == nullptr || ((((p.pass[d >> 5] >> (d & 31)) & 1u) != 0u) != (p.invert_pass != 0));
      if (p.pass2 && !((p.pass2[d >> 5] >> (d & 31)) & 1u)) pass = false;
    }
    if (p.deleted && ((p.deleted[d >> 5] >> (d & 31)) & 1u)) pass = false;
  }
  const uint64_t rej = ~__ballot(pass);  // docs past n_docs are rejected too
  const uint32_t w = d >> 5;
  if ((lane & 31u) == 0 && (d < p.n_docs))
    p.reject[w] = lane == 0 ? (uint32_t)rej : (uint32_t)(rej >> 32);
}

// reject bitmap of a filter after new tombstones: out = a | b (b may be null: no tombstones)
struct BitmapOrParams {
  const uint32_t *a, *b;
  uint32_t *out;
  uint32_t n_words;
};
static __global__ void __launch_bounds__(256) bitmap_or_kernel(BitmapOrParams p) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < p.n_words) p.out[i] = p.a[i] | (p.b ? p.b[i] : 0u);
}

// ---- large k (k > 256): per-query radix select over the candidates the scoring kernel kept ----
// The uniform scoring kernel, instead of keeping a per-slice top-k, writes every doc whose score
// beats the seed threshold to its slice's region of `cand` ({ordered score, doc}; the region of
// slice s starts at slice_cbeg[s] and holds slice_ccnt[s] entries).  One workgroup per query
// then finds the k best under (score desc, segment asc, doc asc) = the 96-bit key
// (ordered score, ~segment, ~doc) descending: byte-wise MSB-first radix select with an LDS
// histogram (a pass per byte until the bucket that holds the k-th key is exactly used up), a
// gather of the k winners into LDS and a bitonic sort.  push_top_k / finalize_heap
// (query/wand.rs:905-926) + the cross-segment sort (api/reader.rs:2776-2778) for large k.
struct SelectParams {
  const QueryRef *queries;
  const uint32_t *slice_seg;
  const uint64_t *slice_cbeg;
  const uint32_t *slice_ccnt;
  uint2 *cand;  // .x ordered score, .y doc (0xFFFFFFFF: dropped, e.g. deleted)
  const SegDev *segs;
  const uint32_t *q_filter;             // [nq] 0 = none, f + 1
  const uint32_t *const *reject_table;  // [n_filters * n_segs] reject bitmaps
  uint32_t n_segs;
  uint32_t *out_doc, *out_seg;
  float *out_score;
  uint32_t *out_count;
  uint32_t nq, k;
  const uint32_t *error_flag;  // see MergeParams
  uint32_t *out_flag;
};

constexpr uint32_t kSelectCap = 2048;      // keys sorted in LDS at a time (a rank range of the result)
constexpr uint32_t kSelectMaxSlices = 512;   // slice table in LDS; more: strided slice loops (34 KB of LDS per workgroup: 4 per CU)
constexpr uint32_t kSelectThreads = 512;

// (8 waves per SIMD: four 512-thread workgroups per CU, so that the 1024 queries of a batch are all resident at once)
static __global__ void __launch_bounds__(kSelectThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
select_topk_kernel(SelectParams p) {
  constexpr uint32_t NT = kSelectThreads;
  __shared__ uint32_t hist[256], hist0[256];  // hist0: top score byte of all live candidates
  __shared__ uint32_t w_ok[kSelectCap], w_sg[kSelectCap], w_dc[kSelectCap];
  __shared__ uint32_t sh_pre[3], sh_need, sh_done, sh_nvalid, sh_nwin, sh_anydel, sh_taken;
  __shared__ uint32_t t_end[kSelectMaxSlices];   // inclusive prefix of the slice counts
  __shared__ uint32_t t_nseg[kSelectMaxSlices];  // ~segment of the slice
  __shared__ uint64_t t_base[kSelectMaxSlices];  // first candidate slot of the slice
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t q = blockIdx.x;
  if (q >= p.nq) return;
  if (q == 0 && tid == 0 && p.out_flag) *p.out_flag = *p.error_flag;
  const uint32_t k = p.k;
  const QueryRef qr = p.queries[q];
  const uint32_t sb = qr.slice_begin, se = qr.slice_end, nsl = se - sb;
  const bool table = nsl <= kSelectMaxSlices;
  const uint32_t flt = p.q_filter ? p.q_filter[q] : 0u;

  if (tid == 0) {
    sh_nvalid = 0;
    sh_nwin = 0;
    sh_anydel = table ? 0u : 1u;
    sh_pre[0] = sh_pre[1] = sh_pre[2] = 0;
    sh_need = k;
    sh_done = 0;
    sh_taken = 0;
  }
  if (tid < 256) hist0[tid] = 0;
  __syncthreads();
  // slice table: the query's candidates form one flat index space [0, n)
  if (table) {
    for (uint32_t i = tid; i < nsl; i += NT) {
      const uint32_t sg = p.slice_seg[sb + i];
      t_end[i] = p.slice_ccnt[sb + i];
      t_base[i] = p.slice_cbeg[sb + i];
      t_nseg[i] = ~sg;
      if (p.segs[sg].deleted || flt) sh_anydel = 1;
    }
    __syncthreads();
    for (uint32_t d = 1; d < nsl; d <<= 1) {  // Hillis-Steele inclusive scan
      uint32_t add[kSelectMaxSlices / NT];
      for (uint32_t i = tid, r = 0; i < nsl; i += NT, r++) add[r] = i >= d ? t_end[i - d] : 0u;
      __syncthreads();
      for (uint32_t i = tid, r = 0; i < nsl; i += NT, r++) t_end[i] += add[r];
      __syncthreads();
    }
  }
  __syncthreads();
  const uint32_t n_flat = table && nsl ? t_end[nsl - 1] : 0u;
  const bool anydel = sh_anydel != 0;

  // visit every live candidate of the query: f(ordered score, ~seg, ~doc, slot)
  auto for_each = [&](auto &&f) {
    if (table) {
      // (four named sets, not arrays: with arrays the compiler keeps the loop over them rolled around the
      //  inlined callback and the arrays in scratch memory — a scratch round trip per candidate)
      auto locate = [&](const uint32_t i, uint2 &c, uint64_t &at, uint32_t &ns) {
        c = make_uint2(0u, 0xFFFFFFFFu);
        at = 0;
        ns = 0;
        if (i < n_flat) {
          uint32_t lo = 0, hi = nsl - 1;  // first slice whose inclusive prefix exceeds i
          while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (t_end[mid] > i)
              hi = mid;
            else
              lo = mid + 1;
          }
          at = t_base[lo] + (i - (lo ? t_end[lo - 1] : 0u));
          ns = t_nseg[lo];
          c = p.cand[at];
        }
      };
      // eight independent loads in flight per thread: the kernel's time is the time of the query with the
      // most candidates (all workgroups are resident at once), and its sweeps are latency-bound
      for (uint32_t i0 = tid; i0 < n_flat; i0 += 8 * NT) {
        uint2 c0, c1, c2, c3, c4, c5, c6, c7;
        uint64_t a0, a1, a2, a3, a4, a5, a6, a7;
        uint32_t n0, n1, n2, n3, n4, n5, n6, n7;
        locate(i0, c0, a0, n0);
        locate(i0 + NT, c1, a1, n1);
        locate(i0 + 2 * NT, c2, a2, n2);
        locate(i0 + 3 * NT, c3, a3, n3);
        locate(i0 + 4 * NT, c4, a4, n4);
        locate(i0 + 5 * NT, c5, a5, n5);
        locate(i0 + 6 * NT, c6, a6, n6);
        locate(i0 + 7 * NT, c7, a7, n7);
        if (c0.y != 0xFFFFFFFFu) f(c0.x, n0, ~c0.y, a0);
        if (c1.y != 0xFFFFFFFFu) f(c1.x, n1, ~c1.y, a1);
        if (c2.y != 0xFFFFFFFFu) f(c2.x, n2, ~c2.y, a2);
        if (c3.y != 0xFFFFFFFFu) f(c3.x, n3, ~c3.y, a3);
        if (c4.y != 0xFFFFFFFFu) f(c4.x, n4, ~c4.y, a4);
        if (c5.y != 0xFFFFFFFFu) f(c5.x, n5, ~c5.y, a5);
        if (c6.y != 0xFFFFFFFFu) f(c6.x, n6, ~c6.y, a6);
        if (c7.y != 0xFFFFFFFFu) f(c7.x, n7, ~c7.y, a7);
      }
    } else {
      for (uint32_t s = sb + wave; s < se; s += NT / 64) {
        const uint64_t base = p.slice_cbeg[s];
        const uint32_t cnt = p.slice_ccnt[s];
        const uint32_t nseg = ~p.slice_seg[s];
        for (uint32_t i = lane; i < cnt; i += 64) {
          const uint2 c = p.cand[base + i];
          if (c.y != 0xFFFFFFFFu) f(c.x, nseg, ~c.y, base + i);
        }
      }
    }
  };

  // ---- sweep 1: accept() (drop deleted docs) + histogram of the top score byte ----
  for_each([&](uint32_t a, uint32_t nseg, uint32_t ndoc, uint64_t at) {
    if (anydel) {
      const uint32_t *del = flt ? p.reject_table[(size_t)(flt - 1) * p.n_segs + ~nseg] : p.segs[~nseg].deleted;
      const uint32_t d = ~ndoc;
      if (del && ((del[d >> 5] >> (d & 31)) & 1u)) {
        p.cand[at].y = 0xFFFFFFFFu;
        return;
      }
    }
    // (scores of one query share their exponent: nearly all candidates fall into one or two bins, and LDS
    //  atomics on one address are served one at a time — 2K of them were 27 us of this kernel.  The
    //  lanes of a wave that hold the same bin add their count once.)
    uint64_t todo = __ballot(true);  // the lanes that are here
    const uint32_t bin = a >> 24;
    while (todo) {
      const uint32_t l = (uint32_t)__builtin_ctzll(todo);
      const uint32_t b = rl(bin, l);
      const uint64_t same = __ballot(bin == b) & todo;
      if (lane == l) atomicAdd(&hist0[b], (uint32_t)__popcll(same));
      todo &= ~same;
    }
  });
  __syncthreads();
  if (wave == 0) {  // (one wave adds the 256 bins: a single thread walking them was ~10 us of this kernel)
    uint32_t nv = hist0[4 * lane] + hist0[4 * lane + 1] + hist0[4 * lane + 2] + hist0[4 * lane + 3];
    for (int o = 32; o > 0; o >>= 1) nv += __shfl_xor(nv, o, 64);
    if (lane == 0) sh_nvalid = nv;
  }
  __syncthreads();
  const uint32_t nvalid = sh_nvalid;
  const uint32_t nout = nvalid < k ? nvalid : k;

  // decided-prefix compare: key (a, b, c) >= prefix (p0, p1, p2) on the top (level + 1) bytes
  auto at_or_above = [](uint32_t a, uint32_t b, uint32_t c, uint32_t p0, uint32_t p1, uint32_t p2,
                        uint32_t level) {
    const uint32_t wd = level >> 2, shift = 24u - 8u * (level & 3u);
    const uint32_t keep = ~((1u << shift) - 1u);  // decided bytes of word wd
    const uint32_t ka = wd == 0 ? (a & keep) : a, kb = wd == 1 ? (b & keep) : b,
                   kc = wd == 2 ? (c & keep) : c;
    if (wd == 0) return ka >= p0;
    if (wd == 1) return ka > p0 || (ka == p0 && kb >= p1);
    return ka > p0 || (ka == p0 && (kb > p1 || (kb == p1 && kc >= p2)));
  };

  // ---- the result is produced in rank ranges of at most kSelectCap keys: for each range a
  //      byte-wise radix select finds the prefix of its last key (exactly, except for the final
  //      range, which may take a few keys more than needed and drops them after the sort), the
  //      keys between this prefix and the previous range's are gathered, sorted, written ----
  uint32_t k_done = 0;
  bool have_prev = false;
  uint32_t q0 = 0, q1 = 0, q2 = 0, q_level = 0;  // prefix of the previous range
  while (k_done < nout) {
    const uint32_t target = nout - k_done > kSelectCap ? k_done + kSelectCap : nout;
    const bool last = target == nout;
    // the final range may take more keys than it emits (dropped after the sort) as long as they fit the
    // sort — whose size is the next power of two: no more than that of the keys still to emit (k = 1001:
    // a 1024-key sort; up to 2048 keys were 66 stages x 2 passes against 55 x 1)
    uint32_t cap_last = 64;
    while (cap_last < nout - k_done) cap_last <<= 1;
    cap_last = cap_last < kSelectCap ? cap_last : kSelectCap;
    if (n_flat > 16u * NT) cap_last = kSelectCap;  // (many candidates: a further select level costs more than the larger sort)
    const bool all = last && nvalid - k_done <= cap_last;  // everything left fits: no select
    __syncthreads();
    if (tid == 0) {
      sh_pre[0] = sh_pre[1] = sh_pre[2] = 0;
      sh_need = target;
      sh_done = all ? 2u : 0u;
      sh_taken = all ? nvalid : 0u;
      sh_nwin = 0;
    }
    __syncthreads();
    uint32_t level = 0;
    if (!all) {
      for (;; level++) {
        const uint32_t wd = level >> 2, shift = 24u - 8u * (level & 3u);
        if (tid < 256) hist[tid] = level == 0 ? hist0[tid] : 0u;
        __syncthreads();
        if (level > 0) {
          const uint32_t p0 = sh_pre[0], p1 = sh_pre[1], p2 = sh_pre[2];
          const uint32_t himask = shift == 24u ? 0u : ~((1u << (shift + 8u)) - 1u);  // bytes above
          for_each([&](uint32_t a, uint32_t b, uint32_t c, uint64_t) {
            const uint32_t w = wd == 0 ? a : (wd == 1 ? b : c), pw = wd == 0 ? p0 : (wd == 1 ? p1 : p2);
            bool m = (w & himask) == (pw & himask);
            if (wd >= 1) m = m && a == p0;
            if (wd >= 2) m = m && b == p1;
            if (m) atomicAdd(&hist[(w >> shift) & 255u], 1u);
          });
          __syncthreads();
        }
        if (wave == 0) {
          // lane l owns bins 255-4l .. 252-4l (descending); inclusive prefix of the lane sums
          const uint32_t b0 = 255u - 4u * lane;
          const uint32_t h0 = hist[b0], h1 = hist[b0 - 1], h2 = hist[b0 - 2], h3 = hist[b0 - 3];
          const uint32_t tot = h0 + h1 + h2 + h3;
          uint32_t incl = tot;
          for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, 64);
            if ((int)lane >= o) incl += v;
          }
          const uint32_t need = sh_need, excl = incl - tot;
          if (excl < need && need <= incl) {  // exactly one lane (need <= matching keys)
            uint32_t cum = excl, b = b0, h = h0;
            if (cum + h < need) { cum += h; b = b0 - 1; h = h1; }
            if (cum + h < need) { cum += h; b = b0 - 2; h = h2; }
            if (cum + h < need) { cum += h; b = b0 - 3; h = h3; }
            sh_pre[wd] = sh_pre[wd] | (b << shift);
            sh_need = need - cum;  // rank of the target key inside the bucket
            const uint32_t taken = (target - (need - cum)) + h;  // keys with prefix >= the chosen one
            sh_taken = taken;
            // exact when the whole bucket is wanted; the final range may overshoot within the buffer
            sh_done = (h == need - cum || level == 11 || (last && taken - k_done <= cap_last)) ? 1u : 0u;
          }
        }
        __syncthreads();
        if (sh_done) break;
      }
    }
      const uint32_t p0 = sh_pre[0], p1 = sh_pre[1], p2 = sh_pre[2];
    const uint32_t count = (sh_taken - k_done) < kSelectCap ? (sh_taken - k_done) : kSelectCap;
    // ---- gather the keys of this range ----
    for_each([&](uint32_t a, uint32_t b, uint32_t c, uint64_t) {
      bool win = all || at_or_above(a, b, c, p0, p1, p2, level);
      if (win && have_prev) win = !at_or_above(a, b, c, q0, q1, q2, q_level);
      const uint64_t wm = __ballot(win);  // (one add per wave: see the histogram above)
      uint32_t wbase = 0;
      if (wm != 0ull) {
        const uint32_t l0 = (uint32_t)__builtin_ctzll(wm);
        if (lane == l0) wbase = atomicAdd(&sh_nwin, (uint32_t)__popcll(wm));
        wbase = rl(wbase, l0);
      }
      if (win) {
        const uint32_t at = wbase + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
        if (at < kSelectCap) {
          w_ok[at] = a;
          w_sg[at] = b;
          w_dc[at] = c;
        }
      }
    });
    __syncthreads();
      // ---- bitonic sort, descending 96-bit key ----
    uint32_t n2 = 1;
    while (n2 < count) n2 <<= 1;
    for (uint32_t i = count + tid; i < n2; i += NT) {
      w_ok[i] = 0;
      w_sg[i] = 0;
      w_dc[i] = 0;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= n2; size <<= 1) {
      for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
        for (uint32_t t = tid; t < (n2 >> 1); t += NT) {
          const uint32_t i = 2 * t - (t & (stride - 1));  // lower index of the pair
          const uint32_t j = i + stride;
          const bool desc = (i & size) == 0;
          const uint32_t ai = w_ok[i], bi = w_sg[i], ci = w_dc[i];
          const uint32_t aj = w_ok[j], bj = w_sg[j], cj = w_dc[j];
          const bool i_lt_j = ai < aj || (ai == aj && (bi < bj || (bi == bj && ci < cj)));
          if (i_lt_j == desc) {
            w_ok[i] = aj; w_sg[i] = bj; w_dc[i] = cj;
            w_ok[j] = ai; w_sg[j] = bi; w_dc[j] = ci;
          }
        }
        __syncthreads();
      }
    }
      const uint32_t emit = target - k_done;  // (the final range drops what it took beyond k)
    for (uint32_t i = tid; i < emit; i += NT) {
      const int32_t tk = (int32_t)(w_ok[i] ^ 0x80000000u);
      p.out_doc[(size_t)q * k + k_done + i] = ~w_dc[i];
      p.out_seg[(size_t)q * k + k_done + i] = ~w_sg[i];
      p.out_score[(size_t)q * k + k_done + i] = key_to_float(tk);
    }
    k_done = target;
    have_prev = true;
    q0 = p0;
    q1 = p1;
    q2 = p2;
    q_level = level;
  }
  for (uint32_t i = nout + tid; i < k; i += NT) {
    p.out_doc[(size_t)q * k + i] = 0u;
    p.out_seg[(size_t)q * k + i] = 0u;
    p.out_score[(size_t)q * k + i] = 0.0f;
  }
  if (tid == 0) p.out_count[q] = nout;
}

}  // namespace slg
