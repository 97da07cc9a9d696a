// slg_score.hpp — shared round-scoring definitions (partition_rounds_kernel, RoundScoreParams,
// wave scans) and the first-generation PACKED kernel score_rounds_kernel: lists packed back to
// back in the register slots with per-lane list ids.  It is superseded by slg_score_uni.hpp
// (<= 4 lists) and slg_score_multi.hpp (5..32 lists, score plans) and now runs only for the
// opt-in MaxScore path (SLG_MAXSCORE=1) or with SLG_NO_UNIFORM=1.
//
// Restates query/wand.rs:459-566 (brute_force: every posting of every term is scored and
// summed per doc, here in ScorePlan leaf order planner.rs:122-135) and push_top_k
// (wand.rs:905-916) for a whole batch of queries.
//
// Work decomposition (built by the host + partition_rounds_kernel):
//   sub-query = (query, segment);  round = a doc-id range of a sub-query holding <= kCap
//   postings over all its lists, with the exact per-list posting ranges known up front;
//   slice = kRoundsPerSlice consecutive rounds, owned by ONE WAVE (no workgroup barriers).
//
// Per round the wave
//   1. loads the round's postings (doc id + precomputed impact) into registers with plain
//      coalesced dword loads; the loads of round r+1 are issued before round r is processed;
//   2. ORs one bit per posting into an LDS bitmap over the round's doc window and
//      prefix-popcounts it: rank(doc) is a dense, collision-free accumulator slot;
//   3. adds weight*impact into vals[rank] one list at a time (program order inside a wave ==
//      term order, so the f32 sum is bit-identical to the reference's leaf-order sum);
//   4. the first list that touched a doc "owns" it, reads the finished sum back and offers
//      it to the wave-wide sorted top-k (registers, DPP shifts).
// MaxScore pruning (strategies Wand/Bmw): the host marks as NON-ESSENTIAL the lists whose summed
// maximum contributions stay below the seed threshold theta0 (slg_api.hip).  A doc that occurs
// only in such lists can never reach the top-k, so those lists set no bitmap bits and own no
// docs: their postings are merely streamed and probed against the bitmap built by the essential
// lists, and the (few) hits are added in their term-order pass.  Results are identical to the
// exhaustive scorer (the reference's own standard: tests/pruning.rs:44-104 Bm25 == Wand == Bmw).
// Integer/f32 VALU + LDS work bounded by the HBM stream of postings; no MFMA on this path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_kernels.hpp"

namespace slg {

constexpr int kNSlot = 8;                 // 64-posting register slots per round
constexpr int kCap = kNSlot * 64;         // postings per round
constexpr int kRoundTarget = 416;         // planned postings per round (host + partition)
constexpr int kMaxRoundsPerSlice = 16;  // and (rounds+1)*T <= 64: cut points live in one VGPR
constexpr int kDefaultRoundsPerSlice = 8;
constexpr int kUniRoundsPerSlice = 4;     // uniform kernel (slg_score_uni.hpp)
constexpr int kSpanWords = 512;           // bitmap words per window
constexpr uint32_t kSpan = kSpanWords * 32;  // docs per window
// per-wave LDS: bitmap words, exclusive prefix popcounts, accumulators
constexpr int kScoreWaveLds = kSpanWords * 4 + kSpanWords * 4 + kCap * 4 + 64 * 4;  // + dump words

// ---- partition: exact per-list cut points of every round --------------------------------------
struct RoundPartParams {
  const RoundQuery *sq;
  const TermRef *terms;
  const uint32_t *bnd_sq;  // [n_boundaries] sub-query of each boundary task
  const SegDev *segs;
  uint32_t *bounds;
  uint32_t *rdoc;
  uint32_t *q_scored;  // [nq] zeroed here (saves a memset node per batch)
  uint32_t nq;
  uint32_t n_boundaries;
};

// 8 threads per boundary: thread u handles lists u, u+8, ...
static __global__ void __launch_bounds__(256) partition_rounds_kernel(RoundPartParams p) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < p.nq) p.q_scored[gid] = 0;
  const uint32_t b = gid >> 3, u = gid & 7;
  if (b >= p.n_boundaries) return;
  const uint32_t sqi = p.bnd_sq[b];
  const RoundQuery s = p.sq[sqi];
  const uint32_t j = b - s.bnd_begin;
  const uint32_t *docs = p.segs[s.seg].docs;
  const TermRef L = p.terms[s.term_begin + s.longest];
  const uint32_t stride = (L.df + s.n_rounds - 1) / s.n_rounds;
  const uint64_t posL = (uint64_t)j * stride;
  const bool first = j == 0, last = j >= s.n_rounds || posL >= L.df;
  uint32_t target = 0;
  if (!first && !last) target = docs[L.off + posL];
  for (uint32_t t = u; t < s.n_terms; t += 8) {
    const TermRef me = p.terms[s.term_begin + t];
    uint32_t out;
    if (first) {
      out = 0;
    } else if (last) {
      out = me.df;
    } else if (t == s.longest) {
      out = (uint32_t)posL;
    } else {
      const uint32_t *d = docs + me.off;
      uint32_t lo = 0, hi = me.df;  // first index with d[idx] >= target
      // bracket around the position a uniform doc-id distribution predicts (widened until it
      // holds the answer), then bisect: ~10 dependent loads instead of ~log2(df)
      {
        const uint32_t nd = p.segs[s.seg].n_docs;
        const uint32_t g = (uint32_t)(((uint64_t)me.df * target) / (nd ? nd : 1u));
        for (uint32_t w = 64; w < me.df; w <<= 3) {
          const uint32_t a = g > w ? g - w : 0u;
          const uint32_t b = (uint64_t)g + w < me.df ? g + w : me.df;
          const bool lo_ok = a == 0u || d[a - 1] < target;   // answer >= a
          const bool hi_ok = b == me.df || d[b - 1] >= target;  // answer <= b - 1 < b
          if (lo_ok && hi_ok) {
            lo = a;
            hi = b == me.df ? me.df : b - 1;  // d[b-1] >= target: answer <= b-1
            if (b != me.df) hi = b - 1;
            break;
          }
        }
      }
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (d[mid] < target)
          lo = mid + 1;
        else
          hi = mid;
      }
      out = lo;
    }
    p.bounds[s.bounds_begin + j * s.n_terms + t] = out;
  }
  if (u == 0) {
    uint32_t rd;
    if (first) {  // smallest first doc over the lists
      rd = 0xFFFFFFFFu;
      for (uint32_t t = 0; t < s.n_terms; t++) {
        const TermRef me = p.terms[s.term_begin + t];
        const uint32_t d0 = docs[me.off];
        rd = d0 < rd ? d0 : rd;
      }
    } else if (last) {  // one past the largest last doc
      rd = 0;
      for (uint32_t t = 0; t < s.n_terms; t++) {
        const TermRef me = p.terms[s.term_begin + t];
        const uint32_t d1 = docs[me.off + me.df - 1] + 1u;
        rd = d1 > rd ? d1 : rd;
      }
    } else {
      rd = target;
    }
    p.rdoc[s.rdoc_begin + j] = rd;
  }
}

// ---- scoring -----------------------------------------------------------------------------------
struct RoundScoreParams {
  const RoundQuery *sq;
  const TermRef *terms;
  const uint32_t *slice_sq;
  const uint32_t *slice_order;  // [n_slices] wave w runs slice slice_order[w] (most rounds first)
  const SegDev *segs;
  const uint32_t *bounds;
  const uint32_t *rdoc;
  int32_t *slice_tk;    // [n_slices * k]
  uint32_t *slice_doc;  // [n_slices * k]
  uint32_t *q_scored;   // [nq] or null
  const uint32_t *const *reject_table;  // [n_filters * n_segs] reject bitmaps (doc filters)
  uint32_t n_segs;
  uint32_t plan_batch;  // multi kernel: some sub-query has a score plan (extra LDS is allocated)
  // large-k mode of the uniform kernel (k > 256): candidates instead of per-slice top-k lists
  uint2 *cand;            // {ordered score, doc}; sub-query region + posting offset of the slice
  uint64_t *slice_cbeg;   // [n_slices] first candidate slot of the slice
  uint32_t *slice_ccnt;   // [n_slices] candidates written
  uint32_t n_slices;
  uint32_t k;
  uint32_t dbg;
  unsigned long long *stamps;  // [n_slices * 8] (SLG_STAMPS builds only)
};

// inclusive wave scan (sum) with DPP row shifts + row broadcasts (gfx9 DPP controls)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  // row_shr:1,2,3 then 4, 8 within rows of 16; then row_bcast:15 and row_bcast:31
  uint32_t x = v;
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, true);   // row_bcast:15
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, true);   // row_bcast:31
  return x;
}

#ifdef SLG_STAMPS
// diagnostic build only: cycle stamps per phase, summed per wave, stored to a debug buffer
#define SLG_STAMP(i)                                                                        \
  do {                                                                                      \
    unsigned long long _t;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    st_acc[i] += _t - st_last;                                                              \
    st_last = _t;                                                                           \
  } while (0)
#else
#define SLG_STAMP(i) \
  do {               \
  } while (0)
#endif

typedef const __attribute__((address_space(1))) uint32_t *gu32_t;
typedef const __attribute__((address_space(1))) float *gf32_t;

template <int TT>
struct ListRegs {  // per-list uniform state for up to TT lists (TT is a compile-time bound)
  uint32_t rel_lo[TT], rel_hi[TT];  // absolute posting index of v == 0 (mod 2^64)
  uint32_t start[TT];               // first v of list t in the round
};

// One round's postings in registers: element v = jj*64 + lane of the concatenated per-list
// ranges.  tp packs the list index of the 8 elements (8 bits each).
struct Elems {
  uint32_t doc[kNSlot];
  float imp[kNSlot];
  uint32_t tp[2];
  __device__ __forceinline__ uint32_t t(int jj) const { return (tp[jj >> 2] >> ((jj & 3) * 8)) & 0xFFu; }
};

template <int KREGS, int TT>
__global__ void __launch_bounds__(256) score_rounds_kernel(RoundScoreParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wib = threadIdx.x >> 6;
  const uint32_t widx = rfl(blockIdx.x * (blockDim.x >> 6) + wib);
  if (widx >= p.n_slices) return;  // waves are independent: no workgroup barrier anywhere
  const uint32_t slice = rfl(p.slice_order[widx]);

  uint32_t *bm = reinterpret_cast<uint32_t *>(smem + (size_t)wib * kScoreWaveLds);
  uint32_t *pre = bm + kSpanWords;
  uint32_t *vals = pre + kSpanWords;
  uint4 *bm4 = reinterpret_cast<uint4 *>(bm);
  uint4 *pre4 = reinterpret_cast<uint4 *>(pre);
  uint4 *vals4 = reinterpret_cast<uint4 *>(vals);

  const uint32_t sqi = rfl(p.slice_sq[slice]);
  const RoundQuery s = p.sq[sqi];
  const uint32_t T = rfl(s.n_terms);
  const uint32_t rps = rfl(s.rounds_per_slice);
  const uint32_t r0 = (slice - rfl(s.slice_begin)) * rps;
  const uint32_t r_end = rfl(s.n_rounds) < r0 + rps ? rfl(s.n_rounds) : r0 + rps;
  const uint32_t n_r = r_end - r0;
  const SegDev sd = p.segs[s.seg];
  // pointers fetched from memory are generic to the compiler; pin them to the global address
  // space so the posting stream uses global_load (flat loads would also occupy lgkmcnt and
  // serialize against the LDS traffic)
  const gu32_t gdocs = (gu32_t)sd.docs;
  const gf32_t gimps = (gf32_t)sd.imps;
  // accept(): tombstones, or the reject bitmap (deleted | ~filter) of the query's doc filter
  const uint32_t fid = rfl(s.filter);
  const gu32_t gdel = (gu32_t)(fid ? p.reject_table[(size_t)(fid - 1) * p.n_segs + s.seg] : sd.deleted);
  const uint32_t k = p.k;
  const uint32_t ess_mask = rfl(s.ess_mask);

  // lane t < T: list t's posting offset; weights go to scalars
  uint64_t my_off = 0;
  float my_w = 0.0f;
  uint32_t my_term = 0;
  if (lane < T) {
    const TermRef tr = p.terms[s.term_begin + lane];
    my_off = tr.off;
    my_w = tr.weight;
    my_term = tr.term;
  }
  // all cut points of the slice in ONE register: lane i holds bounds[r0*T + i] for
  // i < (rounds+1)*T (the host picks rounds_per_slice so that this fits 64 lanes); likewise
  // the rounds' first doc ids.
  const uint32_t bflat = lane < (n_r + 1) * T ? p.bounds[s.bounds_begin + r0 * T + lane] : 0u;
  const uint32_t dflat = lane <= n_r ? p.rdoc[s.rdoc_begin + r0 + lane] : 0u;

  WaveTopK<KREGS, false> top;
  top.init();
  if (sd.champ != nullptr && k <= 1024u && fid == 0) {  // (a filter may reject the champions)
    // threshold seed: >= k postings of term t have impact >= champ[t][k-1], and a doc's total
    // is >= any single (non-negative) contribution, so >= k docs score >= w_t * champ[t][k-1]
    float f = 0.0f;
    if (lane < T && my_w > 0.0f) f = my_w * ((const gf32_t)sd.champ)[(size_t)my_term * kChampions + champ_index(k)];
    float best = 0.0f;
    for (uint32_t t = 0; t < T; t++) best = fmaxf(best, __int_as_float((int)rl((uint32_t)__float_as_int(f), t)));
    // a negative weight would break "total >= single contribution": no seed then
    const bool anyneg = __ballot(lane < T && !(my_w >= 0.0f)) != 0ull;
    if (best > 0.0f && !anyneg) top.set_floor(best);
  }
  uint32_t n_scored = 0;
#ifdef SLG_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

  // ---- load up to kCap postings of [lo, lo+cnt) per list (lane t holds list t's values) ----
  auto issue = [&](Elems &e, const uint32_t lo, const uint32_t cnt, uint32_t &total) {
    ListRegs<TT> L;
    uint32_t run = 0;
#pragma unroll
    for (int t = 0; t < TT; t++) {  // lanes >= T hold zeros: absent lists are empty lists
      const uint32_t c = rl(cnt, t);
      const uint64_t off = ((uint64_t)rl((uint32_t)(my_off >> 32), t) << 32) | rl((uint32_t)my_off, t);
      const uint64_t rel = off + rl(lo, t) - run;
      L.rel_lo[t] = (uint32_t)rel;
      L.rel_hi[t] = (uint32_t)(rel >> 32);
      L.start[t] = run;
      run += c;
    }
    total = run;
    e.tp[0] = 0;
    e.tp[1] = 0;
#pragma unroll
    for (int jj = 0; jj < kNSlot; jj++) {
      const uint32_t v = jj * 64 + lane;
      uint32_t t = 0, rlo = L.rel_lo[0], rhi = L.rel_hi[0];
#pragma unroll
      for (int tt = 1; tt < TT; tt++) {
        const bool ge = v >= L.start[tt];
        t = ge ? (uint32_t)tt : t;
        rlo = ge ? L.rel_lo[tt] : rlo;
        rhi = ge ? L.rel_hi[tt] : rhi;
      }
      e.tp[jj >> 2] |= t << ((jj & 3) * 8);
      e.doc[jj] = kDocEnd;
      e.imp[jj] = 0.0f;
      if (v < total) {
        const uint64_t a = (((uint64_t)rhi << 32) | rlo) + v;
        e.doc[jj] = gdocs[a];
        e.imp[jj] = gimps[a];
      }
    }
  };

  // ---- accumulate the elements of `e` selected by validmask (bit jj per lane) whose docs
  //      lie in [wbase, wbase + kSpan).  Straight-line over the 8 slots: each phase issues its
  //      LDS operations back to back and waits once (measured on gfx950: plain LDS ops cost
  //      ~6 CU-cycles per wave-instruction but ~100 cycles of dependent latency; LDS float
  //      atomics ~190 cycles per instruction, so sums use plain read-add-write).
  //      * bitmap OR with return value: the posting that sets a doc's bit first owns the doc
  //        (slots are laid out list by list, so the owner is the first list in term order);
  //      * rank = prefix popcount: a dense accumulator slot per distinct doc;
  //      * owners store 0.0 + x; the (few) later postings of the same doc add in term order:
  //        ((0.0 + x_a) + x_b) + ... is `or_insert(0.0) += score` (query/wand.rs:539) summed in
  //        ScorePlan leaf order (planner.rs:122-135). ----
  // ne_cur (lane t): cursor of non-essential list t inside the current round; ne_end: its end
  uint32_t ne_cur = 0, ne_end = 0;

  auto accumulate = [&](Elems &e, const uint32_t validmask, const uint32_t wbase,
                        const uint32_t dend) {
    SLG_STAMP(1);
    // P0: clear the bitmap and the accumulators (+0.0f)
    bm4[lane] = make_uint4(0u, 0u, 0u, 0u);
    bm4[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
    vals4[lane] = make_uint4(0u, 0u, 0u, 0u);
    vals4[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
    wave_fence();
    // P1: one bit per posting; the returned old word tells who came first.  Slots are
    // laid out list by list, so across slots "first" is term order.  Inside one slot that
    // straddles two lists the hardware may pick either of two same-doc lanes as first; the
    // sum of two terms commutes, so that is still bit-exact.  Only a slot holding three or
    // more lists (tiny lists) needs the ordered path.
    uint32_t wi[kNSlot], bit[kNSlot], ownmask = 0;
    {
      uint32_t oldw[kNSlot];
      bool tiny = false;
#pragma unroll
      for (int jj = 0; jj < kNSlot; jj++) {
        const uint32_t rel = e.doc[jj] - wbase;
        const bool valid = (validmask >> jj) & 1u;
        // transposed bitmap: doc d -> word d mod 512, bit d / 512 (neighbouring docs of a
        // dense list hit neighbouring words); the prefix popcount over (word, bit) order is
        // still a perfect hash doc -> accumulator slot.
        wi[jj] = rel & (kSpanWords - 1);  // in range even for idle lanes
        bit[jj] = valid ? 1u << ((rel >> 9) & 31) : 0u;
        oldw[jj] = atomicOr(&bm[wi[jj]], bit[jj]);
        const uint32_t tj = e.t(jj);
        tiny = tiny || (bit[jj] != 0u && tj > rfl(tj) + 1u);
      }
#pragma unroll
      for (int jj = 0; jj < kNSlot; jj++)
        ownmask |= (bit[jj] != 0u && (oldw[jj] & bit[jj]) == 0u) ? (1u << jj) : 0u;
      if (__ballot(tiny) != 0ull) {  // rare: redo the claims strictly in term order
        wave_fence();
        bm4[lane] = make_uint4(0u, 0u, 0u, 0u);
        bm4[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
        wave_fence();
        ownmask = 0;
        for (uint32_t tc = 0; tc < T; tc++) {
#pragma unroll
          for (int jj = 0; jj < kNSlot; jj++) {
            const bool mine = bit[jj] != 0u && e.t(jj) == tc;
            const uint32_t o = atomicOr(&bm[wi[jj]], mine ? bit[jj] : 0u);
            ownmask |= (mine && (o & bit[jj]) == 0u) ? (1u << jj) : 0u;
          }
          wave_fence();
        }
      }
    }
    wave_fence();
    SLG_STAMP(2);
    // P2: exclusive prefix popcount.  Lane l owns words 4l..4l+3 and 256+4l..256+4l+3 (two
    // conflict-free ds_read_b128 at a 16-byte lane stride); ranks are numbered lane-major
    // (any bijection doc -> slot works), so one wave scan suffices.
    {
      const uint4 a = bm4[lane], b = bm4[lane + 64];
      const uint32_t c0 = __popc(a.x), c1 = c0 + __popc(a.y), c2 = c1 + __popc(a.z),
                     c3 = c2 + __popc(a.w), c4 = c3 + __popc(b.x), c5 = c4 + __popc(b.y),
                     c6 = c5 + __popc(b.z), c7 = c6 + __popc(b.w);
      const uint32_t incl = wave_incl_scan(c7);
      const uint32_t ex = incl - c7;
      pre4[lane] = make_uint4(ex, ex + c0, ex + c1, ex + c2);
      pre4[lane + 64] = make_uint4(ex + c3, ex + c4, ex + c5, ex + c6);
      n_scored += rl(incl, 63);
    }
    wave_fence();
    SLG_STAMP(3);
    // P3a: rank of every (essential) posting and its weighted impact
    uint32_t slot[kNSlot];
    uint32_t tlo[kNSlot], thi[kNSlot];  // uniform: first / last list present in the slot
    {
      uint32_t wd[kNSlot], pf[kNSlot];
      float w[kNSlot];
#pragma unroll
      for (int jj = 0; jj < kNSlot; jj++) {
        wd[jj] = bm[wi[jj]];
        pf[jj] = pre[wi[jj]];
        // score_tf: base * weight (query/wand.rs:285); cross-lane read of list t's weight
        w[jj] = __shfl(my_w, (int)e.t(jj), 64);
      }
#pragma unroll
      for (int jj = 0; jj < kNSlot; jj++) {
        slot[jj] = (pf[jj] + __popc(wd[jj] & (bit[jj] - 1u))) & (kCap - 1);
        e.imp[jj] = bit[jj] != 0u ? e.imp[jj] * w[jj] : e.imp[jj];  // x, in place
        const uint32_t tj = e.t(jj);
        tlo[jj] = rfl(tj);
        thi[jj] = rl(tj, 63);  // idle lanes past the end report the last list: harmless
      }
    }
    wave_fence();
    // P3b: per-doc sums in term order.  vals started at +0.0, so each doc's sum is
    // ((0.0 + x_a) + x_b) + ... exactly as the reference forms it.
    const uint32_t full_mask = T >= 32 ? 0xFFFFFFFFu : ((1u << T) - 1u);
    if (ess_mask == full_mask) {
      // all lists essential: the owner is the first list (in term order) holding the doc, so it
      // can store 0.0 + x directly; only later postings of the same doc read-add-write
      uint32_t lmask = 0;
#pragma unroll
      for (int jj = 0; jj < kNSlot; jj++) {
        const bool own = (ownmask >> jj) & 1u;
        vals[own ? slot[jj] : kCap + lane] = __float_as_uint(0.0f + e.imp[jj]);
        lmask |= (bit[jj] != 0u && !own) ? (1u << jj) : 0u;
      }
      wave_fence();
      if (__ballot(lmask != 0u) != 0ull) {
        for (uint32_t tc = 0; tc < T; tc++) {
          uint32_t am = 0;
#pragma unroll
          for (int jj = 0; jj < kNSlot; jj++) am |= (((lmask >> jj) & 1u) && e.t(jj) == tc) ? 1u : 0u;
          if (__ballot(am != 0u) == 0ull) continue;
          // only the slots that hold postings of list tc (uniform test): with many short lists
          // that is one or two slots, not all eight
          uint32_t old[kNSlot];
#pragma unroll
          for (int jj = 0; jj < kNSlot; jj++) {
            old[jj] = 0;
            if (tc >= tlo[jj] && tc <= thi[jj]) old[jj] = vals[slot[jj]];
          }
#pragma unroll
          for (int jj = 0; jj < kNSlot; jj++) {
            if (tc >= tlo[jj] && tc <= thi[jj]) {
              const bool act = ((lmask >> jj) & 1u) && e.t(jj) == tc;
              vals[act ? slot[jj] : kCap + lane] = __float_as_uint(__uint_as_float(old[jj]) + e.imp[jj]);
            }
          }
          wave_fence();
        }
      }
    } else
    for (uint32_t tc = 0; tc < T; tc++) {
      if ((ess_mask >> tc) & 1u) {
        // essential list: its postings sit in the register slots
#pragma unroll
        for (int jj = 0; jj < kNSlot; jj++) {
          if (tc >= tlo[jj] && tc <= thi[jj]) {  // uniform
            const bool act = bit[jj] != 0u && e.t(jj) == tc;
            const uint32_t old = vals[slot[jj]];
            vals[act ? slot[jj] : kCap + lane] = __float_as_uint(__uint_as_float(old) + e.imp[jj]);
          }
        }
        wave_fence();
      } else {
        // non-essential list: stream its postings of this doc window and probe the bitmap
        SLG_STAMP(4);
        uint32_t cur = rl(ne_cur, tc);
        const uint32_t end = rl(ne_end, tc);
        const uint64_t off = ((uint64_t)rl((uint32_t)(my_off >> 32), tc) << 32) | rl((uint32_t)my_off, tc);
        const float w = __int_as_float((int)rl((uint32_t)__float_as_int(my_w), tc));
        const gu32_t ld = gdocs + off;
        const gf32_t li = gimps + off;
        // two chunks of 64 postings in flight
        uint32_t d0 = kDocEnd, d1 = kDocEnd;
        float i0 = 0.0f, i1 = 0.0f;
        if (cur + lane < end) {
          d0 = ld[cur + lane];
          i0 = li[cur + lane];
        }
        if (cur + 64 + lane < end) {
          d1 = ld[cur + 64 + lane];
          i1 = li[cur + 64 + lane];
        }
        while (cur < end) {
          uint32_t d2 = kDocEnd;
          float i2 = 0.0f;
          if (cur + 128 + lane < end) {
            d2 = ld[cur + 128 + lane];
            i2 = li[cur + 128 + lane];
          }
          const uint32_t rel = d0 - wbase;
          const bool have = cur + lane < end;
          const bool inwin = have && d0 >= wbase && d0 < dend;
          // postings below dend are finished with (docs before the window exist in no
          // essential list: pruned); sorted, so they form a prefix of the chunk
          const uint32_t adv = (uint32_t)__popcll(__ballot(have && d0 < dend));
          const uint32_t nwi = rel & (kSpanWords - 1);
          const uint32_t nbit = 1u << ((rel >> 9) & 31);
          uint32_t nwd = 0;
          if (inwin) nwd = bm[nwi];
          const bool hit = inwin && (nwd & nbit) != 0u;
          if (__ballot(hit) != 0ull) {
            if (hit) {
              const uint32_t r = (pre[nwi] + __popc(nwd & (nbit - 1u))) & (kCap - 1);
              vals[r] = __float_as_uint(__uint_as_float(vals[r]) + i0 * w);
            }
            wave_fence();
          }
          cur += adv;
          if (adv < 64u) break;  // the next posting is at or past the window end
          d0 = d1;
          i0 = i1;
          d1 = d2;
          i1 = i2;
        }
        ne_cur = lane == tc ? cur : ne_cur;
        SLG_STAMP(7);
      }
    }
    wave_fence();
    SLG_STAMP(4);
    // P4: owners read the finished sums and offer them to the top-k
    int32_t ctk[kNSlot];
    uint32_t passmask = 0;
    {
      uint32_t v[kNSlot];
#pragma unroll
      for (int jj = 0; jj < kNSlot; jj++) v[jj] = vals[slot[jj]];
#pragma unroll
      for (int jj = 0; jj < kNSlot; jj++) {
        const bool own = (ownmask >> jj) & 1u;
        ctk[jj] = own ? total_key(__uint_as_float(v[jj])) : kSentinelTk;
        passmask |= (own && top.passes(ctk[jj], 0u, e.doc[jj])) ? (1u << jj) : 0u;
      }
    }
    if (__ballot(passmask != 0u) != 0ull) {
#pragma unroll
      for (int jj = 0; jj < kNSlot; jj++) {
        // re-evaluate against the current threshold after every insertion: a cold slot
        // costs ~k(1 + ln(64/k)) insertions instead of 64 iterations
        uint64_t m = __ballot(((passmask >> jj) & 1u) && top.passes(ctk[jj], 0u, e.doc[jj]));
        while (m) {
          const uint32_t l = (uint32_t)__builtin_ctzll(m);
          const int32_t c_tk = (int32_t)rl((uint32_t)ctk[jj], l);
          const uint32_t c_doc = rl(e.doc[jj], l);
          if (!(gdel && ((gdel[c_doc >> 5] >> (c_doc & 31)) & 1u)))  // accept()
            top.insert(c_tk, 0u, c_doc, k, lane);
          m &= m - 1;
          m &= __ballot(top.passes(ctk[jj], 0u, e.doc[jj]));
        }
      }
    }
    wave_fence();
    SLG_STAMP(5);
  };

  // lane t < T: cut points of round rr and rr + 1 of this slice
  auto cuts = [&](const uint32_t rr, uint32_t &lo, uint32_t &hi) {
    const uint32_t src = rr * T + lane;
    const uint32_t a = __shfl(bflat, src & 63, 64), b = __shfl(bflat, (src + T) & 63, 64);
    lo = lane < T ? a : 0u;
    hi = lane < T ? b : 0u;
  };
  auto lane_sum_T = [&](const uint32_t v) {
    uint32_t R = 0;
    for (uint32_t t = 0; t < T; t++) R += rl(v, t);
    return R;
  };

  // ---- driver.  Normal rounds (<= kCap postings, planned exactly by the partition kernel)
  // are software-pipelined: round rr+1 is loading into `en` while round rr is processed from
  // `ew`.  An over-full round (skewed data) is streamed through `ew` in bounded chunks cut at
  // a common doc id.  Both paths share ONE accumulate site. ----
  Elems ew, en;
  uint32_t tot_n = 0, lo_n, hi_n;
  const bool my_ess = lane < T && ((ess_mask >> lane) & 1u);
  auto ess_cnt = [&](const uint32_t lo, const uint32_t hi) { return my_ess ? hi - lo : 0u; };
  cuts(0, lo_n, hi_n);
  bool big_n = lane_sum_T(ess_cnt(lo_n, hi_n)) > (uint32_t)kCap;
  if (!big_n) issue(en, lo_n, ess_cnt(lo_n, hi_n), tot_n);
  for (uint32_t rr = 0; rr < n_r; rr++) {
    const bool big = big_n;
    uint32_t ocur = lo_n;
    const uint32_t oend = my_ess ? hi_n : lo_n;  // the streaming path covers essential lists
    ne_cur = lo_n;                               // non-essential lists: probed per doc window
    ne_end = my_ess ? lo_n : hi_n;
    uint32_t total = tot_n;
    if (!big) ew = en;
#ifdef SLG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SLG_STAMP(6);
    if (rr + 1 < n_r) {  // prefetch the next round
      cuts(rr + 1, lo_n, hi_n);
      big_n = lane_sum_T(ess_cnt(lo_n, hi_n)) > (uint32_t)kCap;
      if (!big_n) issue(en, lo_n, ess_cnt(lo_n, hi_n), tot_n);
    }
    uint32_t dlo = rl(dflat, rr), dhi = rl(dflat, rr + 1);
    SLG_STAMP(0);
    uint32_t guard = 0;
    do {
      uint32_t vmask = 0;
      if (big) {  // next chunk of an over-full round
        const uint32_t rem = oend - ocur;
        const uint32_t R = lane_sum_T(rem);
        // every chunk consumes >= 1 posting; the bound only guards against a planner bug
        if (R == 0 || ++guard > (1u << 22)) break;
        uint32_t chunk;
        if (R <= (uint32_t)kCap) {
          chunk = rem;
        } else {
          const float share = (float)(kCap - 2 * (int)T) * ((float)rem / (float)R);
          uint32_t c = (uint32_t)share;
          c = c < 1u ? 1u : c;
          chunk = rem == 0 ? 0u : (c < rem ? c : rem);
        }
        uint32_t lastdoc = kDocEnd, firstdoc = kDocEnd;
        if (chunk < rem) lastdoc = gdocs[my_off + ocur + chunk - 1];
        if (rem > 0) firstdoc = gdocs[my_off + ocur];
        issue(ew, ocur, chunk, total);
        uint32_t bound = kDocEnd;
        dlo = kDocEnd;
        for (uint32_t t = 0; t < T; t++) {
          const uint32_t ld = rl(lastdoc, t), fd = rl(firstdoc, t);
          bound = ld < bound ? ld : bound;
          dlo = fd < dlo ? fd : dlo;
        }
        dhi = bound == kDocEnd ? kDocEnd : bound + 1;
        uint32_t consumed = 0;
#pragma unroll
        for (int jj = 0; jj < kNSlot; jj++) {
          const bool in = (uint32_t)(jj * 64) + lane < total && ew.doc[jj] <= bound;
          vmask |= in ? (1u << jj) : 0u;
          for (uint32_t tt = 0; tt < T; tt++) {
            const uint64_t am = __ballot(in && ew.t(jj) == tt);
            consumed += lane == tt ? (uint32_t)__popcll(am) : 0u;
          }
        }
        ocur += consumed;
      } else {
#pragma unroll
        for (int jj = 0; jj < kNSlot; jj++)
          vmask |= ((uint32_t)(jj * 64) + lane < total) ? (1u << jj) : 0u;
      }
      if (!(p.dbg & 4u) && total != 0) {
        // doc windows: one in the common case (the round spans <= kSpan docs)
        uint32_t wbase = dlo & ~31u;
        const bool single = dhi - wbase <= kSpan;
        uint32_t remain = vmask;
        for (;;) {
          uint32_t vm = remain;
          if (!single) {
            vm = 0;
#pragma unroll
            for (int jj = 0; jj < kNSlot; jj++) {
              const bool in = ((remain >> jj) & 1u) && (ew.doc[jj] - wbase) < kSpan;
              vm |= in ? (1u << jj) : 0u;
            }
          }
          accumulate(ew, vm, wbase, (dhi - wbase) < kSpan ? dhi : wbase + kSpan);
          if (single) break;
          remain &= ~vm;
          uint32_t mn = kDocEnd;
#pragma unroll
          for (int jj = 0; jj < kNSlot; jj++)
            mn = ((remain >> jj) & 1u) && ew.doc[jj] < mn ? ew.doc[jj] : mn;
          mn = wave_min(mn);
          if (mn == kDocEnd) break;
          wbase = mn & ~31u;
        }
      }
    } while (big);
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
#ifdef SLG_STAMPS
  SLG_STAMP(7);
  if (p.stamps && lane == 0)
    for (int i = 0; i < 8; i++) p.stamps[(size_t)slice * 8 + i] = st_acc[i];
#endif
}

}  // namespace slg
