// slg_score.hpp — shared definitions of the round-scoring kernels: the planning kernel
// partition_rounds_kernel, RoundScoreParams, wave scans, in-kernel stamps (diagnostic builds).
// The scoring kernels themselves: slg_score_uni4.hpp (<= 8 lists), slg_score_multi.hpp (9..32
// lists, score plans beyond the few-term kernel's, MaxScore / block-max pruning); the superseded
// few-term forms slg_score_uni.hpp / slg_score_uni3.hpp build only with -DSLG_LEGACY_KERNELS.
//
// Restates query/wand.rs:459-566 (brute_force: every posting of every term is scored and
// summed per doc, in ScorePlan leaf order planner.rs:122-135) and push_top_k
// (wand.rs:905-916) for a whole batch of queries.
//
// Work decomposition (built by the host + partition_rounds_kernel):
//   sub-query = (query, segment);  round = a doc-id range of a sub-query holding about one
//   register set of postings over all its lists, with the exact per-list posting ranges known
//   up front;  slice = consecutive rounds owned by ONE WAVE (no workgroup barriers).
// Integer/f32 VALU + LDS work bounded by the HBM stream of postings; no MFMA on this path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_kernels.hpp"

namespace slg {

// (planning constants: slg_desc.hpp)
// per-wave LDS: bitmap words, exclusive prefix popcounts, accumulators

// ---- partition: exact per-list cut points of every round --------------------------------------
struct RoundPartParams {
  const RoundQuery *sq;
  const TermRef *terms;
  const SegDev *segs;
  const uint32_t *bnd_coarse;  // [ceil(n_boundaries / 32)] sub-query of every 32nd boundary
  uint32_t n_sq;
  uint32_t *bounds;
  uint32_t *rdoc;
  uint32_t *q_scored;  // [nq] zeroed here (saves a memset node per batch)
  unsigned long long *skip_counts;  // [1 + nq] zeroed here, or null
  const uint32_t *slice_sq;     // [n_slices]
  const uint32_t *slice_order;  // [n_slices] launch position -> slice
  SliceDesc *slice_desc;        // [n_slices] out, by launch position
  uint32_t nq;
  uint32_t n_boundaries;
  uint32_t n_slices;
  uint32_t tpb_shift;  // log2(threads per boundary): 2 when no sub-query has more than 4 lists, else 3
  // work counters of the persistent scoring waves (slg_desc.hpp: kWorkQueues), zeroed here
  uint32_t *work_ctr;
  uint32_t n_waves;
};

// first index of d[0, df) with d[idx] >= target, looked for in the 16 NP postings from `a` on: NP + 1
// pivots 16 postings apart (independent loads, one latency) bracket it to 16 postings, which are then
// read whole (one more latency) — two dependent loads instead of a bisection's ~8; false if the
// answer lies outside the window.  The list is
// followed by kListPad sentinels (0xFFFFFFFF), so positions up to df + 63 may be read.
template <int NP = 8, typename DocPtr>
__device__ __forceinline__ bool lower_bound_window(const DocPtr d, uint32_t df, uint32_t target, uint32_t a,
                                                   uint32_t &pos) {
  uint32_t below = 0;  // pivots d[a + 16 i - 1], i = 0..NP, that are < target (i = 0 at a == 0: -inf)
#pragma unroll
  for (uint32_t i = 0; i <= (uint32_t)NP; i++) {
    const uint32_t at = a + 16u * i;
    const uint32_t idx = at - 1u < df ? at - 1u : df;  // (past the end: the first sentinel)
    const uint32_t v = d[at == 0u ? 0u : idx];
    below += (at == 0u || v < target) ? 1u : 0u;
  }
  if (below < 1u || below > (uint32_t)NP) return false;
  const uint32_t base = a + 16u * (below - 1u);  // d[base - 1] < target <= d[base + 15]
  uint32_t cnt = 0;
#pragma unroll
  for (uint32_t i = 0; i < 16; i++) cnt += d[base + i] < target ? 1u : 0u;
  pos = base + cnt;
  return true;
}

// first index of d[0, df) with d[idx] >= target: the window around the position a uniform doc-id
// distribution predicts (doc ids are validated < n_docs at staging, so the guess is <= df); if the
// guess was off by more than 64 postings, a bracket widened until it holds the answer (64-bit
// width: a list may hold up to 2^32 - 2 postings and w grows by 8x per step), then a bisection
template <typename DocPtr>
__device__ __forceinline__ uint32_t lower_bound_guess(const DocPtr d, uint32_t df, uint32_t target, uint32_t n_docs) {
  uint32_t g = (uint32_t)(((uint64_t)df * target) / (n_docs ? n_docs : 1u));
  g = g < df ? g : df;
  uint32_t lo = 0, hi = df;
  if (lower_bound_window<8>(d, df, target, g > 64u ? g - 64u : 0u, lo)) return lo;
  lo = 0;
  for (uint64_t w = 512; w < df; w <<= 3) {
    const uint32_t a = g > w ? (uint32_t)(g - w) : 0u;
    const uint32_t e = (uint64_t)g + w < df ? (uint32_t)(g + w) : df;
    const bool lo_ok = a == 0u || d[a - 1] < target;   // answer >= a
    const bool hi_ok = e == df || d[e - 1] >= target;  // answer <= e - 1
    if (lo_ok && hi_ok) {
      lo = a;
      hi = e == df ? df : e - 1;
      break;
    }
  }
  while (lo < hi) {
    const uint32_t mid = lo + ((hi - lo) >> 1);
    if (d[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// 4 or 8 threads per boundary: thread u handles lists u, u + threads, ...
// (Measured on config 3, where the kernel takes 1.6 ms: it is bound by the HBM lines it fetches —
// with a round's postings of a list inside one or two lines, the cut points of all rounds touch
// every line of every list but the longest, at scattered-access efficiency.  Giving a thread 8
// consecutive boundaries, each searched behind its predecessor, fetches the same lines and was
// slower: 2.2 ms.)
static __global__ void __launch_bounds__(256) partition_rounds_kernel(RoundPartParams p) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < p.nq) p.q_scored[gid] = 0;
  if (gid < kWorkQueues && p.work_ctr) p.work_ctr[gid * kWorkCtrStride] = 0u;
  if (gid <= p.nq && p.skip_counts) p.skip_counts[gid] = 0ull;
  if (gid < p.n_slices) {  // the slice record of launch position gid
    const uint32_t slice = p.slice_order[gid];
    const RoundQuery s = p.sq[p.slice_sq[slice]];
    const uint32_t r0 = (slice - s.slice_begin) * s.rounds_per_slice;
    SliceDesc d;
    d.slice = slice;
    d.term_begin = s.term_begin;
    d.bounds_off = s.bounds_begin + r0 * s.n_terms;
    d.rdoc_off = s.rdoc_begin + r0;
    d.n_terms = s.n_terms;
    d.n_rounds = s.n_rounds - r0 < s.rounds_per_slice ? s.n_rounds - r0 : s.rounds_per_slice;
    d.seg = s.seg;
    d.filter = s.filter;
    d.q = s.q;
    d.theta0 = s.theta0;
    d.cand_lo = s.cand_lo;
    d.cand_hi = s.cand_hi;
    d.first_round = r0;
    d.sq_rounds = s.n_rounds;
    d.longest = s.longest;
    {
      const TermRef L = p.terms[s.term_begin + s.longest];
      d.l_df = L.df;
      d.l_off = L.off;
    }
    d.plan = s.plan;
    d.tie = s.tie;
    d.max_init = s.max_init;
    d.n_leaves = s.n_leaves;
    p.slice_desc[gid] = d;
  }
  const uint32_t tpb = 1u << p.tpb_shift;
  const uint32_t b = gid >> p.tpb_shift, u = gid & (tpb - 1u);
  if (b >= p.n_boundaries) return;
  // the sub-query that owns boundary b: the last one whose first boundary is <= b (bnd_begin
  // ascends).  The host uploads it for every 32nd boundary (a per-boundary table was most of the
  // descriptor upload); from there a short walk (sub-queries have ~85 boundaries on config 2)
  uint32_t sqi = p.bnd_coarse[b >> 5];
  while (sqi + 1 < p.n_sq && p.sq[sqi + 1].bnd_begin <= b) sqi++;
  const RoundQuery s = p.sq[sqi];
  const uint32_t j = b - s.bnd_begin;
  const uint32_t *docs = p.segs[s.seg].docs;
  const TermRef L = p.terms[s.term_begin + s.longest];
  const uint32_t stride = (L.df + s.n_rounds - 1) / s.n_rounds;
  const uint64_t posL = (uint64_t)j * stride;
  const bool first = j == 0, last = j >= s.n_rounds || posL >= L.df;
  uint32_t target = 0;
  if (!first && !last) target = docs[L.off + posL];
  for (uint32_t t = u; t < s.n_terms; t += tpb) {
    const TermRef me = p.terms[s.term_begin + t];
    uint32_t out;
    if (first) {
      out = 0;
    } else if (last) {
      out = me.df;
    } else if (t == s.longest) {
      out = (uint32_t)posL;
    } else {
      out = lower_bound_guess(docs + me.off, me.df, target, p.segs[s.seg].n_docs);
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
  const SliceDesc *slice_desc;  // [n_slices] by launch position (partition_rounds_kernel)
  const SegDev *segs;
  const uint32_t *bounds;
  const uint32_t *rdoc;
  int32_t *slice_tk;    // [n_slices * k]
  uint32_t *slice_doc;  // [n_slices * k]
  uint32_t *q_scored;   // [nq] or null
  const uint32_t *const *reject_table;  // [n_filters * n_segs] reject bitmaps (doc filters)
  uint32_t n_segs;
  uint32_t plan_batch;  // multi kernel: 1 = some sub-query has a score plan, 2 = a two-level one, 4 = a deeper
                        // tree (acc / max arrays in LDS: this many sets of kMultiPlanLds bytes)
  const PlanNode *plan_nodes;  // canonical node tables of the deep trees (RoundQuery::node_begin)
  // large-k mode of the uniform kernel (k > 256): candidates instead of per-slice top-k lists
  uint2 *cand;            // {ordered score, doc}; sub-query region + posting offset of the slice
  uint64_t *slice_cbeg;   // [n_slices] first candidate slot of the slice
  uint32_t *slice_ccnt;   // [n_slices] candidates written
  uint32_t n_slices;
  uint32_t k;
  // block skipping (many-term kernel with classified lists): 64-posting slots of non-essential
  // lists whose doc range holds no doc of an essential list are never loaded
  uint32_t block_skip;
  // postings of non-essential lists that were never loaded: [0] of the batch, [1 + q] of query q; or null
  unsigned long long *skip_counts;
  unsigned long long *stamps;  // [n_slices * 8] (SLG_STAMPS builds only)
  uint32_t *error_flag;        // set (non-zero) by a wave that gave up on a round: slg_batch_fetch fails then
  uint32_t *work_ctr;          // persistent waves: kWorkQueues counters, kWorkCtrStride words apart
  uint32_t n_waves;            // waves launched (few-term kernel: min(n_slices, wave slots of the device))
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

// shared by the scoring kernels
constexpr int kJoinWords = 1024;  // filter words of the few-term kernels = 8192 doc fields; also the join queue
static_assert(kJoinWords * 4 >= kUniCap * 8, "the join queue ({doc, score} per posting) overlays the filter");
// k <= 256 (KREGS <= 4): buffered top-k in LDS (BufTopK); larger k: every doc above the seed
// threshold goes to the slice's candidate region and select_topk_kernel picks the k best
constexpr bool uni_buffered(int kregs) { return kregs <= 4; }

typedef const __attribute__((address_space(1))) uint32_t *gu32_t;
typedef const __attribute__((address_space(1))) float *gf32_t;

}  // namespace slg
