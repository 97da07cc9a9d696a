// slg_score_uni.hpp — the hot kernel for queries with few terms (<= 4 lists per sub-query;
// BASELINE config 2): exact pre-planned rounds, every 64-lane register slot holds postings of ONE
// list, and a FILTER + JOIN accumulate that keeps docs found in a single list entirely in
// registers.
//
// Restates query/wand.rs:459-566 (every posting scored, per-doc sums in ScorePlan leaf order,
// planner.rs:122-135) and push_top_k (wand.rs:905-916).  k <= 256: candidates go to a per-wave
// LDS buffer (BufTopK, slg_kernels.hpp); larger k: to the slice's region of a global candidate
// array, picked per query by select_topk_kernel.  More lists than 4: slg_score_multi.hpp.
//
// Padding each list to a slot boundary makes the per-slot list id, weight, base address and
// lane count wave-uniform scalars (one v_readlane each from a lane-held slot descriptor; the
// descriptors of 8 rounds are computed at once) and lets the posting loads be whole-slot,
// scalar base + lane.
//
// The accumulate.  With the sparse lists of a real query almost every doc of a round occurs in
// ONE of its lists (config 2: ~98 % of the postings); such a doc's score is 0.0 + w*impact
// (`or_insert(0.0) += score`, wand.rs:539) and needs no accumulator at all.  Per round:
//   P0  clear a 1024-word LDS filter (4 wide stores);
//   P1  every posting ORs ONE bit (ds_or, no return): word = doc mod 1024, 4-bit field =
//       (doc / 1024) mod 8, bit = its list (<= 4 lists);
//   P2  every posting reads its word back: the field names the lists that hold this doc (or a
//       doc that aliases it — the filter is one-sided: it never misses a shared doc);
//   P3  postings whose field shows only their own list are SINGLES: scored in registers,
//       compared with the threshold, (rarely) appended to the top-k buffer;
//   P4  the others (both the first and the later postings of a shared doc, plus aliases) are
//       compacted into an LDS queue (it overlays the filter, which is dead by now) in slot =
//       list order; each queue entry then sums the entries with ITS doc id in queue order:
//       ((0.0 + x_a) + x_b) + ..., bit for bit the reference's term-order sum; the entry that
//       comes first for its doc owns the result.  Aliased docs simply find no partner.
// One round is 3 dependent LDS round trips (the bitmap-rank accumulate it replaces had ~10, and
// twice the LDS instructions), there are no doc windows (the filter wraps) and no per-posting
// accumulator traffic.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_score.hpp"

namespace slg {

// (kUniSlots = 8 slots per round, kUniCap, kUniMaxLists = 4 lists: slg_desc.hpp)
// (kJoinWords, uni_buffered: slg_score.hpp)
constexpr int kJoinPairs = 24;               // queue sizes up to this are joined all-pairs in registers
                                             // (16 / 40 / 64 measured: no better)
// per-wave LDS: filter / join queue, top-k buffer, then the slice's cut points (64 words) and the
// lists' posting offsets (2 x 4 words) — values needed once per 8 rounds, kept out of the VGPRs
constexpr int kUniPlanLds = 64 * 4 + 2 * kUniMaxLists * 4;
constexpr int uni_plan_off(int kregs) { return kJoinWords * 4 + (uni_buffered(kregs) ? buftopk_lds(kregs) : 0); }
constexpr int uni_wave_lds(int kregs) { return uni_plan_off(kregs) + kUniPlanLds; }

#ifndef SLG_UNI_WAVES
#define SLG_UNI_WAVES 6  // waves per SIMD the register allocation aims at (80 VGPRs, no spills)
#endif
template <int KREGS>
__global__ void __launch_bounds__(64)
    __attribute__((amdgpu_waves_per_eu(KREGS == 4 ? 5 : SLG_UNI_WAVES, KREGS == 4 ? 5 : SLG_UNI_WAVES)))  // (k 129..256: 16 spills at 6)
score_uniform_kernel(RoundScoreParams p) {
  constexpr int NS = kUniSlots;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t widx = blockIdx.x;
  if (widx >= p.n_slices) return;  // waves are independent: no workgroup barrier anywhere
  const SliceDesc sl = p.slice_desc[widx];  // one scalar load; everything below hangs off it in parallel
  const uint32_t slice = rfl(sl.slice);

  constexpr bool BUF = uni_buffered(KREGS);
  uint32_t *flt = reinterpret_cast<uint32_t *>(smem);
  uint4 *flt4 = reinterpret_cast<uint4 *>(smem);
  uint2 *queue = reinterpret_cast<uint2 *>(smem);  // {doc, score} of queued postings; overlays flt (P4)

  const uint32_t T = rfl(sl.n_terms);
  const uint32_t n_r = rfl(sl.n_rounds);
  const SegDev sd = p.segs[sl.seg];
  const gu32_t gdocs = (gu32_t)sd.docs;
  const gf32_t gimps = (gf32_t)sd.imps;
  // accept(): tombstones, or the reject bitmap (deleted | ~filter) of the query's doc filter
  const uint32_t fid = rfl(sl.filter);
  const gu32_t gdel = (gu32_t)(fid ? p.reject_table[(size_t)(fid - 1) * p.n_segs + sl.seg] : sd.deleted);
  const uint32_t k = p.k;

  // lane t < T: list t's weight; its posting offset and all cut points of the slice (entry
  // r*T + t: where round r starts in list t) go to LDS
  uint32_t *const bflat = reinterpret_cast<uint32_t *>(smem + uni_plan_off(KREGS));
  uint32_t *const off_lo = bflat + 64, *const off_hi = off_lo + kUniMaxLists;
  float my_w = 0.0f;
  if (lane < T) {
    const TermRef tr = p.terms[sl.term_begin + lane];
    my_w = tr.weight;
    off_lo[lane] = (uint32_t)tr.off;
    off_hi[lane] = (uint32_t)(tr.off >> 32);
  }
  bflat[lane] = lane < (n_r + 1) * T ? p.bounds[sl.bounds_off + lane] : 0u;
  wave_fence();
  auto list_off = [&](const uint32_t t) { return ((uint64_t)off_hi[t] << 32) | off_lo[t]; };

  BufTopK<BUF ? KREGS : 1> btop;  // k <= 256; for larger k only its threshold is used
  btop.init(reinterpret_cast<uint64_t *>(smem + kJoinWords * 4));
  // k > 256: the slice's candidate region starts at (sub-query base) + (postings of all lists
  // before the slice's first round) and can hold one entry per posting of the slice
  uint32_t ccur = 0;
  uint64_t cbeg = 0;
  if (!BUF) {
    uint32_t before = 0;
    for (uint32_t t = 0; t < T; t++) before += rfl(bflat[t]);
    cbeg = (((uint64_t)rfl(sl.cand_hi) << 32) | rfl(sl.cand_lo)) + before;
  }
  uint2 *const creg = BUF ? nullptr : p.cand + cbeg;
  {  // threshold seed (RoundQuery::theta0)
    const float th0 = __uint_as_float(rfl(__float_as_uint(sl.theta0)));
    if (th0 > 0.0f) btop.set_floor(th0);
  }
  // distinct docs scored (QueryStats.scored_docs): every posting of the slice is one, except the
  // queued ones, of which only the owners count (added in the join)
  uint32_t n_scored = 0;
  for (uint32_t t = 0; t < T; t++) n_scored += rfl(bflat[n_r * T + t]) - rfl(bflat[t]);
#ifdef SLG_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_ins = 0, st_queued = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
  const unsigned long long st_begin = wall_clock64();  // 100 MHz, device-wide
#endif

  // One round's postings: slot jj holds <= 64 postings of one list.  A slot is described by one
  // lane of a Desc; the slots of a round are the 8 lanes starting at `dbase`.  Two VGPRs per
  // descriptor set (the kernel's time follows its waves per SIMD: every register counts):
  //   meta = list | postings in the slot << 4 | slots the round needs << 12 (saturated: > NS means
  //          it does not fit and is streamed in chunks) | bits 32.. of the first posting's index << 20
  //   lo   = bits 0..31 of that index
  struct Desc {
    uint32_t meta, lo;
  };
  auto mk_meta = [](const uint32_t st, const uint32_t cnt, const uint32_t nsl, const uint32_t hi) {
    return st | (cnt << 4) | ((nsl < 255u ? nsl : 255u) << 12) | (hi << 20);
  };
  struct URound {
    uint32_t doc[NS];
    float imp[NS];
    uint32_t stpack;  // uniform: list of slot jj in bits 4jj..4jj+3 (set by settle)
    uint32_t dbase;   // uniform: first descriptor lane of this round
    uint32_t nslots;  // uniform
  };

  // ---- descriptors of 8 consecutive planned rounds at once: lane 8*i + j = slot j of round
  //      g0 + i (per-round cost: a few readlanes instead of a prefix scan over the lists) ----
  auto describe_group = [&](Desc &d, const uint32_t g0) {
    const uint32_t ri = g0 + (lane >> 3), j = lane & 7u;
    const bool rv = ri < n_r;
    uint32_t run = 0;  // slots of the lists before list t
    uint32_t st = 0, cnt = 0, hi = 0;
    d.lo = 0;
    for (uint32_t t = 0; t < T; t++) {
      const uint32_t src = (ri * T + t) & 63u;
      const uint32_t lo_t = bflat[src];
      const uint32_t c_t = bflat[(src + T) & 63u] - lo_t;  // postings of list t in the round
      const uint32_t m = rv ? (c_t + 63u) >> 6 : 0u;
      const bool mine = j >= run && j < run + m;
      const uint32_t kin = (j - run) * 64u;  // postings of the list before this slot
      const uint64_t base = list_off(t) + lo_t + kin;
      const uint32_t left = c_t - kin;
      st = mine ? t : st;
      cnt = mine ? (left < 64u ? left : 64u) : cnt;
      d.lo = mine ? (uint32_t)base : d.lo;
      hi = mine ? (uint32_t)(base >> 32) : hi;
      run += m;
    }
    d.meta = mk_meta(st, cnt, run, hi);
  };

  // ---- descriptors for per-list ranges [lo, lo+cnt) held in lane t (chunks of an over-full
  //      round); lane j = slot j ----
  auto describe_chunk = [&](Desc &d, const uint32_t lo, const uint32_t cnt) {
    const uint32_t m = (cnt + 63u) >> 6;  // slots of my list
    uint32_t ss = 0, run = 0;             // ss: first slot of my list
    for (uint32_t t = 0; t < T; t++) {
      ss = lane == t ? run : ss;
      run += rl(m, t);
    }
    // lane j: which list owns slot j = the last list whose first slot is <= j
    uint32_t tj = 0;
    for (uint32_t t = 1; t < T; t++) tj = lane >= rl(ss, t) ? t : tj;
    const uint32_t l_ss = __shfl(ss, (int)tj, 64), l_cnt = __shfl(cnt, (int)tj, 64);
    const uint64_t l_abs = list_off(tj) + __shfl(lo, (int)tj, 64);
    const uint32_t kin = (lane - l_ss) * 64u;
    const bool used = lane < run && lane < (uint32_t)NS;
    const uint32_t left = used && l_cnt > kin ? l_cnt - kin : 0u;
    const uint64_t base = l_abs + kin;
    d.meta = mk_meta(tj, left < 64u ? left : 64u, run, (uint32_t)(base >> 32));
    d.lo = (uint32_t)base;
  };

  // ---- issue the loads of the round described by lanes dbase .. dbase+7 of d: whole 64-lane
  //      slots, scalar base + lane (no per-lane predicate; the arrays are padded by 64 entries).
  //      Lanes past the slot's count hold other postings until settle() masks them. ----
  auto issue = [&](URound &r, const Desc &d, const uint32_t dbase) {
    r.dbase = dbase;
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      const uint64_t base = ((uint64_t)(rl(d.meta, dbase + jj) >> 20) << 32) | rl(d.lo, dbase + jj);
      r.doc[jj] = gdocs[base + lane];
      r.imp[jj] = gimps[base + lane];
    }
  };
  // ---- dst = the loaded round src, ready to accumulate: lanes beyond each slot's count become
  //      idle lanes (doc = kDocEnd) and the impacts are multiplied by the slot's list weight
  //      (score_tf, query/wand.rs:285: one scalar per slot).  d = the descriptors src was issued
  //      from (still in place: the next group is described after the settle) ----
  auto settle = [&](URound &dst, const URound &src, const Desc &d) {
    dst.dbase = src.dbase;
    dst.nslots = src.nslots;
    uint32_t sp = 0;
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      const uint32_t m = rl(d.meta, src.dbase + jj);
      const uint32_t st = m & 15u, cnt = (m >> 4) & 255u;
      const float w = __int_as_float((int)rl((uint32_t)__float_as_int(my_w), st));
      dst.doc[jj] = lane < cnt ? src.doc[jj] : kDocEnd;
      dst.imp[jj] = src.imp[jj] * w;
      sp |= st << (4 * jj);
    }
    dst.stpack = sp;
  };
  auto slot_list = [](const URound &e, const int jj) { return (e.stpack >> (4 * jj)) & 15u; };

  // ---- candidates -> top-k.  Cheap necessary condition first (an IEEE compare with the
  //      threshold's score: total_cmp order implies it), exact compare + append in take_checked.
  //      ONE take_checked site per candidate source (the ranking code of BufTopK::compact is large). ----
  auto threshold_score = [&]() {  // score part of the current threshold as a float (-inf: none)
    const uint32_t hi = (uint32_t)(btop.th >> 32);
    return hi < 0x00800000u ? -INFINITY : key_to_float((int32_t)(hi ^ 0x80000000u));
  };
  auto take_checked = [&](const bool cand, const float score, const uint32_t doc) {
    const uint32_t ok = ordered_score(score);
    const bool ps = cand && btop.passes(ok, ~doc);
    if constexpr (BUF) {
      btop.append_checked(ps, ok, ~doc, k, lane, (const uint32_t *)gdel);
    } else {  // the candidate region holds one entry per posting
      const uint64_t m = __ballot(ps);
      const uint32_t at = ccur + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (ps) creg[at] = make_uint2(ok, doc);
      ccur += (uint32_t)__popcll(m);
    }
  };

  // ---- score the postings of `e` (all docs of a doc range; idle lanes hold kDocEnd).  Lane sets
  //      are kept as 64-bit wave masks (scalar registers), the per-slot code is branch-free. ----
  auto accumulate = [&](const URound &e) {
    SLG_STAMP(1);
    uint64_t validm[NS], sharedm[NS];  // sharedm: my doc is (or aliases) a doc of another list
    uint32_t n = 0;                    // queued postings
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      validm[jj] = __ballot(e.doc[jj] != kDocEnd);
      sharedm[jj] = 0ull;
    }
    if (T > 1) {
      // P0: clear the filter
      flt4[lane] = make_uint4(0u, 0u, 0u, 0u);
      flt4[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
      flt4[lane + 128] = make_uint4(0u, 0u, 0u, 0u);
      flt4[lane + 192] = make_uint4(0u, 0u, 0u, 0u);
      wave_fence();
      // P1: one bit per posting: word = doc mod 1024, field = (doc / 1024) mod 8, bit = list
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        const uint32_t lbit = 1u << slot_list(e, jj);  // uniform
        if (e.doc[jj] != kDocEnd)
          atomicOr(&flt[e.doc[jj] & (kJoinWords - 1)], lbit << ((e.doc[jj] >> 8) & 0x1Cu));
      }
      wave_fence();
      SLG_STAMP(2);
      // P2: the lists that hold my doc (or an alias of it).  Idle lanes read word 1023.
      uint32_t fin[NS];
#pragma unroll
      for (int jj = 0; jj < NS; jj++) fin[jj] = flt[e.doc[jj] & (kJoinWords - 1)];
      wave_fence();  // the queue below overlays the filter: all reads are issued before its writes
      SLG_STAMP(3);
      // P3: shared docs (and aliases) are queued, in slot = list order
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        const uint32_t tj = slot_list(e, jj);  // uniform
        const uint32_t others = 0xFu & ~(1u << tj);
        const uint32_t fld = (fin[jj] >> ((e.doc[jj] >> 8) & 0x1Cu)) & others;
        const uint64_t m = __ballot(fld != 0u) & validm[jj];
        sharedm[jj] = m;
        if (m != 0ull) {
          const uint32_t at = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if ((m >> lane) & 1ull) queue[at] = make_uint2(e.doc[jj], __float_as_uint(e.imp[jj]));
          n += (uint32_t)__popcll(m);
        }
      }
      wave_fence();
      n_scored -= n;
    }
    SLG_STAMP(4);
#ifdef SLG_STAMPS
    st_queued += n;
#endif
    // singles: the doc occurs in this list only; score = 0.0 + w*impact (wand.rs:539)
    {
      const float thf = threshold_score();
      uint64_t hotm[NS];
      uint64_t anyhot = 0ull;
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        const uint64_t singles = validm[jj] & ~sharedm[jj];
        hotm[jj] = __ballot(0.0f + e.imp[jj] >= thf) & singles;
        anyhot |= hotm[jj];
      }
      if (anyhot != 0ull) {  // rank-and-trim may be needed between slots: one site, the slot's
                             // registers are selected at run time
#pragma unroll 1
        for (uint32_t it = 0; it < (uint32_t)NS; it++) {
          float x = e.imp[0];
          uint32_t dc = e.doc[0];
          uint64_t hm = hotm[0];
#pragma unroll
          for (int j = 1; j < NS; j++) {
            x = it == (uint32_t)j ? e.imp[j] : x;
            dc = it == (uint32_t)j ? e.doc[j] : dc;
            hm = it == (uint32_t)j ? hotm[j] : hm;
          }
          if (hm == 0ull) continue;
#ifdef SLG_STAMPS
          st_ins += (uint32_t)__popcll(hm);
#endif
          take_checked((hm >> lane) & 1ull, 0.0f + x, dc);
        }
      }
    }
    SLG_STAMP(5);
    // P4: join.  The queue is sorted by (list, doc): slots are in list order and a list's
    // postings in doc order.  A doc's sum is ((0.0 + x_a) + x_b) + ... over the lists that hold it,
    // in list order (= the reference's term order); its entry in the lowest list owns the result.
    if (n != 0u && n <= (uint32_t)kJoinPairs) {
      // few entries (the usual case): all pairs, the senders read from registers lane by lane
      const bool have = lane < n;
      const uint2 me = have ? queue[lane] : make_uint2(kDocEnd, 0u);
      float acc = 0.0f;
      bool lower = false;  // an earlier entry (= a lower list) holds my doc
      auto pair_step = [&](const uint32_t l) {
        const uint32_t dl = rl(me.x, l);
        const float xl = __uint_as_float(rl(me.y, l));
        const bool hit = dl == me.x;
        acc = hit ? acc + xl : acc;
        lower = lower || (hit && l < lane);
      };
      uint32_t l = 0;
      for (; l + 1 < n; l += 2) {  // two senders per iteration: -1.4 % kernel time
        pair_step(l);
        pair_step(l + 1);
      }
      if (l < n) pair_step(l);
      const uint64_t ownerm = __ballot(have && !lower);
      n_scored += (uint32_t)__popcll(ownerm);
      if ((__ballot(acc >= threshold_score()) & ownerm) != 0ull) take_checked((ownerm >> lane) & 1ull, acc, me.x);
    } else if (n != 0u) {
      // many entries (dense lists): binary search of my doc in the queue segment of every list,
      // in list order.  qe[u] = entries of the lists <= u (segment u = [qe[u-1], qe[u])).
      uint32_t qe0 = 0, qe1 = 0, qe2 = 0, qe3 = 0;
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        const uint32_t tj = slot_list(e, jj);
        const uint32_t c = (uint32_t)__popcll(sharedm[jj]);  // (0 for unused slots)
        qe0 += tj <= 0u ? c : 0u;
        qe1 += tj <= 1u ? c : 0u;
        qe2 += tj <= 2u ? c : 0u;
        qe3 += c;
      }
      for (uint32_t rb = 0; rb < n; rb += 64) {  // receivers in blocks of 64 lanes
        const uint32_t idx = rb + lane;
        const bool have = idx < n;
        const uint2 me = have ? queue[idx] : make_uint2(kDocEnd, 0u);
        const uint32_t ml = (idx >= qe0 ? 1u : 0u) + (idx >= qe1 ? 1u : 0u) + (idx >= qe2 ? 1u : 0u);
        float acc = 0.0f;
        bool lower = false;
        uint32_t seg_lo = 0;
        for (uint32_t u = 0; u < T; u++) {
          const uint32_t seg_hi = u == 0u ? qe0 : (u == 1u ? qe1 : (u == 2u ? qe2 : qe3));
          uint32_t lo = seg_lo, hi = seg_hi;
          const uint32_t len = seg_hi - seg_lo;
          const uint32_t steps = len ? 32u - (uint32_t)__builtin_clz(len) : 0u;  // halvings that empty it
          for (uint32_t st = 0; st < steps; st++) {
            const uint32_t mid = (lo + hi) >> 1;  // < seg_hi while lo < hi
            const uint32_t dk = queue[mid < seg_hi ? mid : seg_lo].x;
            const bool less = dk < me.x;
            const bool open = lo < hi;
            lo = open && less ? mid + 1u : lo;
            hi = open && !less ? mid : hi;
          }
          const uint2 kk = queue[lo < seg_hi ? lo : seg_lo];  // (an empty segment reads a neighbour: ignored)
          const bool mine = ml == u;
          const bool hit = have && (mine || (lo < seg_hi && kk.x == me.x));
          acc = hit ? acc + (mine ? __uint_as_float(me.y) : __uint_as_float(kk.y)) : acc;
          lower = lower || (hit && u < ml);
          seg_lo = seg_hi;
        }
        const uint64_t ownerm = __ballot(have && !lower);
        n_scored += (uint32_t)__popcll(ownerm);
        if ((__ballot(acc >= threshold_score()) & ownerm) != 0ull) take_checked((ownerm >> lane) & 1ull, acc, me.x);
      }
    }
    wave_fence();
    SLG_STAMP(6);
  };

  // lane t < T: cut points of round rr and rr + 1 of this slice
  auto cuts = [&](const uint32_t rr, uint32_t &lo, uint32_t &hi) {
    const uint32_t src = rr * T + lane;
    const uint32_t a = bflat[src & 63], b = bflat[(src + T) & 63];
    lo = lane < T ? a : 0u;
    hi = lane < T ? b : 0u;
  };
  auto lane_sum_T = [&](const uint32_t v) {
    uint32_t R = 0;
    for (uint32_t t = 0; t < T; t++) R += rl(v, t);
    return R;
  };

  // ---- driver: planned rounds are prefetched one ahead (`en` loads while `ew` is processed);
  //      a round that needs more than NS slots is streamed in chunks cut at a common doc id.
  //      Both paths share ONE accumulate site (code size / I-cache). ----
  URound ew, en;
  Desc G, C;  // descriptors of the current group of 8 planned rounds / of the current chunk
  describe_group(G, 0);
  en.nslots = (rl(G.meta, 0) >> 12) & 255u;
  en.dbase = 0;
  if (en.nslots <= (uint32_t)NS) issue(en, G, 0);
  for (uint32_t rr = 0; rr < n_r; rr++) {
    const bool big = en.nslots > (uint32_t)NS;
    uint32_t ocur = 0, oend = 0;
    if (big) cuts(rr, ocur, oend);
    if (!big) settle(ew, en, G);
#ifdef SLG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SLG_STAMP(7);
    if (rr + 1 < n_r) {  // prefetch the next round
      const uint32_t nx = rr + 1, db = (nx & 7u) * 8u;
      if (db == 0) describe_group(G, nx);
      en.nslots = (rl(G.meta, db) >> 12) & 255u;
      en.dbase = db;
      if (en.nslots <= (uint32_t)NS) issue(en, G, db);
    }
    SLG_STAMP(0);
    uint32_t guard = 0;
    do {
      if (big) {
        // next chunk of an over-full round: every list gets >= 1 slot, the rest in proportion
        // to what it has left; the chunk ends at the smallest "last loaded doc" of the lists
        // that did not finish, so all postings of a doc are scored in the same chunk
        const uint32_t rem = oend - ocur;
        const uint32_t nne = (uint32_t)__popcll(__ballot(rem != 0u));
        const uint32_t R = lane_sum_T(rem);
        if (R == 0 || ++guard > (1u << 22)) break;
        const uint32_t need = lane_sum_T((rem + 63u) >> 6);
        uint32_t chunk = rem;
        if (need > (uint32_t)NS) {
          const float share = (float)(NS - nne) * ((float)rem / (float)R);
          const uint32_t mslots = rem == 0u ? 0u : 1u + (uint32_t)share;
          chunk = rem < mslots * 64u ? rem : mslots * 64u;
        }
        uint32_t lastdoc = kDocEnd;
        if (chunk < rem) lastdoc = gdocs[list_off(lane < T ? lane : 0u) + ocur + chunk - 1];
        describe_chunk(C, ocur, chunk);
        ew.nslots = (rl(C.meta, 0) >> 12) & 255u;
        issue(ew, C, 0);
        settle(ew, ew, C);
        uint32_t bound = kDocEnd;
        for (uint32_t t = 0; t < T; t++) {
          const uint32_t ld = rl(lastdoc, t);
          bound = ld < bound ? ld : bound;
        }
        const uint32_t dhi = bound == kDocEnd ? kDocEnd : bound + 1u;  // kDocEnd: nothing was cut
        // what each list consumed: its postings with doc < dhi (a prefix of its slots)
        uint32_t consumed = 0;
#pragma unroll
        for (int jj = 0; jj < NS; jj++) {
          const uint32_t cnt = (uint32_t)__popcll(__ballot(ew.doc[jj] < dhi));
          consumed += lane == slot_list(ew, jj) ? cnt : 0u;
          ew.doc[jj] = ew.doc[jj] < dhi ? ew.doc[jj] : kDocEnd;  // the rest: next chunk
        }
        ocur += consumed;
      }
      if (ew.nslots == 0) break;
      accumulate(ew);
    } while (big);
  }

  // ---- write this slice's candidates ----
  if constexpr (BUF) {  // k entries, sentinel-padded, for merge_topk_kernel
    btop.write_out(p.slice_tk + (size_t)slice * k, p.slice_doc + (size_t)slice * k, k, lane,
                   (const uint32_t *)gdel);
  } else if (lane == 0) {  // region already written; deleted docs are dropped by the select
    p.slice_cbeg[slice] = cbeg;
    p.slice_ccnt[slice] = ccur;
  }
  if (p.q_scored && lane == 0 && n_scored) atomicAdd(&p.q_scored[sl.q], n_scored);
#ifdef SLG_STAMPS
  const unsigned long long st_extra = st_ins | (st_queued << 32);
  if (p.stamps && lane == 0) {
    for (int i = 0; i < 8; i++) p.stamps[(size_t)slice * 12 + i] = st_acc[i];
    p.stamps[(size_t)slice * 12 + 8] = st_extra;
    p.stamps[(size_t)slice * 12 + 9] = st_begin;
    p.stamps[(size_t)slice * 12 + 10] = wall_clock64();
    p.stamps[(size_t)slice * 12 + 11] = ((unsigned long long)n_r << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
  }
#endif
}

}  // namespace slg
