// slg_score_multi.hpp — scoring kernel for queries with many terms (5..32 lists), the shape a
// multi-field query string produces (one scored term per field and word, api/reader.rs:2971-3005).
//
// Same algorithm as slg_score_uni.hpp — exact pre-planned rounds, LDS bitmap + prefix popcount =
// dense accumulator slot per doc, f32 sums in term order (query/wand.rs:459-566, planner.rs:122-135),
// buffered top-k — and the same register shape: a 64-lane slot holds postings of ONE list, so a
// slot's list, weight, count and address are wave-uniform scalars.  What differs: a round's lists
// need up to 8 + T slots, more than the 8 a wave keeps in registers, so a round is processed in
// two sweeps over batches of 8 consecutive slots:
//   sweep A  load each batch, set one bitmap bit per posting (ds_or, no return value);
//   P2       prefix popcount over the bitmap -> rank(doc) = accumulator slot;
//   sweep C  load each batch again (L2 hits), rank every posting, add weight*impact into
//            vals[rank] slot by slot = in list order; vals starts at +0.0, so a doc's sum is
//            ((0.0 + x_a) + x_b) + ... exactly as the reference forms it; docid[rank] = doc;
//   P4       walk the ranks in order (conflict-free LDS reads): finished sum + doc -> top-k.
// No posting "owns" a doc, so P1 needs no returning atomics and P4 no per-posting reads.
// A round whose postings exceed the 512 accumulators, or whose docs span more than the 16 384
// doc window, is cut at a common doc id and finished in further chunks.
//
// MaxScore pruning (strategies Wand / Bmw; query/wand.rs:659-903 reaches the same top-k by
// skipping): the host marks as NON-ESSENTIAL the lists whose summed maximum contributions stay
// below the seed threshold (slg_api.hip) — a doc found only in them cannot reach the top-k.  Such
// lists set no bitmap bits in sweep A; in sweep C their postings are only probed against the
// bitmap of the essential lists and added (in list order) where the doc is present.
// Block skipping (wand.rs:205-265 advance_to / skip_to_block: postings between candidate docs are
// never visited): a 64-posting slot of a non-essential list is a block; its first and last doc id
// are read (2 sectors) before sweep A, and after P2 the bitmap's rank structure says how many
// essential docs lie in [first, last] — none: the slot is dropped from sweep C, its 512 B of
// postings are never loaded.  A round with no essential posting at all is skipped outright.
//
// Score plans (query/planner.rs:113-153): when several terms share a ScorePlan leaf
// (multi-field query strings) or the root is a DisMax, the lists arrive sorted by leaf; vals[]
// then holds the CURRENT leaf's partial sums, and at every leaf change all accumulators are
// closed in bulk: Sum: acc += leaf (acc starts at -0.0, the f32 Sum identity); DisMax:
// max = max(max, leaf), sum += leaf (every leaf counts, absent ones as 0.0), final
// max + tie * (sum - max).  Adding an untouched leaf's +0.0 is exact, so the bulk close gives
// bit for bit what the reference computes per doc.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_score.hpp"

namespace slg {

// (kMultiCap = 512 accumulators per chunk, kMultiTarget: slg_desc.hpp)
constexpr int kMultiFill = 448;      // postings taken when a round has to be cut
constexpr int multi_wave_lds(int kregs) {
  return kSpanWords * 4 + kSpanWords * 4 + (kMultiCap + 64) * 4 + kMultiCap * 4 +
         (uni_buffered(kregs) ? buftopk_lds(kregs) : 0);
}
constexpr int kMultiPlanLds = 2 * kMultiCap * 4;  // acc[] and max[] of the leaf close (x2 for two-level plans)

// MODE 0: flat sums; 1: the batch has MaxScore-classified sub-queries (non-essential lists are
// only probed); 2: the batch has score plans (leaf close); 3: some of them two-level (group close);
// 4: some of them deeper trees (a close per level, slg_desc.hpp: PlanNode).
// Separate instantiations: the extra code of one mode costs the others registers.
// (the tree modes hold 18 / 26 KB of LDS per wave — 2 / 1.5 waves per SIMD — so they may as well have
//  the registers of 2 waves per SIMD: no spills)
#ifndef SLG_MULTI_WAVES
#define SLG_MULTI_WAVES 4  // waves per SIMD of the flat / classified / flat-plan modes (114-126 VGPRs).  At 5 (96 VGPRs) the
                           // compiler spills 24-35 registers to scratch: 0.40 against 0.29 ms on the multi-field workload
#endif
template <int KREGS, int MODE>
__global__ void __launch_bounds__(64, (MODE >= 3 ? 2 : SLG_MULTI_WAVES)) score_multi_kernel(RoundScoreParams p) {
  constexpr bool MS = MODE == 1;
  constexpr bool PL = MODE >= 2;
  constexpr bool NE = MODE >= 3;  // two-level plans (its own instantiation: the group close costs registers)
  constexpr bool DEEP = MODE == 4;  // trees of 3 .. SLG_MAX_PLAN_DEPTH levels
  constexpr int NS = kUniSlots;
  constexpr bool BUF = uni_buffered(KREGS);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t widx = blockIdx.x;
  if (widx >= p.n_slices) return;
  const uint32_t slice = rfl(p.slice_order[widx]);

  uint32_t *bm = reinterpret_cast<uint32_t *>(smem);
  uint32_t *pre = bm + kSpanWords;
  uint32_t *vals = pre + kSpanWords;          // [kMultiCap] + 64 dump words
  uint32_t *docid = vals + kMultiCap + 64;    // [kMultiCap]
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
  const gu32_t gdocs = (gu32_t)sd.docs;
  const gf32_t gimps = (gf32_t)sd.imps;
  const uint64_t null_idx = sd.null_idx;
  const uint32_t fid = rfl(s.filter);
  const gu32_t gdel = (gu32_t)(fid ? p.reject_table[(size_t)(fid - 1) * p.n_segs + s.seg] : sd.deleted);
  const uint32_t k = p.k;

  // lane t < T: list t's posting offset, weight, term id
  uint64_t my_off = 0;
  float my_w = 0.0f;
  uint32_t my_leaf = 0, my_gmeta = 0;
  float my_gtie = 0.0f;
  if (lane < T) {
    const TermRef tr = p.terms[s.term_begin + lane];
    my_off = tr.off;
    my_w = tr.weight;
    my_leaf = tr.leaf;
    my_gmeta = tr.gmeta;
    my_gtie = tr.gtie;
  }
  const uint32_t my_off_lo = (uint32_t)my_off, my_off_hi = (uint32_t)(my_off >> 32);
  const uint32_t ess_mask = MS ? rfl(s.ess_mask) : 0xFFFFFFFFu;  // bit t: list t is essential
  const bool my_ess = lane < T && ((ess_mask >> lane) & 1u);
  // score plan: 0 flat sum, 1 Sum of multi-term leaves, 2 DisMax of leaves
  const uint32_t plan = PL ? rfl(s.plan) & 0xFFu : 0u;  // (bits 8..: min_match, few-term plan kernel only)
  const float tie = __uint_as_float(rfl(__float_as_uint(s.tie)));
  const uint32_t max_init = rfl(__float_as_uint(s.max_init));
  const uint32_t n_leaves = rfl(s.n_leaves);
  float *acc = reinterpret_cast<float *>(smem + multi_wave_lds(KREGS));  // only if plan_batch
  float *mxv = acc + kMultiCap;
  // two-level plans (plan_batch == 2): acc / mxv accumulate the CURRENT GROUP's leaves, racc / rmx
  // the root over the closed groups
  const uint32_t n_groups = NE ? rfl(s.n_groups) : 0u;
  const bool nested = n_groups != 0u;
  float *racc = mxv + kMultiCap;
  float *rmx = racc + kMultiCap;
  // deep trees: level l of the canonical tree (0 = the root) accumulates in lacc(l) / lmx(l); the
  // deepest level shares acc / mxv's place in the numbering: level l at acc + 2 * l * kMultiCap
  const uint32_t depth = DEEP ? rfl(s.depth) : 0u;
  const PlanNode *const nodes = DEEP ? p.plan_nodes + rfl(s.node_begin) : nullptr;
  auto lacc = [&](const uint32_t l) { return acc + 2u * l * (uint32_t)kMultiCap; };
  auto lmx = [&](const uint32_t l) { return acc + (2u * l + 1u) * (uint32_t)kMultiCap; };
  const gu32_t gbounds = (gu32_t)p.bounds + s.bounds_begin + (size_t)r0 * T;
  const gu32_t grdoc = (gu32_t)p.rdoc + s.rdoc_begin + r0;

  BufTopK<BUF ? KREGS : 1> btop;
  btop.init(reinterpret_cast<uint64_t *>(docid + kMultiCap));
  uint32_t ccur = 0;
  uint64_t cbeg = 0;
  if (!BUF) {  // k > 256: candidate region (see slg_score_uni.hpp)
    const uint32_t b0 = lane < T ? gbounds[lane] : 0u;
    cbeg = (((uint64_t)rfl(s.cand_hi) << 32) | rfl(s.cand_lo)) + wave_sum(b0);
  }
  uint2 *const creg = BUF ? nullptr : p.cand + cbeg;
  {  // threshold seed (RoundQuery::theta0, set by the host planner)
    const float th0 = __uint_as_float(rfl(__float_as_uint(s.theta0)));
    if (th0 > 0.0f) btop.set_floor(th0);
  }
  uint32_t n_scored = 0;
  const uint32_t skip_mask = MS && p.block_skip ? rfl(s.skip_mask) : 0u;
  const bool skipping = skip_mask != 0u;
  uint32_t n_skipped = 0;  // postings of non-essential lists that were never loaded
#ifdef SLG_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
  const unsigned long long st_begin = wall_clock64();
#endif

  // slot descriptors of the current chunk: lane G = global slot G (list, count, 64-bit index)
  uint32_t d_st = 0, d_cnt = 0, d_lo = 0, d_hi = 0;

  // registers of one batch of 8 slots, and of the next one (in flight while this one is used)
  uint32_t doc[NS], ndoc[NS];
  float imp[NS], nimp[NS];
  auto issue_batch = [&](const uint32_t b, const uint32_t lo, const uint32_t hi) {
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      const uint64_t base = ((uint64_t)rl(hi, b * 8u + jj) << 32) | rl(lo, b * 8u + jj);
      ndoc[jj] = gdocs[base + lane];
      nimp[jj] = gimps[base + lane];
    }
  };
  auto take_batch = [&](const uint32_t b, const uint32_t dhi) {
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {  // lanes past the slot's count / docs past the cut: idle
      // (no lane count: the arrays are padded per list, so lanes past the slot's postings hold
      //  later postings of the same list — doc >= the cut — or sentinels; unused slots load sentinels)
      const bool live = ndoc[jj] < dhi;
      doc[jj] = live ? ndoc[jj] : kDocEnd;
      imp[jj] = nimp[jj];
    }
  };
  // slot descriptors for per-list counts c (lane t): list t takes ceil(c/64) consecutive global
  // slots; lane G of (st, cnt, lo, hi) describes slot G; returns the number of slots
  auto describe = [&](const uint32_t c, const uint32_t cur, uint32_t &st, uint32_t &cnt, uint32_t &lo,
                      uint32_t &hi) {
    const uint32_t m = (c + 63u) >> 6;
    const uint32_t gs_incl = wave_incl_scan(m);
    const uint32_t gs = gs_incl - m;        // first global slot of my list
    const uint32_t S = rl(gs_incl, 63);      // slots in use
    // lane G: the list that owns slot G = number of lists that end at or before G
    uint32_t tG = 0;
    for (uint32_t t = 0; t < T; t++) tG += (rl(gs_incl, t) <= lane) ? 1u : 0u;
    tG = tG < T ? tG : T - 1;
    const uint32_t l_gs = __shfl(gs, (int)tG, 64), l_c = __shfl(c, (int)tG, 64);
    const uint64_t l_abs = (((uint64_t)__shfl(my_off_hi, (int)tG, 64) << 32) |
                            __shfl(my_off_lo, (int)tG, 64)) +
                           __shfl(cur, (int)tG, 64);
    const uint32_t kin = (lane - l_gs) * 64u;
    const bool used = lane < S;
    const uint32_t left = used && l_c > kin ? l_c - kin : 0u;
    const uint64_t base = used ? l_abs + kin : null_idx;  // unused slot: a run of sentinels
    st = tG;
    cnt = left < 64u ? left : 64u;
    lo = (uint32_t)base;
    hi = (uint32_t)(base >> 32);
    return S;
  };

  // cut points and doc starts of rounds rr / rr + 1 (lane t: list t); the pair for the round after
  // is loaded while this round is processed
  uint32_t cutA = 0, cutB = 0;
  if (lane < T) {
    cutA = gbounds[lane];
    cutB = gbounds[T + lane];
  }
  uint32_t rdA = grdoc[0], rdB = grdoc[1];
  for (uint32_t rr = 0; rr < n_r; rr++) {
    // lane t: this round's range [cur, end) of list t; the round's doc range [dlo, rdhi)
    uint32_t cur = cutA;
    const uint32_t end = cutB;
    uint32_t dlo = rfl(rdA);
    const uint32_t rdhi = rfl(rdB);
    cutA = cutB;
    rdA = rdB;
    if (rr + 1 < n_r) {
      if (lane < T) cutB = gbounds[(rr + 2) * T + lane];
      rdB = grdoc[rr + 2];
    }

    for (uint32_t guard = 0; guard < (1u << 22); guard++) {  // chunks of the round (usually one)
      const uint32_t rem = end - cur;
      const uint32_t R = wave_sum(rem);
      if (R == 0) break;
      // essential postings need accumulators; all slots' descriptors live in the 64 lanes
      // (without pruning every posting is essential and R <= 512 already implies <= 8 + T slots)
      const uint32_t R_ess = MS ? wave_sum(my_ess ? rem : 0u) : R;
      if (MS && p.block_skip && R_ess == 0u) {  // no essential posting left in this round: nothing can score
        n_skipped += R;
        break;
      }
      const uint32_t S_all = MS ? wave_sum((rem + 63u) >> 6) : 0u;
      uint32_t chunk = rem, dhi = rdhi;
      bool cut = false;
      if (R_ess > (uint32_t)kMultiCap || S_all > 60u) {
        // too many postings for the accumulators (or slots): take a proportional part of every
        // list and cut at the smallest "last loaded doc" of the lists that were not taken whole
        float share = 1.0f;
        if (R_ess > (uint32_t)kMultiCap) share = (float)kMultiFill / (float)R_ess;
        // the slot descriptors of a chunk live in the wave's 64 lanes: list t takes
        // ceil(c_t / 64) <= c_t / 64 + 1 slots with c_t <= max(1, rem_t * share), so a share of
        // (63 - T) * 64 / R bounds the chunk by (63 - T) + T = 63 slots whatever the lists' mix
        // (a bound in postings alone, 48 * 64 / R, let 18..32 lists reach 48 + T > 64 slots)
        if (S_all > 60u) share = fminf(share, (float)((63u - T) * 64u) / (float)R);
        uint32_t c = (uint32_t)((float)rem * share);
        c = c < 1u ? 1u : c;
        chunk = rem < c ? rem : c;
        uint32_t lastdoc = kDocEnd;
        if (chunk < rem) lastdoc = gdocs[my_off + cur + chunk - 1];
        const uint32_t bound = wave_min(lastdoc);
        dhi = bound == kDocEnd ? rdhi : bound + 1u;
        cut = true;
      }
      const uint32_t wbase = dlo & ~31u;
      if (dhi - wbase > kSpan) {  // docs span more than one bitmap window
        dhi = wbase + kSpan;
        cut = true;
      }
      // ---- slots of the chunk (all lists, term order); with pruning also the essential lists
      //      alone: sweep A touches nothing else ----
      const uint32_t S = describe(chunk, cur, d_st, d_cnt, d_lo, d_hi);
      uint32_t a_st = d_st, a_cnt = d_cnt, a_lo = d_lo, a_hi = d_hi, S_a = S;
      if constexpr (MS) S_a = describe(my_ess ? chunk : 0u, cur, a_st, a_cnt, a_lo, a_hi);
      const uint32_t nb_a = (S_a + 7u) >> 3;
      uint32_t nb = (S + 7u) >> 3;
      const uint32_t wspan = dhi - wbase;
      // block skipping: lane G = slot G of a non-essential list reads the slot's first and last doc
      // id now (the loads fly during sweep A); the skip test follows P2
      bool probe = false;
      uint32_t p_fd = 0, p_ld = 0;
      if (skipping) {
        probe = lane < S && d_cnt != 0u && ((skip_mask >> d_st) & 1u) != 0u;
        if (probe) {
          const uint64_t base = ((uint64_t)d_hi << 32) | d_lo;
          p_fd = gdocs[base];
          p_ld = gdocs[base + d_cnt - 1u];
        }
      }
      SLG_STAMP(0);

      // ---- P0: clear bitmap and accumulators ----
      bm4[lane] = make_uint4(0u, 0u, 0u, 0u);
      bm4[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
      vals4[lane] = make_uint4(0u, 0u, 0u, 0u);
      vals4[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
      if (plan) {
        const uint32_t a0 = plan == 2u ? 0u : 0x80000000u;  // DisMax sum starts at 0.0, Sum at -0.0
        uint4 *acc4 = reinterpret_cast<uint4 *>(acc), *mx4 = reinterpret_cast<uint4 *>(mxv);
        acc4[lane] = make_uint4(a0, a0, a0, a0);
        acc4[lane + 64] = make_uint4(a0, a0, a0, a0);
        mx4[lane] = make_uint4(max_init, max_init, max_init, max_init);
        mx4[lane + 64] = make_uint4(max_init, max_init, max_init, max_init);
        if (nested) {  // the root: a DisMax sum starts at 0.0, a Sum at -0.0; its max at -inf
          uint4 *ra4 = reinterpret_cast<uint4 *>(racc), *rm4 = reinterpret_cast<uint4 *>(rmx);
          ra4[lane] = make_uint4(a0, a0, a0, a0);
          ra4[lane + 64] = make_uint4(a0, a0, a0, a0);
          rm4[lane] = make_uint4(0xFF800000u, 0xFF800000u, 0xFF800000u, 0xFF800000u);
          rm4[lane + 64] = make_uint4(0xFF800000u, 0xFF800000u, 0xFF800000u, 0xFF800000u);
        }
      }
      wave_fence();
      SLG_STAMP(1);
      // ---- sweep A: one bit per posting of the essential lists.  Only the doc ids are needed
      //      here (the impacts are read once, in sweep C), so the loads are half as wide and two
      //      batches are kept in flight in the registers sweep C uses for doc ids + impacts ----
      {
        uint32_t da[NS], db[NS];
        auto issue_docs = [&](uint32_t (&dst)[NS], const uint32_t bb) {
#pragma unroll
          for (int jj = 0; jj < NS; jj++) {
            const uint64_t base = ((uint64_t)rl(a_hi, bb * 8u + jj) << 32) | rl(a_lo, bb * 8u + jj);
            dst[jj] = gdocs[base + lane];
          }
        };
        auto set_bits = [&](const uint32_t (&src)[NS], const uint32_t bb) {
#pragma unroll
          for (int jj = 0; jj < NS; jj++) {
            const bool live = src[jj] < dhi;
            const uint32_t rel = src[jj] - wbase;
            if (live && rel < wspan) atomicOr(&bm[rel >> 5], 1u << (rel & 31u));
          }
        };
        if (nb_a > 0) issue_docs(da, 0);
        if (nb_a > 1) issue_docs(db, 1);
        for (uint32_t bb = 0; bb < nb_a; bb += 2) {
          set_bits(da, bb);
          if (bb + 2 < nb_a) issue_docs(da, bb + 2);
          if (bb + 1 < nb_a) {
            set_bits(db, bb + 1);
            if (bb + 3 < nb_a) issue_docs(db, bb + 3);
          }
        }
      }
      SLG_STAMP(2);
      issue_batch(0, d_lo, d_hi);  // sweep C's first batch
      wave_fence();
      // ---- P2: exclusive prefix popcount (lane l owns words 8l..8l+7: bit order = doc order, so
      //      rank(doc) is monotone and a doc range's population is a difference of two ranks) ----
      uint32_t ndocs;
      {
        const uint4 a = bm4[2 * lane], bb = bm4[2 * lane + 1];
        const uint32_t c0 = __popc(a.x), c1 = c0 + __popc(a.y), c2 = c1 + __popc(a.z),
                       c3 = c2 + __popc(a.w), c4 = c3 + __popc(bb.x), c5 = c4 + __popc(bb.y),
                       c6 = c5 + __popc(bb.z), c7 = c6 + __popc(bb.w);
        const uint32_t incl = wave_incl_scan(c7);
        const uint32_t ex = incl - c7;
        pre4[2 * lane] = make_uint4(ex, ex + c0, ex + c1, ex + c2);
        pre4[2 * lane + 1] = make_uint4(ex + c3, ex + c4, ex + c5, ex + c6);
        ndocs = rl(incl, 63);
        n_scored += ndocs;
      }
      wave_fence();
      // leaf close (score plans): fold the current leaf's partial sums into the per-doc totals
      // A DisMax takes the max over ALL leaves of the plan; a leaf none of whose lists has a
      // posting in this chunk is never closed here and counts as 0.0 (planner.rs:138-150).
      uint32_t n_closed = 0;
      auto close_leaf = [&](const bool final) {
        n_closed++;
        const bool some_leaf_idle = final && n_closed < n_leaves;
        for (uint32_t r = lane; r < ndocs; r += 64) {
          const float c = __uint_as_float(vals[r]);
          const float a = acc[r] + c;
          float m = fmaxf(mxv[r], c);
          if (final) {
            if (some_leaf_idle) m = fmaxf(m, 0.0f);
            vals[r] = __float_as_uint(plan == 2u ? m + tie * (a - m) : a);
          } else {
            acc[r] = a;
            mxv[r] = m;
            vals[r] = 0u;
          }
        }
        wave_fence();
      };
      // Two-level plans (ScoreExpr::evaluate recursion, planner.rs:122-153): the lists arrive sorted
      // by leaf and a group's leaves are consecutive.  A leaf change folds vals into the GROUP's
      // (acc, mxv); a group change folds the group's value — Sum: acc; DisMax: max + tie * (sum -
      // max), a leaf of the group without a posting in this chunk counting as 0.0 — into the root's
      // (racc, rmx) and resets the group accumulators for the next group's kind.
      uint32_t g_closed = 0, groups_closed = 0, cur_group = 0xFFFFFFFFu, cur_gmeta = 0;
      float cur_gtie = 0.0f;
      auto open_group = [&](const uint32_t gmeta) {  // bulk init of the group accumulators
        const float a0 = (gmeta >> 16) & 1u ? 0.0f : -0.0f;
        for (uint32_t r = lane; r < ndocs; r += 64) {
          acc[r] = a0;
          mxv[r] = -INFINITY;
        }
        wave_fence();
      };
      auto close_group_leaf = [&]() {  // the current leaf's sums -> the group
        g_closed++;
        for (uint32_t r = lane; r < ndocs; r += 64) {
          const float c = __uint_as_float(vals[r]);
          acc[r] = acc[r] + c;
          mxv[r] = fmaxf(mxv[r], c);
          vals[r] = 0u;
        }
        wave_fence();
      };
      auto close_group = [&](const bool final) {  // the group's value -> the root (-> vals when final)
        groups_closed++;
        const bool g_dismax = (cur_gmeta >> 16) & 1u;
        const bool leaf_idle = g_closed < ((cur_gmeta >> 8) & 0xFFu);
        const bool group_idle = final && groups_closed < n_groups;
        for (uint32_t r = lane; r < ndocs; r += 64) {
          const float a = acc[r];
          float m = mxv[r];
          if (leaf_idle) m = fmaxf(m, 0.0f);
          const float gv = g_dismax ? m + cur_gtie * (a - m) : a;
          const float ra = racc[r] + gv;
          float rm = fmaxf(rmx[r], gv);
          if (final) {
            if (group_idle) rm = fmaxf(rm, 0.0f);
            vals[r] = __float_as_uint(plan == 2u ? rm + tie * (ra - rm) : ra);
          } else {
            racc[r] = ra;
            rmx[r] = rm;
          }
        }
        wave_fence();
      };
      // Deeper trees (slg_score_plans::q_node_offsets; canonical form: slg_desc.hpp PlanNode): level l
      // of the tree (0 = the root, depth - 1 = the nodes the leaves hang off) has one OPEN node at a time
      // — the lists arrive in leaf = traversal order — with its (sum, max) of the children closed so far
      // in lacc(l) / lmx(l).  A leaf change closes the leaf into level depth - 1, then every level whose
      // node changes, bottom up, into the level above (Sum: the sum; DisMax: max + tie * (sum - max), a
      // child without a posting in this chunk counting as 0.0: closed children < the node's n_children),
      // and opens the new path's nodes from the first level that differs.
      // (every loop over the levels is unrolled with constant indices: d_path / d_closed stay in scalar
      //  registers instead of scratch memory)
      constexpr int MAXD = (int)kMaxPlanDepth;
      uint32_t d_path[MAXD] = {0u, 0u, 0u, 0u}, d_closed[MAXD] = {0u, 0u, 0u, 0u};
      auto deep_open = [&](const int l, const uint32_t node) {
        const float a0 = rfl(nodes[node].kind) ? 0.0f : -0.0f;  // DisMax sums from 0.0, Sum from -0.0
        float *A = lacc((uint32_t)l), *M = lmx((uint32_t)l);
        for (uint32_t r = lane; r < ndocs; r += 64) {
          A[r] = a0;
          M[r] = -INFINITY;
        }
        d_path[l] = node;
        d_closed[l] = 0u;
        wave_fence();
      };
      auto deep_close_leaf = [&]() {  // the current leaf's sums -> level depth - 1
        float *A = lacc(depth - 1u), *M = lmx(depth - 1u);
        for (uint32_t r = lane; r < ndocs; r += 64) {
          const float c = __uint_as_float(vals[r]);
          A[r] = A[r] + c;
          M[r] = fmaxf(M[r], c);
          vals[r] = 0u;
        }
#pragma unroll
        for (int l = 0; l < MAXD; l++) d_closed[l] += (uint32_t)l + 1u == depth ? 1u : 0u;
        wave_fence();
      };
      // the open node of level l -> level l - 1 (l >= 1), or -> vals when l == 0 (the root, at the end)
      auto deep_close_node = [&](const int l) {
        const PlanNode nd = nodes[d_path[l]];
        const bool dismax = rfl(nd.kind) != 0u;
        const float ntie = __uint_as_float(rfl(__float_as_uint(nd.tie)));
        const bool idle = d_closed[l] < rfl(nd.n_children);
        float *A = lacc((uint32_t)l), *M = lmx((uint32_t)l);
        float *PA = l ? lacc((uint32_t)l - 1u) : nullptr, *PM = l ? lmx((uint32_t)l - 1u) : nullptr;
        for (uint32_t r = lane; r < ndocs; r += 64) {
          const float a = A[r];
          float m = M[r];
          if (idle) m = fmaxf(m, 0.0f);
          const float v = dismax ? m + ntie * (a - m) : a;
          if (l) {
            PA[r] = PA[r] + v;
            PM[r] = fmaxf(PM[r], v);
          } else {
            vals[r] = __float_as_uint(v);
          }
        }
        if (l) d_closed[l > 0 ? l - 1 : 0]++;
        wave_fence();
      };
      auto deep_leaf_change = [&](const uint32_t leaf_parent, const bool had_leaf) {
        uint32_t np[MAXD] = {0u, 0u, 0u, 0u};  // the new leaf's path, root first
        uint32_t walk = leaf_parent;
#pragma unroll
        for (int l = MAXD - 1; l >= 0; l--)
          if ((uint32_t)l < depth) {
            np[l] = walk;
            walk = rfl(nodes[walk].parent);
          }
        uint32_t div = 0u;  // first level whose node changes (all of them for the chunk's first leaf)
        if (had_leaf) {
          deep_close_leaf();
          div = depth;
#pragma unroll
          for (int l = MAXD - 1; l >= 0; l--)
            if ((uint32_t)l < depth && np[l] != d_path[l]) div = (uint32_t)l;
#pragma unroll
          for (int l = MAXD - 1; l >= 1; l--)
            if ((uint32_t)l < depth && (uint32_t)l >= div) deep_close_node(l);
        }
#pragma unroll
        for (int l = 0; l < MAXD; l++)
          if ((uint32_t)l < depth && (uint32_t)l >= div) deep_open(l, np[l]);
      };
      auto deep_finish = [&]() {
        deep_close_leaf();
#pragma unroll
        for (int l = MAXD - 1; l >= 1; l--)
          if ((uint32_t)l < depth) deep_close_node(l);
        deep_close_node(0);
      };
      uint32_t cur_leaf = 0xFFFFFFFFu;
      uint32_t consumed = 0;  // what every list consumes: its postings below the cut
      if (skipping) {
        // essential docs in [first, last] of my slot = rank(last + 1) - rank(first); rank(x) =
        // docs of the window below x (bits at or past wspan are never set)
        auto rank_below = [&](const uint32_t x) {
          const uint32_t wi = (x >> 5) & (kSpanWords - 1);
          const uint32_t r = pre[wi] + __popc(bm[wi] & ((1u << (x & 31u)) - 1u));
          return x >= wspan ? ndocs : r;
        };
        // a slot that begins at or past the cut has no posting in this chunk; one that straddles
        // the cut is kept (its consumed count comes from the loaded doc ids)
        const bool past = probe && p_fd >= dhi;
        const bool whole = probe && p_ld < dhi;
        const uint32_t hits = probe && !past ? rank_below((p_ld < dhi ? p_ld : dhi - 1u) - wbase + 1u) -
                                                   rank_below(p_fd - wbase)
                                             : 0u;
        const bool skip = probe && (past || (whole && hits == 0u));
        const uint64_t skipm = __ballot(skip);
        if (skipm != 0ull) {
          n_skipped += wave_sum(skip && whole ? d_cnt : 0u);  // (slots past the cut come again)
          if (cut)  // a skipped slot below the cut is consumed whole
            for (uint32_t t = 0; t < T; t++) {
              const uint32_t c = wave_sum(skip && whole && d_st == t ? d_cnt : 0u);
              consumed += lane == t ? c : 0u;
            }
          // compact the descriptor lanes of the kept slots (order preserved = list order); the
          // dropped lanes all land on lane 63 with count 0, lanes nobody writes read 0
          const uint64_t keepm = ~skipm;
          const uint32_t pos = skip ? 63u
                                    : __builtin_amdgcn_mbcnt_hi((uint32_t)(keepm >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((uint32_t)keepm, 0u));
          const uint32_t kept = (uint32_t)__popcll(keepm & (S >= 64u ? ~0ull : ((1ull << S) - 1ull)));
          const bool first8_same = (skipm & 0xFFull) == 0ull;
          d_cnt = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos * 4u), (int)(skip ? 0u : d_cnt));
          d_st = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos * 4u), (int)d_st);
          d_lo = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos * 4u), (int)d_lo);
          d_hi = (uint32_t)__builtin_amdgcn_ds_permute((int)(pos * 4u), (int)d_hi);
          if (kept < 64u && lane >= kept) {  // (lane 63 may hold a dropped slot's leftovers)
            d_cnt = 0u;
            d_lo = (uint32_t)null_idx;
            d_hi = (uint32_t)(null_idx >> 32);
          }
          nb = (kept + 7u) >> 3;
          if (!first8_same && nb != 0u) issue_batch(0, d_lo, d_hi);  // batch 0 changed: load it again
        }
      }
      SLG_STAMP(3);
      // ---- sweep C: rank every posting, accumulate slot by slot (= in list order) ----
      for (uint32_t b = 0; b < nb; b++) {
        take_batch(b, dhi);
        if (b + 1 < nb) issue_batch(b + 1, d_lo, d_hi);
        uint32_t rank[NS];
        bool in[NS];
        {
          uint32_t wd[NS], pf[NS], bit[NS];
#pragma unroll
          for (int jj = 0; jj < NS; jj++) {
            const uint32_t rel = doc[jj] - wbase;
            const uint32_t wi = (rel >> 5) & (kSpanWords - 1);
            in[jj] = rel < wspan;
            if (cut) {  // (the ballot must run in all lanes: keep it out of the lane select)
              const uint32_t n_in = (uint32_t)__popcll(__ballot(in[jj]));
              consumed += lane == rl(d_st, b * 8u + jj) ? n_in : 0u;
            }
            bit[jj] = 1u << (rel & 31u);
            wd[jj] = bm[wi];
            pf[jj] = pre[wi];
          }
#pragma unroll
          for (int jj = 0; jj < NS; jj++) {
            rank[jj] = pf[jj] + __popc(wd[jj] & (bit[jj] - 1u));
            // a posting of a non-essential list counts only if an essential list has the doc
            if constexpr (MS) {
              const bool ess_slot = (ess_mask >> rl(d_st, b * 8u + jj)) & 1u;  // uniform
              in[jj] = in[jj] && (ess_slot || (wd[jj] & bit[jj]) != 0u);
            }
          }
        }
#pragma unroll
        for (int jj = 0; jj < NS; jj++) {
          if (__ballot(in[jj]) == 0ull) continue;  // unused slot
          const uint32_t lst = rl(d_st, b * 8u + jj);
          if (plan) {
            const uint32_t lf = rl(my_leaf, lst);
            if (lf != cur_leaf) {
              if (DEEP && depth != 0u) {
                deep_leaf_change(rl(my_gmeta, lst), cur_leaf != 0xFFFFFFFFu);
              } else if (nested) {
                const uint32_t gm = rl(my_gmeta, lst);
                if (cur_leaf != 0xFFFFFFFFu) close_group_leaf();
                if ((gm & 0xFFu) != cur_group) {
                  if (cur_group != 0xFFFFFFFFu) close_group(false);
                  open_group(gm);
                  cur_group = gm & 0xFFu;
                  cur_gmeta = gm;
                  cur_gtie = __uint_as_float(rl(__float_as_uint(my_gtie), lst));
                  g_closed = 0;
                }
              } else if (cur_leaf != 0xFFFFFFFFu) {
                close_leaf(false);
              }
              cur_leaf = lf;
            }
          }
          // score_tf: impact * weight (query/wand.rs:285); the slot's list weight is a scalar.
          // (Pairing the read-modify-writes of two consecutive slots of one list — distinct docs, so
          // independent — was measured: +5 % on config 3, and the plan variant spills.)
          const float w = __int_as_float((int)rl((uint32_t)__float_as_int(my_w), lst));
          const uint32_t at = in[jj] ? rank[jj] : (uint32_t)kMultiCap + lane;
          const uint32_t old = vals[at];
          vals[at] = __float_as_uint(__uint_as_float(old) + imp[jj] * w);
          if (in[jj]) docid[rank[jj]] = doc[jj];
          wave_fence();
        }
      }
      wave_fence();
      SLG_STAMP(4);
      if (plan) {
        if (DEEP && depth != 0u) {
          deep_finish();
        } else if (nested) {
          close_group_leaf();
          close_group(true);
        } else {
          close_leaf(true);
        }
      }
      // ---- P4: the docs of the chunk in rank order -> top-k ----
      for (uint32_t base = 0; base < ndocs; base += 64) {
        const uint32_t r = base + lane;
        const bool have = r < ndocs;
        const uint32_t v = vals[have ? r : 0u];
        const uint32_t d = docid[have ? r : 0u];
        const uint32_t ok = ordered_score(__uint_as_float(v));
        const bool ps = have && btop.passes(((uint64_t)ok << 32) | (uint32_t)~d);
        if constexpr (BUF) {
          btop.append_checked(ps, ok, ~d, k, lane, (const uint32_t *)gdel);
        } else {
          const uint64_t mm = __ballot(ps);
          if (mm != 0ull) {
            const uint32_t at = ccur + __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0u));
            if (ps) creg[at] = make_uint2(ok, d);
            ccur += (uint32_t)__popcll(mm);
          }
        }
      }
      wave_fence();
      SLG_STAMP(5);
      // ---- advance ----
      if (!cut) break;  // the whole rest of the round was in this chunk
      cur += consumed;
      uint32_t firstdoc = kDocEnd;  // next chunk starts at the smallest doc not yet scored
      if (cur < end) firstdoc = gdocs[my_off + cur];
      dlo = wave_min(firstdoc);
      if (dlo == kDocEnd) break;
    }
  }

  // ---- write this slice's candidates ----
  if constexpr (BUF) {
    btop.write_out(p.slice_tk + (size_t)slice * k, p.slice_doc + (size_t)slice * k, k, lane,
                   (const uint32_t *)gdel);
  } else if (lane == 0) {
    p.slice_cbeg[slice] = cbeg;
    p.slice_ccnt[slice] = ccur;
  }
  if (p.q_scored && lane == 0 && n_scored) atomicAdd(&p.q_scored[s.q], n_scored);
  if (p.skip_counts && lane == 0 && n_skipped) {
    atomicAdd(&p.skip_counts[0], (unsigned long long)n_skipped);
    atomicAdd(&p.skip_counts[1u + s.q], (unsigned long long)n_skipped);
  }
#ifdef SLG_STAMPS
  SLG_STAMP(6);
  if (p.stamps && lane == 0) {
    for (int i = 0; i < 8; i++) p.stamps[(size_t)slice * 12 + i] = st_acc[i];
    p.stamps[(size_t)slice * 12 + 8] = 0;
    p.stamps[(size_t)slice * 12 + 9] = st_begin;
    p.stamps[(size_t)slice * 12 + 10] = wall_clock64();
    p.stamps[(size_t)slice * 12 + 11] = ((unsigned long long)n_r << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
  }
#endif
}

}  // namespace slg
