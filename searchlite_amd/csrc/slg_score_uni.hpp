// slg_score_uni.hpp — the hot kernel for queries with few terms (the host uses it up to 4 lists;
// BASELINE config 2): exact pre-planned rounds, LDS bitmap-rank accumulate, buffered top-k, and
// every 64-lane register slot holds postings of ONE list.
//
// Padding each list to a slot boundary makes the per-slot list id, weight, base address and
// lane count wave-uniform scalars (one v_readlane each from a lane-held slot descriptor; the
// descriptors of 8 rounds are computed at once), removes the per-lane list selects and the
// mixed-slot ordering paths of the older packed kernel (slg_score.hpp), and lets the posting
// loads be whole-slot, scalar base + lane.
//
// Restates query/wand.rs:459-566 (every posting scored, per-doc sums in ScorePlan leaf order,
// planner.rs:122-135) and push_top_k (wand.rs:905-916); phases P0..P4 are described in
// DESIGN.md section 4.  k <= 256: candidates go to a per-wave LDS buffer (BufTopK,
// slg_kernels.hpp); larger k: to the slice's region of a global candidate array, picked per
// query by select_topk_kernel.  More lists than 4: slg_score_multi.hpp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_score.hpp"

#ifndef SLG_ABL
#define SLG_ABL 0  // diagnostic builds only (tools/ablate.py): bit i skips a phase; results are wrong
#endif

namespace slg {

constexpr int kUniSlots = 8;                 // 64-posting slots per round; also max lists
constexpr int kUniCap = kUniSlots * 64;
constexpr int kUniWaveLdsBase = kSpanWords * 4 + kSpanWords * 4 + kUniCap * 4 + 64 * 4;
// k <= 256 (KREGS <= 4): buffered top-k in LDS (BufTopK); larger k: every doc above the seed
// threshold goes to the slice's candidate region and select_topk_kernel picks the k best
constexpr bool uni_buffered(int kregs) { return kregs <= 4; }
constexpr int uni_wave_lds(int kregs) {
  return kUniWaveLdsBase + (uni_buffered(kregs) ? 128 * kregs * 8 : 0);
}

template <int KREGS>
__global__ void __launch_bounds__(256, 4) score_uniform_kernel(RoundScoreParams p) {
  constexpr int NS = kUniSlots;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wib = threadIdx.x >> 6;
  const uint32_t widx = rfl(blockIdx.x * (blockDim.x >> 6) + wib);
  if (widx >= p.n_slices) return;  // waves are independent: no workgroup barrier anywhere
  const uint32_t slice = rfl(p.slice_order[widx]);

  constexpr bool BUF = uni_buffered(KREGS);
  uint32_t *bm = reinterpret_cast<uint32_t *>(smem + (size_t)wib * uni_wave_lds(KREGS));
  uint32_t *pre = bm + kSpanWords;
  uint32_t *vals = pre + kSpanWords;
  uint4 *bm4 = reinterpret_cast<uint4 *>(bm);
  uint4 *pre4 = reinterpret_cast<uint4 *>(pre);

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
  // accept(): tombstones, or the reject bitmap (deleted | ~filter) of the query's doc filter
  const uint32_t fid = rfl(s.filter);
  const gu32_t gdel = (gu32_t)(fid ? p.reject_table[(size_t)(fid - 1) * p.n_segs + s.seg] : sd.deleted);
  const uint32_t k = p.k;

  // lane t < T: list t's posting offset, weight, term id
  uint64_t my_off = 0;
  float my_w = 0.0f;
  uint32_t my_term = 0;
  if (lane < T) {
    const TermRef tr = p.terms[s.term_begin + lane];
    my_off = tr.off;
    my_w = tr.weight;
    my_term = tr.term;
  }
  const uint32_t my_off_lo = (uint32_t)my_off, my_off_hi = (uint32_t)(my_off >> 32);
  // all cut points of the slice in one register (lane i: bounds[r0*T + i]); round doc starts
  const uint32_t bflat = lane < (n_r + 1) * T ? p.bounds[s.bounds_begin + r0 * T + lane] : 0u;
  const uint32_t dflat = lane <= n_r ? p.rdoc[s.rdoc_begin + r0 + lane] : 0u;
  // lane i = r*T + t: postings of list t in round r
  const uint32_t dcnt = __shfl(bflat, (lane + T) & 63u, 64) - bflat;

  BufTopK<BUF ? KREGS : 1> btop;  // k <= 256; for larger k only its threshold is used
  btop.init(reinterpret_cast<uint64_t *>(vals + kUniCap + 64));
  // k > 256: the slice's candidate region starts at (sub-query base) + (postings of all lists
  // before the slice's first round) and can hold one entry per posting of the slice
  uint32_t ccur = 0;
  uint64_t cbeg = 0;
  if (!BUF) {
    uint32_t before = 0;
    for (uint32_t t = 0; t < T; t++) before += rl(bflat, t);
    cbeg = (((uint64_t)rfl(s.cand_hi) << 32) | rfl(s.cand_lo)) + before;
  }
  uint2 *const creg = BUF ? nullptr : p.cand + cbeg;
  if (sd.champ != nullptr && k <= 1024u && fid == 0) {  // (a filter may reject the champions)  // threshold seed (see slg_score.hpp)
    float f = 0.0f;
    if (lane < T && my_w > 0.0f)
      f = my_w * ((const gf32_t)sd.champ)[(size_t)my_term * kChampions + champ_index(k)];
    float best = 0.0f;
    for (uint32_t t = 0; t < T; t++)
      best = fmaxf(best, __int_as_float((int)rl((uint32_t)__float_as_int(f), t)));
    const bool anyneg = __ballot(lane < T && !(my_w >= 0.0f)) != 0ull;
    if (best > 0.0f && !anyneg) btop.set_floor(best);
  }
  uint32_t n_scored = 0;
#ifdef SLG_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_ins = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif

  // One round's postings: slot jj holds <= 64 postings of one list.  A slot is described by one
  // lane of a Desc (list, postings in the slot, 64-bit index of its first posting); the slots of
  // a round are the 8 lanes starting at `dbase`.
  struct Desc {
    uint32_t st, cnt, lo, hi;
    uint32_t nsl;  // slots the round needs (> NS: it does not fit and is streamed in chunks)
  };
  struct URound {
    uint32_t doc[NS];
    float imp[NS];
    uint32_t st, cnt;  // copies of the Desc's list ids / counts (it may be rebuilt meanwhile)
    uint32_t dbase;   // uniform: first descriptor lane of this round
    uint32_t nslots;  // uniform
  };

  // ---- descriptors of 8 consecutive planned rounds at once: lane 8*i + j = slot j of round
  //      g0 + i (per-round cost: a few readlanes instead of a prefix scan over the lists) ----
  auto describe_group = [&](Desc &d, const uint32_t g0) {
    const uint32_t ri = g0 + (lane >> 3), j = lane & 7u;
    const bool rv = ri < n_r;
    uint32_t run = 0;  // slots of the lists before list t
    d.st = 0;
    d.cnt = 0;
    d.lo = 0;
    d.hi = 0;
    for (uint32_t t = 0; t < T; t++) {
      const uint32_t src = (ri * T + t) & 63u;
      const uint32_t lo_t = __shfl(bflat, (int)src, 64);
      const uint32_t c_t = __shfl(dcnt, (int)src, 64);
      const uint32_t m = rv ? (c_t + 63u) >> 6 : 0u;
      const bool mine = j >= run && j < run + m;
      const uint32_t kin = (j - run) * 64u;  // postings of the list before this slot
      const uint64_t base = (((uint64_t)rl(my_off_hi, t) << 32) | rl(my_off_lo, t)) + lo_t + kin;
      const uint32_t left = c_t - kin;
      d.st = mine ? t : d.st;
      d.cnt = mine ? (left < 64u ? left : 64u) : d.cnt;
      d.lo = mine ? (uint32_t)base : d.lo;
      d.hi = mine ? (uint32_t)(base >> 32) : d.hi;
      run += m;
    }
    d.nsl = run;
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
    d.nsl = run;
    // lane j: which list owns slot j = the last list whose first slot is <= j
    uint32_t tj = 0;
    for (uint32_t t = 1; t < T; t++) tj = lane >= rl(ss, t) ? t : tj;
    const uint32_t l_ss = __shfl(ss, (int)tj, 64), l_cnt = __shfl(cnt, (int)tj, 64);
    const uint64_t l_abs = (((uint64_t)__shfl(my_off_hi, (int)tj, 64) << 32) |
                            __shfl(my_off_lo, (int)tj, 64)) +
                           __shfl(lo, (int)tj, 64);
    const uint32_t kin = (lane - l_ss) * 64u;
    const bool used = lane < run && lane < (uint32_t)NS;
    const uint32_t left = used && l_cnt > kin ? l_cnt - kin : 0u;
    d.st = tj;
    d.cnt = left < 64u ? left : 64u;
    const uint64_t base = l_abs + kin;
    d.lo = (uint32_t)base;
    d.hi = (uint32_t)(base >> 32);
  };

  // ---- issue the loads of the round described by lanes dbase .. dbase+7 of d: whole 64-lane
  //      slots, scalar base + lane (no per-lane predicate; the arrays are padded by 64 entries).
  //      Lanes past the slot's count hold other postings until settle() masks them. ----
  auto issue = [&](URound &r, const Desc &d, const uint32_t dbase) {
    r.st = d.st;
    r.cnt = d.cnt;
    r.dbase = dbase;
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      const uint64_t base = ((uint64_t)rl(d.hi, dbase + jj) << 32) | rl(d.lo, dbase + jj);
      r.doc[jj] = gdocs[base + lane];
      r.imp[jj] = gimps[base + lane];
    }
  };
  // ---- dst = the loaded round src, ready to accumulate: lanes beyond each slot's count become
  //      idle lanes (kDocEnd is never inside a doc window) and the impacts are multiplied by
  //      the slot's list weight (score_tf, query/wand.rs:285: one scalar per slot) ----
  auto settle = [&](URound &dst, const URound &src) {
    dst.st = src.st;
    dst.cnt = src.cnt;
    dst.dbase = src.dbase;
    dst.nslots = src.nslots;
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      const float w = __int_as_float((int)rl((uint32_t)__float_as_int(my_w), rl(src.st, src.dbase + jj)));
      dst.doc[jj] = lane < rl(src.cnt, src.dbase + jj) ? src.doc[jj] : kDocEnd;
      dst.imp[jj] = src.imp[jj] * w;
    }
  };

  // ---- accumulate the postings of `e` whose docs lie in [wbase, wbase + wspan) ----
  auto accumulate = [&](URound &e, const uint32_t wbase, const uint32_t wspan) {
    SLG_STAMP(1);
    if (SLG_ABL & 32) {
      n_scored += e.doc[0] & 1u;
      return;
    }
    // P0: clear the bitmap
    bm4[lane] = make_uint4(0u, 0u, 0u, 0u);
    bm4[lane + 64] = make_uint4(0u, 0u, 0u, 0u);
    wave_fence();
    // P1: one bit per posting (transposed bitmap: doc d -> word d mod 512, bit d / 512); the
    // returned old word tells which posting of a doc came first = the owner.  Slots are in
    // list order and a slot holds one list, so the owner is the first list in term order.
    uint32_t wi[NS], bit[NS];
    bool own[NS];
    {
      uint32_t oldw[NS];
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        const uint32_t rel = e.doc[jj] - wbase;
        wi[jj] = (SLG_ABL & 64) ? ((lane * 8u + jj) & (kSpanWords - 1)) : (rel & (kSpanWords - 1));
        bit[jj] = rel < wspan ? 1u << (rel >> 9) : 0u;  // rel < 16384 => rel >> 9 < 32
        if (SLG_ABL & 4) {
          oldw[jj] = 0;
          bm[wi[jj]] = bit[jj];
        } else {
          oldw[jj] = atomicOr(&bm[wi[jj]], bit[jj]);
        }
      }
#pragma unroll
      for (int jj = 0; jj < NS; jj++) own[jj] = (bit[jj] & ~oldw[jj]) != 0u;
    }
    wave_fence();
    SLG_STAMP(2);
    // P2: exclusive prefix popcount (lane l owns words 4l..4l+3 and 256+4l..256+4l+3)
    if (!(SLG_ABL & 16)) {
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
    // P3a: rank(doc) = accumulator slot (< kUniCap: a round holds <= kUniCap postings; idle
    // lanes index at most kUniCap + 31, inside the dump words); x = impact * weight (settle());
    // owners store 0.0 + x (`or_insert(0.0) += score`, query/wand.rs:539)
    uint32_t slot[NS];
    if (SLG_ABL & 8) {
#pragma unroll
      for (int jj = 0; jj < NS; jj++) slot[jj] = wi[jj];
    } else {
      uint32_t wd[NS], pf[NS];
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        wd[jj] = bm[wi[jj]];
        pf[jj] = pre[wi[jj]];
      }
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        slot[jj] = pf[jj] + __popc(wd[jj] & (bit[jj] - 1u));
        vals[own[jj] ? slot[jj] : kUniCap + lane] = __float_as_uint(0.0f + e.imp[jj]);
      }
    }
    wave_fence();
    // P3b: later postings of a doc (it was first seen in an earlier list = an earlier slot)
    // add to the owner's value, slot by slot: a wave's LDS operations execute in program
    // order and slots are in list order, so the sum is ((0.0 + x_a) + x_b) + ... in term order.
#pragma unroll
    for (int jj = 1; jj < NS; jj++) {
      const bool later = bit[jj] != 0u && !own[jj];
      if (!(SLG_ABL & 1) && __ballot(later) != 0ull) {
        const uint32_t old = vals[slot[jj]];
        vals[later ? slot[jj] : kUniCap + lane] = __float_as_uint(__uint_as_float(old) + e.imp[jj]);
        wave_fence();
      }
    }
    wave_fence();
    SLG_STAMP(4);
    // P4: owners read the finished sums and offer them to the top-k
    if (SLG_ABL & 2) {
      n_scored += slot[0] & 1u;
      return;
    }
    uint32_t v[NS];
#pragma unroll
    for (int jj = 0; jj < NS; jj++) v[jj] = vals[slot[jj]];
    if constexpr (BUF) {
      uint32_t okey[NS];
      bool pass[NS];
      bool any = false;
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        okey[jj] = ordered_score(__uint_as_float(v[jj]));
        pass[jj] = own[jj] && btop.passes(((uint64_t)okey[jj] << 32) | (uint32_t)~e.doc[jj]);
        any = any || pass[jj];
      }
      if (__ballot(any) != 0ull) {
        uint32_t tot = 0;
#pragma unroll
        for (int jj = 0; jj < NS; jj++) tot += (uint32_t)__popcll(__ballot(pass[jj]));
#ifdef SLG_STAMPS
        st_ins += tot;
#endif
        if (btop.count + tot <= btop.kEntries) {  // the common case: everything fits
#pragma unroll
          for (int jj = 0; jj < NS; jj++) btop.append(pass[jj], okey[jj], ~e.doc[jj], lane);
        } else {  // rank-and-trim between slots; one site, slot registers selected at run time
#pragma unroll 1
          for (uint32_t it = 0; it < (uint32_t)NS; it++) {
            uint32_t ok = okey[0], dc = e.doc[0];
            bool ps = pass[0];
#pragma unroll
            for (int j = 1; j < NS; j++) {
              ok = it == (uint32_t)j ? okey[j] : ok;
              dc = it == (uint32_t)j ? e.doc[j] : dc;
              ps = it == (uint32_t)j ? pass[j] : ps;
            }
            btop.append_checked(ps && btop.passes(ok, ~dc), ok, ~dc, k, lane, (const uint32_t *)gdel);
          }
        }
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        const uint32_t ok = ordered_score(__uint_as_float(v[jj]));
        const bool ps = own[jj] && btop.passes(((uint64_t)ok << 32) | (uint32_t)~e.doc[jj]);
        const uint64_t m = __ballot(ps);
        if (m != 0ull) {
          const uint32_t at = ccur + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                               __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (ps) creg[at] = make_uint2(ok, e.doc[jj]);
          ccur += (uint32_t)__popcll(m);
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

  // ---- driver: planned rounds are prefetched one ahead (`en` loads while `ew` is processed);
  //      a round that needs more than NS slots is streamed in chunks cut at a common doc id.
  //      Both paths and all doc windows share ONE accumulate site (code size / I-cache). ----
  URound ew, en;
  Desc G, C;  // descriptors of the current group of 8 planned rounds / of the current chunk
  describe_group(G, 0);
  en.nslots = rl(G.nsl, 0);
  en.st = G.st;
  en.dbase = 0;
  if (en.nslots <= (uint32_t)NS) issue(en, G, 0);
  for (uint32_t rr = 0; rr < n_r; rr++) {
    const bool big = en.nslots > (uint32_t)NS;
    uint32_t ocur = 0, oend = 0;
    if (big) cuts(rr, ocur, oend);
    if (!big) settle(ew, en);
#ifdef SLG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SLG_STAMP(6);
    if (rr + 1 < n_r) {  // prefetch the next round
      const uint32_t nx = rr + 1, db = (nx & 7u) * 8u;
      if (db == 0) describe_group(G, nx);
      en.nslots = rl(G.nsl, db);
      en.st = G.st;
      en.dbase = db;
      if (en.nslots <= (uint32_t)NS) issue(en, G, db);
    }
    uint32_t dlo = rl(dflat, rr), dhi = rl(dflat, rr + 1);
    SLG_STAMP(0);
    const uint32_t rhi = dhi;
    if (p.dbg & 4u) continue;
    uint32_t guard = 0;
    do {
      if (big) {
        // next chunk of an over-full round: every list gets >= 1 slot, the rest in proportion
        // to what it has left; the chunk ends at the smallest "last loaded doc" of the lists
        // that did not finish
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
        uint32_t lastdoc = kDocEnd, firstdoc = kDocEnd;
        if (chunk < rem) lastdoc = gdocs[my_off + ocur + chunk - 1];
        if (rem > 0) firstdoc = gdocs[my_off + ocur];
        describe_chunk(C, ocur, chunk);
        ew.nslots = C.nsl;
        issue(ew, C, 0);
        settle(ew, ew);
        uint32_t bound = kDocEnd;
        dlo = kDocEnd;
        for (uint32_t t = 0; t < T; t++) {
          const uint32_t ld = rl(lastdoc, t), fd = rl(firstdoc, t);
          bound = ld < bound ? ld : bound;
          dlo = fd < dlo ? fd : dlo;
        }
        dhi = bound == kDocEnd ? rhi : bound + 1u;
        // what each list consumed: its postings with doc < dhi (a prefix of its slots)
        uint32_t consumed = 0;
#pragma unroll
        for (int jj = 0; jj < NS; jj++) {
          const uint32_t cnt = (uint32_t)__popcll(__ballot(ew.doc[jj] < dhi));
          consumed += lane == rl(ew.st, jj) ? cnt : 0u;  // chunk descriptors: dbase 0
          ew.doc[jj] = ew.doc[jj] < dhi ? ew.doc[jj] : kDocEnd;  // the rest: next chunk
        }
        ocur += consumed;
      }
      if (ew.nslots == 0) break;
      // doc windows: one in the common case (the postings span <= kSpan docs)
      uint32_t wbase = dlo & ~31u;
      for (;;) {
        const uint32_t wend = (dhi - wbase) <= kSpan ? dhi : wbase + kSpan;
        accumulate(ew, wbase, wend - wbase);
        if (wend == dhi) break;
        uint32_t mn = kDocEnd;  // next window starts at the smallest doc not yet covered
#pragma unroll
        for (int jj = 0; jj < NS; jj++) mn = (ew.doc[jj] >= wend && ew.doc[jj] < mn) ? ew.doc[jj] : mn;
        mn = wave_min(mn);
        if (mn >= dhi) break;
        wbase = mn & ~31u;
      }
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
  if (p.q_scored && lane == 0 && n_scored) atomicAdd(&p.q_scored[s.q], n_scored);
#ifdef SLG_STAMPS
  SLG_STAMP(7);
  st_acc[7] = st_ins;
  if (p.stamps && lane == 0)
    for (int i = 0; i < 8; i++) p.stamps[(size_t)slice * 8 + i] = st_acc[i];
#endif
}

}  // namespace slg
