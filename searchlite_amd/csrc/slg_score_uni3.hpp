// slg_score_uni3.hpp — the few-term scoring kernel (<= 4 lists per sub-query; BASELINE config 2),
// third form: the round-2 kernel (slg_score_uni.hpp) rebuilt around what bounds it on gfx950.
//
// Same algorithm — exact pre-planned rounds of 8 one-list slots, FILTER + JOIN accumulate, buffered
// top-k; restates query/wand.rs:459-566 (every posting scored, per-doc sums in ScorePlan leaf order,
// planner.rs:122-135) and push_top_k (wand.rs:905-916) — but the round-2 kernel spent its time on
// instructions that touch the SCALAR register file: 37.6M SALU + ~18M v_readlane / v_cmp->SGPR /
// v_cndmask<-SGPR wave-instructions per launch against 0.55 such instructions per ns per SIMD
// (tools/micro/issue_rates.hip: plain VALU issues at 0.85 / ns / SIMD, anything that reads or
// writes an SGPR at 0.55, and the two kinds overlap only partly) = the kernel's 97 us.  Per-slot
// uniform values (posting index, list weight, list bit, lane count) lived in lanes of a descriptor
// VGPR and were brought to SGPRs by v_readlane, lane sets were 64-bit SGPR masks, idle lanes were
// rewritten to a sentinel by v_cmp + v_cndmask.  Here:
//   * slot descriptors are written to LDS once per 8 rounds and read back as BROADCAST ds_reads:
//     the uniform values arrive in VGPRs and every per-posting instruction is plain VALU;
//   * there are no idle lanes to mask.  Posting arrays are padded per list (SegDev, kListPad), so
//     a whole slot loaded at any posting holds only postings of ITS list or sentinels; lanes past
//     the round's cut hold later postings of the same list (doc >= the round's end doc).  Such
//     lanes take part in the filter like everybody else (the filter is one-sided: extra bits only
//     cost a spurious queue entry) and are told apart by `doc < end` only where a candidate or a
//     queue entry is actually produced;
//   * the per-posting tests produce VGPR values, OR-ed over the round (x != 0: my doc is, or
//     aliases, a doc of another list; u != 0: my score may reach the threshold — an integer
//     saturating subtract on the score bits), and ONE ballot per round decides whether the queue /
//     candidate code runs for it.
// One round = 3 dependent LDS round trips as before; ~120 plain VALU instructions for 8 slots where
// the round-2 kernel issued ~420 VALU + ~430 SALU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_score_uni.hpp"

namespace slg {

// per-wave LDS: [filter / join queue 4 KB][top-k buffer][cut points + list offsets (kUniPlanLds)]
//               [slot descriptors: 8 rounds x 8 slots + 8 chunk slots, 16 B each][round end docs]
constexpr int kU3Rows = 9;  // descriptor rows of 8 slots: 8 planned rounds + the chunk row
constexpr int kU3DescBytes = kU3Rows * 8 * 16;
constexpr int kU3EndBytes = (kMaxRoundsPerSlice + 1) * 4;
// ML = lists a sub-query may have: 4 (a filter field is 4 list bits) or 8 (8 list bits per field:
// half as many fields per word, so the filter gets twice the words)
constexpr int u3_filter_words(int ml) { return ml <= 4 ? kJoinWords : 2 * kJoinWords; }
constexpr int u3_plan_off(int kregs, int ml) { return u3_filter_words(ml) * 4 + (uni_buffered(kregs) ? buftopk_lds(kregs) : 0); }
constexpr int u3_plan_lds(int ml) { return 64 * 4 + 2 * ml * 4; }
constexpr int u3_desc_off(int kregs, int ml) { return u3_plan_off(kregs, ml) + u3_plan_lds(ml); }
constexpr int u3_wave_lds(int kregs, int ml) {
  return u3_desc_off(kregs, ml) + kU3DescBytes + ((kU3EndBytes + 15) & ~15);
}

#ifndef SLG_U3_WAVES
#define SLG_U3_WAVES 6  // 80 VGPRs, no spill; 24 waves x 6.6 KB of LDS per CU
#endif

#ifndef SLG_U3_WAVES8
#define SLG_U3_WAVES8 4  // ML = 8: 10.7 KB of LDS per wave = 15 waves per CU
#endif
constexpr int u3_waves(int kregs, int ml) { return ml > 4 ? SLG_U3_WAVES8 : (kregs >= 4 ? 5 : SLG_U3_WAVES); }

template <int KREGS, int ML>
__global__ void __launch_bounds__(64)
    __attribute__((amdgpu_waves_per_eu(u3_waves(KREGS, ML), u3_waves(KREGS, ML))))
score_uniform3_kernel(RoundScoreParams p) {
  constexpr int NS = kUniSlots;
  constexpr int FW = u3_filter_words(ML);       // filter words
  constexpr uint32_t LB = ML <= 4 ? 4u : 8u;    // list bits per filter field
  constexpr uint32_t LBM = (1u << LB) - 1u;
  // field of a doc: word = doc mod FW, shift = LB * ((doc / FW) mod (32 / LB))
  constexpr uint32_t FSH = (FW == 1024 ? 10u : 11u) - (LB == 4u ? 2u : 3u);
  constexpr uint32_t FSM = LB == 4u ? 0x1Cu : 0x18u;
  static_assert(FW == 1024 || FW == 2048, "filter size");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t widx = blockIdx.x;
  if (widx >= p.n_slices) return;  // waves are independent: no workgroup barrier anywhere
  const SliceDesc sl = p.slice_desc[widx];
  const uint32_t slice = rfl(sl.slice);

  constexpr bool BUF = uni_buffered(KREGS);
  uint32_t *flt = reinterpret_cast<uint32_t *>(smem);
  uint4 *flt4 = reinterpret_cast<uint4 *>(smem);
  uint2 *queue = reinterpret_cast<uint2 *>(smem);  // {doc, score} of queued postings; overlays flt
  uint4 *const sdesc = reinterpret_cast<uint4 *>(smem + u3_desc_off(KREGS, ML));  // {idx lo, idx hi, weight, meta}
  uint32_t *const rend = reinterpret_cast<uint32_t *>(smem + u3_desc_off(KREGS, ML) + kU3DescBytes);

  const uint32_t T = rfl(sl.n_terms);
  const uint32_t n_r = rfl(sl.n_rounds);
  const SegDev sd = p.segs[sl.seg];
  const gu32_t gdocs = (gu32_t)sd.docs;
  const gf32_t gimps = (gf32_t)sd.imps;
  // per-lane bases: a slot's loads are lanebase[idx] — one v_lshl_add_u64 each, no carry chain
  const gu32_t ldocs = gdocs + lane;
  const gf32_t limps = gimps + lane;
  const uint64_t null_idx = sd.null_idx;
  const uint32_t fid = rfl(sl.filter);
  const gu32_t gdel = (gu32_t)(fid ? p.reject_table[(size_t)(fid - 1) * p.n_segs + sl.seg] : sd.deleted);
  const uint32_t k = p.k;

  // lane t < T: list t's weight and posting offset; all cut points of the slice (entry r*T + t:
  // where round r starts in list t); the end doc of every round
  uint32_t *const bflat = reinterpret_cast<uint32_t *>(smem + u3_plan_off(KREGS, ML));
  uint32_t *const off_lo = bflat + 64, *const off_hi = off_lo + ML;
  float my_w = 0.0f;
  if (lane < T) {
    const TermRef tr = p.terms[sl.term_begin + lane];
    my_w = tr.weight;
    off_lo[lane] = (uint32_t)tr.off;
    off_hi[lane] = (uint32_t)(tr.off >> 32);
  }
  bflat[lane] = lane < (n_r + 1) * T ? p.bounds[sl.bounds_off + lane] : 0u;
  if (lane < n_r) rend[lane] = p.rdoc[sl.rdoc_off + lane + 1];
  wave_fence();
  auto list_off = [&](const uint32_t t) { return ((uint64_t)off_hi[t] << 32) | off_lo[t]; };

  BufTopK<BUF ? KREGS : 1> btop;  // k <= 256; for larger k only its threshold is used
  btop.init(reinterpret_cast<uint64_t *>(smem + FW * 4));
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
  uint32_t n_scored = 0;
  for (uint32_t t = 0; t < T; t++) n_scored += rfl(bflat[n_r * T + t]) - rfl(bflat[t]);

#ifdef SLG_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_ins = 0, st_queued = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
  const unsigned long long st_begin = wall_clock64();  // 100 MHz, device-wide
#endif

  struct URound {
    uint32_t doc[NS];
    float sc[NS];  // in flight: impact; settled: weight * impact (score_tf, wand.rs:285)
  };

  // a round's 8 list bits packed LB bits per slot (0: unused slot): an OR over the round's 8 lanes
  // (j = the lane's slot).  ML = 8: slots 0..3 in `lo`, 4..7 in `hi`
  auto pack_list_bits = [&](const uint32_t lbit, const uint32_t j, uint32_t &lo, uint32_t &hi) {
    if constexpr (ML <= 4) {
      lo = lbit << (4u * j);
      hi = 0u;
    } else {
      lo = j < 4u ? lbit << (8u * j) : 0u;
      hi = j >= 4u ? lbit << (8u * (j - 4u)) : 0u;
      hi |= (uint32_t)__shfl_xor((int)hi, 1, 64);
      hi |= (uint32_t)__shfl_xor((int)hi, 2, 64);
      hi |= (uint32_t)__shfl_xor((int)hi, 4, 64);
    }
    lo |= (uint32_t)__shfl_xor((int)lo, 1, 64);
    lo |= (uint32_t)__shfl_xor((int)lo, 2, 64);
    lo |= (uint32_t)__shfl_xor((int)lo, 4, 64);
  };
  auto slot_bit = [&](const uint64_t lbits, const int jj) { return (uint32_t)(lbits >> (LB * jj)) & LBM; };
  // ---- descriptors of 8 consecutive planned rounds: lane 8*i + j = slot j of round g0 + i.
  //      Written to LDS row i; meta = list bit | slots the round needs (saturated) << 8 ----
  auto describe_group = [&](const uint32_t g0) {
    const uint32_t ri = g0 + (lane >> 3), j = lane & 7u;
    const bool rv = ri < n_r;
    uint32_t run = 0;  // slots of the lists before list t
    uint32_t lbit = 0;
    uint64_t idx = null_idx;
    float w = 0.0f;
    for (uint32_t t = 0; t < T; t++) {
      const uint32_t src = (ri * T + t) & 63u;
      const uint32_t lo_t = bflat[src];
      const uint32_t c_t = bflat[(src + T) & 63u] - lo_t;  // postings of list t in the round
      const uint32_t m = rv ? (c_t + 63u) >> 6 : 0u;
      const bool mine = j >= run && j < run + m;
      const uint64_t base = list_off(t) + lo_t + (j - run) * 64u;
      const float wt = __int_as_float((int)rl((uint32_t)__float_as_int(my_w), t));
      lbit = mine ? 1u << t : lbit;
      idx = mine ? base : idx;
      w = mine ? wt : w;
      run += m;
    }
    uint32_t lbpack, lbpack_hi;
    pack_list_bits(lbit, j, lbpack, lbpack_hi);
    // meta: slot 0 = slots the round needs (saturated), slot 1 = the packed list bits
    sdesc[lane] = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), __float_as_uint(w),
                             j == 1u ? lbpack : (ML > 4 && j == 2u ? lbpack_hi : (run < 255u ? run : 255u)));
  };
  // ---- chunk of an over-full round: per-list ranges [lo, lo + cnt) held in lane t -> row 8;
  //      returns the slots in use.  lane j also keeps its slot's list (consumed counts) ----
  auto describe_chunk = [&](const uint32_t lo, const uint32_t cnt, uint32_t &slot_list_of_lane) {
    const uint32_t m = (cnt + 63u) >> 6;  // slots of my list
    uint32_t ss = 0, run = 0;             // ss: first slot of my list
    for (uint32_t t = 0; t < T; t++) {
      ss = lane == t ? run : ss;
      run += rl(m, t);
    }
    uint32_t tj = 0;  // lane j: the list that owns slot j = the last list whose first slot is <= j
    for (uint32_t t = 1; t < T; t++) tj = lane >= rl(ss, t) ? t : tj;
    const uint32_t l_ss = __shfl(ss, (int)tj, 64);
    const uint64_t l_abs = list_off(tj) + __shfl(lo, (int)tj, 64);
    const bool used = lane < run && lane < (uint32_t)NS;
    const uint64_t idx = used ? l_abs + (lane - l_ss) * 64u : null_idx;
    const float wt = __int_as_float((int)__shfl((int)__float_as_int(my_w), (int)tj, 64));
    uint32_t lbpack, lbpack_hi;
    pack_list_bits(lane < (uint32_t)NS && used ? 1u << tj : 0u, lane & 7u, lbpack, lbpack_hi);
    if (lane < (uint32_t)NS)
      sdesc[64 + lane] = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), used ? __float_as_uint(wt) : 0u,
                                    lane == 1u ? lbpack : (ML > 4 && lane == 2u ? lbpack_hi : (run < 255u ? run : 255u)));
    slot_list_of_lane = used ? tj : 0xFFu;
    return run;
  };

  // ---- issue the loads of descriptor row `row`: 8 whole slots, lanebase[idx] (no predicate, no
  //      scalar address arithmetic; the descriptor arrives by a broadcast LDS read) ----
  auto issue = [&](URound &r, const uint32_t row) {
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      const uint2 d = reinterpret_cast<const uint2 *>(sdesc)[(row * 8u + jj) * 2u];
      const uint64_t idx = ((uint64_t)d.y << 32) | d.x;
      r.doc[jj] = ldocs[idx];
      r.sc[jj] = limps[idx];
    }
  };
  // ---- dst = the loaded round src with the slot's list weight applied ----
  auto settle = [&](URound &dst, const URound &src, const uint32_t row) {
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      const uint32_t wj = reinterpret_cast<const uint32_t *>(sdesc)[(row * 8u + jj) * 4u + 2u];
      dst.doc[jj] = src.doc[jj];
      dst.sc[jj] = src.sc[jj] * __uint_as_float(wj);
    }
  };
  auto row_slots = [&](const uint32_t row) { return rfl(sdesc[row * 8u].w); };
  auto row_lbits = [&](const uint32_t row) {  // LB bits per slot
    const uint64_t hi = ML > 4 ? rfl(sdesc[row * 8u + 2u].w) : 0u;
    return (hi << 32) | rfl(sdesc[row * 8u + 1u].w);
  };

  // ---- candidates -> top-k (one take_checked site per source; BufTopK::compact is large) ----
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
  // the per-posting threshold test works on score BITS: for scores and thresholds > 0 the float
  // order is the unsigned integer order, so "score >= threshold" is a saturating subtract (plain
  // VALU, no compare into an SGPR mask).  thr_m1 = bits(threshold) - 1; 0xFFFFFFFF... a threshold
  // that is not positive (none yet / zero / negative weights in play) sends every round through the
  // exact candidate code instead (hot_all).
  uint32_t thr_m1 = 0;
  bool hot_all = true;
  auto refresh_threshold = [&]() {
    const float thf = threshold_score();
    hot_all = !(thf > 0.0f);
    thr_m1 = hot_all ? 0u : __float_as_uint(thf) - 1u;
  };
  refresh_threshold();

  // ---- score the postings of `e`: all postings with doc < end are this round's (or chunk's);
  //      the others (later postings of the same lists, sentinels) only ever add filter bits.
  //      lbits: the list bit of every slot, LB bits each (uniform) ----
  auto accumulate = [&](const URound &e, const uint32_t end, const uint64_t lbits) {
#ifdef SLG_U3_LOADS_ONLY  // diagnostic build: what the load stream alone costs (results are wrong)
    {
      uint32_t chk = 0;
#pragma unroll
      for (int jj = 0; jj < NS; jj++) chk ^= e.doc[jj] ^ __float_as_uint(e.sc[jj]);
      if (chk == 0x12345678u && end == 7u && lbits == 9ull) n_scored++;
      return;
    }
#endif
    SLG_STAMP(1);
    uint32_t x[NS];
    uint32_t accx = hot_all ? 1u : 0u;
    if (T > 1) {
      // P0: clear the filter
#pragma unroll
      for (int c = 0; c < FW / 256; c++) flt4[lane + 64 * c] = make_uint4(0u, 0u, 0u, 0u);
      wave_fence();
      // P1: one bit per posting: word = doc mod FW, field = (doc / FW) mod (32 / LB), bit = list
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        atomicOr(&flt[e.doc[jj] & (FW - 1)], slot_bit(lbits, jj) << ((e.doc[jj] >> FSH) & FSM));
      }
      wave_fence();
      SLG_STAMP(2);
      // P2: the lists that hold my doc (or an alias of it)
#pragma unroll
      for (int jj = 0; jj < NS; jj++) x[jj] = flt[e.doc[jj] & (FW - 1)];
      wave_fence();  // the queue overlays the filter: all reads are issued before its writes
      // P3: x != 0: another list's bit is set in my field (my own always is)
#pragma unroll
      for (int jj = 0; jj < NS; jj++)
        x[jj] = ((x[jj] >> ((e.doc[jj] >> FSH) & FSM)) & LBM) ^ slot_bit(lbits, jj);
    } else {
#pragma unroll
      for (int jj = 0; jj < NS; jj++) x[jj] = 0u;
    }
    // postings whose score may reach the threshold are queued too: a single is a doc without a
    // partner (the exact compare happens once, at the join's candidate site)
    if (!hot_all) {
#pragma unroll
      for (int jj = 0; jj < NS; jj++) x[jj] |= __builtin_elementwise_sub_sat(__float_as_uint(e.sc[jj]), thr_m1);
    }
#pragma unroll
    for (int jj = 0; jj < NS; jj++) accx |= x[jj];

    SLG_STAMP(3);
    uint32_t cnt[NS];  // queued postings per slot (uniform)
#pragma unroll
    for (int jj = 0; jj < NS; jj++) cnt[jj] = 0u;
    uint32_t n = 0;  // queued postings
    bool touched = false;
    if (__ballot(accx != 0u) != 0ull) {
      if (hot_all) {
        // no positive threshold yet (no seed; filtered query; candidates mode without a seed): every
        // single of the round is a candidate: score = 0.0 + w*impact (wand.rs:539).  One site; the
        // slot's registers are selected at run time
        touched = true;
#pragma unroll 1
        for (uint32_t it = 0; it < (uint32_t)NS; it++) {
          float xs = e.sc[0];
          uint32_t dc = e.doc[0], xf = x[0];
#pragma unroll
          for (int j = 1; j < NS; j++) {
            xs = it == (uint32_t)j ? e.sc[j] : xs;
            dc = it == (uint32_t)j ? e.doc[j] : dc;
            xf = it == (uint32_t)j ? x[j] : xf;
          }
          const bool single = xf == 0u && dc < end;
          if (__ballot(single) == 0ull) continue;
          take_checked(single, 0.0f + xs, dc);
        }
      }
      // shared docs (and aliases), and hot singles, of THIS round are queued in slot = list order.
      // All lane sets first (independent compares), then their scalar prefix sums, then the writes:
      // no per-slot compare -> branch -> count chain
      uint64_t qm[NS];
#pragma unroll
      for (int jj = 0; jj < NS; jj++) qm[jj] = __ballot(x[jj] != 0u && e.doc[jj] < end);
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        const uint32_t at = n + __builtin_amdgcn_mbcnt_hi((uint32_t)(qm[jj] >> 32),
                                                          __builtin_amdgcn_mbcnt_lo((uint32_t)qm[jj], 0u));
        if ((qm[jj] >> lane) & 1ull) queue[at] = make_uint2(e.doc[jj], __float_as_uint(e.sc[jj]));
        cnt[jj] = (uint32_t)__popcll(qm[jj]);
        n += cnt[jj];
      }
      // the all-pairs join reads the queue in groups of 8 entries: pad the last group with entries
      // no doc matches (what lies behind the queue is filter words, i.e. arbitrary bit patterns)
      if (lane >= n && lane < ((n + 7u) & ~7u) && n <= (uint32_t)kJoinPairs) queue[lane] = make_uint2(kDocEnd, 0u);
      wave_fence();
      n_scored -= n;
    }
    SLG_STAMP(4);
#ifdef SLG_STAMPS
    st_queued += n;
#endif
    // P4: join.  The queue is sorted by (list, doc): slots are in list order and a list's
    // postings in doc order.  A doc's sum is ((0.0 + x_a) + x_b) + ... over the lists that hold it,
    // in list order (= the reference's term order); its entry in the lowest list owns the result.
    if (n != 0u && n <= (uint32_t)kJoinPairs) {
      // few entries (the usual case): all pairs.  Sender l is read by a BROADCAST ds_read (uniform
      // address), so its doc and score arrive in VGPRs and a pair costs 7 plain VALU instructions:
      // same = (doc_l == my doc) as an all-ones mask; acc += same ? x_l : +0.0 (adding +0.0 is
      // exact: a sum that starts at +0.0 is never -0.0); first = the lowest l holding my doc (the
      // entry of the lowest list: it owns the result)
      const bool have = lane < n;
      const uint2 me = have ? queue[lane] : make_uint2(kDocEnd, 0u);
      float acc = 0.0f;
      uint32_t first = 64u;
#pragma unroll
      for (int g = 0; g < kJoinPairs; g += 8) {
        if ((uint32_t)g < n) {
#pragma unroll
          for (int l = g; l < g + 8; l++) {
            const uint2 sq = queue[l];
            const uint32_t diff = sq.x ^ me.x;
            const uint32_t nm = 0u - (diff < 1u ? diff : 1u);  // 0: same doc, ~0: another doc
            acc += __uint_as_float(sq.y & ~nm);
            const uint32_t cand = (uint32_t)l | nm;
            first = cand < first ? cand : first;
          }
        }
      }
      const uint64_t ownerm = __ballot(have && first == lane);
      n_scored += (uint32_t)__popcll(ownerm);
      if ((__ballot(acc >= threshold_score()) & ownerm) != 0ull) {
        touched = true;
        take_checked((ownerm >> lane) & 1ull, acc, me.x);
      }
    } else if (n != 0u) {
      // many entries (dense lists): binary search of my doc in the queue segment of every list,
      // in list order.  qe[u] = entries of the lists <= u (segment u = [qe[u-1], qe[u])).
      uint32_t qe[ML + 1];
      qe[0] = 0u;
#pragma unroll
      for (int u = 0; u < ML; u++) {
        uint32_t c = 0;
#pragma unroll
        for (int jj = 0; jj < NS; jj++) c += slot_bit(lbits, jj) <= (1u << u) ? cnt[jj] : 0u;  // (0: unused slot, cnt 0)
        qe[u + 1] = c;
      }
      uint32_t maxlen = 0;
#pragma unroll
      for (int u = 0; u < ML; u++) maxlen = qe[u + 1] - qe[u] > maxlen ? qe[u + 1] - qe[u] : maxlen;
      const uint32_t steps = maxlen ? 32u - (uint32_t)__builtin_clz(maxlen) : 0u;  // halvings that empty the longest
      for (uint32_t rb = 0; rb < n; rb += 64) {  // receivers in blocks of 64 lanes
        const uint32_t idx = rb + lane;
        const bool have = idx < n;
        const uint2 me = have ? queue[idx] : make_uint2(kDocEnd, 0u);
        uint32_t ml = 0;  // my list
#pragma unroll
        for (int u = 1; u < ML; u++) ml += idx >= qe[u] ? 1u : 0u;
        float acc = 0.0f;
        bool lower = false;
        // the searches in the lists' segments are independent: one LDS read of each per step, four
        // lists at a time (the sum stays in list order)
#pragma unroll
        for (int h = 0; h < ML; h += 4) {
          uint32_t lo[4], hi[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            lo[u] = qe[h + u];
            hi[u] = qe[h + u + 1];
          }
          for (uint32_t st = 0; st < steps; st++) {
            uint32_t mid[4], dk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) mid[u] = (lo[u] + hi[u]) >> 1;  // < qe[h + u + 1] while lo < hi
#pragma unroll
            for (int u = 0; u < 4; u++) dk[u] = queue[mid[u] < qe[h + u + 1] ? mid[u] : 0u].x;
            wave_fence();  // (all four reads are in flight before the first compare)
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const bool less = dk[u] < me.x;
              const bool open = lo[u] < hi[u];
              lo[u] = open && less ? mid[u] + 1u : lo[u];
              hi[u] = open && !less ? mid[u] : hi[u];
            }
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {  // the sum, in list order
            const uint2 kk = queue[lo[u] < qe[h + u + 1] ? lo[u] : 0u];  // (an empty segment reads entry 0: ignored)
            const bool mine = ml == (uint32_t)(h + u);
            const bool hit = have && (mine || (lo[u] < qe[h + u + 1] && kk.x == me.x));
            acc = hit ? acc + (mine ? __uint_as_float(me.y) : __uint_as_float(kk.y)) : acc;
            lower = lower || (hit && (uint32_t)(h + u) < ml);
          }
        }
        const uint64_t ownerm = __ballot(have && !lower);
        n_scored += (uint32_t)__popcll(ownerm);
        if ((__ballot(acc >= threshold_score()) & ownerm) != 0ull) {
          touched = true;
          take_checked((ownerm >> lane) & 1ull, acc, me.x);
        }
      }
    }
    wave_fence();
    if (touched) refresh_threshold();
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
  //      ONE accumulate site and ONE planned-issue site (code size / I-cache): iteration rr = -1
  //      only issues round 0 ----
  URound ew, en;
  uint32_t en_slots = 0;
  for (uint32_t rr = 0xFFFFFFFFu; rr == 0xFFFFFFFFu || rr < n_r; rr++) {
    const bool first = rr == 0xFFFFFFFFu;
    const bool big = !first && en_slots > (uint32_t)NS;
    const uint32_t row = rr & 7u;
    const uint32_t rend_r = first ? 0u : rfl(rend[rr]);
    uint64_t lbits = first ? 0ull : row_lbits(row);
    if (!first && !big) settle(ew, en, row);
#ifdef SLG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SLG_STAMP(7);
    if (first || rr + 1 < n_r) {  // prefetch the next round
      const uint32_t nx = rr + 1u, nrow = nx & 7u;
      if (nrow == 0) {
        wave_fence();  // (settle's reads of row 7 precede the rewrite)
        describe_group(nx);
        wave_fence();
      }
      en_slots = row_slots(nrow);
      if (en_slots <= (uint32_t)NS) issue(en, nrow);
    }
    SLG_STAMP(0);
    if (first) continue;
    uint32_t ocur = 0, oend = 0, end = rend_r;
    if (big) cuts(rr, ocur, oend);
    uint32_t guard = 0;
    do {
      if (big) {
        // next chunk of an over-full round: every list gets >= 1 slot, the rest in proportion to
        // what it has left; the chunk ends at the smallest "last loaded doc" of the lists that did
        // not finish, so all postings of a doc are scored in the same chunk
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
        uint32_t my_slot_list;
        wave_fence();
        describe_chunk(ocur, chunk, my_slot_list);
        wave_fence();
        issue(ew, 8);
        settle(ew, ew, 8);
        lbits = row_lbits(8);
        uint32_t bound = kDocEnd;
        for (uint32_t t = 0; t < T; t++) {
          const uint32_t ld = rl(lastdoc, t);
          bound = ld < bound ? ld : bound;
        }
        // (kDocEnd: nothing was cut, the chunk is the rest of the round)
        end = bound == kDocEnd ? rend_r : (bound + 1u < rend_r ? bound + 1u : rend_r);
        // what each list consumed: its postings with doc < end (a prefix of its slots)
        uint32_t consumed = 0;
#pragma unroll
        for (int jj = 0; jj < NS; jj++) {
          const uint32_t c = (uint32_t)__popcll(__ballot(ew.doc[jj] < end));
          consumed += lane == rl(my_slot_list, jj) ? c : 0u;
        }
        ocur += consumed;
      }
      accumulate(ew, end, lbits);
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
