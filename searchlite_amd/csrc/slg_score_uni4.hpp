// slg_score_uni4.hpp — the few-term scoring kernel (<= 8 lists per sub-query; BASELINE configs 2, 3
// and 5), fourth form: slg_score_uni3.hpp with the round's postings laid out BLOCKED over the wave.
//
// Same algorithm — exact pre-planned rounds, FILTER + JOIN accumulate, buffered top-k; restates
// query/wand.rs:459-566 (every posting scored, per-doc sums in ScorePlan leaf order,
// planner.rs:122-135) and push_top_k (wand.rs:905-916).  What changed is which posting a lane holds.
// The earlier forms gave every list whole 64-lane slots (register j of all lanes = 64 consecutive
// postings of ONE list): a list with 10 postings in the round still cost a slot, so a 5-list round
// of config 3 carried ~300 postings in its 512 lanes x registers, and the slot's list (posting
// index, weight, list bit) was a wave-uniform value that had to be fetched per slot.  Here lane l
// holds 8 CONSECUTIVE postings of one list in its 8 registers: the round's lists are laid end to
// end, each padded to a multiple of 8 postings, and lane l takes positions 8l .. 8l+7.  So
//   * a list wastes at most 7 lane-registers per round (mean 3.5) instead of half a slot (32): the
//     planner fills rounds to ~470 postings whatever the number of lists;
//   * list, weight, list bit and posting index are PER-LANE values, found once per round (8 byte
//     boundaries compared in two SIMD-in-register subtractions + one LDS table read), not per slot;
//   * a lane's postings are contiguous in memory: four 16-byte loads per round (doc ids, impacts)
//     instead of sixteen 4-byte ones — the wave still reads contiguous runs of 2 KB per list;
//   * the join queue must be in (list, doc) order = position order = lane-major: the queued entries
//     are placed by a wave prefix sum over the lanes' counts (no per-slot ballots).
// Posting arrays are padded per list (SegDev, kListPad) and end in kNullRun sentinels (idle lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_score.hpp"

namespace slg {

// per-wave LDS: [filter / join queue][top-k buffer][cut points (64 / 128 words), list offsets, weights, lengths]
//               [list table: 9 rows (8 planned rounds + the chunk row) x (ML + 1) entries of 16 B]
//               [row headers: 9 x 16 B][round end docs]
constexpr int kU4Rows = 9;
constexpr int u4_plan_off(int kregs, int fw) { return fw * 4 + (uni_buffered(kregs) ? buftopk_lds(kregs) : 0); }
constexpr int u4_cut_words(int ml) { return ml <= 4 ? 64 : 128; }  // a slice's cut points: (rounds + 1) * lists
constexpr int u4_plan_lds(int ml) { return u4_cut_words(ml) * 4 + 4 * ml * 4; }
constexpr int u4_tbl_off(int kregs, int ml, int fw) { return u4_plan_off(kregs, fw) + ((u4_plan_lds(ml) + 15) & ~15); }
constexpr int u4_tbl_bytes(int ml) { return kU4Rows * (ml + 1) * 16 + kU4Rows * 16; }
constexpr int u4_end_bytes() { return ((kMaxRoundsPerSlice + 1) * 4 + 15) & ~15; }
constexpr int u4_wave_lds(int kregs, int ml, int fw) { return u4_tbl_off(kregs, ml, fw) + u4_tbl_bytes(ml) + u4_end_bytes(); }

#ifndef SLG_U4_WAVES
#define SLG_U4_WAVES 6
#endif
#ifndef SLG_U4_WAVES8
#define SLG_U4_WAVES8 5
#endif
#ifndef SLG_U4_FW8
#define SLG_U4_FW8 1024  // filter words of the 5..8-list form (2048 at 4 waves: 6.37 vs 5.68 ms on config 3)
#endif
#ifndef SLG_U4_TWO_PHASE_MIN
#define SLG_U4_TWO_PHASE_MIN 9  // slices of this many rounds cut their inner boundaries by interpolation
#endif
#ifndef SLG_U4_JOIN_PAIRS
#define SLG_U4_JOIN_PAIRS 64
#endif
constexpr int kU4JoinPairs = SLG_U4_JOIN_PAIRS;  // queues up to this many entries are joined all-pairs (<= 64: one lane per entry)
constexpr int u4_filter_words(int ml) { return ml <= 4 ? kJoinWords : SLG_U4_FW8; }
// (k 129..256: LDS; the plan instantiation's leaf close needs a few registers more than 6 waves leave)
constexpr int u4_waves(int kregs, int ml, bool plan = false, bool persist = false) {
  return (ml > 4 || plan || persist) ? SLG_U4_WAVES8 : (kregs >= 4 ? 5 : SLG_U4_WAVES);  // (k > 256: the direct candidates of singles)
}
#ifndef SLG_U4_WPB
#define SLG_U4_WPB 1  // waves per workgroup of the persistent launch (waves never synchronise with each other)
#endif
constexpr int kU4WavesPerBlock = SLG_U4_WPB;
// persistent launch: workgroups to start so that every wave slot the kernel can occupy holds one wave
// (n_cu compute units x 4 SIMDs x u4_waves), or one wave per slice if the batch has fewer
inline uint32_t u4_launch_blocks(int kregs, int ml, bool plan, uint32_t n_slices, uint32_t n_cu, uint32_t waves_per_simd = 0) {
  const uint32_t fit = (uint32_t)u4_waves(kregs, ml, plan, true);
  const uint32_t slots = n_cu * 4u * (waves_per_simd != 0u && waves_per_simd < fit ? waves_per_simd : fit);
  const uint32_t waves = n_slices < slots ? n_slices : slots;
  return (waves + (uint32_t)kU4WavesPerBlock - 1u) / (uint32_t)kU4WavesPerBlock;
}

// 16-byte loads at 4-byte alignment (a lane's 8 postings start at any posting)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef u32x4_t U4x4 __attribute__((aligned(4)));

// *src read through the constant address space (scalar loads when the address is wave-uniform): for
// data written before the kernel starts and never during it
template <typename T>
__device__ __forceinline__ T load_const(const T *src) {
  static_assert(sizeof(T) % 4 == 0, "whole words");
  typedef const __attribute__((address_space(4))) uint32_t *c_u32_t;
  const c_u32_t w = (c_u32_t)(uintptr_t)src;
  T out;
  uint32_t *dst = reinterpret_cast<uint32_t *>(&out);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) dst[i] = w[i];
  return out;
}

// PLAN: the batch has score plans (query/planner.rs:113-153, flat: Sum or DisMax over leaves that sum
// one or more terms each): the lists arrive sorted by leaf and the join closes a doc's leaves in leaf
// order.  Its own instantiation: the flat-sum batches (BASELINE configs 2, 3, 5) keep their registers.
// PERSIST: the persistent-waves form (its own instantiation: the one-wave-per-slice kernel keeps the
// straight-line code and register allocation it had before the slice loop existed — with the loop
// compiled in, the default launch measured 5 % slower, 0.0817 against 0.0778 ms on config 2).
template <int KREGS, int ML, bool PLAN = false, bool PERSIST = false>
__global__ void __launch_bounds__(64 * kU4WavesPerBlock)
    __attribute__((amdgpu_waves_per_eu(u4_waves(KREGS, ML, PLAN, PERSIST), u4_waves(KREGS, ML, PLAN, PERSIST))))
score_uniform4_kernel(RoundScoreParams p_arg) {
  constexpr int NS = kUniSlots;            // postings per lane and round
  constexpr int FW = u4_filter_words(ML);  // filter words
  constexpr int TE = ML + 1;               // table entries per row: the lists + the idle-lane entry
  constexpr uint32_t BW = u4_cut_words(ML);  // words of the slice's cut-point row
  constexpr uint32_t LB = ML <= 4 ? 4u : 8u;  // list bits per filter field
  constexpr uint32_t LBM = (1u << LB) - 1u;
  // field of a doc: word = doc mod FW, shift = LB * ((doc / FW) mod (32 / LB))
  constexpr uint32_t FSH = (FW == 1024 ? 10u : 11u) - (LB == 4u ? 2u : 3u);
  constexpr uint32_t FSM = LB == 4u ? 0x1Cu : 0x18u;
  static_assert(FW == 1024 || FW == 2048, "filter size");
  static_assert(FW * 4 >= kUniCap * 8, "the join queue ({doc, score} per posting) overlays the filter");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_wg[];
  (void)p_arg;
  // waves are independent (no workgroup barrier anywhere): each wave of the workgroup owns its own
  // LDS region
  const uint32_t wave_in_wg = kU4WavesPerBlock > 1 ? rfl(threadIdx.x >> 6) : 0u;  // (uniform: a scalar register)
  unsigned char *const smem = smem_wg + wave_in_wg * (uint32_t)u4_wave_lds(KREGS, ML, FW);
  // Two launch forms (RoundScoreParams::work_ctr):
  //  * one wave per slice (work_ctr == nullptr; the default): wave w of the grid runs launch position w,
  //    the longest slices first; the hardware dispatcher is the work queue.
  //  * PERSISTENT WAVES (slg_tuning.score_waves_per_simd): the launch holds that many waves per SIMD and
  //    every wave pulls launch positions from the batch's work queues (slg_desc.hpp: kWorkQueues) until
  //    they are empty.  Built because the slice timelines show the wave slots only 63 % / 72 % occupied
  //    (configs 2 / 3) and measured: it does not pay on MI355X.  One counter hands out a ticket every
  //    ~11-16 ns (config 2: 0.198 ms); 64 counters with an all-counter scan at the end 0.109 ms (every
  //    wave reads every counter: same-address reads are served one at a time too); 64 counters with
  //    bounded stealing 0.0846 ms at 6 waves per SIMD, 0.0876 at 5, 0.0945 at 4 — against 0.0776 ms for
  //    the dispatcher on the same box; config 3: 5.23 against 5.06 ms.  Instruction counts and HBM
  //    fetch are identical (profiles/r04_persistent_waves.txt): what the dispatcher does for free — a
  //    new wave the moment a slot and its LDS are free, no ticket latency in the slice's start-up chain,
  //    no end-game — costs more as software than the occupancy it recovers.
  //    Every wave reaches the exit: the grid always drains.
  const uint32_t wave_id = blockIdx.x * (uint32_t)kU4WavesPerBlock + wave_in_wg;
  uint32_t wq = wave_id % kWorkQueues;  // the queue this wave pulls from
  // (in the persistent form EVERY slice is pulled, the first one too: a wave whose workgroup is not
  //  resident when the launch starts must not own a slice that then waits for a wave slot until the
  //  other waves have drained the queues)
  for (uint32_t widx = wave_id, done_slices = 0;; done_slices++) {
  // (per slice, the launch parameters are read again from the kernel-argument segment: what a slice
  //  derives from them then lives in registers for that slice only, instead of being hoisted out of
  //  this loop and held — spilled — across it)
  typedef const __attribute__((address_space(4))) RoundScoreParams *kparams_t;
  kparams_t pk = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
  if constexpr (PERSIST) asm volatile("" : "+s"(pk));
  const __attribute__((address_space(4))) RoundScoreParams &p = *pk;
  const uint32_t lane = threadIdx.x & 63;
  if constexpr (!PERSIST) {
    if (done_slices != 0u) break;  // one wave per slice: the loop is gone at compile time
  } else
  {
    // A queue that has run dry is left for another one at most kWorkSteals times, then the wave exits.
    // (Looking at ALL counters to find the queues that still hold work was measured and is a trap: at
    //  the end of the launch every wave reads every counter, 6144 reads of each of the 64 lines, and
    //  reads of one address are served one at a time like the atomics — the tail grew by ~30 us.)
    const uint32_t n_sl = p.n_slices;
    for (uint32_t tries = 0;; tries++) {
      uint32_t c = 0;
      if (lane == 0) c = atomicAdd(p.work_ctr + wq * kWorkCtrStride, 1u);
      widx = wq + kWorkQueues * rfl(c);
      if (widx < n_sl || tries >= kWorkSteals) break;
      wq = (wq + 21u) % kWorkQueues;
    }
  }
  if (widx >= p.n_slices) break;
  // The slice record, the segment descriptor and the reject-table row are written before this kernel
  // starts and never during it: they are read through the CONSTANT address space, i.e. by scalar
  // loads (a few hundred cycles from the scalar cache).  Inside this loop the compiler cannot prove
  // that for a plain global pointer — earlier slices stored to global memory — and would use vector
  // loads, which put two more HBM-latency steps into every slice's dependent start-up chain
  // (record -> segment -> term references -> cut-point searches -> first round): measured on config
  // 2, whose slices are 4 rounds long, 0.111 ms against 0.078 ms for the one-wave-per-slice launch.
  typedef const __attribute__((address_space(4))) uint64_t *c_u64_t;
  const SliceDesc sl = load_const(p.slice_desc + widx);
  const uint32_t slice = rfl(sl.slice);

  constexpr bool BUF = uni_buffered(KREGS);
  uint32_t *flt = reinterpret_cast<uint32_t *>(smem);
  uint4 *flt4 = reinterpret_cast<uint4 *>(smem);
  uint2 *queue = reinterpret_cast<uint2 *>(smem);  // {doc, score} of queued postings; overlays flt
  uint4 *const tbl = reinterpret_cast<uint4 *>(smem + u4_tbl_off(KREGS, ML, FW));  // {idx lo, idx hi, weight, list bit}
  uint4 *const hdr = tbl + kU4Rows * TE;  // {first lanes of lists 1..4, of lists 5..8 (bytes), lanes in use, -}
  uint32_t *const rend = reinterpret_cast<uint32_t *>(smem + u4_tbl_off(KREGS, ML, FW) + u4_tbl_bytes(ML));

  const uint32_t T = rfl(sl.n_terms);
  const uint32_t n_r = rfl(sl.n_rounds);
  // (inside the slice loop the compiler cannot prove these loads invariant — earlier slices stored to
  //  global memory — so they are vector loads: everything wave-uniform is moved to scalar registers by hand)
  auto uni64 = [](const uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v); };
  const uint32_t seg_id = rfl(sl.seg);
  const SegDev sd_v = load_const(p.segs + seg_id);
  const gu32_t gdocs = (gu32_t)uni64((uint64_t)sd_v.docs);
  const gf32_t gimps = (gf32_t)uni64((uint64_t)sd_v.imps);
  const uint32_t seg_n_docs = rfl(sd_v.n_docs);
  // per-lane bases: lane l's postings are lanebase[idx .. idx + 7]
  const gu32_t ldocs = gdocs + 8u * lane;
  const gf32_t limps = gimps + 8u * lane;
  const uint64_t null_idx = uni64(sd_v.null_idx);
  const uint32_t fid = rfl(sl.filter);
  const gu32_t gdel = (gu32_t)uni64(fid ? ((c_u64_t)(uintptr_t)p.reject_table)[(size_t)(fid - 1) * p.n_segs + seg_id]
                                        : (uint64_t)sd_v.deleted);
  const uint32_t k = p.k;

  // lane t < T: list t's weight and posting offset; all cut points of the slice (entry r*T + t:
  // where round r starts in list t); the end doc of every round
  uint32_t *const bflat = reinterpret_cast<uint32_t *>(smem + u4_plan_off(KREGS, FW));
  uint32_t *const off_lo = bflat + BW, *const off_hi = off_lo + ML, *const wts = off_hi + ML, *const dfs = wts + ML;
  const bool inline_cuts = p.bounds == nullptr;
  TermRef tr{};
  if (lane < T) tr = p.terms[rfl(sl.term_begin) + lane];
  // score plan (PLAN instantiation): 0 = flat sum in term order; 1 = Sum of leaves; 2 = DisMax of leaves.
  // leaf_start: bit t = list t is the first list of its leaf (the lists are sorted by leaf)
  // min_match (bits 8.. of the plan word; RoundQuery::plan): a doc counts only if at least that many LEAVES
  // hold it — the query-string matcher's minimum_should_match over its term groups (api/reader.rs:1509-1517:
  // matched_terms >= required).  0 / 1: any doc of any list.
  const uint32_t plan_word = PLAN ? rfl(sl.plan) : 0u;
  const uint32_t plan = plan_word & 0xFFu;
  const uint32_t min_match = PLAN ? plan_word >> 8 : 0u;
  const bool need_many = PLAN && min_match > 1u;  // a doc found in ONE list is never accepted
  const float plan_tie = __uint_as_float(rfl(__float_as_uint(sl.tie)));
  const float plan_max0 = __uint_as_float(rfl(__float_as_uint(sl.max_init)));
  uint32_t leaf_start = 1u;
  if (PLAN) {
    const uint32_t prev_leaf = (uint32_t)wave_shr1((int32_t)tr.leaf);
    leaf_start = (uint32_t)__ballot(lane < T && (lane == 0u || tr.leaf != prev_leaf));
  }
  const uint32_t plan_leaves = (uint32_t)__popc(leaf_start);  // leaves with a list in this sub-query
  // The wave cuts its own slice (what partition_rounds_kernel does for every boundary of the batch,
  // restricted to this slice's (rounds + 1) x lists boundaries): boundary j of the sub-query is the
  // doc id at position j * stride of its longest list, every other list is cut at its first
  // posting with doc >= that id.  The lines the searches fetch are the lines the rounds below
  // load: on config 3 the separate kernel read every list but the longest a second time at
  // scattered-access efficiency (1.6 ms per batch).
  const uint32_t lg = rfl(sl.longest), sq_rounds = rfl(sl.sq_rounds), r0 = rfl(sl.first_round);
  const uint64_t l_off = ((uint64_t)rfl((uint32_t)(sl.l_off >> 32)) << 32) | rfl((uint32_t)sl.l_off);
  const uint32_t l_df = rfl(sl.l_df);
  const uint32_t stride = inline_cuts ? (l_df + sq_rounds - 1u) / sq_rounds : 0u;
  const float inv_t = 1.0f / (float)T;
  // (the boundary docs are loaded straight off the slice record, beside the TermRef loads)
  constexpr int NTASK = (int)(BW / 64u);
  uint32_t tgt[NTASK];
  if (inline_cuts) {
#pragma unroll
    for (int u = 0; u < NTASK; u++) {
      const uint32_t task = lane + 64u * u;
      const uint32_t i = (uint32_t)(((float)task + 0.5f) * inv_t);  // (exact: task < 128, T <= 8)
      const uint64_t pos_l = (uint64_t)(r0 + i) * stride;
      const bool mid = task < (n_r + 1u) * T && r0 + i != 0u && r0 + i < sq_rounds && pos_l < l_df;
      tgt[u] = mid ? gdocs[l_off + pos_l] : 0u;
    }
  }
  if (lane < T) {
    wts[lane] = __float_as_uint(tr.weight);
    off_lo[lane] = (uint32_t)tr.off;
    off_hi[lane] = (uint32_t)(tr.off >> 32);
    dfs[lane] = tr.df;
  }
  if (!inline_cuts) {  // cut points from partition_rounds_kernel
#pragma unroll
    for (uint32_t i = 0; i < BW; i += 64) bflat[i + lane] = i + lane < (n_r + 1) * T ? p.bounds[rfl(sl.bounds_off) + i + lane] : 0u;
    if (lane < n_r) rend[lane] = p.rdoc[rfl(sl.rdoc_off) + lane + 1];
  } else {
    wave_fence();
    // the boundaries' docs: rend[i - 1] = end doc of round i - 1 (sentinels are 0xFFFFFFFF: never below
    // kDocEnd); rend[kMaxRoundsPerSlice + 1] = the doc the slice starts at
    uint32_t *const row0_doc = rend + kMaxRoundsPerSlice + 1;
#pragma unroll
    for (int u = 0; u < NTASK; u++) {
      const uint32_t task = lane + 64u * u;
      const uint32_t i = (uint32_t)(((float)task + 0.5f) * inv_t);
      const uint64_t pos_l = (uint64_t)(r0 + i) * stride;
      const bool last_b = r0 + i >= sq_rounds || pos_l >= l_df;
      if (task < (n_r + 1u) * T && task == i * T) {
        if (i >= 1u)
          rend[i - 1u] = last_b ? kDocEnd : tgt[u];
        else
          *row0_doc = tgt[u];  // (0 for the sub-query's first boundary)
      }
    }
    // one boundary of one list: first posting with doc >= the boundary's doc
    auto cut_one = [&](const uint32_t i, const uint32_t t, const uint32_t target, const bool inner) {
      const uint32_t j = r0 + i;
      const uint64_t pos_l = (uint64_t)j * stride;
      const bool first_b = j == 0u, last_b = j >= sq_rounds || pos_l >= l_df;
      const uint32_t df_t = dfs[t];
      if (first_b) return 0u;
      if (last_b) return df_t;
      if (t == lg) return (uint32_t)pos_l;
      const gu32_t d = gdocs + (((uint64_t)off_hi[t] << 32) | off_lo[t]);
      if (!inner) return lower_bound_guess(d, df_t, target, seg_n_docs);
      // between the slice's own first and last cut points the postings are spread evenly enough for a
      // 64-posting window around the interpolated position (half the lines of the global guess's
      // window; a miss bisects between the two known cuts)
      uint32_t lo = bflat[t], hi = bflat[n_r * T + t];
      const uint32_t d0 = *row0_doc, d1e = rend[n_r - 1u];
      const uint32_t d1 = d1e < seg_n_docs ? d1e : seg_n_docs;
      const float fr = d1 > d0 ? (float)(target - d0) / (float)(d1 - d0) : 0.0f;
      uint32_t g = lo + (uint32_t)((float)(hi - lo) * fr);
      g = g < hi ? g : hi;
      uint32_t pos;
      if (lower_bound_window<4>(d, df_t, target, g > lo + 32u ? g - 32u : lo, pos)) return pos;
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (d[mid] < target)
          lo = mid + 1;
        else
          hi = mid;
      }
      return lo;
    };
    const bool two_phase = n_r >= (uint32_t)SLG_U4_TWO_PHASE_MIN;
    if (two_phase) {
      wave_fence();
      if (lane < 2u * T) {  // the slice's first and last boundary, every list
        const uint32_t t = lane < T ? lane : lane - T, i = lane < T ? 0u : n_r;
        const uint32_t target = i == 0u ? *row0_doc : rend[n_r - 1u];
        bflat[i * T + t] = cut_one(i, t, target, false);
      }
      wave_fence();
    }
#pragma unroll
    for (int u = 0; u < NTASK; u++) {
      const uint32_t task = lane + 64u * u;
      if (task >= (n_r + 1u) * T) continue;
      const uint32_t i = (uint32_t)(((float)task + 0.5f) * inv_t);
      const uint32_t t = task - i * T;
      if (two_phase && (i == 0u || i == n_r)) continue;
      bflat[task] = cut_one(i, t, tgt[u], two_phase);
    }
  }
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

  struct BRound {
    uint32_t doc[NS];  // lane l: 8 consecutive postings of ONE list
    float sc[NS];      // in flight: impact; settled: weight * impact (score_tf, wand.rs:285)
    uint32_t w, lb;    // the lane's list: weight bits, list bit (0: idle lane)
  };

  // ---- list table of one round.  A list with c postings in the round takes m = ceil(c / 8) lanes
  //      right after the lists before it; entry t = {A, weight, list bit} with A = the posting index
  //      lane 0 WOULD read (list offset + cut point - 8 * first lane), so lane l reads A + 8 l.
  //      Entry T serves the idle lanes behind the last list: sentinels at null_idx.  The header holds
  //      the first lane of lists 1..8 as bytes (127: no such list) and the lanes in use. ----
  auto boundary_byte = [&](const uint32_t u, const uint32_t first, const uint32_t used) {
    // the byte list u >= 1 contributes: its first lane; the idle entry (u == T) starts at `used`
    // (127 - x with x selected against 0: no constant has to live in a register)
    const uint32_t f = first < 127u ? first : 127u;
    return 127u - (u < T ? 127u - f : (u == T ? 127u - used : 0u));
  };
  // ---- 8 consecutive planned rounds at once: lane 8 i + t = list t of round g0 + i -> row i ----
  // (rare paths take their own copy of the lane id through an empty asm: values derived from it are
  //  then computed where they are used instead of being hoisted into registers held over the loop)
  auto fresh_lane = [&]() {
    uint32_t l = lane;
    asm volatile("" : "+v"(l));
    return l;
  };
  // (__shfl* derive the lane id again and the compiler keeps that copy in a register over the loop)
  auto from_lane = [&](const uint32_t v, const uint32_t src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
  };
  auto describe_group = [&](const uint32_t g0) {
    const uint32_t lane = fresh_lane();
    const uint32_t i = lane >> 3, t = lane & 7u, ri = g0 + i;
    const bool rv = ri < n_r && t < T;
    const uint32_t src = (ri * T + t) & (BW - 1u);
    const uint32_t lo_t = bflat[src];
    const uint32_t c = rv ? bflat[(src + T) & (BW - 1u)] - lo_t : 0u;  // postings of list t in the round
    const uint32_t m = (c + 7u) >> 3;
    uint32_t incl = m;  // prefix sum over the round's 8 lanes
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      const uint32_t v = from_lane(incl, (lane - (uint32_t)d) & 63u);
      incl += t >= (uint32_t)d ? v : 0u;
    }
    const uint32_t first = incl - m;
    const uint32_t total = from_lane(incl, lane | 7u);
    const uint32_t used = total < 64u ? total : 64u;
    if (t < T) {
      const uint64_t a = list_off(t) + lo_t - 8ull * first;
      tbl[i * TE + t] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), wts[t], 1u << t);
    }
    if (t == 0u) {
      const uint64_t a = null_idx - 8ull * used;
      tbl[i * TE + T] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), 0u, 0u);
    }
    const uint32_t u = t == 0u ? 8u : t;  // lane t speaks for list t; lane 0 for list 8
    const uint32_t val = boundary_byte(u, first, used);
    uint32_t bl = u <= 4u ? val << (8u * (u - 1u)) : 0u;
    uint32_t bh = u > 4u ? val << (8u * (u - 5u)) : 0u;
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
      bl |= from_lane(bl, lane ^ (uint32_t)d);
      bh |= from_lane(bh, lane ^ (uint32_t)d);
    }
    if (t == 0u) hdr[i] = make_uint4(bl, bh, total < 255u ? total : 255u, 0u);
  };
  // ---- chunk of an over-full round: per-list ranges [lo, lo + cnt) held in lane t -> row 8;
  //      lane t gets its list's lanes [first, first + m) ----
  auto describe_chunk = [&](const uint32_t lo, const uint32_t cnt, uint32_t &first, uint32_t &m) {
    const uint32_t lane = fresh_lane();
    m = lane < T ? (cnt + 7u) >> 3 : 0u;
    const uint32_t incl = wave_incl_scan(m);
    first = incl - m;
    const uint32_t total = rl(incl, 63);  // <= 64 (the chunk was sized for it)
    if (lane < T) {
      const uint64_t a = list_off(lane) + lo - 8ull * first;
      tbl[8 * TE + lane] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), wts[lane], 1u << lane);
    }
    if (lane == 0u) {
      const uint64_t a = null_idx - 8ull * total;
      tbl[8 * TE + T] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), 0u, 0u);
    }
    const uint32_t val = lane >= 1u && lane <= 8u ? boundary_byte(lane, first, total) : 0u;
    uint32_t bl = lane >= 1u && lane <= 4u ? val << (8u * (lane - 1u)) : 0u;
    uint32_t bh = lane >= 5u && lane <= 8u ? val << (8u * (lane - 5u)) : 0u;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
      bl |= from_lane(bl, lane ^ (uint32_t)d);
      bh |= from_lane(bh, lane ^ (uint32_t)d);
    }
    if (lane == 0u) hdr[8] = make_uint4(bl, bh, total, 0u);
  };

  // ---- issue the loads of row `row`: my list = the number of boundary bytes <= my lane (two
  //      SIMD-in-register subtractions: byte = 0x80 + lane - boundary keeps its top bit iff
  //      boundary <= lane; boundaries are <= 127, so no byte borrows), then its table entry ----
  const uint32_t lanev = 0x80808080u | (lane * 0x01010101u);
  auto issue = [&](BRound &r, const uint32_t row) {
    const uint4 h = hdr[row];  // (uniform address: a broadcast read)
    uint32_t t = (uint32_t)__popc((lanev - h.x) & 0x80808080u);
    if constexpr (ML > 4) t += (uint32_t)__popc((lanev - h.y) & 0x80808080u);
    const uint4 en = tbl[row * TE + t];
    const uint64_t idx = ((uint64_t)en.y << 32) | en.x;
    typedef const __attribute__((address_space(1))) U4x4 *gq_t;
    const gq_t pd = (gq_t)(ldocs + idx);
    const gq_t pi = (gq_t)(limps + idx);
    const U4x4 d0 = pd[0], d1 = pd[1], i0 = pi[0], i1 = pi[1];
    r.doc[0] = d0.x, r.doc[1] = d0.y, r.doc[2] = d0.z, r.doc[3] = d0.w;
    r.doc[4] = d1.x, r.doc[5] = d1.y, r.doc[6] = d1.z, r.doc[7] = d1.w;
    r.sc[0] = __uint_as_float(i0.x), r.sc[1] = __uint_as_float(i0.y), r.sc[2] = __uint_as_float(i0.z),
    r.sc[3] = __uint_as_float(i0.w), r.sc[4] = __uint_as_float(i1.x), r.sc[5] = __uint_as_float(i1.y),
    r.sc[6] = __uint_as_float(i1.z), r.sc[7] = __uint_as_float(i1.w);
    r.w = en.z;
    r.lb = en.w;
  };
  // ---- dst = the loaded round src with the lane's list weight applied ----
  auto settle = [&](BRound &dst, const BRound &src) {
    const float w = __uint_as_float(src.w);
#pragma unroll
    for (int jj = 0; jj < NS; jj++) {
      dst.doc[jj] = src.doc[jj];
      dst.sc[jj] = src.sc[jj] * w;
    }
    dst.lb = src.lb;
  };
  auto row_lanes = [&](const uint32_t row) { return rfl(hdr[row].z); };

  // ---- candidates -> top-k (one take_checked site per source; BufTopK::compact is large) ----
  auto threshold_score = [&]() {  // score part of the current threshold as a float (-inf: none)
    // (uniform; the empty asm keeps the compare a 32-bit scalar one: folded into a 64-bit compare of
    //  the whole threshold it runs on the vector unit against constants held in registers)
    uint32_t hi = rfl((uint32_t)(btop.th >> 32));
    asm volatile("" : "+s"(hi));
    const uint32_t bits = hi < 0x00800000u ? 0xFF800000u : __float_as_uint(key_to_float((int32_t)(hi ^ 0x80000000u)));
    return __uint_as_float(rfl(bits));
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
  // the per-posting threshold test works on score BITS (slg_score_uni3.hpp): thr_m1 = bits(threshold)
  // - 1; a threshold that is not positive sends every round through the exact candidate code (hot_all)
  uint32_t thr_m1 = 0;
  bool hot_all = true;
  auto refresh_threshold = [&]() {
    const float thf = threshold_score();
    hot_all = !(thf > 0.0f);
    thr_m1 = rfl(hot_all ? 0u : __float_as_uint(thf) - 1u);
  };
  refresh_threshold();

  // ---- score the postings of `e`: all postings with doc < end are this round's (or chunk's);
  //      the others (later postings of the same lists, sentinels) only ever add filter bits.
  //      bnd_*_v: the round's list boundaries (header words, uniform; for the dense join) ----
  // score of a doc found in ONE list (w*impact = xs): its only leaf
  auto single_score = [&](const float xs) {
    float one = 0.0f + xs;  // Sum: -0.0 + one + 0.0 ... = one
    if (PLAN && plan == 2u) {
      // DisMax of a doc found in one list: that leaf = one, every other leaf of the plan = 0.0
      // (planner.rs:138-150): max over all of them, sum = one
      float m = fmaxf(plan_max0, one);
      if (plan_leaves > 1u) m = fmaxf(m, 0.0f);
      one = m + plan_tie * (one - m);
    }
    return one;
  };
  auto accumulate = [&](const BRound &e, const uint32_t end, const uint32_t bnd_lo_v, const uint32_t bnd_hi_v) {
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
      for (int jj = 0; jj < NS; jj++)
        atomicOr(&flt[e.doc[jj] & (FW - 1)], e.lb << ((e.doc[jj] >> FSH) & FSM));
      wave_fence();
      SLG_STAMP(2);
      // P2: the lists that hold my doc (or an alias of it)
#pragma unroll
      for (int jj = 0; jj < NS; jj++) x[jj] = flt[e.doc[jj] & (FW - 1)];
      wave_fence();  // the queue overlays the filter: all reads are issued before its writes
      // P3: x != 0: another list's bit is set in my field (my own always is)
#pragma unroll
      for (int jj = 0; jj < NS; jj++) x[jj] = ((x[jj] >> ((e.doc[jj] >> FSH) & FSM)) & LBM) ^ e.lb;
    } else {
#pragma unroll
      for (int jj = 0; jj < NS; jj++) x[jj] = 0u;
    }
    // postings whose score may reach the threshold are queued too: a single is a doc without a
    // partner (the exact compare happens once, at the join's candidate site)
    if (!hot_all && !need_many) {
      if constexpr (BUF) {
#pragma unroll
        for (int jj = 0; jj < NS; jj++) x[jj] |= __builtin_elementwise_sub_sat(__float_as_uint(e.sc[jj]), thr_m1);
      } else {
        // CANDIDATES mode (k > 256; BASELINE config 5: k = 1001): the threshold is the planner's seed and
        // stays there — a bound on the 1024th best impact of a list, which a tenth of all postings reach.
        // Queued, they made the join three times as long as config 2's (~65 entries per round: the dense
        // path); a posting that is ALONE in its filter field needs no join: it is this round's candidate
        // as it stands (score = 0.0 + w*impact, wand.rs:539).  Only shared docs and aliases are queued.
        // (placed like the queue entries: the lanes' counts are prefix-summed over the wave)
        uint32_t hcnt = 0;
#pragma unroll
        for (int jj = 0; jj < NS; jj++)
          hcnt += (x[jj] == 0u && __float_as_uint(e.sc[jj]) > thr_m1 && e.doc[jj] < end) ? 1u : 0u;
        if (__ballot(hcnt != 0u) != 0ull) {
          const uint32_t hincl = wave_incl_scan(hcnt);
          uint32_t at = ccur + hincl - hcnt;
#pragma unroll
          for (int jj = 0; jj < NS; jj++) {
            const uint32_t ok = ordered_score(single_score(e.sc[jj]));
            const bool h = x[jj] == 0u && __float_as_uint(e.sc[jj]) > thr_m1 && e.doc[jj] < end;
            if (h) creg[at] = make_uint2(btop.passes(ok, ~e.doc[jj]) ? ok : 0u, btop.passes(ok, ~e.doc[jj]) ? e.doc[jj] : 0xFFFFFFFFu);
            at += h ? 1u : 0u;
          }
          ccur += rl(hincl, 63);
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < NS; jj++) accx |= x[jj];

    SLG_STAMP(3);
    uint32_t n = 0;  // queued postings
    uint32_t qincl = 0;  // inclusive prefix sum of the lanes' queued postings
    bool touched = false;
    if (__ballot(accx != 0u) != 0ull) {
      if (hot_all && !need_many) {
        // no positive threshold yet (no seed; filtered query; candidates mode without a seed): every
        // single of the round is a candidate: score = 0.0 + w*impact (wand.rs:539).  One site; the
        // register is selected at run time
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
          take_checked(single, single_score(xs), dc);
        }
      }
      // shared docs (and aliases), and hot singles, of THIS round are queued in position order =
      // (list, doc) order: lane-major.  qf = 1 for a queued posting (plain VALU: x != 0 and
      // doc < end as saturating subtracts), the lanes' counts are prefix-summed over the wave
      uint32_t qf[NS], cnt = 0;
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        const uint32_t below = __builtin_elementwise_sub_sat(end, e.doc[jj]);  // != 0: doc < end
        const uint32_t both = x[jj] < below ? x[jj] : below;                    // != 0: both
        qf[jj] = both < 1u ? both : 1u;
        cnt += qf[jj];
      }
      qincl = wave_incl_scan(cnt);
      n = rl(qincl, 63);
      uint32_t at = qincl - cnt;
#pragma unroll
      for (int jj = 0; jj < NS; jj++) {
        if (qf[jj] != 0u) queue[at] = make_uint2(e.doc[jj], __float_as_uint(e.sc[jj]));
        at += qf[jj];
      }
      // the all-pairs join reads the queue in groups of 8 entries: pad the last group with entries
      // no doc matches (what lies behind the queue is filter words, i.e. arbitrary bit patterns)
      if (lane >= n && lane < ((n + 7u) & ~7u) && n <= (uint32_t)kU4JoinPairs) queue[lane] = make_uint2(kDocEnd, 0u);
      wave_fence();
      n_scored -= n;
    }
    SLG_STAMP(4);
#ifdef SLG_STAMPS
    st_queued += n;
#endif
    // P4: join.  The queue is sorted by (list, doc).  A doc's sum is ((0.0 + x_a) + x_b) + ... over
    // the lists that hold it, in list order (= the reference's term order); its entry in the lowest
    // list owns the result.
    if (n != 0u && n <= (uint32_t)kU4JoinPairs) {
      // up to a wave of entries (the usual case): all pairs.  Sender l is read by a BROADCAST ds_read
      // (uniform address); same = (doc_l == my doc) as an all-ones mask; acc += same ? x_l : +0.0
      // (adding +0.0 is exact: a sum that starts at +0.0 is never -0.0); first = the lowest l holding
      // my doc.  7 plain VALU instructions per sender and no dependent LDS chain: cheaper than the
      // binary searches below up to 64 entries (config 3 queues ~34 entries per round: with the
      // searches from 25 entries on, the join was 61 % of the kernel)
      const bool have = lane < n;
      const uint2 me = have ? queue[lane] : make_uint2(kDocEnd, 0u);
      float acc = 0.0f;
      uint32_t first = 64u;
      auto sender = [&](const uint32_t at, float &sum) {
        const uint2 sq = queue[at];
        const uint32_t diff = sq.x ^ me.x;
        const uint32_t nm = 0u - (diff < 1u ? diff : 1u);  // 0: same doc, ~0: another doc
        sum += __uint_as_float(sq.y & ~nm);
        const uint32_t cand = at | nm;
        first = cand < first ? cand : first;
      };
      if (PLAN && plan != 0u) {
        // Score plan: the senders of one LEAF are a contiguous run of the queue (lists sorted by leaf,
        // queue sorted by list).  Per leaf: its senders add into `la` (from +0.0, in term order:
        // wand.rs:488-497 `buf[leaf] += score`), then the leaf closes into the root exactly as
        // ScoreExpr::evaluate does (planner.rs:122-153) — every leaf of the sub-query, the ones without
        // a posting of this doc as 0.0: Sum: tot += la (from -0.0); DisMax: max = max(max, la),
        // tot += la (from 0.0).  Leaves of the plan without a term in this segment are in max_init.
        const uint32_t bnd_lo = rfl(bnd_lo_v), bnd_hi = rfl(bnd_hi_v);
        float tot = plan == 2u ? 0.0f : -0.0f, mx = plan_max0;
        uint32_t lo = 0u, rest = leaf_start >> 1;  // bits of the leaf starts still ahead (bit 0 = list 1)
        uint32_t t_at = 0u;                        // first list of the current leaf
        uint32_t present = 0u;                     // leaves that hold my doc (min_match)
        for (;;) {
          // next leaf start after t_at, or T
          const uint32_t skip = rest != 0u ? (uint32_t)__builtin_ctz(rest) + 1u : T - t_at;
          const uint32_t t2 = t_at + skip;
          uint32_t hi = n;
          if (t2 < T) {  // entries of the lists < t2: the prefix sum at the last lane before list t2's first
            const uint32_t f = ((t2 <= 4u ? bnd_lo >> (8u * (t2 - 1u)) : bnd_hi >> (8u * (t2 - 5u))) & 0xFFu);
            hi = f == 0u ? 0u : rl(qincl, (f < 64u ? f : 64u) - 1u);
          }
          float la = 0.0f;
          uint32_t lfirst = 64u;  // lowest sender of THIS leaf that holds my doc (64: none: the leaf is not present)
          // whole groups of 8 broadcast reads; the leaf's last group is predicated per sender (a uniform
          // compare) instead of a rolled loop over what remains: a rolled loop exposes the LDS latency of
          // every single sender (measured on the multi-field workload: the join was 54 % of the wave-cycles)
          for (uint32_t g = lo; g < hi; g += 8u) {
#pragma unroll
            for (uint32_t l = 0; l < 8; l++) {
              const uint2 sq = queue[g + l];  // (past the leaf's end: the next leaf's entries, or whatever lies behind the queue)
              const uint32_t off = g + l < hi ? 0u : ~0u;  // uniform
              const uint32_t diff = sq.x ^ me.x;
              const uint32_t nm = (0u - (diff < 1u ? diff : 1u)) | off;  // 0: same doc and inside the leaf
              la += __uint_as_float(sq.y & ~nm);
              const uint32_t cand = (g + l) | nm;
              lfirst = cand < lfirst ? cand : lfirst;
            }
          }
          first = lfirst < first ? lfirst : first;
          present += lfirst < 64u ? 1u : 0u;
          tot += la;
          mx = fmaxf(mx, la);
          if (t2 >= T) break;
          rest = skip >= 32u ? 0u : rest >> skip;
          t_at = t2;
          lo = hi;
        }
        acc = plan == 2u ? mx + plan_tie * (tot - mx) : tot;
        if (need_many && present < min_match) first = 64u;  // not accepted: nobody owns it
      } else {
        for (uint32_t g = 0; g < n; g += 8) {
#pragma unroll
          for (uint32_t l = 0; l < 8; l++) sender(g + l, acc);
        }
      }
      const bool own = have && first == lane;
      n_scored += (uint32_t)__popcll(__ballot(own));
      if (__ballot(own && acc >= threshold_score()) != 0ull) {
        touched = true;
        take_checked(own, acc, me.x);
      }
    } else if (n != 0u) {
      // many entries (dense lists): binary search of my doc in the queue segment of every list,
      // in list order.  qe[u] = entries of the lists < u = the prefix sum at the last lane before
      // list u's first lane (segment u = [qe[u], qe[u + 1])).
      const uint32_t bnd_lo = rfl(bnd_lo_v), bnd_hi = rfl(bnd_hi_v);
      uint32_t qe[ML + 1];
      qe[0] = 0u;
#pragma unroll
      for (int u = 1; u <= ML; u++) {
        const uint32_t f = ((u <= 4 ? bnd_lo >> (8 * (u - 1)) : bnd_hi >> (8 * (u - 5))) & 0xFFu);
        qe[u] = f == 0u ? 0u : rl(qincl, (f < 64u ? f : 64u) - 1u);
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
        float tot = (PLAN && plan == 2u) ? 0.0f : -0.0f, mx = plan_max0;  // score plan: root sum / max (see the all-pairs join)
        uint32_t present = 0u;  // leaves that hold my doc (min_match)
        bool leaf_hit = false;
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
            if (PLAN && plan != 0u && (uint32_t)(h + u) != 0u && (uint32_t)(h + u) < T && ((leaf_start >> (h + u)) & 1u)) {
              tot += acc;  // list h + u starts a new leaf: the one before it closes (acc = that leaf's sum)
              mx = fmaxf(mx, acc);
              acc = 0.0f;
              present += leaf_hit ? 1u : 0u;
              leaf_hit = false;
            }
            leaf_hit = leaf_hit || hit;
            acc = hit ? acc + (mine ? __uint_as_float(me.y) : __uint_as_float(kk.y)) : acc;
            lower = lower || (hit && (uint32_t)(h + u) < ml);
          }
        }
        if (PLAN && plan != 0u) {  // the last leaf, then the root
          tot += acc;
          mx = fmaxf(mx, acc);
          acc = plan == 2u ? mx + plan_tie * (tot - mx) : tot;
          present += leaf_hit ? 1u : 0u;
        }
        const bool own = have && !lower && !(need_many && present < min_match);
        n_scored += (uint32_t)__popcll(__ballot(own));
        if (__ballot(own && acc >= threshold_score()) != 0ull) {
          touched = true;
          take_checked(own, acc, me.x);
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
    const uint32_t a = bflat[src & (BW - 1u)], b = bflat[(src + T) & (BW - 1u)];
    lo = lane < T ? a : 0u;
    hi = lane < T ? b : 0u;
  };
  auto lane_sum_T = [&](const uint32_t v) {
    uint32_t R = 0;
    for (uint32_t t = 0; t < T; t++) R += rl(v, t);
    return R;
  };

  // ---- driver: planned rounds are prefetched one ahead (`en` loads while `ew` is processed);
  //      a round that needs more than 64 lanes is streamed in chunks cut at a common doc id.
  //      ONE accumulate site and ONE planned-issue site (code size / I-cache): iteration rr = -1
  //      only issues round 0 ----
  BRound ew, en;
  uint32_t en_lanes = 0;
  for (uint32_t rr = 0xFFFFFFFFu; rr == 0xFFFFFFFFu || rr < n_r; rr++) {
    const bool first = rr == 0xFFFFFFFFu;
    const bool big = !first && en_lanes > 64u;
    const uint32_t rend_r = first ? 0u : rfl(rend[rr]);
    uint4 hcur = hdr[rr & 7u];  // (read before the next group's descriptors replace the row)
    if (!first && !big) settle(ew, en);
#ifdef SLG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SLG_STAMP(7);
    if (first || rr + 1 < n_r) {  // prefetch the next round
      const uint32_t nx = rr + 1u, nrow = nx & 7u;
      if (nrow == 0) {
        wave_fence();  // (the read of row 7's header precedes the rewrite)
        describe_group(nx);
        wave_fence();
      }
      en_lanes = row_lanes(nrow);
      if (en_lanes <= 64u) issue(en, nrow);
    }
    SLG_STAMP(0);
    if (first) continue;
    uint32_t ocur = 0, oend = 0, end = rend_r;
    if (big) cuts(rr, ocur, oend);
    uint32_t guard = 0;
    do {
      if (big) {
        // next chunk of an over-full round: every list with postings left gets >= 1 lane, the rest
        // in proportion to what it has left; the chunk ends at the smallest "last loaded doc" of
        // the lists that did not finish, so all postings of a doc are scored in the same chunk
        const uint32_t lane = fresh_lane();
        const uint32_t rem = oend - ocur;
        const uint32_t nne = (uint32_t)__popcll(__ballot(rem != 0u));
        const uint32_t R = lane_sum_T(rem);
        if (R == 0) break;
        if (++guard > (1u << 22)) {  // (cannot happen: every chunk consumes >= 1 posting of a round of < 2^22) — reported, not silent
          if (lane == 0 && p.error_flag) atomicOr(p.error_flag, 1u);
          break;
        }
        const uint32_t need = lane_sum_T((rem + 7u) >> 3);
        uint32_t chunk = rem;
        if (need > 64u) {
          const float share = (float)(64u - nne) * ((float)rem / (float)R);
          const uint32_t mlanes = rem == 0u ? 0u : 1u + (uint32_t)share;
          chunk = rem < mlanes * 8u ? rem : mlanes * 8u;
        }
        uint32_t lastdoc = kDocEnd;
        if (chunk < rem) lastdoc = gdocs[list_off(lane < T ? lane : 0u) + ocur + chunk - 1];
        uint32_t my_first, my_m;
        wave_fence();
        describe_chunk(ocur, chunk, my_first, my_m);
        wave_fence();
        hcur = hdr[8];
        issue(ew, 8);
        settle(ew, ew);
        uint32_t bound = kDocEnd;
        for (uint32_t t = 0; t < T; t++) {
          const uint32_t ld = rl(lastdoc, t);
          bound = ld < bound ? ld : bound;
        }
        // (kDocEnd: nothing was cut, the chunk is the rest of the round)
        end = bound == kDocEnd ? rend_r : (bound + 1u < rend_r ? bound + 1u : rend_r);
        // what each list consumed: its postings with doc < end (a prefix of its lanes' postings)
        uint32_t below = 0;
#pragma unroll
        for (int jj = 0; jj < NS; jj++) below += ew.doc[jj] < end ? 1u : 0u;
        const uint32_t bincl = wave_incl_scan(below);
        const uint32_t hi_v = from_lane(bincl, (my_first + my_m - 1u) & 63u);
        const uint32_t lo_v = from_lane(bincl, (my_first - 1u) & 63u);
        ocur += my_m == 0u ? 0u : hi_v - (my_first == 0u ? 0u : lo_v);
      }
      accumulate(ew, end, hcur.x, hcur.y);
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
#ifdef SLG_FOLD_PROBE
  // Experiment (tools/ab_tags.sh, -DSLG_FOLD_PROBE): what folding merge_topk_kernel into the last-arriving
  // slice of a query would pay per slice BEFORE any merging — an agent-scope release of this slice's
  // candidate list (the slices of a query run on different XCDs, whose L2s are not coherent with each
  // other for plain stores) and an arrival ticket on a per-query counter.  q_scored doubles as the counter
  // (its value is not read in this build).
  if constexpr (BUF) {
    __atomic_thread_fence(__ATOMIC_RELEASE);  // (HIP: agent scope) buffer_wbl2 + waits
    uint32_t prev = 0;
    if (lane == 0) prev = atomicAdd(&p.q_scored[rfl(sl.q)], 1u);
    if (rfl(prev) == 0xFFFFFFF0u) __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (the merging wave would acquire here)
  }
#endif
  if (p.q_scored && lane == 0 && n_scored) atomicAdd(&p.q_scored[rfl(sl.q)], n_scored);
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
  if constexpr (!PERSIST) break;
  wave_fence();  // (the next slice rewrites the LDS tables this one read)
  }  // next slice of this wave
}

}  // namespace slg
