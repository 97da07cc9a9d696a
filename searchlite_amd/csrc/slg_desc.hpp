// slg_desc.hpp — the descriptors a query batch is planned into (host) and consumed from (device),
// and the constants both sides share.  Plain C++: included by the HIP translation units through
// slg_kernels.hpp and by the host-only planner (slg_plan.cpp, built with g++ for the CPU unit tests).
#pragma once

#include <stdint.h>
#include <climits>

#if defined(__HIPCC__)
#define SLG_HD __host__ __device__
#else
#define SLG_HD
#endif

namespace slg {

constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;
constexpr uint32_t kDocEnd = 0xFFFFFFFFu;
constexpr int32_t kSentinelTk = INT32_MIN;
constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr uint32_t kMaxTerms = 32;
constexpr int kChampSorted = 64;  // exact-rank lower bounds (sorted lane maxima)
constexpr int kChampions = 68;    // + bounds for ranks 128, 256, 512, 1024
// LDS bytes of the buffered top-k of a scoring wave (BufTopK<KREGS>::kEntries 64-bit keys)
constexpr int buftopk_lds(int kregs) { return 64 * (kregs + 1) * 8; }
// index of the champion entry that bounds the k-th largest impact of a term from below
SLG_HD inline int champ_index(uint32_t k) {
  return k <= 64 ? (int)k - 1 : k <= 128 ? 64 : k <= 256 ? 65 : k <= 512 ? 66 : 67;
}

// ---- device-side descriptors (built on the host per batch) ---------------------------
// Posting arrays are PADDED per list: list t starts at term_offsets[t] + kListPad * t and is
// followed by kListPad sentinel entries (doc = kDocEnd, impact = 0), so a whole 64-lane slot
// loaded at ANY posting of a list never reads another list's postings: lanes past the list's end
// see sentinels, lanes past a round's cut see later postings of the same list (docs >= the round's
// end).  The scoring kernels therefore need no per-slot lane count to tell real postings from
// foreign ones.  null_idx = a 64-entry run of sentinels (the slot unused descriptor lanes load).
constexpr uint32_t kListPad = 64;
constexpr uint32_t kNullRun = 576;  // sentinels at null_idx: a whole round of idle lanes (64 x 8) + a slot
struct SegDev {
  const uint32_t *docs;     // [P + kListPad * V + kNullRun] doc ids
  const float *imps;        // same layout: precomputed bm25 (weight == 1) per posting
  const uint32_t *deleted;  // bitmap words or nullptr
  const float *champ;       // [V * kChampions] per-term descending impact lower bounds
  uint32_t n_docs;
  uint32_t pad;
  uint64_t null_idx;        // index of kNullRun sentinel entries
};

struct TermRef {  // one scored term of one sub-query
  uint64_t off;   // posting offset inside the segment arrays
  uint32_t df;    // list length
  float weight;
  uint32_t term;  // term id inside the segment (champion table row)
  uint32_t leaf;  // ScorePlan leaf the term's scores add to (non-decreasing inside a sub-query)
  // two-level plans (RoundQuery::n_groups != 0): the group of the leaf = bits 0..7, leaves the plan
  // gives that group (present in this segment or not) = bits 8..15, bit 16 = the group is a DisMax
  uint32_t gmeta;
  float gtie;     // the group's tie breaker
};

struct RoundQuery {  // sub-query = (query, segment) pair with >= 1 non-empty term
  uint32_t q, seg;
  uint32_t term_begin, n_terms;
  uint32_t slice_begin, n_slices;
  uint32_t n_rounds;      // doc-range rounds of <= ~kRoundTarget postings
  uint32_t rounds_per_slice;
  uint32_t bounds_begin;  // bounds[bounds_begin + j*n_terms + t], j = 0..n_rounds
  uint32_t rdoc_begin;    // rdoc[rdoc_begin + j]: first doc id of round j (j = n_rounds: end)
  uint32_t bnd_begin;     // first boundary task of this sub-query (partition kernel)
  uint32_t longest;       // index of the longest ESSENTIAL list (splitter source)
  uint32_t ess_mask;      // bit t: list t is essential (MaxScore); the others are only probed
  uint32_t skip_mask;     // bit t: non-essential list t is sparse-probed: its 64-posting blocks are
                          // tested against the candidate docs before they are loaded (block skipping)
  uint32_t filter;        // 0: none; f + 1: docs must also pass filter f (reject table row f)
  uint32_t cand_lo, cand_hi;  // large-k mode: first candidate slot of this sub-query (u64)
  // score plan (query/planner.rs:113-153): 0 = every term its own leaf, summed (the flat sum in
  // term order); 1 = Sum of leaves that group several terms; 2 = DisMax of leaves
  uint32_t plan;
  float tie;       // DisMax tie breaker
  float max_init;  // DisMax: 0.0 if some leaf of the plan has no term in this segment, else -inf
  uint32_t n_leaves;  // leaves of the plan (a DisMax counts every one, absent ones as 0.0)
  uint32_t n_groups;  // two-level plan: the root combines this many groups of leaves (0: flat plan)
  // threshold seed (0: none): theta0 = max_t w_t * champ[t][rank(k)].  At least k live docs have
  // a single contribution >= theta0 and a doc's total is >= any one of its non-negative
  // contributions (Sum, or DisMax with tie in [0, 1]), so nothing below theta0 reaches the top-k.
  // Set by the host planner from its mirror of the champion table; never with a doc filter (it
  // may reject the champions) or a negative weight.
  float theta0;
  // deep score trees (slg_score_plans::q_node_offsets, > 2 levels): depth = levels of Sum / DisMax nodes
  // above the leaves (0: not a deep tree), node_begin = the query's first PlanNode in the image
  uint32_t depth, node_begin;
};

// One Sum / DisMax node of a deep score tree, in the CANONICAL form the planner builds: every leaf hangs
// at the same depth (a leaf higher up gets a chain of one-child Sum nodes: Sum of one child is the
// child, bit for bit), so level l of the tree = the nodes at distance l from the root.  TermRef::gmeta
// of a term = the node its leaf hangs off (level depth - 1).
constexpr uint32_t kMaxPlanDepth = 4;  // = SLG_MAX_PLAN_DEPTH (searchlite_gpu.h; checked in slg_plan.cpp)
struct PlanNode {
  uint32_t parent;      // node index inside the query's table (the root: itself)
  uint32_t n_children;  // children the plan gives the node (present in this segment or not)
  uint32_t kind;        // 0 Sum, 1 DisMax
  float tie;
};

// What a scoring wave needs to start its slice, gathered in one record per launch position by
// partition_rounds_kernel (a wave then starts with ONE dependent load instead of a chain of four).
struct SliceDesc {
  uint32_t slice;       // slice index (candidate output arrays)
  uint32_t term_begin;  // first TermRef of the sub-query
  uint32_t bounds_off;  // bounds[] index of the slice's first cut points
  uint32_t rdoc_off;    // rdoc[] index of the slice's first round
  uint32_t n_terms, n_rounds, seg, filter;
  uint32_t q;
  float theta0;
  uint32_t cand_lo, cand_hi;
  // for waves that cut their slice themselves (slg_score_uni4.hpp with RoundScoreParams::bounds ==
  // nullptr): the slice's first round, the sub-query's rounds, its splitter list
  uint32_t first_round, sq_rounds, longest;
  uint32_t l_df;   // the splitter list's length and posting offset: its stride positions are read
  uint64_t l_off;  // straight off this record, beside the TermRef loads instead of behind them
  // score plan of the sub-query (RoundQuery::plan / tie / max_init / n_leaves), for the few-term
  // kernel's plan instantiation (flat plans: Sum or DisMax over leaves of one or more terms)
  uint32_t plan;
  float tie, max_init;
  uint32_t n_leaves;
};

struct QueryRef {
  uint32_t slice_begin, slice_end;  // all slices of all sub-queries of this query
};


// ---- work queues of the persistent scoring waves (slg_score_uni4.hpp) ---------------------------
// One returning atomic per slice on ONE counter does not scale: MI355X hands out a ticket of a single
// address every ~11-16 ns whatever the number of waves asking (tools/micro/atomic_queue.hip: 6144 waves,
// 195 us for 12 288 tickets; 8 counters 1.8 ns per ticket, 64 counters 0.43 ns) — config 2's 13.4K
// slices would take longer to hand out than to score.  So the launch order is dealt round-robin over
// kWorkQueues queues (position p: queue p % kWorkQueues, index p / kWorkQueues — every queue gets the
// same mix of long and short slices), each with its own counter on its own 256-byte line; wave w pulls
// from queue w % kWorkQueues and, when that has run dry, tries kWorkSteals other queues before it exits.
constexpr uint32_t kWorkQueues = 64;
constexpr uint32_t kWorkCtrStride = 64;  // words between two counters (256 bytes)
constexpr uint32_t kWorkSteals = 2;      // other queues a wave tries when its own has run dry

// ---- planning constants (the kernels that consume them: slg_score*.hpp) ------------------------
constexpr int kMaxRoundsPerSlice = 16;  // and (rounds+1)*T <= 64: cut points live in one VGPR
constexpr int kDefaultRoundsPerSlice = 8;
#ifndef SLG_UNI_RPS
#define SLG_UNI_RPS 4
#endif
constexpr int kUniRoundsPerSlice = SLG_UNI_RPS;  // few-term kernel, k <= 64
constexpr int kSpanWords = 512;           // bitmap words per window (many-term kernel)
constexpr uint32_t kSpan = kSpanWords * 32;  // docs per window
constexpr int kUniSlots = 8;                 // 64-posting slots per round
constexpr int kUniCap = kUniSlots * 64;      // postings per round
constexpr int kUniMaxLists = 4;              // lists per sub-query on the round-2 few-term kernel and
                                             // on the 4-bit-field form of the current one
constexpr int kU3MaxLists = 8;               // lists per sub-query on the few-term kernel (8-bit fields)
constexpr int kMultiCap = 512;       // accumulators (= distinct docs) per chunk
constexpr int kMultiTarget = 448;    // planned postings per round (host; measured optimum 448-480)

}  // namespace slg
