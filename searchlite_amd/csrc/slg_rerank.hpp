// slg_rerank.hpp — dense f32 vector rerank (the slot of searchlite-core/src/gpu/rerank.rs:3).
//
// Arithmetic restated: vectors/mod.rs:63-71 (VectorStore::vector), :107-120
// (metric_similarity), :122-129 (blend_scores); api/reader.rs:217-223 (missing vector
// score) and :225-254 (compute_hybrid_score, one clause).
//
// The path is an HBM row gather (one D-float row per candidate, 0.5 flop/byte): each wave
// streams whole rows with 16-byte lane loads, reduces in-register, and a wave-wide sorted
// top-k (slg_kernels.hpp) picks the k_out best blended scores.  One clause: rerank_kernel
// (VALU; the contraction is a GEMV).  Several clauses sharing a candidate set:
// rerank_multi_kernel, whose [candidates x clauses] products run on v_mfma_f32_16x16x4_f32.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "slg_kernels.hpp"

namespace slg {

struct VecSegDev {
  const uint32_t *offsets;  // [n_docs] row index or 0xFFFFFFFF
  const float *values;      // [rows * dim]
  uint32_t n_docs;
  uint32_t dim;
  int32_t metric;  // 0 cosine, 1 l2
  uint32_t pad;
};

struct RerankParams {
  const VecSegDev *vsegs;
  uint32_t n_segs, dim;
  const float *qvecs;  // [nq * dim]
  const float *alpha;  // [nq]
  const uint32_t *cand_doc, *cand_seg;
  const float *cand_bm25;
  const uint32_t *cand_count;  // [nq]
  uint32_t max_cand, k_out;
  uint32_t *out_doc, *out_seg;
  float *out_score, *out_vec;
  uint32_t *out_count;
  uint32_t nq;
};

constexpr uint32_t kRerankMaxCand = 8192;  // blended + vec scores staged in LDS

// wave-wide f32 sum with DPP row shifts / broadcasts (no LDS round trips); result in lane 63
__device__ __forceinline__ float wave_sum_f(float v) {
  int x = __float_as_int(v);
#define SLG_DPP_ADD(ctrl, rmask)                                                                \
  x = __float_as_int(__int_as_float(x) +                                                        \
                     __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, true)))
  SLG_DPP_ADD(0x111, 0xf);  // row_shr:1
  SLG_DPP_ADD(0x112, 0xf);  // row_shr:2
  SLG_DPP_ADD(0x114, 0xf);  // row_shr:4
  SLG_DPP_ADD(0x118, 0xf);  // row_shr:8
  SLG_DPP_ADD(0x142, 0xa);  // row_bcast:15
  SLG_DPP_ADD(0x143, 0xc);  // row_bcast:31
#undef SLG_DPP_ADD
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// The k_out best blended scores of query q by (blended desc, segment asc, doc asc), their vector
// scores, the count: wave 0 of the workgroup, after the blends are in LDS.
template <int KREGS>
__device__ __forceinline__ void rerank_emit_topk(const RerankParams &p, const uint32_t q, const uint32_t n,
                                                 const float *s_blend, const float *s_vec,
                                                 const uint32_t *cdoc, const uint32_t *cseg,
                                                 const uint32_t lane) {
  const uint32_t k = p.k_out;
  if (k == 0) {
    if (lane == 0) p.out_count[q] = 0;
    return;
  }
  WaveTopK<KREGS, true> top;
  top.init();
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t i = base + lane;
    int32_t ctk = kSentinelTk;
    uint32_t d = 0xFFFFFFFFu, sg = 0xFFFFFFFFu;
    if (i < n) {
      ctk = total_key(s_blend[i]);
      d = cdoc[i];
      sg = cseg[i];
    }
    uint64_t m = __ballot(i < n && top.passes(ctk, sg, d));
    while (m) {
      const uint32_t l = (uint32_t)__builtin_ctzll(m);
      top.insert((int32_t)rl((uint32_t)ctk, l), rl(sg, l), rl(d, l), k, lane);
      m &= m - 1;
      m &= __ballot(top.passes(ctk, sg, d));
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
  if (p.out_vec) {
    // vector score of each winner: all lanes scan the candidate list for its (seg, doc)
    const uint32_t nout = top.count < k ? top.count : k;
    for (uint32_t pos = 0; pos < nout; pos++) {
      const uint32_t pl = pos / KREGS, pr = pos % KREGS;
      uint32_t wd = top.doc[0], ws = top.seg[0];
#pragma unroll
      for (int r = 1; r < KREGS; r++) {
        wd = pr == (uint32_t)r ? top.doc[r] : wd;
        ws = pr == (uint32_t)r ? top.seg[r] : ws;
      }
      wd = rl(wd, pl);
      ws = rl(ws, pl);
      for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const bool match = i < n && cdoc[i] == wd && cseg[i] == ws;
        const uint64_t mm = __ballot(match);
        if (mm) {
          if (lane == (uint32_t)__builtin_ctzll(mm)) p.out_vec[(size_t)q * k + pos] = s_vec[i];
          break;
        }
      }
    }
    for (uint32_t pos = nout + lane; pos < k; pos += 64) p.out_vec[(size_t)q * k + pos] = 0.0f;
  }
  if (lane == 0) p.out_count[q] = top.count;
}

// vector score + blend of candidate c (lane 0 stores): api/reader.rs:217-223, :240-246,
// vectors/mod.rs:112-118, :128
__device__ __forceinline__ void rerank_finish(const float sum, const bool have_row, const int32_t metric,
                                              const float alpha, const float bm, float *s_blend,
                                              float *s_vec, const uint32_t c, const bool store) {
  float vs;
  if (!have_row)
    vs = metric == 0 ? -1.0f : -3.40282347e+38f;
  else if (metric == 0)
    vs = (sum != sum) ? 0.0f : sum;
  else
    vs = -sqrtf(sum);
  if (store) {
    float blended;
    if (alpha >= 1.0f)
      blended = bm;
    else if (alpha <= 0.0f)
      blended = vs;
    else
      blended = alpha * bm + (1.0f - alpha) * vs;
    s_blend[c] = blended;
    s_vec[c] = vs;
  }
}

// Whole-row scan for dim <= 256 * CH, dim % 4 == 0 (config 5: 768 = 3 chunks).  A wave takes 4
// candidates at a time and issues ALL 16-byte lane loads of their 4 rows at once (4 x 3 KiB
// contiguous: whole DRAM pages instead of 1 KiB pieces of 8 rows, and no load -> FMA -> load
// round trips inside a row); the query vector stays in registers; the next group's row pointers
// (candidate -> segment -> offsets[doc], a chain of dependent scalar loads) are resolved while
// this group's rows are in flight.
template <int CH>
__device__ __forceinline__ void rerank_scan_rows(const RerankParams &p, const uint32_t q, const uint32_t n,
                                                 const float alpha, float *s_blend, float *s_vec,
                                                 const uint32_t lane, const uint32_t wave) {
  constexpr int U = 4;
  if (n == 0) return;  // (the clamped loads below read candidate 0)
  const uint32_t dim = p.dim;
  const float *__restrict__ qv = p.qvecs + (size_t)q * dim;
  const uint32_t *__restrict__ cdoc = p.cand_doc + (size_t)q * p.max_cand;
  const uint32_t *__restrict__ cseg = p.cand_seg + (size_t)q * p.max_cand;
  const float *__restrict__ cbm = p.cand_bm25 + (size_t)q * p.max_cand;
  // lane's piece of chunk ch: floats idx[ch] .. idx[ch]+3 of a row (lanes past the row's end
  // re-read its first floats and are masked out)
  typedef float fvec4 __attribute__((ext_vector_type(4)));
  fvec4 a[CH];
  uint32_t idx[CH];
  bool valid[CH];
#pragma unroll
  for (int ch = 0; ch < CH; ch++) {
    const uint32_t i = ch * 256 + lane * 4;
    valid[ch] = i < dim;
    idx[ch] = valid[ch] ? i : 0u;
    const fvec4 v = *reinterpret_cast<const fvec4 *>(qv + idx[ch]);
    a[ch] = valid[ch] ? v : (fvec4){0.f, 0.f, 0.f, 0.f};
  }
  // candidate -> row pointer, branch-free on the scalar unit (every index is clamped to something
  // readable; a candidate without a vector reads the query vector and its sum is discarded)
  // (pointers read from memory are generic: cast to the global address space, or the row loads
  // become flat_load and their waits couple with the scalar chain's)
  typedef const __attribute__((address_space(1))) fvec4 *grow_t;
  typedef const __attribute__((address_space(1))) uint32_t *gword_t;
  grow_t rowN[U];
  bool haveN[U];
  int32_t metN[U];
  float bmN[U];
  auto resolve = [&](const uint32_t c0) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t c = c0 + u;
      const bool c_ok = c < n;
      const uint32_t cc = c_ok ? c : 0u;
      const uint32_t doc = cdoc[cc], seg = cseg[cc];
      bmN[u] = cbm[cc];
      const bool seg_ok = seg < p.n_segs;
      const VecSegDev vd = p.vsegs[seg_ok ? seg : 0u];
      const bool doc_ok = c_ok && seg_ok && vd.dim == dim && doc < vd.n_docs;
      const gword_t optr = doc_ok ? (gword_t)(vd.offsets + doc) : (gword_t)p.cand_count;  // (any readable word)
      const uint32_t off = *optr;
      haveN[u] = doc_ok && off != 0xFFFFFFFFu;
      metN[u] = seg_ok ? vd.metric : 0;
      rowN[u] = (grow_t)(haveN[u] ? vd.values + (size_t)off * dim : qv);
    }
  };
  resolve(wave * U);
  for (uint32_t c0 = wave * U; c0 < n; c0 += 4 * U) {
    grow_t row[U];
    bool have[U];
    int32_t met[U];
    float bm[U];
    fvec4 b[U][CH];
#pragma unroll
    for (int u = 0; u < U; u++) {
      row[u] = rowN[u];
      have[u] = haveN[u];
      met[u] = metN[u];
      bm[u] = bmN[u];
#pragma unroll
      for (int ch = 0; ch < CH; ch++) b[u][ch] = row[u][idx[ch] >> 2];
    }
    if (c0 + 4 * U < n) resolve(c0 + 4 * U);  // the next group's pointers, under this group's rows
#pragma unroll
    for (int u = 0; u < U; u++) {
      float acc = 0.0f;
#pragma unroll
      for (int ch = 0; ch < CH; ch++) {
        const fvec4 bb = valid[ch] ? b[u][ch] : a[ch];  // masked lanes: a = 0 and a - a = 0
        if (met[u] == 0) {
          acc += a[ch].x * bb.x;
          acc += a[ch].y * bb.y;
          acc += a[ch].z * bb.z;
          acc += a[ch].w * bb.w;
        } else {
          const float d0 = a[ch].x - bb.x, d1 = a[ch].y - bb.y, d2 = a[ch].z - bb.z, d3 = a[ch].w - bb.w;
          acc += d0 * d0;
          acc += d1 * d1;
          acc += d2 * d2;
          acc += d3 * d3;
        }
      }
      const uint32_t c = c0 + u;
      const float sum = wave_sum_f(acc);
      rerank_finish(sum, have[u], met[u], alpha, bm[u], s_blend, s_vec, c, lane == 0 && c < n);
    }
  }
}

// The same whole-row scan for SEVERAL clause vectors over one vector field (multi-clause hybrid
// requests, api/reader.rs:225-254): a wave takes 4 candidates at a time, has their 4 whole rows in
// flight (dim <= 256 * CH, dim % 4 == 0), resolves the next 4 row pointers under them, and runs
// every clause against the rows it holds — each row is read ONCE whatever the clause count.  Clause
// cc's vector lies in LDS at s_qc + cc * q_stride.  Writes the similarity (cosine: dot, NaN -> 0;
// L2: -sqrt sum (x - y)^2, vectors/mod.rs:107-120) of clause cc and candidate c to
// s_vs[cc * vs_stride + c]; with s_has != nullptr, candidates that have a vector get has_bits OR-ed
// into s_has[c] (the caller zeroes it).
template <int CH>
__device__ __forceinline__ void rerank_scan_clauses(const VecSegDev *vsegs, const uint32_t n_segs, const uint32_t dim,
                                                    const int32_t metric, const float *s_qc, const uint32_t q_stride,
                                                    const uint32_t n_cl, const uint32_t *cdoc, const uint32_t *cseg,
                                                    const uint32_t n, const float *readable, float *s_vs,
                                                    const uint32_t vs_stride, uint32_t *s_has, const uint32_t has_bits,
                                                    const uint32_t lane, const uint32_t wave) {
  constexpr int U = 4;
  if (n == 0) return;
  typedef float fvec4 __attribute__((ext_vector_type(4)));
  typedef const __attribute__((address_space(1))) fvec4 *grow_t;
  typedef const __attribute__((address_space(1))) uint32_t *gword_t;
  uint32_t idx[CH];
  bool valid[CH];
#pragma unroll
  for (int ch = 0; ch < CH; ch++) {
    const uint32_t i = ch * 256 + lane * 4;
    valid[ch] = i < dim;
    idx[ch] = valid[ch] ? i : 0u;
  }
  grow_t rowN[U];
  bool haveN[U];
  auto resolve = [&](const uint32_t c0) {  // branch-free on the scalar unit (clamped indices)
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t c = c0 + u;
      const bool c_ok = c < n;
      const uint32_t cc = c_ok ? c : 0u;
      const uint32_t doc = cdoc[cc], seg = cseg[cc];
      const bool seg_ok = seg < n_segs;
      const VecSegDev vd = vsegs[seg_ok ? seg : 0u];
      const bool doc_ok = c_ok && seg_ok && vd.dim == dim && doc < vd.n_docs;
      const gword_t optr = doc_ok ? (gword_t)(vd.offsets + doc) : (gword_t)cdoc;  // (any readable word)
      const uint32_t off = *optr;
      haveN[u] = doc_ok && off != 0xFFFFFFFFu;
      rowN[u] = (grow_t)(haveN[u] ? vd.values + (size_t)off * dim : readable);
    }
  };
  resolve(wave * U);
  for (uint32_t c0 = wave * U; c0 < n; c0 += 4 * U) {
    bool have[U];
    fvec4 b[U][CH];
#pragma unroll
    for (int u = 0; u < U; u++) {
      have[u] = haveN[u];
#pragma unroll
      for (int ch = 0; ch < CH; ch++) b[u][ch] = rowN[u][idx[ch] >> 2];
    }
    if (c0 + 4 * U < n) resolve(c0 + 4 * U);  // the next group's pointers, under this group's rows
    for (uint32_t cc = 0; cc < n_cl; cc++) {
      fvec4 a[CH];
#pragma unroll
      for (int ch = 0; ch < CH; ch++) {
        const fvec4 v = *reinterpret_cast<const fvec4 *>(s_qc + cc * q_stride + idx[ch]);
        a[ch] = valid[ch] ? v : (fvec4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        float acc = 0.0f;
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
          const fvec4 bb = valid[ch] ? b[u][ch] : a[ch];  // masked lanes: a = 0 and a - a = 0
          if (metric == 0) {
            acc += a[ch].x * bb.x;
            acc += a[ch].y * bb.y;
            acc += a[ch].z * bb.z;
            acc += a[ch].w * bb.w;
          } else {
            const float d0 = a[ch].x - bb.x, d1 = a[ch].y - bb.y, d2 = a[ch].z - bb.z, d3 = a[ch].w - bb.w;
            acc += d0 * d0;
            acc += d1 * d1;
            acc += d2 * d2;
            acc += d3 * d3;
          }
        }
        const float sum = wave_sum_f(acc);
        const uint32_t c = c0 + u;
        if (lane == 0 && c < n && have[u])
          s_vs[cc * vs_stride + c] = metric == 0 ? (sum != sum ? 0.0f : sum) : -sqrtf(sum);
      }
    }
    if (s_has && lane == 0) {
#pragma unroll
      for (int u = 0; u < U; u++)
        if (c0 + u < n && have[u]) s_has[c0 + u] |= has_bits;
    }
  }
}

// One workgroup (4 waves) per query.  dim <= 768 and a multiple of 4: rerank_scan_rows (whole
// rows, 4 candidates per wave and step).  Other dims: 8 candidates per wave and step, 256 floats
// of each row at a time.
template <int KREGS>
__global__ void __launch_bounds__(256) rerank_kernel(RerankParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *s_blend = reinterpret_cast<float *>(smem);
  float *s_vec = s_blend + p.max_cand;
  const uint32_t lane = threadIdx.x & 63;
  // (a scalar: candidate indices derived from it are wave-uniform, so the candidate -> row pointer
  // chain runs on the scalar unit and row pointers live in SGPRs)
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t q = blockIdx.x;
  uint32_t n = p.cand_count[q];
  n = n < p.max_cand ? n : p.max_cand;
  const float alpha = p.alpha[q];
  const float *__restrict__ qv = p.qvecs + (size_t)q * p.dim;
  const uint32_t dim = p.dim;
  const uint32_t *cdoc = p.cand_doc + (size_t)q * p.max_cand;
  const uint32_t *cseg = p.cand_seg + (size_t)q * p.max_cand;
  const float *cbm = p.cand_bm25 + (size_t)q * p.max_cand;
  constexpr int U = 8;
  const bool whole_rows = (dim & 3u) == 0 && dim <= 768u;
  if (whole_rows) {
    if (dim <= 256u)
      rerank_scan_rows<1>(p, q, n, alpha, s_blend, s_vec, lane, wave);
    else if (dim <= 512u)
      rerank_scan_rows<2>(p, q, n, alpha, s_blend, s_vec, lane, wave);
    else
      rerank_scan_rows<3>(p, q, n, alpha, s_blend, s_vec, lane, wave);
  }
  for (uint32_t c0 = whole_rows ? n : wave * U; c0 < n; c0 += 4 * U) {
    const float *row[U];
    int32_t metric[U];
    float acc[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      row[u] = nullptr;
      metric[u] = 0;
      acc[u] = 0.0f;
      const uint32_t c = c0 + u;
      if (c < n) {
        const uint32_t doc = cdoc[c], seg = cseg[c];
        if (seg < p.n_segs) {
          const VecSegDev vd = p.vsegs[seg];
          metric[u] = vd.metric;
          if (vd.dim == dim && doc < vd.n_docs) {
            const uint32_t off = vd.offsets[doc];
            if (off != 0xFFFFFFFFu) row[u] = vd.values + (size_t)off * dim;
          }
        }
      }
    }
    if ((dim & 3u) == 0) {
      for (uint32_t i = lane * 4; i < dim; i += 256) {
        const float4 a = *reinterpret_cast<const float4 *>(qv + i);
        float4 b[U];
#pragma unroll
        for (int u = 0; u < U; u++)
          b[u] = row[u] ? *reinterpret_cast<const float4 *>(row[u] + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < U; u++) {
          if (metric[u] == 0) {
            acc[u] += a.x * b[u].x;
            acc[u] += a.y * b[u].y;
            acc[u] += a.z * b[u].z;
            acc[u] += a.w * b[u].w;
          } else {
            const float d0 = a.x - b[u].x, d1 = a.y - b[u].y, d2 = a.z - b[u].z, d3 = a.w - b[u].w;
            acc[u] += d0 * d0;
            acc[u] += d1 * d1;
            acc[u] += d2 * d2;
            acc[u] += d3 * d3;
          }
        }
      }
    } else {
      for (uint32_t i = lane; i < dim; i += 64) {
        const float a = qv[i];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const float bb = row[u] ? row[u][i] : 0.0f;
          if (metric[u] == 0) {
            acc[u] += a * bb;
          } else {
            const float d = a - bb;
            acc[u] += d * d;
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t c = c0 + u;
      const float sum = wave_sum_f(acc[u]);
      float vs;
      if (row[u] == nullptr)
        vs = metric[u] == 0 ? -1.0f : -3.40282347e+38f;  // api/reader.rs:217-223
      else if (metric[u] == 0)
        vs = (sum != sum) ? 0.0f : sum;  // vectors/mod.rs:112-116 NaN -> 0
      else
        vs = -sqrtf(sum);  // vectors/mod.rs:118
      if (lane == 0 && c < n) {
        const float bm = cbm[c];
        float blended;  // api/reader.rs:240-246
        if (alpha >= 1.0f)
          blended = bm;
        else if (alpha <= 0.0f)
          blended = vs;
        else
          blended = alpha * bm + (1.0f - alpha) * vs;  // vectors/mod.rs:128
        s_blend[c] = blended;
        s_vec[c] = vs;
      }
    }
  }
  __syncthreads();
  if (wave != 0) return;

  rerank_emit_topk<KREGS>(p, q, n, s_blend, s_vec, cdoc, cseg, lane);
}

// ---- hybrid rerank with several vector clauses (api/reader.rs:225-254, MAX_VECTOR_CLAUSES = 8) ----
// score(doc) = mean over clauses c of blend(alpha_c, bm25, vec_c), vec_c = boost_c * similarity
// (api/reader.rs:2421 `vscore *= clause.boost`), a missing vector counts as -1.0 / f32::MIN per
// clause (:217-223); the vector score reported is the sum over the clauses (:236-238).
// All clauses of a query share its candidate set, so the [candidates x clauses] similarities are
// a small GEMM: for cosine (= dot of pre-normalized vectors, vectors/mod.rs:107-117) it runs on
// the f32 matrix cores, one v_mfma_f32_16x16x4_f32 tile = 16 candidates x 16 clause columns
// (<= 8 live) per wave.  Lane l feeds A[cand l&15][k-group l>>4] and B[k-group l>>4][clause l&15]
// with one float4 each per 16 k values (four MFMAs: component c of every lane is one k-step, the
// same k permutation on both operands), so a candidate row is read as four 16-byte lanes = whole
// 64-byte sectors.  D[row 4*(l>>4)+r][col l&15] = the dot of candidate row / clause col.
// L2 (-sqrt sum (x - y)^2, vectors/mod.rs:98-105,118) and dimensions that are not a multiple of 16
// take the VALU path.  The path is an HBM row gather either way.
struct RerankMultiParams {
  RerankParams base;       // qvecs: [nq][n_clauses][dim]; alpha: [nq][n_clauses]
  const float *boost;      // [nq][n_clauses] or nullptr (1.0)
  uint32_t n_clauses;      // 1..8
  uint32_t q_stride;       // padded LDS row of one clause vector (floats): dim + 4
};

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr uint32_t kRerankMultiLdsFloats = 36 * 1024;  // clause vectors + per-clause scores + blends

template <int KREGS>
__global__ void __launch_bounds__(256) rerank_multi_kernel(RerankMultiParams mp) {
  const RerankParams &p = mp.base;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t q = blockIdx.x;
  const uint32_t NC = mp.n_clauses, dim = p.dim, qs = mp.q_stride;
  uint32_t n = p.cand_count[q];
  n = n < p.max_cand ? n : p.max_cand;
  float *s_q = reinterpret_cast<float *>(smem);     // [NC][qs]
  float *s_vs = s_q + NC * qs;                      // [NC][max_cand] similarity (boost applied)
  float *s_blend = s_vs + NC * p.max_cand;          // [max_cand]
  float *s_vsum = s_blend + p.max_cand;             // [max_cand]
  const uint32_t *cdoc = p.cand_doc + (size_t)q * p.max_cand;
  const uint32_t *cseg = p.cand_seg + (size_t)q * p.max_cand;
  const float *cbm = p.cand_bm25 + (size_t)q * p.max_cand;
  const int32_t metric = p.vsegs[0].metric;  // one vector field: the same metric in every segment

  for (uint32_t i = threadIdx.x; i < NC * dim; i += 256)
    s_q[(i / dim) * qs + (i % dim)] = p.qvecs[(size_t)q * NC * dim + i];
  __syncthreads();

  // row of candidate c, or nullptr (missing vector / out of range)
  auto row_of = [&](const uint32_t c) -> const float * {
    if (c >= n) return nullptr;
    const uint32_t doc = cdoc[c], seg = cseg[c];
    if (seg >= p.n_segs) return nullptr;
    const VecSegDev vd = p.vsegs[seg];
    if (vd.dim != dim || doc >= vd.n_docs) return nullptr;
    const uint32_t off = vd.offsets[doc];
    return off == 0xFFFFFFFFu ? nullptr : vd.values + (size_t)off * dim;
  };

  // L2 with >= 3 clauses also runs its products on the matrix cores: |q - x|^2 = |q|^2 + |x|^2 - 2 q.x
  // (|x|^2 from the row pieces a lane holds anyway, |q|^2 once per clause).  The identity cancels for
  // near-duplicate vectors, so a pair whose distance comes out below 5 % of |q|^2 + |x|^2 is
  // recomputed as the plain sum of squared differences (vectors/mod.rs:98-105) by the whole wave.
  const bool l2_mfma = metric != 0 && NC >= 3u && (dim & 15u) == 0;
  float *s_qq = s_vsum + p.max_cand;  // [NC] squared norms of the clause vectors (l2_mfma)
  if (l2_mfma) {
    for (uint32_t cc = wave; cc < NC; cc += 4) {
      float a = 0.0f;
      for (uint32_t i = lane; i < dim; i += 64) a = __builtin_fmaf(s_q[cc * qs + i], s_q[cc * qs + i], a);
      const float t = wave_sum_f(a);
      if (lane == 0) s_qq[cc] = t;
    }
    __syncthreads();
  }
  if ((metric == 0 || l2_mfma) && (dim & 15u) == 0) {
    // ---- cosine (and many-clause L2) on the matrix cores: tiles of 16 candidates, round-robin over the 4 waves.
    //      Row loads go through global-address-space pointers (a pointer read from memory is
    //      generic: flat loads would share the LDS counter with the clause-vector reads), a
    //      candidate without a vector reads the query (its products are never used), four
    //      16-byte pieces per lane are in flight, and the next tile's row pointers (a chain of
    //      three dependent loads) are fetched under this tile's rows ----
    typedef const __attribute__((address_space(1))) f32x4_t *grow_t;
    const uint32_t g = lane >> 4, cl = lane & 15u;
    const f32x4_t *qrow = reinterpret_cast<const f32x4_t *>(s_q + (cl < NC ? cl : 0u) * qs);
    const float *dummy = p.qvecs + (size_t)q * NC * dim;
    const float *nrow = row_of(wave * 16 + cl);
    for (uint32_t t0 = wave * 16; t0 < n; t0 += 64) {
      const grow_t row = (grow_t)(nrow ? nrow : dummy);
      const bool has_row = nrow != nullptr;
      float xx = 0.0f;  // l2_mfma: squared norm of the pieces of my candidate's row that I hold
      f32x4_t acc = {0.0f, 0.0f, 0.0f, 0.0f};
      uint32_t kb = g;  // in units of 4 floats; lane g takes pieces g, g + 4, ...
      const uint32_t kend = dim >> 2;
      f32x4_t a0 = row[kb], a1 = a0, a2 = a0, a3 = a0;
      if (kb + 4 < kend) a1 = row[kb + 4];
      if (kb + 8 < kend) a2 = row[kb + 8];
      if (kb + 12 < kend) a3 = row[kb + 12];
      if (t0 + 64 < n) nrow = row_of(t0 + 64 + cl);
      for (; kb < kend; kb += 16) {
        const f32x4_t c0 = a0, c1 = a1, c2 = a2, c3 = a3;
        const uint32_t nk = kb + 16;
        if (nk < kend) a0 = row[nk];  // (uniform conditions: dim is a multiple of 16)
        if (nk + 4 < kend) a1 = row[nk + 4];
        if (nk + 8 < kend) a2 = row[nk + 8];
        if (nk + 12 < kend) a3 = row[nk + 12];
        auto step = [&](const f32x4_t a, const uint32_t at) {
          const f32x4_t b = qrow[at];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
          if (l2_mfma)
            xx = __builtin_fmaf(a.x, a.x, __builtin_fmaf(a.y, a.y, __builtin_fmaf(a.z, a.z, __builtin_fmaf(a.w, a.w, xx))));
        };
        step(c0, kb);
        if (kb + 4 < kend) step(c1, kb + 4);
        if (kb + 8 < kend) step(c2, kb + 8);
        if (kb + 12 < kend) step(c3, kb + 12);
      }
      if (!l2_mfma) {
        if (cl < NC) {  // lane holds D[candidate t0 + 4g + r][clause cl]
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const uint32_t c = t0 + 4 * g + r;
            const float d = acc[r];
            if (c < n) s_vs[cl * p.max_cand + c] = d != d ? 0.0f : d;  // vectors/mod.rs:112-116 NaN -> 0
          }
        }
      } else {
        // |x|^2 of candidate t0 + cl: the four lanes of its column hold a quarter of the row each
        xx += __shfl_xor(xx, 16, 64);
        xx += __shfl_xor(xx, 32, 64);
        const float qq = s_qq[cl < NC ? cl : 0u];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const uint32_t c = t0 + 4 * g + r;
          const float xc = __shfl(xx, (int)(4 * g + r), 64);
          const bool hc = __shfl((int)has_row, (int)(4 * g + r), 64) != 0;
          const float ssum = qq + xc;
          const float d2 = ssum - 2.0f * acc[r];
          const bool mine = cl < NC && c < n && hc;
          const bool near = mine && !(d2 >= 0.05f * ssum);  // (also: NaN)
          if (mine && !near) s_vs[cl * p.max_cand + c] = -sqrtf(d2);
          // near-duplicates: the exact sum, one pair at a time, all 64 lanes on it
          uint64_t todo = __ballot(near);
          while (todo) {
            const uint32_t l = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t pc = t0 + 4 * (l >> 4) + r, pcl = l & 15u;  // the pair's candidate and clause
            const uint32_t src = pc - t0;                              // a lane that holds its row pointer
            const uint64_t rp = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)((uint64_t)(uintptr_t)row >> 32), (int)src) << 32) |
                                (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uintptr_t)row, (int)src);
            typedef const __attribute__((address_space(1))) float *gf_t;
            const gf_t prow = (gf_t)(uintptr_t)rp;
            float a = 0.0f;
            for (uint32_t i = lane; i < dim; i += 64) {
              const float df = s_q[pcl * qs + i] - prow[i];
              a += df * df;
            }
            const float t = wave_sum_f(a);
            if (lane == 0) s_vs[pcl * p.max_cand + pc] = -sqrtf(t);
          }
        }
      }
    }
  } else if ((dim & 3u) == 0 && dim <= 768u) {
    // ---- VALU path (L2, or cosine at dimensions the MFMA tiling does not take): four candidates
    //      per wave and step with their whole rows in flight, every clause against the rows held ----
    const float *readable = p.qvecs + (size_t)q * NC * dim;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    if (dim <= 256u)
      rerank_scan_clauses<1>(p.vsegs, p.n_segs, dim, metric, s_q, qs, NC, cdoc, cseg, n, readable, s_vs, p.max_cand,
                             nullptr, 0u, lane, wv);
    else if (dim <= 512u)
      rerank_scan_clauses<2>(p.vsegs, p.n_segs, dim, metric, s_q, qs, NC, cdoc, cseg, n, readable, s_vs, p.max_cand,
                             nullptr, 0u, lane, wv);
    else
      rerank_scan_clauses<3>(p.vsegs, p.n_segs, dim, metric, s_q, qs, NC, cdoc, cseg, n, readable, s_vs, p.max_cand,
                             nullptr, 0u, lane, wv);
  } else {
    // ---- other dimensions: one candidate per wave at a time; its row is read ONCE into registers
    //      (dim <= 1024) and every clause runs against it ----
    typedef const __attribute__((address_space(1))) float *gf_t;
    constexpr int RV = 16;
    for (uint32_t c = wave; c < n; c += 4) {
      const float *row = row_of(c);
      float rv[RV];
      const bool cached = dim <= 64u * RV;
      if (row && cached) {
#pragma unroll
        for (int j = 0; j < RV; j++) {
          const uint32_t i = lane + 64u * j;
          rv[j] = i < dim ? ((gf_t)row)[i] : 0.0f;
        }
      }
      for (uint32_t cc = 0; cc < NC; cc++) {
        float acc = 0.0f;
        if (row && cached) {
#pragma unroll
          for (int j = 0; j < RV; j++) {
            const uint32_t i = lane + 64u * j;
            if (64u * j < dim) {  // uniform
              const float a = i < dim ? s_q[cc * qs + i] : 0.0f;
              if (metric == 0) {
                acc += a * rv[j];
              } else {
                const float d = i < dim ? a - rv[j] : 0.0f;
                acc += d * d;
              }
            }
          }
        } else if (row) {
          for (uint32_t i = lane; i < dim; i += 64) {
            const float a = s_q[cc * qs + i], bb = ((gf_t)row)[i];
            if (metric == 0) {
              acc += a * bb;
            } else {
              const float d = a - bb;
              acc += d * d;
            }
          }
        }
        const float sum = wave_sum_f(acc);
        if (lane == 0) s_vs[cc * p.max_cand + c] = metric == 0 ? (sum != sum ? 0.0f : sum) : -sqrtf(sum);
      }
    }
  }
  __syncthreads();
  // ---- compute_hybrid_score (api/reader.rs:225-254) per candidate, clauses in order ----
  for (uint32_t c = threadIdx.x; c < n; c += 256) {
    const bool has = row_of(c) != nullptr;
    const float bm = cbm[c];
    float blended_sum = 0.0f, vector_sum = 0.0f;
    for (uint32_t cc = 0; cc < NC; cc++) {
      const float alpha = p.alpha[(size_t)q * NC + cc];
      float vs;
      if (has) {
        vs = s_vs[cc * p.max_cand + c] * (mp.boost ? mp.boost[(size_t)q * NC + cc] : 1.0f);
        vector_sum += vs;
      } else {
        vs = metric == 0 ? -1.0f : -3.40282347e+38f;  // missing_vector_score
      }
      float blended;
      if (alpha >= 1.0f)
        blended = bm;
      else if (alpha <= 0.0f)
        blended = vs;
      else
        blended = alpha * bm + (1.0f - alpha) * vs;
      blended_sum += blended;
    }
    s_blend[c] = blended_sum / (float)NC;
    s_vsum[c] = has ? vector_sum : (metric == 0 ? -1.0f : -3.40282347e+38f);
  }
  __syncthreads();
  if (wave != 0) return;

  rerank_emit_topk<KREGS>(p, q, n, s_blend, s_vsum, cdoc, cseg, lane);
}

// ---- hybrid rerank whose clauses name DIFFERENT vector fields (api/reader.rs:225-254: every clause
// has its own field, metric and dimension; a doc may have a vector in one field and none in
// another — then only that clause takes the missing-vector score, and the reported vector score
// sums the clauses that found one).  One candidate per wave at a time, clause after clause, each
// against its own field's row: a plain VALU path (requests of this shape are rare; clauses over ONE
// field take rerank_multi_kernel). ----
struct RerankFieldsParams {
  RerankParams base;           // vsegs / dim unused; qvecs: [nq][q_floats]; alpha: [nq][n_clauses]
  const float *boost;          // [nq][n_clauses] or nullptr (1.0)
  uint32_t n_clauses;          // 1..8
  uint32_t q_floats;           // floats of one query's clause vectors, clause after clause
  const VecSegDev *cvsegs[8];  // per clause: the per-segment stores of its field
  uint32_t cdim[8], coff[8];   // per clause: dimension, offset of its vector inside q_floats
  int32_t cmetric[8];
};

template <int KREGS>
__global__ void __launch_bounds__(256) rerank_fields_kernel(RerankFieldsParams fp) {
  const RerankParams &p = fp.base;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef const __attribute__((address_space(1))) float *gf_t;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t q = blockIdx.x;
  const uint32_t NC = fp.n_clauses;
  uint32_t n = p.cand_count[q];
  n = n < p.max_cand ? n : p.max_cand;
  float *s_q = reinterpret_cast<float *>(smem);             // [q_floats]
  float *s_vs = s_q + ((fp.q_floats + 3u) & ~3u);           // [NC][max_cand] similarity (no boost yet)
  float *s_blend = s_vs + NC * p.max_cand;                  // [max_cand]
  float *s_vsum = s_blend + p.max_cand;                     // [max_cand]
  uint32_t *s_has = reinterpret_cast<uint32_t *>(s_vsum + p.max_cand);  // [max_cand] bit c: clause c found a vector
  const uint32_t *cdoc = p.cand_doc + (size_t)q * p.max_cand;
  const uint32_t *cseg = p.cand_seg + (size_t)q * p.max_cand;
  const float *cbm = p.cand_bm25 + (size_t)q * p.max_cand;
  for (uint32_t i = threadIdx.x; i < fp.q_floats; i += 256) s_q[i] = p.qvecs[(size_t)q * fp.q_floats + i];
  __syncthreads();

  for (uint32_t c = threadIdx.x; c < n; c += 256) s_has[c] = 0u;
  __syncthreads();
  // clause after clause: every clause is a row gather from ITS field's store; where the dimension
  // allows, with the four-rows-in-flight scan of the single-field kernels
  uint32_t slow_clauses = 0;
  for (uint32_t cc = 0; cc < NC; cc++) {
    const uint32_t dim = fp.cdim[cc];
    if ((dim & 3u) != 0 || dim > 768u || (fp.coff[cc] & 3u) != 0) {
      slow_clauses |= 1u << cc;
      continue;
    }
    const float *readable = p.qvecs + (size_t)q * fp.q_floats + fp.coff[cc];
    if (dim <= 256u)
      rerank_scan_clauses<1>(fp.cvsegs[cc], p.n_segs, dim, fp.cmetric[cc], s_q + fp.coff[cc], 0u, 1u, cdoc, cseg, n,
                             readable, s_vs + cc * p.max_cand, p.max_cand, s_has, 1u << cc, lane, wave);
    else if (dim <= 512u)
      rerank_scan_clauses<2>(fp.cvsegs[cc], p.n_segs, dim, fp.cmetric[cc], s_q + fp.coff[cc], 0u, 1u, cdoc, cseg, n,
                             readable, s_vs + cc * p.max_cand, p.max_cand, s_has, 1u << cc, lane, wave);
    else
      rerank_scan_clauses<3>(fp.cvsegs[cc], p.n_segs, dim, fp.cmetric[cc], s_q + fp.coff[cc], 0u, 1u, cdoc, cseg, n,
                             readable, s_vs + cc * p.max_cand, p.max_cand, s_has, 1u << cc, lane, wave);
  }
  __syncthreads();  // (the two loops spread the candidates over the waves differently: s_has is shared)
  for (uint32_t c = wave; slow_clauses != 0u && c < n; c += 4) {
    const uint32_t doc = cdoc[c], seg = cseg[c];
    uint32_t has = 0;
    for (uint32_t cc = 0; cc < NC; cc++) {
      if (!((slow_clauses >> cc) & 1u)) continue;
      const uint32_t dim = fp.cdim[cc];
      const int32_t metric = fp.cmetric[cc];
      const float *row = nullptr;
      if (seg < p.n_segs) {
        const VecSegDev vd = fp.cvsegs[cc][seg];
        if (vd.dim == dim && doc < vd.n_docs) {
          const uint32_t off = vd.offsets[doc];
          if (off != 0xFFFFFFFFu) row = vd.values + (size_t)off * dim;
        }
      }
      if (row == nullptr) continue;
      const float *qc = s_q + fp.coff[cc];
      float acc = 0.0f;
      for (uint32_t i = lane; i < dim; i += 64) {
        const float a = qc[i], bb = ((gf_t)row)[i];
        if (metric == 0) {
          acc += a * bb;
        } else {
          const float d = a - bb;
          acc += d * d;
        }
      }
      const float sum = wave_sum_f(acc);
      if (lane == 0) s_vs[cc * p.max_cand + c] = metric == 0 ? (sum != sum ? 0.0f : sum) : -sqrtf(sum);
      has |= 1u << cc;
    }
    if (lane == 0) s_has[c] |= has;
  }
  __syncthreads();
  // ---- compute_hybrid_score (api/reader.rs:225-254) per candidate, clauses in order ----
  for (uint32_t c = threadIdx.x; c < n; c += 256) {
    const uint32_t has = s_has[c];
    const float bm = cbm[c];
    float blended_sum = 0.0f, vector_sum = 0.0f;
    for (uint32_t cc = 0; cc < NC; cc++) {
      const float alpha = p.alpha[(size_t)q * NC + cc];
      float vs;
      if ((has >> cc) & 1u) {
        vs = s_vs[cc * p.max_cand + c] * (fp.boost ? fp.boost[(size_t)q * NC + cc] : 1.0f);
        vector_sum += vs;
      } else {
        vs = fp.cmetric[cc] == 0 ? -1.0f : -3.40282347e+38f;  // missing_vector_score of THIS clause
      }
      float blended;
      if (alpha >= 1.0f)
        blended = bm;
      else if (alpha <= 0.0f)
        blended = vs;
      else
        blended = alpha * bm + (1.0f - alpha) * vs;
      blended_sum += blended;
    }
    s_blend[c] = blended_sum / (float)NC;
    s_vsum[c] = has ? vector_sum : (fp.cmetric[0] == 0 ? -1.0f : -3.40282347e+38f);
  }
  __syncthreads();
  if (wave != 0) return;
  rerank_emit_topk<KREGS>(p, q, n, s_blend, s_vsum, cdoc, cseg, lane);
}

inline size_t rerank_fields_lds_floats(uint32_t n_clauses, uint32_t q_floats, uint32_t max_cand) {
  return (size_t)((q_floats + 3u) & ~3u) + (size_t)n_clauses * max_cand + 3 * (size_t)max_cand;
}

inline size_t rerank_multi_lds_floats(uint32_t n_clauses, uint32_t dim, uint32_t max_cand) {
  return (size_t)n_clauses * (dim + 4) + (size_t)n_clauses * max_cand + 2 * (size_t)max_cand + 8;  // (+ clause norms)
}

template <typename K, typename P>
inline hipError_t launch_with_lds(K kernel, const P &params, uint32_t nq, size_t lds, hipStream_t st) {
  if (lds > 48 * 1024) {  // above the default dynamic-LDS limit: opt in (up to the CU's 160 KiB)
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kernel, dim3(nq), dim3(256), lds, st, params);
  return hipGetLastError();
}

inline hipError_t launch_rerank_multi(const RerankMultiParams &mp, int kregs, hipStream_t st) {
  const size_t lds = rerank_multi_lds_floats(mp.n_clauses, mp.base.dim, mp.base.max_cand) * 4 + 16;
  switch (kregs) {
    case 1: return launch_with_lds(rerank_multi_kernel<1>, mp, mp.base.nq, lds, st);
    case 2: return launch_with_lds(rerank_multi_kernel<2>, mp, mp.base.nq, lds, st);
    case 4: return launch_with_lds(rerank_multi_kernel<4>, mp, mp.base.nq, lds, st);
    case 8: return launch_with_lds(rerank_multi_kernel<8>, mp, mp.base.nq, lds, st);
    default: return launch_with_lds(rerank_multi_kernel<16>, mp, mp.base.nq, lds, st);
  }
}

inline hipError_t launch_rerank_fields(const RerankFieldsParams &fp, int kregs, hipStream_t st) {
  const size_t lds = rerank_fields_lds_floats(fp.n_clauses, fp.q_floats, fp.base.max_cand) * 4 + 16;
  switch (kregs) {
    case 1: return launch_with_lds(rerank_fields_kernel<1>, fp, fp.base.nq, lds, st);
    case 2: return launch_with_lds(rerank_fields_kernel<2>, fp, fp.base.nq, lds, st);
    case 4: return launch_with_lds(rerank_fields_kernel<4>, fp, fp.base.nq, lds, st);
    case 8: return launch_with_lds(rerank_fields_kernel<8>, fp, fp.base.nq, lds, st);
    default: return launch_with_lds(rerank_fields_kernel<16>, fp, fp.base.nq, lds, st);
  }
}

inline hipError_t launch_rerank(const RerankParams &rp, int kregs, hipStream_t st) {
  const size_t lds = (size_t)rp.max_cand * 8 + 16;
  switch (kregs) {
    case 1: return launch_with_lds(rerank_kernel<1>, rp, rp.nq, lds, st);
    case 2: return launch_with_lds(rerank_kernel<2>, rp, rp.nq, lds, st);
    case 4: return launch_with_lds(rerank_kernel<4>, rp, rp.nq, lds, st);
    case 8: return launch_with_lds(rerank_kernel<8>, rp, rp.nq, lds, st);
    default: return launch_with_lds(rerank_kernel<16>, rp, rp.nq, lds, st);
  }
}

}  // namespace slg
