// slg_rerank.hpp — dense f32 vector rerank (the slot of searchlite-core/src/gpu/rerank.rs:3).
//
// Arithmetic restated: vectors/mod.rs:63-71 (VectorStore::vector), :107-120
// (metric_similarity), :122-129 (blend_scores); api/reader.rs:217-223 (missing vector
// score) and :225-254 (compute_hybrid_score, one clause).
//
// The path is an HBM row gather (one D-float row per candidate, 0.5 flop/byte): each wave
// streams whole rows with 16-byte lane loads, reduces in-register, and a wave-wide sorted
// top-k (slg_kernels.hpp) picks the k_out best blended scores.
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

// One workgroup (4 waves) per query.  Each wave takes candidates c = wave, wave+4, ... four at
// a time: the 4 rows' 16-byte lane loads are issued together (rows are >= 1 KiB apart: an HBM
// row gather), then reduced in registers.
template <int KREGS>
__global__ void __launch_bounds__(256) rerank_kernel(RerankParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *s_blend = reinterpret_cast<float *>(smem);
  float *s_vec = s_blend + p.max_cand;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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

  for (uint32_t c0 = wave * U; c0 < n; c0 += 4 * U) {
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

inline void launch_rerank(const RerankParams &rp, int kregs, hipStream_t st) {
  const size_t lds = (size_t)rp.max_cand * 8 + 16;
  dim3 grid(rp.nq), block(256);
  switch (kregs) {
    case 1: hipLaunchKernelGGL((rerank_kernel<1>), grid, block, lds, st, rp); break;
    case 2: hipLaunchKernelGGL((rerank_kernel<2>), grid, block, lds, st, rp); break;
    case 4: hipLaunchKernelGGL((rerank_kernel<4>), grid, block, lds, st, rp); break;
    case 8: hipLaunchKernelGGL((rerank_kernel<8>), grid, block, lds, st, rp); break;
    default: hipLaunchKernelGGL((rerank_kernel<16>), grid, block, lds, st, rp); break;
  }
}

}  // namespace slg
