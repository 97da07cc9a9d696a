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

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one workgroup (4 waves) per query
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

  for (uint32_t c = wave; c < n; c += 4) {
    const uint32_t doc = cdoc[c], seg = cseg[c];
    float vs;
    int32_t metric = 0;
    bool have = false;
    const float *row = nullptr;
    if (seg < p.n_segs) {
      const VecSegDev vd = p.vsegs[seg];
      metric = vd.metric;
      if (vd.dim == dim && doc < vd.n_docs) {
        const uint32_t off = vd.offsets[doc];
        if (off != 0xFFFFFFFFu) {
          have = true;
          row = vd.values + (size_t)off * dim;
        }
      }
    }
    if (have) {
      float acc = 0.0f;
      if ((dim & 3u) == 0) {
        for (uint32_t i = lane * 4; i < dim; i += 256) {
          const float4 a = *reinterpret_cast<const float4 *>(qv + i);
          const float4 b = *reinterpret_cast<const float4 *>(row + i);
          if (metric == 0) {
            acc += a.x * b.x;
            acc += a.y * b.y;
            acc += a.z * b.z;
            acc += a.w * b.w;
          } else {
            float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
            acc += d0 * d0;
            acc += d1 * d1;
            acc += d2 * d2;
            acc += d3 * d3;
          }
        }
      } else {
        for (uint32_t i = lane; i < dim; i += 64) {
          const float a = qv[i], b = row[i];
          if (metric == 0) {
            acc += a * b;
          } else {
            const float d = a - b;
            acc += d * d;
          }
        }
      }
      acc = wave_sum_f(acc);
      if (metric == 0)
        vs = (acc != acc) ? 0.0f : acc;  // vectors/mod.rs:112-116 NaN -> 0
      else
        vs = -sqrtf(acc);  // vectors/mod.rs:118
    } else {
      vs = metric == 0 ? -1.0f : -3.40282347e+38f;  // api/reader.rs:217-223
    }
    if (lane == 0) {
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
  __syncthreads();
  if (wave != 0) return;

  const uint32_t k = p.k_out;
  if (k == 0) {
    if (lane == 0) p.out_count[q] = 0;
    return;
  }
  WaveTopK<KREGS, true> top;
  top.init();
  // payload: candidate index rides in a parallel array keyed by (seg,doc) lookup below
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
      m &= m - 1;
      const int32_t c_tk = (int32_t)rl((uint32_t)ctk, l);
      const uint32_t c_doc = rl(d, l), c_seg = rl(sg, l);
      if (!top.passes(c_tk, c_seg, c_doc)) continue;
      top.insert(c_tk, c_seg, c_doc, k, lane);
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
      if (p.out_vec) {
        // recover the vector score of this (seg, doc): linear scan of the candidate list
        float v = 0.0f;
        if (real)
          for (uint32_t i = 0; i < n; i++)
            if (cdoc[i] == top.doc[r] && cseg[i] == top.seg[r]) {
              v = s_vec[i];
              break;
            }
        p.out_vec[(size_t)q * k + pos] = v;
      }
    }
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
