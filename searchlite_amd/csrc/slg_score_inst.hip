// slg_score_inst.hip — one translation unit per top-k register width (SLG_INST_KREGS), so
// the score_rounds_kernel instantiations compile in parallel.  slg_api.hip calls
// slg::launch_score_kregs<N>() declared below.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "slg_score.hpp"
#include "slg_score_uni.hpp"
#include "slg_score_multi.hpp"

#ifndef SLG_INST_KREGS
#error "compile with -DSLG_INST_KREGS={1,2,4,8,16}"
#endif

namespace slg {

template <int KREGS, int TT>
static void launch_tt(const RoundScoreParams &sp, hipStream_t st) {
  // waves are independent (no workgroup barrier): one-wave workgroups, as for the uniform kernel
  static const uint32_t wpb = [] {
    const char *e = getenv("SLG_PACKED_WAVES_PER_BLOCK");
    const uint32_t v = e ? (uint32_t)atoi(e) : 1u;
    return v >= 1 && v <= 4 ? v : 1u;
  }();
  const uint32_t blocks = (sp.n_slices + wpb - 1) / wpb;
  const size_t lds = (size_t)wpb * kScoreWaveLds;
  hipLaunchKernelGGL((score_rounds_kernel<KREGS, TT>), dim3(blocks), dim3(64 * wpb), lds, st, sp);
}

template <int KREGS>
void launch_score_kregs(const RoundScoreParams &sp, uint32_t max_terms, int kind, hipStream_t st);

template <>
void launch_score_kregs<SLG_INST_KREGS>(const RoundScoreParams &sp, uint32_t max_terms,
                                        int kind, hipStream_t st) {
  if (kind == 2 || kind == 3) {  // many lists: slots of one list each, 8 at a time (slg_score_multi.hpp)
    const size_t lds = (size_t)multi_wave_lds(SLG_INST_KREGS) + (sp.plan_batch ? kMultiPlanLds : 0);
    if (sp.plan_batch)  // score plans (never together with MaxScore)
      hipLaunchKernelGGL((score_multi_kernel<SLG_INST_KREGS, 2>), dim3(sp.n_slices), dim3(64), lds, st, sp);
    else if (kind == 3)  // MaxScore-classified batch
      hipLaunchKernelGGL((score_multi_kernel<SLG_INST_KREGS, 1>), dim3(sp.n_slices), dim3(64), lds, st, sp);
    else
      hipLaunchKernelGGL((score_multi_kernel<SLG_INST_KREGS, 0>), dim3(sp.n_slices), dim3(64), lds, st, sp);
    return;
  }
  if (kind == 1) {  // one list per register slot (slg_score_uni.hpp); waves are independent, so
                  // one-wave workgroups: a finished wave frees its slot and LDS at once
    static const uint32_t wpb = [] {
      const char *e = getenv("SLG_UNI_WAVES_PER_BLOCK");
      const uint32_t v = e ? (uint32_t)atoi(e) : 1u;
      return v >= 1 && v <= 4 ? v : 1u;
    }();
    const uint32_t blocks = (sp.n_slices + wpb - 1) / wpb;
    const size_t lds = (size_t)wpb * uni_wave_lds(SLG_INST_KREGS);
    hipLaunchKernelGGL((score_uniform_kernel<SLG_INST_KREGS>), dim3(blocks), dim3(64 * wpb), lds, st, sp);
    return;
  }
  if (max_terms <= 4)
    launch_tt<SLG_INST_KREGS, 4>(sp, st);
  else if (max_terms <= 8)
    launch_tt<SLG_INST_KREGS, 8>(sp, st);
  else
    launch_tt<SLG_INST_KREGS, 32>(sp, st);
}

}  // namespace slg
