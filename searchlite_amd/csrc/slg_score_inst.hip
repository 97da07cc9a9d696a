// slg_score_inst.hip — one translation unit per top-k register width (SLG_INST_KREGS), so the
// scoring-kernel instantiations compile in parallel.  slg_api.hip calls
// slg::launch_score_kregs<N>() declared below.
#include <hip/hip_runtime.h>

#include "slg_score.hpp"
#ifdef SLG_LEGACY_KERNELS  // the superseded few-term forms, for A/B timing on one device (tools/ab_uniform.sh)
#include "slg_score_uni.hpp"
#include "slg_score_uni3.hpp"
#endif
#include "slg_score_uni4.hpp"
#include "slg_score_multi.hpp"

#ifndef SLG_INST_KREGS
#error "compile with -DSLG_INST_KREGS={1,2,4,8,16}"
#endif

namespace slg {

template <int KREGS>
void launch_score_kregs(const RoundScoreParams &sp, int kind, hipStream_t st);

// Waves are independent in both kernels (no workgroup barrier anywhere), so workgroups are ONE
// wave: a finished wave frees its wave slot and its LDS at once.
template <>
void launch_score_kregs<SLG_INST_KREGS>(const RoundScoreParams &sp, int kind, hipStream_t st) {
#ifdef SLG_LEGACY_KERNELS
  if (kind == 1) {  // <= 4 lists (slg_score_uni3.hpp)
    hipLaunchKernelGGL((score_uniform3_kernel<SLG_INST_KREGS, 4>), dim3(sp.n_slices), dim3(64),
                       u3_wave_lds(SLG_INST_KREGS, 4), st, sp);
    return;
  }
  if (kind == 5) {  // 5..8 lists: the same kernel with 8 list bits per filter field
    hipLaunchKernelGGL((score_uniform3_kernel<SLG_INST_KREGS, 8>), dim3(sp.n_slices), dim3(64),
                       u3_wave_lds(SLG_INST_KREGS, 8), st, sp);
    return;
  }
  if (kind == 4) {  // the round-2 form of the same kernel (slg_tuning.uniform_kernel = 2: A/B timing)
    hipLaunchKernelGGL((score_uniform_kernel<SLG_INST_KREGS>), dim3(sp.n_slices), dim3(64),
                       uni_wave_lds(SLG_INST_KREGS), st, sp);
    return;
  }
#endif
  if (kind == 6 || kind == 7) {  // <= 4 / 5..8 lists, blocked layout (slg_score_uni4.hpp)
    // one wave per slice (sp.work_ctr == nullptr), or persistent waves: sp.n_waves / kU4WavesPerBlock
    // workgroups (u4_launch_blocks, slg_api.hip), each wave pulls slices from sp.work_ctr
    const uint32_t blocks = (sp.n_waves + (uint32_t)kU4WavesPerBlock - 1u) / (uint32_t)kU4WavesPerBlock;
    const dim3 grid(blocks), wg(64 * kU4WavesPerBlock);
    const size_t lds4 = kU4WavesPerBlock * u4_wave_lds(SLG_INST_KREGS, 4, u4_filter_words(4));
    const size_t lds8 = kU4WavesPerBlock * u4_wave_lds(SLG_INST_KREGS, 8, u4_filter_words(8));
    if (sp.work_ctr == nullptr) {
      if (kind == 6)
        hipLaunchKernelGGL((score_uniform4_kernel<SLG_INST_KREGS, 4, false, false>), grid, wg, lds4, st, sp);
      else
        hipLaunchKernelGGL((score_uniform4_kernel<SLG_INST_KREGS, 8, false, false>), grid, wg, lds8, st, sp);
    } else {
      if (kind == 6)
        hipLaunchKernelGGL((score_uniform4_kernel<SLG_INST_KREGS, 4, false, true>), grid, wg, lds4, st, sp);
      else
        hipLaunchKernelGGL((score_uniform4_kernel<SLG_INST_KREGS, 8, false, true>), grid, wg, lds8, st, sp);
    }
    return;
  }
  if (kind == 8 || kind == 9) {  // the same kernel with score plans (flat Sum / DisMax over multi-term leaves)
    const uint32_t blocks = (sp.n_waves + (uint32_t)kU4WavesPerBlock - 1u) / (uint32_t)kU4WavesPerBlock;
    if (kind == 8)
      hipLaunchKernelGGL((score_uniform4_kernel<SLG_INST_KREGS, 4, true>), dim3(blocks), dim3(64 * kU4WavesPerBlock),
                         kU4WavesPerBlock * u4_wave_lds(SLG_INST_KREGS, 4, u4_filter_words(4)), st, sp);
    else
      hipLaunchKernelGGL((score_uniform4_kernel<SLG_INST_KREGS, 8, true>), dim3(blocks), dim3(64 * kU4WavesPerBlock),
                         kU4WavesPerBlock * u4_wave_lds(SLG_INST_KREGS, 8, u4_filter_words(8)), st, sp);
    return;
  }
  // many lists: slots of one list each, 8 at a time (slg_score_multi.hpp)
  const size_t lds = (size_t)multi_wave_lds(SLG_INST_KREGS) + (size_t)sp.plan_batch * kMultiPlanLds;
  if (sp.plan_batch == 4)  // score trees of more than two levels
    hipLaunchKernelGGL((score_multi_kernel<SLG_INST_KREGS, 4>), dim3(sp.n_slices), dim3(64), lds, st, sp);
  else if (sp.plan_batch == 2)  // two-level score plans
    hipLaunchKernelGGL((score_multi_kernel<SLG_INST_KREGS, 3>), dim3(sp.n_slices), dim3(64), lds, st, sp);
  else if (sp.plan_batch)  // score plans (never together with pruning)
    hipLaunchKernelGGL((score_multi_kernel<SLG_INST_KREGS, 2>), dim3(sp.n_slices), dim3(64), lds, st, sp);
  else if (kind == 3)  // pruning-classified batch
    hipLaunchKernelGGL((score_multi_kernel<SLG_INST_KREGS, 1>), dim3(sp.n_slices), dim3(64), lds, st, sp);
  else
    hipLaunchKernelGGL((score_multi_kernel<SLG_INST_KREGS, 0>), dim3(sp.n_slices), dim3(64), lds, st, sp);
}

}  // namespace slg
