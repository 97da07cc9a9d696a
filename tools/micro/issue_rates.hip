// Microbenchmark: wave-instruction issue rates on gfx950 for the instruction classes the scoring
// kernels are made of (VALU, SALU, v_readlane, v_cmp -> SGPR mask, v_cndmask with an SGPR mask,
// LDS ops) at 1..8 waves per SIMD, alone and interleaved.  Answers: is a kernel with N VALU and
// M SALU wave-instructions bounded by the sum or by the max of the two, and what does one
// wave-instruction cost per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o issue_rates tools/micro/issue_rates.hip && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

constexpr int kUnroll = 16;  // asm groups per loop iteration; each group = 4 instructions of the pattern

// One group = 4 wave-instructions (what "instr" counts below).
template <int PAT>
__device__ __forceinline__ void group(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d, uint32_t &s0,
                                      uint32_t &s1, uint32_t &s2, uint32_t &s3, uint64_t &m0, uint64_t &m1,
                                      uint32_t lds_addr) {
  if constexpr (PAT == 0) {  // 4 independent VALU adds
    asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(lds_addr));
  } else if constexpr (PAT == 1) {  // 4 independent SALU adds
    asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1"
                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");
  } else if constexpr (PAT == 2) {  // 2 VALU + 2 SALU interleaved
    asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 %2, %2, 1\n v_add_u32 %1, %1, %4\n s_add_u32 %3, %3, 1"
                 : "+v"(a), "+v"(b), "+s"(s0), "+s"(s1) : "v"(lds_addr) : "scc");
  } else if constexpr (PAT == 3) {  // v_readlane x4 (VALU op writing an SGPR)
    asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 7\n v_readlane_b32 %2, %6, 11\n v_readlane_b32 %3, %7, 13"
                 : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a), "v"(b), "v"(c), "v"(d));
  } else if constexpr (PAT == 4) {  // v_cmp -> SGPR pair x4
    asm volatile("v_cmp_lt_u32 %0, %2, %3\n v_cmp_lt_u32 %1, %3, %4\n v_cmp_lt_u32 %0, %4, %5\n v_cmp_lt_u32 %1, %5, %2"
                 : "=s"(m0), "=s"(m1) : "v"(a), "v"(b), "v"(c), "v"(d));
  } else if constexpr (PAT == 5) {  // v_cndmask with SGPR mask x4
    asm volatile("v_cndmask_b32 %0, %0, %4, %5\n v_cndmask_b32 %1, %1, %4, %6\n v_cndmask_b32 %2, %2, %4, %5\n v_cndmask_b32 %3, %3, %4, %6"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(lds_addr), "s"(m0), "s"(m1));
  } else if constexpr (PAT == 6) {  // v_cmp -> mask, then dependent v_cndmask (the kernels' usual pair) x2
    asm volatile("v_cmp_lt_u32 %2, %0, %4\n v_cndmask_b32 %0, %0, %1, %2\n v_cmp_lt_u32 %3, %1, %4\n v_cndmask_b32 %1, %1, %0, %3"
                 : "+v"(a), "+v"(b), "=&s"(m0), "=&s"(m1) : "v"(c));
  } else if constexpr (PAT == 7) {  // s_and_b64 / s_bcnt1 / s_or_b64 / s_andn2_b64 (mask arithmetic)
    asm volatile("s_and_b64 %0, %0, %1\n s_bcnt1_i32_b64 %2, %0\n s_or_b64 %1, %1, %0\n s_andn2_b64 %0, %1, %0"
                 : "+s"(m0), "+s"(m1), "=s"(s0) :: "scc");
  } else if constexpr (PAT == 8) {  // ds_read_b32 x4, waited per group
    asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(lds_addr) : "memory");
  } else if constexpr (PAT == 9) {  // ds_or_b32 without return x4
    asm volatile("ds_or_b32 %0, %1\n ds_or_b32 %0, %2 offset:256\n ds_or_b32 %0, %3 offset:512\n ds_or_b32 %0, %1 offset:768"
                 :: "v"(lds_addr), "v"(a), "v"(b), "v"(c) : "memory");
  } else if constexpr (PAT == 10) {  // 1 ds_read + 3 VALU (LDS beside VALU)
    asm volatile("ds_read_b32 %3, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4"
                 : "+v"(a), "+v"(b), "+v"(c), "=v"(d) : "v"(lds_addr) : "memory");
  } else if constexpr (PAT == 11) {  // v_readlane + dependent SALU + VALU using the SGPR (hazard pattern)
    asm volatile("v_readlane_b32 %2, %0, 5\n s_add_u32 %3, %2, 1\n v_add_u32 %0, %0, %3\n v_add_u32 %1, %1, %2"
                 : "+v"(a), "+v"(b), "=&s"(s0), "=&s"(s1) :: "scc");
  } else if constexpr (PAT == 12) {  // 3 VALU + 1 SALU
    asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n s_add_u32 %3, %3, 1\n v_add_u32 %2, %2, %4"
                 : "+v"(a), "+v"(b), "+v"(c), "+s"(s0) : "v"(lds_addr) : "scc");
  } else if constexpr (PAT == 13) {  // 1 VALU + 3 SALU
    asm volatile("v_add_u32 %0, %0, %4\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1"
                 : "+v"(a), "+s"(s0), "+s"(s1), "+s"(s2) : "v"(lds_addr) : "scc");
  } else if constexpr (PAT == 14) {  // v_mul_f32 / v_cmp_ge_f32 / v_lshrrev / v_and (the per-posting mix)
    asm volatile("v_mul_f32 %0, %0, %3\n v_lshrrev_b32 %1, 8, %0\n v_and_b32 %2, 0x3ff, %1\n v_lshlrev_b32 %1, %2, %3"
                 : "+v"(a), "+v"(b), "+v"(c) : "v"(d));
  }
}

template <int PAT>
__global__ void __launch_bounds__(256) k(uint32_t *out, unsigned long long *cyc, int iters) {
  extern __shared__ uint32_t lds[];
  const uint32_t lane = threadIdx.x & 63;
  uint32_t a = lane, b = lane * 3, c = lane * 5, d = lane * 7;
  uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
  uint64_t m0 = 0x5555555555555555ull, m1 = 0x3333333333333333ull;
  const uint32_t lds_addr = (threadIdx.x & 255) * 4;
  lds[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < kUnroll; u++) group<PAT>(a, b, c, d, s0, s1, s2, s3, m0, m1, lds_addr);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + s0 + s1 + s2 + s3 + (uint32_t)m0 + (uint32_t)m1;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int PAT>
int run(const char *name, uint32_t *d_out, unsigned long long *d_cyc) {
  const int iters = 2000;
  printf("%-44s", name);
  for (int W : {1, 2, 4, 6, 8}) {
    // W 256-thread workgroups per CU = W waves per SIMD: LDS sized so that exactly W fit
    const int lds_bytes = (160 * 1024 / W) & ~1023;
    const int blocks = 256 * W;
    CHECK(hipFuncSetAttribute((const void *)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), lds_bytes, 0, d_out, d_cyc, 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(blocks), dim3(256), lds_bytes, 0, d_out, d_cyc, iters);
    hipEventRecord(e1);
    CHECK(hipEventSynchronize(e1));
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(blocks * 4);
    CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    std::sort(cyc.begin(), cyc.end());
    const double med = (double)cyc[cyc.size() / 2];
    const double instr_per_wave = (double)iters * kUnroll * 4;
    // SIMD-cycles per wave-instruction = (cycles a wave took) / (instr per wave) / (waves per SIMD)
    printf("  W%d: %5.2f cyc/instr/SIMD (wave: %5.2f; %.0f us)", W, med / instr_per_wave / W, med / instr_per_wave, ms * 1e3);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  printf("\n");
  return 0;
}

int main() {
  uint32_t *d_out;
  unsigned long long *d_cyc;
  CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
  CHECK(hipMalloc(&d_cyc, 256 * 8 * 4 * 8));
  printf("cycles per wave-instruction per SIMD (s_memtime cycles of the median wave / instructions / waves per SIMD)\n");
  run<0>("valu: 4 x v_add_u32", d_out, d_cyc);
  run<1>("salu: 4 x s_add_u32", d_out, d_cyc);
  run<2>("2 valu + 2 salu interleaved", d_out, d_cyc);
  run<12>("3 valu + 1 salu", d_out, d_cyc);
  run<13>("1 valu + 3 salu", d_out, d_cyc);
  run<3>("4 x v_readlane_b32", d_out, d_cyc);
  run<4>("4 x v_cmp -> sgpr mask", d_out, d_cyc);
  run<5>("4 x v_cndmask (sgpr mask)", d_out, d_cyc);
  run<6>("2 x (v_cmp -> mask -> v_cndmask)", d_out, d_cyc);
  run<7>("salu 64-bit mask ops (and/bcnt/or/andn2)", d_out, d_cyc);
  run<11>("readlane -> s_add -> v_add (dependent)", d_out, d_cyc);
  run<14>("v_mul_f32 / shift / and / shift", d_out, d_cyc);
  run<8>("4 x ds_read_b32 + wait", d_out, d_cyc);
  run<9>("4 x ds_or_b32 (no return)", d_out, d_cyc);
  run<10>("1 ds_read_b32 + 3 valu", d_out, d_cyc);
  return 0;
}
