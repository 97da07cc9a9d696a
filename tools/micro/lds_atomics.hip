// Microbenchmark: cost of LDS operations per wave-instruction on gfx950, 16 waves/CU.
// pattern 0: every lane a distinct word (stride 1); 1: 8 consecutive lanes share a word;
// 2: 32 lanes share a word; 3: random words.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
typedef __attribute__((address_space(3))) float* ldsf;
template<int OP> __global__ void __launch_bounds__(256) k(unsigned* out, int pattern, int iters){
  __shared__ unsigned lds[4096];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0;
  __syncthreads();
  unsigned base = wave * 1024;
  unsigned idx;
  if (pattern == 0) idx = lane;
  else if (pattern == 1) idx = lane >> 3;
  else if (pattern == 2) idx = lane >> 5;
  else idx = (lane * 2654435761u >> 7) & 1023;
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
    unsigned a = base + ((idx + it * 67) & 1023);
    if (OP == 0) acc += lds[a];                                   // ds_read_b32
    else if (OP == 1) lds[a] = acc + it;                          // ds_write_b32
    else if (OP == 2) atomicOr(&lds[a], 1u << (lane & 31));       // ds_or_b32 (no rtn)
    else if (OP == 3) acc += atomicOr(&lds[a], 1u << (lane & 31));// ds_or_rtn_b32
    else if (OP == 4) __builtin_amdgcn_ds_faddf((ldsf)(float*)&lds[a], 1.0f, 0, 0, false); // ds_add_f32
    else if (OP == 5) acc += atomicCAS(&lds[a], 0u, lane + 1);    // ds_cmpst_rtn
    else if (OP == 6) acc += __shfl(acc + it, (lane * 7 + it) & 63, 64); // ds_bpermute
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): dependent chain like the real kernel
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc + lds[base + lane];
}
template<int OP> int run(const char* name, unsigned* d){
  const int iters = 2000, blocks = 256 * 4;
  for (int pat = 0; pat < 4; pat++) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, pat, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, pat, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: 4 blocks x 4 waves = 16 waves resident, each does iters dependent ops
    double cyc_per_op_per_wave = ms * 1e-3 * 2.4e9 / iters;          // latency seen by one wave
    double cu_cycles_per_wave_instr = ms * 1e-3 * 2.4e9 / (iters * 16.0); // CU throughput
    printf("%-14s pattern %d: %.0f cycles/op per wave (16 waves/CU) => %.1f CU-cycles per wave-instr\n",
           name, pat, cyc_per_op_per_wave, cu_cycles_per_wave_instr);
  }
  return 0;
}
int main(){
  unsigned* d; CHECK(hipMalloc(&d, 256*4*256*4));
  run<0>("ds_read_b32", d); run<1>("ds_write_b32", d); run<2>("ds_or_b32", d); run<3>("ds_or_rtn_b32", d);
  run<4>("ds_add_f32", d); run<5>("ds_cmpst_rtn", d); run<6>("ds_bpermute", d);
  return 0;
}
