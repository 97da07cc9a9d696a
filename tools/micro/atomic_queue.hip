// Microbenchmark: a work counter shared by all waves of the device (persistent waves pulling slices).
// 6144 one-wave workgroups; every wave pulls `per_wave` tickets with a returning global atomic add by
// lane 0, spaced by `work` dependent VALU iterations.  Reports ns per ticket for 1 .. 64 counters
// (each on its own 256-byte line; a wave uses counter wave % n_counters) and checks that every ticket
// was handed out exactly once (sum of tickets).
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_queue atomic_queue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

__global__ void __launch_bounds__(64) pull(unsigned *ctr, unsigned n_ctr, unsigned per_wave, unsigned work,
                                           unsigned long long *sum, float *sink) {
  const unsigned lane = threadIdx.x & 63;
  unsigned *my = ctr + (blockIdx.x % n_ctr) * 64;  // 256 bytes apart
  unsigned long long acc = 0;
  float x = (float)lane;
  for (unsigned i = 0; i < per_wave; i++) {
    unsigned t = 0;
    if (lane == 0) t = atomicAdd(my, 1u);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    acc += t;
    for (unsigned j = 0; j < work; j++) x = x * 1.0001f + 0.5f;
  }
  if (lane == 0) atomicAdd(sum, acc);
  if (x == 12345.678f) sink[0] = x;
}

int main() {
  unsigned *ctr;
  unsigned long long *sum;
  float *sink;
  CHECK(hipMalloc(&ctr, 64 * 256));
  CHECK(hipMalloc(&sum, 8));
  CHECK(hipMalloc(&sink, 4));
  const unsigned waves = 6144;
  for (unsigned work : {0u, 2000u, 20000u}) {
    for (unsigned per_wave : {2u, 50u}) {
      for (unsigned n_ctr : {1u, 8u, 64u}) {
        CHECK(hipMemset(ctr, 0, 64 * 256));
        CHECK(hipMemset(sum, 0, 8));
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipLaunchKernelGGL(pull, dim3(waves), dim3(64), 0, 0, ctr, n_ctr, 1u, 0u, sum, sink);  // warm
        CHECK(hipMemset(ctr, 0, 64 * 256));
        CHECK(hipMemset(sum, 0, 8));
        hipEventRecord(a);
        hipLaunchKernelGGL(pull, dim3(waves), dim3(64), 0, 0, ctr, n_ctr, per_wave, work, sum, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long got = 0;
        CHECK(hipMemcpy(&got, sum, 8, hipMemcpyDeviceToHost));
        // expected: per counter c, tickets 0 .. n_c*per_wave-1 where n_c = waves using it
        unsigned long long want = 0;
        for (unsigned c = 0; c < n_ctr; c++) {
          unsigned long long n = (unsigned long long)((waves - c + n_ctr - 1) / n_ctr) * per_wave;
          want += n * (n - 1) / 2;
        }
        printf("work %5u per_wave %3u counters %2u: %8.1f us total, %7.2f ns per ticket, tickets %s\n", work, per_wave,
               n_ctr, ms * 1e3, ms * 1e6 / ((double)waves * per_wave), got == want ? "unique" : "DUPLICATED/LOST");
      }
    }
  }
  return 0;
}
