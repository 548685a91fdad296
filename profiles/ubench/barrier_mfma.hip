// Micro-benchmark (gfx950): cost of s_barrier in an 8-wave workgroup, and of 8 v_mfma_f32_32x32x16_f16 per wave
// between barriers with 2 vs 4 accumulators in rotation, lockstep vs two groups staggered by one barrier.
//   hipcc --offload-arch=gfx950 -O3 barrier_mfma.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define BAR() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int NACC, int NMFMA, bool STAGGER>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  f16v acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f16v{0};
  if (STAGGER && wave >= 4) BAR();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    BAR();
#pragma unroll
    for (int j = 0; j < NMFMA; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j % NACC], 0, 0, 0);
    BAR();
  }
  long long t1 = __builtin_readcyclecounter();
  if (STAGGER && wave < 4) BAR();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int NMFMA, bool STAGGER>
void run(const char* name, int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, NMFMA, STAGGER><<<blocks, 512>>>(out, cyc, 10);
  hipEventRecord(e0);
  k<NACC, NMFMA, STAGGER><<<blocks, 512>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s blocks=%d: %.1f ns / iteration (2 barriers + %d MFMA per wave); cycle counter %.0f ticks/iter\n", name, blocks,
         ms * 1e6 / iters, NMFMA, (double)c / iters);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int blocks : {1, 256}) {
    run<1, 0, false>("barriers only", blocks);
    run<2, 8, false>("8 MFMA, 2 accumulators, lockstep", blocks);
    run<4, 8, false>("8 MFMA, 4 accumulators, lockstep", blocks);
    run<2, 8, true>("8 MFMA, 2 accumulators, staggered groups", blocks);
    run<4, 8, true>("8 MFMA, 4 accumulators, staggered groups", blocks);
    run<4, 16, true>("16 MFMA, 4 accumulators, staggered groups", blocks);
    run<1, 8, false>("8 MFMA, 1 accumulator, lockstep", blocks);
  }
  return 0;
}
