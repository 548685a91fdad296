// Diagnostic (gfx950): cycle stamps inside the K loop of dec_gemm_big_kernel (the LDS-staged decoder linear of merged
// decode runs) — where do the ~1 800 cycles of a k-step go?  Waves 0 and 4 (the two waves of one SIMD) of workgroup 0
// record (slot, cycle) for their first 192 stamps; the slots are the DGB_TL(n) marks in csrc/dec_kernels.hip:
//   lockstep loop (cfg 0 / 1 / 2):  0 top | 1 after the counted vmcnt wait | 2 after the barrier | 3 after the LDS-read issue
//                                   | (DMA issue) | 4 before the first MFMA | 5 after the MFMAs
//   (the staggered form of round 6 — commit 1d52bf7 — had its own marks; its timeline is in profiles/r06_dec_big_timeline.txt)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../faster_whisper_amd/csrc -I../../include dec_big_timeline.hip -o /tmp/dbt && /tmp/dbt
#include <hip/hip_runtime.h>
#define DGB_NTL 192
__device__ long long gb_tl[2][DGB_NTL];
__device__ int gb_slot[2][DGB_NTL];
#define DGB_TL_DECL int dgb_tl_i = 0
#define DGB_TL(slot_)                                                                        \
  do {                                                                                       \
    if (blockIdx.x == 0 && (threadIdx.x & 255) == 0 && dgb_tl_i < DGB_NTL) {                 \
      gb_tl[threadIdx.x >> 8][dgb_tl_i] = __builtin_readcyclecounter();                      \
      gb_slot[threadIdx.x >> 8][dgb_tl_i] = (slot_);                                         \
      ++dgb_tl_i;                                                                            \
    }                                                                                        \
  } while (0)
#include "../../faster_whisper_amd/csrc/dec_kernels.hip"
#include <stdio.h>
#include <string.h>
#include <vector>

static void run(int cfg, int R, int N, int K, bool lnf) {
  const size_t nx = (size_t)((R + 15) / 16 * 16) * K, nw = (size_t)N * K;
  std::vector<_Float16> h(nw > nx ? nw : nx);
  unsigned st = 1;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (_Float16)(((int)(st >> 16) % 2001 - 1000) * 1e-3f); }
  half_t *X, *W, *O, *B;
  float *S1, *CF;
  hipMalloc(&X, nx * 2); hipMalloc(&W, nw * 2 * 8); hipMalloc(&O, (size_t)R * N * 2); hipMalloc(&B, N * 2);
  hipMalloc(&S1, N * 4); hipMalloc(&CF, N * 4);
  hipMemcpy(X, h.data(), nx * 2, hipMemcpyHostToDevice);
  for (int c = 0; c < 8; ++c) hipMemcpy(W + c * nw, h.data(), nw * 2, hipMemcpyHostToDevice);
  hipMemset(B, 0, N * 2); hipMemset(S1, 0, N * 4); hipMemset(CF, 0, N * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto go = [&](int i) {
    return fwd::launch_dec_gemm_big(nullptr, cfg, X, W + (size_t)(i % 8) * nw, lnf ? nullptr : B, lnf ? S1 : nullptr,
                                    lnf ? CF : nullptr, nullptr, 0, O, N, nullptr, R, N, K, 0);
  };
  if (go(0) != 0) { printf("cfg %d: shape refused\n", cfg); return; }
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) go(i + 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long tl[2][DGB_NTL]; int sl[2][DGB_NTL];
  hipMemcpyFromSymbol(tl, HIP_SYMBOL(gb_tl), sizeof(tl));
  hipMemcpyFromSymbol(sl, HIP_SYMBOL(gb_slot), sizeof(sl));
  printf("cfg %d  R=%d N=%d K=%d lnf=%d: %.2f us per launch (instrumented build) = %.0f TFLOP/s\n", cfg, R, N, K, (int)lnf,
         ms * 20.0, 2.0 * R * N * K / (ms / 50 * 1e-3) / 1e12);
  for (int g = 0; g < 2; ++g) {
    // mean cycles from each stamp to the next one, by the slot of the FIRST of the pair, stamps 40.. (steady state)
    double sum[8] = {0}; int cnt[8] = {0};
    for (int i = 40; i + 1 < DGB_NTL; ++i) {
      if (tl[g][i + 1] == 0) break;
      sum[sl[g][i] & 7] += (double)(tl[g][i + 1] - tl[g][i]);
      cnt[sl[g][i] & 7] += 1;
    }
    printf("  wave %d: cycles from stamp s to the next stamp:", g * 4);
    double tot = 0; int per = 0;
    for (int s = 0; s < 6; ++s) if (cnt[s]) { printf("  s%d %.0f", s, sum[s] / cnt[s]); tot += sum[s] / cnt[s]; per = cnt[s]; }
    printf("   | per loop iteration %.0f cycles (%d iterations)\n", tot, per);
  }
  hipFree(X); hipFree(W); hipFree(O); hipFree(B); hipFree(S1); hipFree(CF);
}

int main() {
  for (int R : {1280}) {
    run(0, R, 3840, 1280, true);    // qkv, 256 x 128 lockstep (2 k-steps per stage, ring of 3)
    run(2, R, 3840, 1280, true);    // 128 x 128, 4 waves (1 k-step per stage, ring of 4)
    run(1, R, 1280, 1280, false);   // d x d, 128 x 64
    run(0, R, 5120, 1280, true);    // ffn1
    run(1, R, 1280, 5120, false);   // ffn2, 128 x 64
  }
  return 0;
}
