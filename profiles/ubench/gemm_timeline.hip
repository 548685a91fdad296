// Diagnostic (gfx950): s_memtime stamps at the segment boundaries of the encoder GEMM's K loop, for wave 0 (group 0)
// and wave 4 (group 1) of workgroup 0, K tiles 6..9 — where do the cycles of a phase go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_timeline.hip -o /tmp/tl && /tmp/tl
#include <hip/hip_runtime.h>
__device__ long long gb_tl[2][4][16];
#define GB_TL(slot)                                                                                  \
  do {                                                                                               \
    if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && kt >= 6 && kt < 10)                       \
      gb_tl[wave >> 2][kt - 6][slot] = __builtin_readcyclecounter();                                 \
  } while (0)
#include "../../faster_whisper_amd/csrc/gemm.hip"
#include <stdio.h>
#include <vector>

int main() {
  const int M = 1500, N = 2560, K = 1280, batch = 16;
  std::vector<_Float16> h((size_t)batch * M * K);
  unsigned st = 1;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = (_Float16)(((int)(st >> 16) % 2001 - 1000) * 1e-3f); }
  half_t *A, *W, *C;
  hipMalloc(&A, h.size() * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&C, (size_t)batch * M * N * 2);
  hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(W, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  fwk::GemmParams p = {};
  p.A = A; p.lda = K; p.a_bstride = (int64_t)M * K; p.W = W; p.ldw = K; p.C = C; p.ldc = N; p.c_bstride = (int64_t)M * N;
  p.M = M; p.N = N; p.K = K;
  for (int i = 0; i < 3; ++i) fwk::launch_gemm(nullptr, p, batch, false);
  hipDeviceSynchronize();
  long long tl[2][4][16];
  hipMemcpyFromSymbol(tl, HIP_SYMBOL(gb_tl), sizeof(tl));
  const char* names[4] = {"load", "barrier", "compute", "barrier"};
  for (int g = 0; g < 2; ++g) {
    printf("wave %d (group %d): cycles per segment, K tiles 6..9\n", g * 4, g);
    for (int kt = 0; kt < 4; ++kt) {
      printf("  kt %d:", kt + 6);
      for (int s = 0; s < 16; ++s) {
        const long long nxt = s < 15 ? tl[g][kt][s + 1] : (kt < 3 ? tl[g][kt + 1][0] : tl[g][kt][15]);
        printf(" %s%lld", s % 4 == 0 ? "| " : "", nxt - tl[g][kt][s]);
      }
      printf("   (tile total %lld)\n", (kt < 3 ? tl[g][kt + 1][0] : tl[g][kt][15]) - tl[g][kt][0]);
    }
  }
  printf("segments per phase: %s %s %s %s\n", names[0], names[1], names[2], names[3]);
  return 0;
}
