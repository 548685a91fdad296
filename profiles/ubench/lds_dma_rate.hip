// Micro-benchmark (gfx950): what rate does the L2 -> LDS DMA path (global_load_lds_dwordx4) sustain per CU with the
// access pattern of the encoder GEMM (256 x 256 x 128-byte tiles: 64 KB per K tile per workgroup, operands shared
// between workgroups through the XCD's L2), as a function of the DMA instructions kept in flight per wave?
//   hipcc --offload-arch=gfx950 -O3 lds_dma_rate.hip -o /tmp/dma && /tmp/dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

static __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// DEPTH = DMA instructions outstanding per wave (8 per K tile are issued: 4 units x 2)
template <int DEPTH, bool SHARED>
__global__ __launch_bounds__(512) void k(const char* A, const char* W, int M, int N, int K2 /*bytes per row*/, int nNt,
                                         int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = bid / nNt, nt = bid - mt * nNt;
  // SHARED: the GEMM's sharing (A panel shared by the n tiles, W by the m tiles); else every workgroup its own rows
  const size_t arow0 = SHARED ? (size_t)mt * 256 : (size_t)bid * 256;
  const size_t wrow0 = SHARED ? (size_t)nt * 256 : (size_t)bid * 256;
  const char* src[8];
  for (int i = 0; i < 2; ++i) {
    const int u = (i * 8 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((u >> 1) & 7);
    for (int h = 0; h < 2; ++h) {
      src[h * 2 + i] = A + ((arow0 + (u >> 6) * 128 + (u & 63) + h * 64) % M) * K2 + c * 16;
      src[4 + h * 2 + i] = W + ((wrow0 + (u >> 5) * 64 + (u & 31) + h * 32) % N) * K2 + c * 16;
    }
  }
  const int nk = K2 / 128;
  for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      char* dst = smem + (kt & 1) * 65536 + (j >> 1) * 16384 + ((j & 1) * 8 + wave) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + kt * 128),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (DEPTH == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (DEPTH == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      if (DEPTH == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = smem[lane];
}

template <int DEPTH, bool SHARED>
void run(const char* A, const char* W, int M, int N, int K, int* sink) {
  const int nMt = M / 256, nNt = N / 256, grid = nMt * nNt;
  hipFuncSetAttribute((const void*)k<DEPTH, SHARED>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<DEPTH, SHARED><<<grid, 512, 131072>>>(A, W, M, N, K * 2, nNt, sink);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<DEPTH, SHARED><<<grid, 512, 131072>>>(A, W, M, N, K * 2, nNt, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)grid * (K * 2 / 128) * 65536.0;
  printf("M=%d N=%d K=%d %s depth %2d: %.1f us, L2->LDS %.2f TB/s = %.1f GB/s per CU (GEMM-equivalent %.0f TFLOP/s)\n", M, N, K,
         SHARED ? "shared panels" : "private rows ", DEPTH, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
         2.0 * M * N * K / ms / 1e9);
}

int main() {
  const int M = 24064, N = 5120, K = 5120;
  char *A, *W; int* sink;
  hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&sink, 1 << 20);
  hipMemset(A, 1, (size_t)M * K * 2); hipMemset(W, 1, (size_t)N * K * 2);
  for (int n : {1280, 2560, 5120}) {
    run<4, true>(A, W, M, n, 1280, sink);
    run<8, true>(A, W, M, n, 1280, sink);
    run<16, true>(A, W, M, n, 1280, sink);
    run<32, true>(A, W, M, n, 1280, sink);
  }
  run<8, true>(A, W, M, 1280, 5120, sink);
  run<32, true>(A, W, M, 1280, 5120, sink);
  run<8, false>(A, W, M, 1280, 1280, sink);
  run<32, false>(A, W, M, 1280, 1280, sink);
  return 0;
}
