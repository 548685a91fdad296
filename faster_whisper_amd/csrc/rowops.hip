// K5 — LayerNorm (eps 1e-5, affine), fp16 in/out, fp32 statistics; the int8 row quantiser; dtype casts.
// One 64-lane wave per row, the whole row lives in registers as 16-byte chunks (8 halves per lane per access:
// scalar / half2 accesses run these HBM-bound kernels at less than half the rate), mean and variance are two
// shuffle reductions (no LDS, no second HBM pass).  HBM-bound: 2*d*2 bytes per row (LayerNorm).
#include "common.h"
#include "kernels.h"

#define LN_MAXC 3    // 16-byte chunks per lane for a LayerNorm row: d <= 1536
#define QR_MAXC 10   // ... for a plain row handed to the quantiser: d <= 5120

// loads the row's chunks lane, lane + 64, ... into v[][8]; returns their sum
template <int MAXC>
static __device__ __forceinline__ float load_row(const half_t* xr, int nc, int lane, float (&v)[MAXC][8]) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < nc) {
      const half8_t h = *reinterpret_cast<const half8_t*>(xr + (size_t)c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = (float)h[e]; s += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  return s;
}

// y = LN(x) * g + b in place in v (rounded to fp16, as the tensor is stored); returns the row's absmax
template <int MAXC>
static __device__ __forceinline__ float normalise_row(float (&v)[MAXC][8], int nc, int lane, int d, float sum,
                                                      const half_t* g, const half_t* b) {
  const float mean = wave_sum(sum) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
    if (i * 64 + lane < nc) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float a = v[i][e] - mean; q += a * a; }
    }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < nc) {
      const half8_t gg = *reinterpret_cast<const half8_t*>(g + (size_t)c * 8);
      const half8_t bb = *reinterpret_cast<const half8_t*>(b + (size_t)c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (float)(half_t)((v[i][e] - mean) * rstd * (float)gg[e] + (float)bb[e]);
        amax = fmaxf(amax, fabsf(v[i][e]));
      }
    }
  }
  return amax;
}

// frag != 0: y is MFMA-fragment-major (the layout the decoder linears read, dec_kernels.hip frag_off): the 8 halves
// k = 8c .. 8c+7 of row r stay one 16-byte chunk at ((r/16 * d/32 + c/4) * 64 + 16*(c%4) + r%16) * 8
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const half_t* __restrict__ g,
                                                        const half_t* __restrict__ b, half_t* __restrict__ y,
                                                        int rows, int d, int frag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nc = d >> 3;
  float v[LN_MAXC][8];
  const float s = load_row<LN_MAXC>(x + (size_t)row * d, nc, lane, v);
  (void)normalise_row<LN_MAXC>(v, nc, lane, d, s, g, b);
  half_t* yr = y + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < nc) {
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)v[i][e];
      if (frag)
        *reinterpret_cast<half8_t*>(y + ((size_t)((row >> 4) * (d >> 5) + (c >> 2)) * 64 + (c & 3) * 16 + (row & 15)) * 8) = o;
      else
        *reinterpret_cast<half8_t*>(yr + (size_t)c * 8) = o;
    }
  }
}

// K25 (activation side): per-row dynamic int8 quantisation, optionally fused with the LayerNorm that
// produces the row.  scale[r] = absmax / 127 (de-quantisation factor), xq = rint(y / scale).
// One wave per row, the row stays in registers (LayerNorm rows d <= 1536, plain rows d <= 5120): one pass.
// frag != 0: int8 MFMA-fragment-major destination (dec_gemm_frag_i8_kernel): element (row, k) lives at byte
// ((row/16 * d/64 + k/64) * 64 + 16*((k/16)%4) + row%16) * 16 + k%16; a chunk of 8 consecutive k stays contiguous.
template <int MAXC, bool LN>
static __device__ __forceinline__ void quant_row(const half_t* xr, const half_t* g, const half_t* b, int8_t* xq,
                                                 float* scale, int row, int d, int frag, int lane) {
  const int nc = d >> 3;
  float v[MAXC][8];
  const float s = load_row<MAXC>(xr, nc, lane, v);
  float amax = 0.f;
  if (LN) {
    amax = normalise_row<MAXC>(v, nc, lane, d, s, g, b);
  } else {
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[i][e]));
  }
  amax = wave_max(amax);
  const float inv = amax > 0.f ? 127.0f / amax : 0.f;
  if (lane == 0) scale[row] = amax > 0.f ? amax / 127.0f : 1.0f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < nc) {
      union { signed char q[8]; intx2 w; } o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.q[e] = (signed char)__float2int_rn(v[i][e] * inv);
      const int k = c * 8;
      const size_t off = frag ? ((size_t)((row >> 4) * (d >> 6) + (k >> 6)) * 64 + ((k >> 4) & 3) * 16 + (row & 15)) * 16 + (k & 15)
                              : (size_t)row * d + k;
      *reinterpret_cast<intx2*>(xq + off) = o.w;
    }
  }
}

__global__ __launch_bounds__(256) void quant_rows_kernel(const half_t* __restrict__ x, int64_t ldx,
                                                         const half_t* __restrict__ g, const half_t* __restrict__ b,
                                                         int8_t* __restrict__ xq, float* __restrict__ scale, int rows,
                                                         int d, int frag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const half_t* xr = x + (size_t)row * ldx;
  if (g) quant_row<LN_MAXC, true>(xr, g, b, xq, scale, row, d, frag, lane);
  else if (d <= 8 * 64 * LN_MAXC) quant_row<LN_MAXC, false>(xr, g, b, xq, scale, row, d, frag, lane);
  else quant_row<QR_MAXC, false>(xr, g, b, xq, scale, row, d, frag, lane);
}

__global__ void f32_to_f16_kernel(const float* __restrict__ x, half_t* __restrict__ y, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = (half_t)x[i];
}
__global__ void f16_to_f32_kernel(const half_t* __restrict__ x, float* __restrict__ y, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = (float)x[i];
}

namespace fwk {
void launch_layernorm(hipStream_t st, const half_t* x, const half_t* g, const half_t* b, half_t* y, int rows, int d,
                      int frag) {
  layernorm_kernel<<<(rows + 3) / 4, 256, 0, st>>>(x, g, b, y, rows, d, frag);
}
static int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}
void launch_quant_rows(hipStream_t st, const half_t* x, int64_t ldx, const half_t* g, const half_t* b, int8_t* xq,
                       float* scale, int rows, int d, int frag) {
  quant_rows_kernel<<<(rows + 3) / 4, 256, 0, st>>>(x, ldx, g, b, xq, scale, rows, d, frag);
}
void launch_f32_to_f16(hipStream_t st, const float* x, half_t* y, int64_t n) {
  f32_to_f16_kernel<<<grid_for(n), 256, 0, st>>>(x, y, n);
}
void launch_f16_to_f32(hipStream_t st, const half_t* x, float* y, int64_t n) {
  f16_to_f32_kernel<<<grid_for(n), 256, 0, st>>>(x, y, n);
}
}  // namespace fwk
