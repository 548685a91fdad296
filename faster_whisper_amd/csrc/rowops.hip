// K5 — LayerNorm (eps 1e-5, affine), fp16 in/out, fp32 statistics; plus dtype casts.
// One 64-lane wave per row, the whole row lives in registers (d <= 1280 -> <= 10
// half2 per lane), mean and variance are two shuffle reductions (no LDS, no
// second HBM pass).  HBM-bound: 2*d*2 bytes per row.
#include "common.h"
#include "kernels.h"

#define LN_MAXV 10

__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const half_t* __restrict__ g,
                                                        const half_t* __restrict__ b, half_t* __restrict__ y,
                                                        int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d >> 7;  // half2 per lane
  const half2_t* xr = reinterpret_cast<const half2_t*>(x + (size_t)row * d);
  float v0[LN_MAXV], v1[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    if (i < nv) {
      const half2_t h = xr[i * 64 + lane];
      v0[i] = (float)h[0]; v1[i] = (float)h[1];
      s += v0[i] + v1[i];
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    if (i < nv) {
      const float a = v0[i] - mean, c = v1[i] - mean;
      q += a * a + c * c;
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
  const half2_t* gr = reinterpret_cast<const half2_t*>(g);
  const half2_t* br = reinterpret_cast<const half2_t*>(b);
  half2_t* yr = reinterpret_cast<half2_t*>(y + (size_t)row * d);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    if (i < nv) {
      const half2_t gg = gr[i * 64 + lane], bb = br[i * 64 + lane];
      half2_t o;
      o[0] = (half_t)((v0[i] - mean) * rstd * (float)gg[0] + (float)bb[0]);
      o[1] = (half_t)((v1[i] - mean) * rstd * (float)gg[1] + (float)bb[1]);
      yr[i * 64 + lane] = o;
    }
  }
}

// K25 (activation side): per-row dynamic int8 quantisation, optionally fused with the LayerNorm that
// produces the row.  scale[r] = absmax / 127 (de-quantisation factor), xq = rint(y / scale).
// One wave per row; LayerNorm rows (d <= 1280) stay in registers, plain rows (d up to 5120) take a
// second pass over L2-resident data.
__global__ __launch_bounds__(256) void quant_rows_kernel(const half_t* __restrict__ x, int64_t ldx,
                                                         const half_t* __restrict__ g, const half_t* __restrict__ b,
                                                         int8_t* __restrict__ xq, float* __restrict__ scale, int rows,
                                                         int d, int frag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const half2_t* xr = reinterpret_cast<const half2_t*>(x + (size_t)row * ldx);
  char2* qr = reinterpret_cast<char2*>(xq + (size_t)row * d);
  // frag != 0: int8 MFMA-fragment-major destination (dec_gemm_frag_i8_kernel): element (row, k) lives at byte
  // ((row/16 * d/64 + k/64) * 64 + 16*((k/16)%4) + row%16) * 16 + k%16; pair index p = k/2 stays contiguous
  auto dst = [&](int p) -> char2* {
    if (!frag) return qr + p;
    const int k = 2 * p;
    const size_t off = ((size_t)((row >> 4) * (d >> 6) + (k >> 6)) * 64 + ((k >> 4) & 3) * 16 + (row & 15)) * 16 + (k & 15);
    return reinterpret_cast<char2*>(xq + off);
  };
  const int nv = d >> 7;  // half2 per lane
  if (g) {
    float v0[LN_MAXV], v1[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (i < nv) {
        const half2_t h = xr[i * 64 + lane];
        v0[i] = (float)h[0]; v1[i] = (float)h[1];
        s += v0[i] + v1[i];
      }
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (i < nv) {
        const float a = v0[i] - mean, c = v1[i] - mean;
        q += a * a + c * c;
      }
    const float rstd = rsqrtf(wave_sum(q) / (float)d + 1e-5f);
    const half2_t* gr = reinterpret_cast<const half2_t*>(g);
    const half2_t* br = reinterpret_cast<const half2_t*>(b);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (i < nv) {
        const half2_t gg = gr[i * 64 + lane], bb = br[i * 64 + lane];
        // the LayerNorm output is an fp16 tensor in the reference pipeline: round before quantising
        v0[i] = (float)(half_t)((v0[i] - mean) * rstd * (float)gg[0] + (float)bb[0]);
        v1[i] = (float)(half_t)((v1[i] - mean) * rstd * (float)gg[1] + (float)bb[1]);
        amax = fmaxf(amax, fmaxf(fabsf(v0[i]), fabsf(v1[i])));
      }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 127.0f / amax : 0.f;
    if (lane == 0) scale[row] = amax > 0.f ? amax / 127.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (i < nv) {
        char2 o;
        o.x = (signed char)__float2int_rn(v0[i] * inv);
        o.y = (signed char)__float2int_rn(v1[i] * inv);
        *dst(i * 64 + lane) = o;
      }
  } else {
    float amax = 0.f;
    for (int i = lane; i < d / 2; i += 64) {
      const half2_t h = xr[i];
      amax = fmaxf(amax, fmaxf(fabsf((float)h[0]), fabsf((float)h[1])));
    }
    amax = wave_max(amax);
    const float inv = amax > 0.f ? 127.0f / amax : 0.f;
    if (lane == 0) scale[row] = amax > 0.f ? amax / 127.0f : 1.0f;
    for (int i = lane; i < d / 2; i += 64) {
      const half2_t h = xr[i];
      char2 o;
      o.x = (signed char)__float2int_rn((float)h[0] * inv);
      o.y = (signed char)__float2int_rn((float)h[1] * inv);
      *dst(i) = o;
    }
  }
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
void launch_layernorm(hipStream_t st, const half_t* x, const half_t* g, const half_t* b, half_t* y, int rows, int d) {
  layernorm_kernel<<<(rows + 3) / 4, 256, 0, st>>>(x, g, b, y, rows, d);
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
