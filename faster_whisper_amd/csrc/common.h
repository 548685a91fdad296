// Shared device/host helpers for the gfx950 (MI355X, CDNA4) Whisper engine.
// Wave = 64 lanes everywhere in this code base; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef int intx2 __attribute__((ext_vector_type(2)));

#define FW_WAVE 64

static __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
static __device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Exchange with the lane 32 (16) away on the VALU: gfx950's v_permlane32_swap / v_permlane16_swap instead of the
// ds_bpermute round trip through the LDS crossbar that __shfl_xor compiles to.  With both operands the same value v,
// the two results hold {own, partner's} in some order in every lane; max and + do not care which is which, and the
// two lanes of a pair compute the same operation on the same two numbers: the same bits in both, as before.
// (The two results are copied into scalars before the bit cast: __builtin_bit_cast applied to a vector ELEMENT
//  expression reads the vector's first element whatever the index — hipcc 7.2 emitted max(r0, r0) / r0 + r0 for
//  bit_cast(r[0]) op bit_cast(r[1]).)
static __device__ __forceinline__ void pair_swap32(float v, float& a, float& b) {
  const unsigned x = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
static __device__ __forceinline__ void pair_swap16(float v, float& a, float& b) {
  const unsigned x = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
static __device__ __forceinline__ float pair32_max(float v) { float a, b; pair_swap32(v, a, b); return fmaxf(a, b); }
static __device__ __forceinline__ float pair32_sum(float v) { float a, b; pair_swap32(v, a, b); return a + b; }
static __device__ __forceinline__ float pair16_max(float v) { float a, b; pair_swap16(v, a, b); return fmaxf(a, b); }
static __device__ __forceinline__ float pair16_sum(float v) { float a, b; pair_swap16(v, a, b); return a + b; }

// exact (erf) GELU, as in openai-whisper / CTranslate2
static __device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// Bijective XCD-aware block remap: consecutive logical ids land on the same
// XCD (hardware places block b on XCD b % 8), so neighbouring tiles share an L2.
static __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// order-preserving float <-> int mapping for atomicMax on floats of either sign
static __device__ __forceinline__ int float_to_ordered(float f) {
  int i = __float_as_int(f);
  return (i >= 0) ? i : (i ^ 0x7fffffff);
}
static __device__ __host__ __forceinline__ float ordered_to_float(int i) {
  int j = (i >= 0) ? i : (i ^ 0x7fffffff);
  union { int i; float f; } u; u.i = j; return u.f;
}
