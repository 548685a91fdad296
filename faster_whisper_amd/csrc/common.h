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

// ---- whole-wave reductions on the VALU (DPP + permlane swaps), no LDS crossbar ----
// The butterfly of wave_sum / wave_max above, partner distance 32, 16, 8, 4, 2, 1, with every exchange done by a DPP
// modifier or a gfx950 permlane swap instead of ds_bpermute (~100 cycles of LDS round trip per step, six dependent steps).
// Same partners in the same order, and a + b == b + a bit for bit, so wave_sum_v returns exactly wave_sum's bits:
//   32 / 16: v_permlane32_swap / v_permlane16_swap;  8: row_ror:8 (lane i of a 16-lane row reads lane (i + 8) % 16 = i ^ 8);
//   4: row_ror:4 — lane (i + 4) % 16 is i ^ 4 or (i ^ 4) ^ 8, and after the distance-8 step lanes j and j ^ 8 hold the same
//   value;  2 / 1: quad_perm [2,3,0,1] / [1,0,3,2].
template <int CTRL>
static __device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
static __device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
#define FW_DPP_XOR1 0xB1   /* quad_perm [1,0,3,2] */
#define FW_DPP_XOR2 0x4E   /* quad_perm [2,3,0,1] */
#define FW_DPP_ROR4 0x124  /* row_ror:4 */
#define FW_DPP_ROR8 0x128  /* row_ror:8 */
static __device__ __forceinline__ float wave_sum_v(float v) {
  v = pair32_sum(v);
  v = pair16_sum(v);
  v += dpp_f<FW_DPP_ROR8>(v);
  v += dpp_f<FW_DPP_ROR4>(v);
  v += dpp_f<FW_DPP_XOR2>(v);
  v += dpp_f<FW_DPP_XOR1>(v);
  return v;
}
static __device__ __forceinline__ float wave_max_v(float v) {
  v = pair32_max(v);
  v = pair16_max(v);
  v = fmaxf(v, dpp_f<FW_DPP_ROR8>(v));
  v = fmaxf(v, dpp_f<FW_DPP_ROR4>(v));
  v = fmaxf(v, dpp_f<FW_DPP_XOR2>(v));
  v = fmaxf(v, dpp_f<FW_DPP_XOR1>(v));
  return v;
}
// sum over the lanes 16 and 32 away (the k-octet lanes of an MFMA operand row): xor-16 then xor-32 partners, as
// `v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);` adds them
static __device__ __forceinline__ float sum_x16_x32_v(float v) { return pair32_sum(pair16_sum(v)); }
// sum over the lanes 8, 16 and 32 away, in that order
static __device__ __forceinline__ float sum_x8_x16_x32_v(float v) {
  v += dpp_f<FW_DPP_ROR8>(v);
  return pair32_sum(pair16_sum(v));
}
// arg-max of (key desc, token asc) pairs over the wave; the order relation is total and the operation idempotent,
// so any exchange pattern that connects all 64 lanes gives every lane the same winner
static __device__ __forceinline__ void argmax_pair_step(float& k, int& t, float ok, int ot) {
  if (ok > k || (ok == k && ot < t)) { k = ok; t = ot; }
}
static __device__ __forceinline__ void wave_argmax_v(float& k, int& t) {
  argmax_pair_step(k, t, dpp_f<FW_DPP_XOR1>(k), dpp_i<FW_DPP_XOR1>(t));
  argmax_pair_step(k, t, dpp_f<FW_DPP_XOR2>(k), dpp_i<FW_DPP_XOR2>(t));
  argmax_pair_step(k, t, dpp_f<FW_DPP_ROR4>(k), dpp_i<FW_DPP_ROR4>(t));
  argmax_pair_step(k, t, dpp_f<FW_DPP_ROR8>(k), dpp_i<FW_DPP_ROR8>(t));
  {
    float a, b; pair_swap16(k, a, b);
    float ta, tb; pair_swap16(__builtin_bit_cast(float, t), ta, tb);
    // {a, b} = {own, partner} in some order, (ta, tb) in the SAME order: pick the better of the two pairs
    float k2 = a; int t2 = __builtin_bit_cast(int, ta);
    argmax_pair_step(k2, t2, b, __builtin_bit_cast(int, tb));
    k = k2; t = t2;
  }
  {
    float a, b; pair_swap32(k, a, b);
    float ta, tb; pair_swap32(__builtin_bit_cast(float, t), ta, tb);
    float k2 = a; int t2 = __builtin_bit_cast(int, ta);
    argmax_pair_step(k2, t2, b, __builtin_bit_cast(int, tb));
    k = k2; t = t2;
  }
}

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
