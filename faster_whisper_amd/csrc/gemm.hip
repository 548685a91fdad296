// K2/K3/K6/K8-K11 — the dense fp16 MFMA GEMM of the encoder (and of every
// "many rows" linear: conv1d-as-GEMM, fused QKV, out-proj, FFN, cross-K/V projection).
//
//   C[z][m][n] = epi( sum_k A[z][m][k] * W[n][k] )      A, W, C fp16; fp32 accumulate
//   epi(v)     = act(v + bias[n]) + res[z][m][n]         act = identity | exact-erf GELU
//
// Both operands are K-contiguous ("B^T" form), which is the natural MFMA fragment
// order: lane (row = l&31, k-octet = l>>5) reads 16 contiguous bytes.
//
// gfx950 mapping
//  * 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 2x2 tiles of
//    v_mfma_f32_32x32x16_f16 (64 accumulator VGPRs).
//  * global -> LDS by direct DMA (global_load_lds, 16 B/lane, coalesced 128 B rows, no VGPR
//    round trip) -> MFMA fragments (ds_read_b128). LDS rows are 128 B; the 16-byte chunk
//    index is XOR-swizzled with (row>>1)&7 (applied to the DMA source address, the LDS
//    image itself is lane-linear), which makes every ds_read_b128 lane group
//    conflict-free (MI355X_MICROARCH.md LDS table).
//  * double-buffered LDS, ONE barrier per K tile; the next tile's DMA is issued right
//    after that barrier and lands while the current tile is multiplied.
//  * normal mode computes D = W_tile * A_tile^T so that each lane ends up with four
//    consecutive n of one row m -> 8-byte stores; TRANS mode computes D = A_tile *
//    W_tile^T, each lane holds four consecutive m of one column n, and the tile is
//    stored transposed (Ct[z][n][m]) — used to emit V^T for the attention kernels
//    without a separate transpose pass.
//  * conv1d (k=3, stride s) over a channel-last, zero-padded image is this GEMM with
//    lda = s*C and K = 3*C: the three taps of an output row are contiguous in memory.
//  * 1-D grid with a bijective XCD remap; n fastest inside an m panel so the blocks
//    resident on one XCD share the A panel and the whole W in that XCD's L2.
#include "common.h"
#include "kernels.h"

#define GB_M 128
#define GB_N 128
#define GB_K 64
#define GB_TILE_HALVES (128 * 64)

typedef int intx16 __attribute__((ext_vector_type(16)));

// I8: the int8_float16 path (K25): A and W are int8 (per-row dequant scales a_scale[m], w_scale[n]),
// v_mfma_i32_32x32x32_i8 accumulates in int32, the epilogue de-quantises.  The tile is defined in BYTES
// (128-byte rows = 64 halves or 128 int8), so staging, swizzle and fragment reads are shared.
template <bool TRANS, bool I8>
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(fwk::GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int ES = I8 ? 1 : 2;        // element size
  constexpr int TILE_BYTES = 128 * 128;
  char* sA = smem_raw;                  // [2][128 rows][128 B]
  char* sW = smem_raw + 2 * TILE_BYTES; // [2][128 rows][128 B]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  const int per_z = p.nMt * p.nNt;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int z = bid / per_z;
  bid -= z * per_z;
  const int mt = bid / p.nNt, nt = bid - mt * p.nNt;
  const int m0 = mt * GB_M, n0 = nt * GB_N;

  const char* Ab = reinterpret_cast<const char*>(p.A) + (size_t)z * p.a_bstride * ES;
  const char* Wb = reinterpret_cast<const char*>(p.W);
  // staging: direct HBM/L2 -> LDS DMA (global_load_lds, 16 B per lane, no VGPR round trip, no ds_write
  // pass).  A wave instruction fills 64 consecutive 16-byte LDS slots = 8 tile rows; the LDS image must
  // stay lane-linear, so the XOR swizzle is applied to the SOURCE address: LDS slot (row, c') receives
  // global chunk c = c' ^ ((row >> 1) & 7)  (the fragment reads below apply the same involution).
  const int wuni = __builtin_amdgcn_readfirstlane(wave);
  const char* gA[4];
  const char* gW[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + wuni) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    int am = m0 + row; if (am > p.M - 1) am = p.M - 1;
    int wn_ = n0 + row; if (wn_ > p.N - 1) wn_ = p.N - 1;
    gA[i] = Ab + (size_t)am * p.lda * ES + c * 16;
    gW[i] = Wb + (size_t)wn_ * p.ldw * ES + c * 16;
  }
  const int nk = p.K * ES / 128;
  auto stage = [&](int kt, int buf) {
    const int koff = kt * 128;
    char* dA = sA + buf * TILE_BYTES;
    char* dW = sW + buf * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA[i] + koff),
                                       (__attribute__((address_space(3))) void*)(dA + (i * 4 + wuni) * 1024), 16, 0,
                                       0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gW[i] + koff),
                                       (__attribute__((address_space(3))) void*)(dW + (i * 4 + wuni) * 1024), 16, 0,
                                       0);
    }
  };
  stage(0, 0);

  floatx16 accf[2][2];
  intx16 acci[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { accf[i][j] = floatx16{0}; acci[i][j] = intx16{0}; }

  // fragment rows and their swizzle keys
  int arow[2], wrow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    arow[i] = wm * 64 + i * 32 + l31;
    wrow[i] = wn * 64 + i * 32 + l31;
  }

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // tile kt has landed (this wave's DMAs drained, then the barrier covers the other waves'); the same
    // barrier proves every wave is done reading buffer cur^1, which the next tile's DMA overwrites while
    // this tile is multiplied.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
    const char* cA = sA + cur * TILE_BYTES;
    const char* cW = sW + cur * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + hi;
      intx4 fa[2], fw[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const intx4*>(cA + arow[i] * 128 + ((chunk ^ ((arow[i] >> 1) & 7)) << 4));
        fw[i] = *reinterpret_cast<const intx4*>(cW + wrow[i] * 128 + ((chunk ^ ((wrow[i] >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // TRANS: D[m][n] -> acc[mi][ni];  else D[n][m] -> acc[ni][mi]
          const intx4 opa = TRANS ? fa[i] : fw[i];
          const intx4 opb = TRANS ? fw[j] : fa[j];
          if (I8) {
            acci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(opa, opb, acci[i][j], 0, 0, 0);
          } else {
            accf[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, opa),
                                                              __builtin_bit_cast(half8_t, opb), accf[i][j], 0, 0, 0);
          }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---------------------------------- epilogue ----------------------------------
  const float* sa = I8 ? p.a_scale + (size_t)z * p.as_bstride : nullptr;
  if (!TRANS) {
    half_t* Cb = p.C + (size_t)z * p.c_bstride;
    const half_t* Rb = p.res ? p.res + (size_t)z * p.r_bstride : nullptr;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wm * 64 + mi * 32 + l31;
        if (m >= p.M) continue;
        const float sam = I8 ? sa[m] : 1.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = n0 + wn * 64 + ni * 32 + 8 * g + 4 * hi;
          if (n >= p.N) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = I8 ? (float)acci[ni][mi][g * 4 + e] * sam * p.w_scale[n + e] : accf[ni][mi][g * 4 + e];
          if (p.bias) {
            const half4_t bv = *reinterpret_cast<const half4_t*>(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
          }
          if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
          }
          if (Rb) {
            const half4_t rv = *reinterpret_cast<const half4_t*>(Rb + (size_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
          }
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
          half_t* dst;
          if (p.head_rows > 0) {
            // cross-attention K, MFMA-fragment-major per (chunk, head): a 32-key group is 4 runs of 64 lanes x 16 B,
            // run q = 2*sub + s, lane = 16*g + j  <->  key 32*gi + 8*(j>>2) + 4*sub + (j&3), dims 32*s + 8*g + [0,8)
            const int c = n & 63, r = m & 31;
            const int run = (m >> 5) * 4 + 2 * ((r >> 2) & 1) + (c >> 5);
            const int ln = ((c >> 3) & 3) * 16 + (((r >> 3) << 2) | (r & 3));
            dst = Cb + (size_t)(n >> 6) * p.head_rows * 64 + ((size_t)run * 64 + ln) * 8 + (c & 7);
          } else {
            dst = Cb + (size_t)m * p.ldc + n;
          }
          *reinterpret_cast<half4_t*>(dst) = o;
        }
      }
  } else {
    half_t* Cb = p.C + (size_t)z * p.c_bstride;  // Ct[z][n][m], ldc = row stride of Ct
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + l31;
        if (n >= p.N) continue;
        const float bv = p.bias ? (float)p.bias[n] : 0.f;
        const float swn = I8 ? p.w_scale[n] : 1.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = m0 + wm * 64 + mi * 32 + 8 * g + 4 * hi;
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int mm = m + e; if (mm > p.M - 1) mm = p.M - 1;
            float v = (I8 ? (float)acci[mi][ni][g * 4 + e] * swn * sa[mm] : accf[mi][ni][g * 4 + e]) + bv;
            if (p.act == 1) v = gelu_erf(v);
            o[e] = (half_t)v;
          }
          half_t* dst;
          if (p.head_rows > 0) {
            // cross-attention V^T, fragment-major per (chunk, head): run = 4*gi + dt, lane = 16*g + j  <->
            // dim 16*dt + j, keys 32*gi + 8*g + [0,8)
            const int c = n & 63;
            dst = Cb + (size_t)(n >> 6) * p.head_rows * 64 +
                  ((size_t)((m >> 5) * 4 + (c >> 4)) * 64 + ((m >> 3) & 3) * 16 + (c & 15)) * 8 + (m & 7);
          } else {
            dst = Cb + (size_t)n * p.ldc + m;
          }
          if (m + 3 < p.M) {
            *reinterpret_cast<half4_t*>(dst) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (m + e < p.M) dst[e] = o[e];
          }
        }
      }
  }
}

namespace fwk {

int launch_gemm(hipStream_t st, const GemmParams& pin, int batch, bool trans) {
  GemmParams p = pin;
  const bool i8 = p.a_scale != nullptr;
  const int es = i8 ? 1 : 2;
  if (p.K <= 0 || (p.K * es) % 128 != 0) return -1;
  if (((p.lda * es) % 16) || ((p.ldw * es) % 16) || ((p.a_bstride * es) % 16)) return -1;
  if (!trans && ((p.N % 4) || (p.ldc % 4) || (p.c_bstride % 4))) return -1;
  if (i8 && !p.w_scale) return -1;
  p.nMt = (p.M + GB_M - 1) / GB_M;
  p.nNt = (p.N + GB_N - 1) / GB_N;
  const int lds = 4 * 128 * 128;  // 64 KiB
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<true, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int grid = p.nMt * p.nNt * batch;
  if (i8) {
    if (trans) gemm_f16_kernel<true, true><<<grid, 256, lds, st>>>(p);
    else gemm_f16_kernel<false, true><<<grid, 256, lds, st>>>(p);
  } else {
    if (trans) gemm_f16_kernel<true, false><<<grid, 256, lds, st>>>(p);
    else gemm_f16_kernel<false, false><<<grid, 256, lds, st>>>(p);
  }
  return 0;
}

}  // namespace fwk
