// K12-K20, K22-K23 — decoder-step kernels for gfx950.
//
// One decode step works on R = B*K rows (B chunks x K beams).  Everything a step needs
// (step counter, beam parents, token history, KV-slot indirection) lives in HBM, so the
// whole step is a fixed launch sequence that is captured once in a hipGraph and replayed;
// the host only polls a "chunks done" counter every few steps.
//
// The step is HBM-bound: weights are streamed once per step by the
// skinny MFMA GEMM below, the cross-attention K/V of a chunk is streamed once per step and
// SHARED by the K beams of that chunk (they are the 16 columns of one MFMA tile), and the
// self-attention cache is never copied on a beam reorder — rows read their ancestors'
// slots through a [row][position] -> slot table instead.
#include "common.h"
#include "dec_kernels.h"
#include <stdlib.h>
#include <atomic>
#include <type_traits>

namespace fwd {
static std::atomic<int> g_self_attn_form{0};   // fw_test_knob(2, ..): A/B of the self-attention forms
static std::atomic<int> g_forms_epoch{0};      // bumped by every knob that changes which kernels a decode step launches
static std::atomic<int> g_cross_regs{0};       // fw_test_knob(7, ..): 0 = uncapped (110 registers), 1 = capped at 96, 2 = at 80
}

// ------------------------------------------------------------------------------------
// K12: token + learned-position embedding  x[r] = E[tok[r]] + pos[p]
// ------------------------------------------------------------------------------------
// Fragment-major activations (decoder rows feeding the register-streaming skinny GEMM below): element
// (row, k) of a [R][K] matrix lives at ((row/16 * K/32 + k/32) * 64 + lane) * 8 + k%8 with
// lane = 16 * ((k/8) % 4) + row % 16 — i.e. the 16 bytes lane l needs for MFMA k-step ks of row tile rt are
// at ((rt*KS + ks)*64 + l)*16 B: one wave load = one contiguous 1 KB run.
__device__ __forceinline__ size_t frag_off(int row, int k, int KS) {
  return ((size_t)((row >> 4) * KS + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (row & 15)) * 8 + (k & 7);
}

// blk_n > 0 (position blocks: prompt forward, align): row r = chunk * blk_n + j is position pos_fixed + j
__global__ void dec_embed_kernel(const int* __restrict__ tok, const half_t* __restrict__ emb,
                                 const half_t* __restrict__ pos_emb, half_t* __restrict__ x,
                                 half_t* __restrict__ xfrag, int d, const int* __restrict__ d_step, int pos_fixed,
                                 int P, int blk_n) {
  const int r = blockIdx.x;
  const int pos = blk_n > 0 ? pos_fixed + r % blk_n : (pos_fixed >= 0 ? pos_fixed : P - 1 + *d_step);
  const half2_t* e = reinterpret_cast<const half2_t*>(emb + (size_t)tok[r] * d);
  const half2_t* pe = reinterpret_cast<const half2_t*>(pos_emb + (size_t)pos * d);
  half2_t* xo = reinterpret_cast<half2_t*>(x + (size_t)r * d);
  for (int i = threadIdx.x; i < d / 2; i += blockDim.x) {
    const half2_t a = e[i], b = pe[i];
    half2_t o;
    o[0] = (half_t)((float)a[0] + (float)b[0]);
    o[1] = (half_t)((float)a[1] + (float)b[1]);
    xo[i] = o;
    if (xfrag) *reinterpret_cast<half2_t*>(xfrag + frag_off(r, 2 * i, d >> 5)) = o;
  }
}

// ------------------------------------------------------------------------------------
// The decoder linear's epilogue arithmetic, PINNED operation by operation (no compiler contraction choices: explicit
// fma where one is meant, separate roundings elsewhere).  The skinny kernel of solo runs and the GEMM-shaped kernel
// of merged runs both end in these two functions, so that which kernel a run takes never changes a bit of its result.
//   LNF:  y = rstd * (v - mu * s1[n]) + cf[n]     (LayerNorm folded into the weights: s1 = rowsum(W o g), cf = W b + bias)
//   else: y = v + bias[n]
//   then  act == 1: exact-erf GELU;   then + residual;   ONE rounding to fp16
// ------------------------------------------------------------------------------------
static __device__ __forceinline__ void dec_ln_stats(float sa, float sb, int K, float& mu, float& rstd) {
#pragma clang fp contract(off)
  const float kf = (float)K;
  mu = sa / kf;
  const float ex2 = sb / kf;
  const float var = ex2 - mu * mu;
  rstd = rsqrtf(fmaxf(var, 0.f) + 1e-5f);
}
template <bool LNF>
static __device__ __forceinline__ half4_t dec_epilogue4(const floatx4 v, float mu, float rstd,
                                                        const float* __restrict__ s1, const float* __restrict__ cf,
                                                        const half_t* __restrict__ bias, bool has_res, half4_t r4, int n, int act) {
#pragma clang fp contract(off)
  floatx4 a4 = {0.f, 0.f, 0.f, 0.f}, c4 = {0.f, 0.f, 0.f, 0.f};
  half4_t b4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  if (LNF) {
    a4 = *reinterpret_cast<const floatx4*>(s1 + n);
    c4 = *reinterpret_cast<const floatx4*>(cf + n);
  } else if (bias) {
    b4 = *reinterpret_cast<const half4_t*>(bias + n);
  }
  half4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float tv = v[e];
    if (LNF) {
      const float t = tv - mu * a4[e];
      tv = rstd * t + c4[e];
    } else if (bias) {
      tv = tv + (float)b4[e];
    }
    if (act == 1) {
      const float er = erff(tv * 0.70710678118654752440f);
      tv = (0.5f * tv) * (1.0f + er);
    }
    if (has_res) tv = tv + (float)r4[e];
    o[e] = (half_t)tv;
  }
  return o;
}

// int8_float16 form: de-quantise (exact integer sum x row scale x column scale), bias, GELU, residual, one rounding;
// pinned like the fp16 form so that every tile grouping of the int8 kernel returns the same bits
static __device__ __forceinline__ half4_t dec_epilogue4_i8(const int (&v)[4], float sx, const float* __restrict__ w_scale,
                                                           const half_t* __restrict__ bias, const half_t* __restrict__ res,
                                                           int ldr, int row, int n, int act) {
#pragma clang fp contract(off)
  const floatx4 w4 = *reinterpret_cast<const floatx4*>(w_scale + n);
  half4_t b4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f}, r4 = b4;
  if (bias) b4 = *reinterpret_cast<const half4_t*>(bias + n);
  if (res) r4 = *reinterpret_cast<const half4_t*>(res + (size_t)row * ldr + n);
  half4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float tv = ((float)v[e] * sx) * w4[e];
    if (bias) tv = tv + (float)b4[e];
    if (act == 1) {
      const float er = erff(tv * 0.70710678118654752440f);
      tv = (0.5f * tv) * (1.0f + er);
    }
    if (res) tv = tv + (float)r4[e];
    o[e] = (half_t)tv;
  }
  return o;
}

// ------------------------------------------------------------------------------------
// Skinny GEMM, register-streaming form over FRAGMENT-MAJOR operands (frag_off above).
//
// Both the weights (permuted once at pack time) and the activations (written in this order by their
// producers: embed, the attention kernels, this kernel's own epilogue) are stored so that the 16 bytes a lane
// feeds to v_mfma_f32_16x16x32_f16 for k-step ks are at (tile*KS + ks)*1 KB + lane*16 B.  Every wave load is
// one contiguous 1 KB run, operands go straight from L2/HBM to VGPRs: no LDS staging, no DMA ring, no
// barriers in the K loop, so a workgroup holds no LDS while it waits on memory (the LDS-staged form above is
// latency-bound single stream AND LDS-residency-bound with several batches in flight).
//   * one workgroup = RT x NT tiles of 16 x 16 (2 x 2 in the product: the x / W fragments of a k-step are re-used
//     for 4 MFMAs); its WAVES waves split K and keep CH k-steps of loads in flight each ((RT + NT) * CH * 16 B per
//     lane): two load rounds per launch for K <= 1280;
//   * grid (N/16/NT, row groups): the row groups of a column group land on the same XCD (the grid's x extent is a
//     multiple of 8), so a weight tile is fetched from HBM once and re-read from that XCD's L2 — with merged decode
//     runs (320-1 680 rows) that L2 traffic, not HBM, is what bounds the kernel (profiles/NOTES.md);
//   * fixed-order reduction of the WAVES partial tiles through LDS, epilogue spread over the waves.
// ------------------------------------------------------------------------------------
// (Round 5 measured two ways of touching the NEXT linear's weights ahead of its launch for solo runs — a prefetch kernel on a
//  shadow branch of the step graph, and an extra wave inside this kernel that reads the consumer's tiles on the XCD that
//  will use them.  Both lost: a fork edge of a HIP graph costs ~10 us (2.3x slower steps), and the L2 does not keep clean
//  lines across a kernel boundary (the acquire that makes other XCDs' writes visible drops them), so the consumer
//  re-fetches anyway and the prefetch only doubles the traffic: 192 -> 289 ms single utterance.
//  profiles/r05_ab_wprefetch_shadow_branch.jsonl, r05_ab_wprefetch_wave.jsonl; the code is in the history: commits eeb074e (wave) and d097c11^ (branch))
template <int WAVES, bool LNF, int RT, int NT, int CH_ = 0>
__global__ __launch_bounds__(WAVES * 64) void dec_gemm_frag_kernel(
    const half_t* __restrict__ xf, const half_t* __restrict__ Wf, const half_t* __restrict__ bias,
    const float* __restrict__ s1, const float* __restrict__ cf, const half_t* __restrict__ res, int ldr,
    half_t* __restrict__ out, int ldo, half_t* __restrict__ out_frag, int R, int N, int K, int act) {
  // RT x NT tiles of 16 x 16 per workgroup (larger tiles re-use the x / W fragments in registers and cut the L2
  // re-reads at the price of fewer, fatter workgroups: launch_dec_gemm_frag_variant / profiles/dec_linear_bench.py)
  __shared__ float red[WAVES][RT * NT][64][4];
  __shared__ float red_s[WAVES][RT][16][2];
  constexpr int CH = CH_ ? CH_ : 20 / (RT + NT);   // k-steps in flight per wave: (RT + NT) * CH * 16 B per lane
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int n_rt = (R + 15) >> 4;
  const int ct0 = blockIdx.x * NT, rt0 = blockIdx.y * RT;
  const int KS = K >> 5;
  const int per = (KS + WAVES - 1) / WAVES;
  const int ks0 = wave * per;
  int nks = KS - ks0;
  if (nks > per) nks = per;
  floatx4 acc[RT][NT];
  float rs[RT], rq[RT];
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    rs[a] = 0.f; rq[a] = 0.f;
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = floatx4{0, 0, 0, 0};
  }
  // the residual values of the tiles THIS wave finishes in the epilogue, requested now: behind the barrier the load was
  // one more dependent round trip of a launch that is nothing but round trips (solo runs)
  half4_t resv[(RT * NT + WAVES - 1) / WAVES];
#pragma unroll
  for (int q = 0; q < (RT * NT + WAVES - 1) / WAVES; ++q) resv[q] = half4_t{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
  if (res) {
#pragma unroll
    for (int a = 0; a < RT; ++a) {
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        if ((a * NT + b) % WAVES != wave) continue;
        const int row = (rt0 + a) * 16 + i, n = (ct0 + b) * 16 + 4 * g;
        if (row < R && n < N) resv[(a * NT + b) / WAVES] = *reinterpret_cast<const half4_t*>(res + (size_t)row * ldr + n);
      }
    }
  }
  if (nks > 0) {
    const half8_t* wp[NT];
    const half8_t* xp[RT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      int ct = ct0 + b;
      if (ct > (N >> 4) - 1) ct = (N >> 4) - 1;   // a missing column tile re-reads the last one; never stored
      wp[b] = reinterpret_cast<const half8_t*>(Wf) + ((size_t)ct * KS + ks0) * 64 + lane;
    }
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      int rt = rt0 + a;
      if (rt > n_rt - 1) rt = n_rt - 1;   // a missing row tile re-reads the last one; its result is dropped
      xp[a] = reinterpret_cast<const half8_t*>(xf) + ((size_t)rt * KS + ks0) * 64 + lane;
    }
    for (int c = 0; c < nks; c += CH) {
      half8_t wv[NT][CH], xv[RT][CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int jj = (c + j < nks) ? c + j : nks - 1;   // clamped: a tail slot re-reads the last step, unused
#pragma unroll
        for (int b = 0; b < NT; ++b) wv[b][j] = wp[b][(size_t)jj * 64];
#pragma unroll
        for (int a = 0; a < RT; ++a) xv[a][j] = xp[a][(size_t)jj * 64];
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (c + j < nks) {
#pragma unroll
          for (int a = 0; a < RT; ++a) {
#pragma unroll
            for (int b = 0; b < NT; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[b][j], xv[a][j], acc[a][b], 0, 0, 0);
            if (LNF) {
              const half2_t one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const half2_t h2 = {xv[a][j][2 * e], xv[a][j][2 * e + 1]};
                rs[a] = __builtin_amdgcn_fdot2(h2, one2, rs[a], false);
                rq[a] = __builtin_amdgcn_fdot2(h2, h2, rq[a], false);
              }
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    if (LNF) {   // row i's statistics: the 4 k-octet lanes of the row, then the waves
      float sa = rs[a], sb = rq[a];
      sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
      sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
      if (g == 0) { red_s[wave][a][i][0] = sa; red_s[wave][a][i][1] = sb; }
    }
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][a * NT + b][lane][e] = acc[a][b][e];
  }
  __syncthreads();
  // fixed-order reduction + epilogue, tile t by wave t % WAVES
#pragma unroll
  for (int a = 0; a < RT; ++a) {
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      if ((a * NT + b) % WAVES != wave) continue;
      const int row = (rt0 + a) * 16 + i;
      if (row >= R) continue;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < WAVES; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += red[w][a * NT + b][lane][e];
      float mu = 0.f, rstd = 1.f;
      if (LNF) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { sa += red_s[w][a][i][0]; sb += red_s[w][a][i][1]; }
        dec_ln_stats(sa, sb, K, mu, rstd);
      }
      const int n = (ct0 + b) * 16 + 4 * g;   // this lane: D[n + e][row], e = 0..3
      if (n >= N) continue;
      const floatx4 v4 = {v[0], v[1], v[2], v[3]};
      const half4_t r4 = resv[(a * NT + b) / WAVES];
      const half4_t o = dec_epilogue4<LNF>(v4, mu, rstd, s1, cf, bias, res != nullptr, r4, n, act);
      if (out) *reinterpret_cast<half4_t*>(out + (size_t)row * ldo + n) = o;
      if (out_frag) *reinterpret_cast<half4_t*>(out_frag + frag_off(row, n, N >> 5)) = o;
    }
  }
}

// ------------------------------------------------------------------------------------
// GEMM-shaped, LDS-staged form of the decoder linear for MERGED decode runs (a thousand rows and more).  The
// register-streaming kernel above re-reads every operand fragment once per workgroup tile and runs into the ~33 B/clk a
// CU's vector-memory path delivers; here a workgroup of WM x WN waves stages the fragments of a (64 WM) x (16 FB WN)
// tile ONCE in LDS for all its waves:
//   * one wave owns 64 rows x 16 FB columns (4 x FB tiles of 16 x 16: every fragment read from LDS feeds 4 or FB MFMAs);
//   * operands are FRAGMENT-MAJOR in HBM, so a k-step of the workgroup tile is 4 WM + FB WN contiguous 1 KB pieces: one
//     global_load_lds per piece (a wave instruction moves exactly one MFMA fragment), the LDS image is the fragment
//     itself and every ds_read_b128 is lane-linear: no swizzle, no bank conflicts;
//   * ring of NST stages of KC k-steps, ONE barrier per stage: wait (counted vmcnt, NST - 2 stages stay in flight)
//     -> barrier -> refill the stage read last -> 4 FB KC MFMAs per wave;
//   * epilogue through the idle ring: the accumulator layout gives a lane 4 consecutive columns of ONE row (16 stores
//     of 8 bytes into 16 different lines per instruction, twice: row-major and fragment-major — store-issue bound,
//     measured ~5 us per launch).  Instead the residual tile is fetched in whole 128-byte row segments into the wave's
//     LDS patch, every lane computes its four values there (the pinned dec_epilogue4 arithmetic, unchanged), and the
//     finished tile leaves in 16-byte pieces: row segments for the row-major copy, whole 1 KB fragments for the
//     fragment-major copy the next linear reads;
//   * BIT-IDENTICAL to dec_gemm_frag_kernel<S waves>: that kernel cuts K into S slices of consecutive k-steps, runs one
//     MFMA chain per slice from a zero accumulator and adds the slice partials in slice order starting from 0.0f
//     (likewise the LayerNorm statistics).  The same chains and the same additions are made here by one wave — at a
//     slice boundary  total += acc; acc = 0  — and both kernels share the pinned epilogue arithmetic.  So a merged run
//     returns exactly what each caller's solo run returns (tests/test_gpu_kernels.py::test_dec_linear_big_bit_identical).
// What was measured on the way (profiles/r03_dec_linear_bench.txt, NOTES.md): staging alone costs ~670 cycles per 16 KB
// k-step with four waves issuing (25 B/clk per CU: the LDS-DMA issue, not its latency — a deeper ring, or touching the
// lines of later stages into L2 ahead of time, made it slower), the MFMA side alone ~550.
// ------------------------------------------------------------------------------------
// cycle stamps for profiles/ubench/dec_big_timeline.hip (defined there before this file is included); nothing otherwise
#ifndef DGB_TL
#define DGB_TL(slot) do {} while (0)
#define DGB_TL_DECL
#endif
template <int N_>
static __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
// workgroup barrier without the fence of __syncthreads() (which would drain the DMA queue: vmcnt(0)); the counted
// vmcnt before it is what orders the staged bytes, as in gemm.hip
#define DGB_BARRIER()                      \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();          \
    asm volatile("" ::: "memory");         \
  } while (0)

// workgroup barrier WITH the LDS fence (plain data exchange through LDS, outside the K loop)
#define DGB_BARRIER_LDS() __syncthreads()

template <bool LNF, int S, int WM, int WN, int FB, int KC, int NST>
__global__ __launch_bounds__(WM * WN * 64, 2) void dec_gemm_big_kernel(
    const half_t* __restrict__ xf, const half_t* __restrict__ Wf, const half_t* __restrict__ bias,
    const float* __restrict__ s1, const float* __restrict__ cf, const half_t* __restrict__ res, int ldr,
    half_t* __restrict__ out, int ldo, half_t* __restrict__ out_frag, int R, int N, int K, int act, int nNt) {
  constexpr int NW = WM * WN;
  DGB_TL_DECL;
  constexpr int PX = 4 * WM, PW = FB * WN, PCS = PX + PW;   // fragments (1 KB pieces) of one k-step: x row tiles, W column tiles
  static_assert((PCS * KC) % NW == 0, "pieces of a stage must divide over the waves");
  static_assert(FB == 2 || FB == 4, "column tiles per wave");
  constexpr int PPW = PCS * KC / NW;                       // pieces a wave issues per stage
  constexpr int STAGE_BYTES = PCS * KC * 1024;
  static_assert(NST >= 3 && NST <= 6, "ring depth");
  // bytes per staged output row: the patch row + 16 (conflict-free 8-byte column writes, 16-byte aligned rows)
  constexpr int DGB_EP_STRIDE = FB * 32 + 16;
  static_assert(NST * STAGE_BYTES >= NW * 64 * DGB_EP_STRIDE, "the epilogue stages one 64-row patch per wave through the ring");
  extern __shared__ __attribute__((aligned(16))) char dgb_smem[];   // NST stages (the only LDS object of the kernel)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  // consecutive tiles run on one XCD (xcd_remap) and share the W column tile: the cold operand (a step's weights are
  // 1.5 GB, never cached from one use to the next) is fetched once per XCD
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int nMt = gridDim.x / nNt;
  const int nt = bid / nMt, mt = bid - nt * nMt;
  const int rt0 = mt * PX, ct0 = nt * PW;
  const int n_rt = (R + 15) >> 4;
  const int KS = K >> 5;
  const int per = KS / S;                 // k-steps per slice (the launcher guarantees KS % S == 0, per % KC == 0)
  const int nch = KS / KC;                // stages' worth of K (the launcher guarantees nch >= NST)
  const int ch_per_slice = per / KC;

  // staging: piece p of a stage = (tile t = p / KC, k-step p % KC); tiles 0 .. PX-1 are x row tiles, PX .. PCS-1 W
  // column tiles; wave w issues pieces w * PPW .. + PPW - 1.  32-bit offsets from the wave-uniform operand bases (the
  // DMA instruction takes SGPR base + VGPR offset); which base a piece uses is a compile-time fact per (wave, q) only
  // when PPW divides the x / W boundary, so the base is picked with a wave-uniform select.
  unsigned soff[PPW];
  const char* sbase[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int p = wave * PPW + q;
    const int t = p / KC, ks = p % KC;
    if (t < PX) {
      int rt = rt0 + t;
      if (rt > n_rt - 1) rt = n_rt - 1;   // a missing row tile re-reads the last one; its result is dropped
      sbase[q] = reinterpret_cast<const char*>(xf);
      soff[q] = (unsigned)((((size_t)rt * KS + ks) * 64 + lane) * 16);
    } else {
      int ct = ct0 + t - PX;
      if (ct > (N >> 4) - 1) ct = (N >> 4) - 1;   // a missing column tile re-reads the last one; never stored
      sbase[q] = reinterpret_cast<const char*>(Wf);
      soff[q] = (unsigned)((((size_t)ct * KS + ks) * 64 + lane) * 16);
    }
  }
  auto issue = [&](int c, int slot) {   // K stage c -> ring slot
    char* dst = dgb_smem + slot * STAGE_BYTES + wave * PPW * 1024;
    const unsigned koff = (unsigned)c * (KC * 1024);   // KC k-steps further along every tile's fragment run
#pragma unroll
    for (int q = 0; q < PPW; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sbase[q] + (size_t)(soff[q] + koff)),
                                       (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
  };

  static_assert(!LNF || WN == 2, "the LayerNorm statistics are shared by the TWO waves of a row block");
  floatx4 acc[4][FB], tot[4][FB];
  float rs[2] = {0.f, 0.f}, rq[2] = {0.f, 0.f}, sal[2] = {0.f, 0.f}, sbl[2] = {0.f, 0.f};   // this wave's two row tiles
  float sa[4], sb[4];                                                                     // all four, after the swap
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    sa[a] = 0.f; sb[a] = 0.f;
#pragma unroll
    for (int b = 0; b < FB; ++b) { acc[a][b] = floatx4{0, 0, 0, 0}; tot[a][b] = floatx4{0, 0, 0, 0}; }
  }

  half8_t xv[KC][4], wv[KC][FB];          // the stage's fragments: fetch() -> LDS reads, mma() -> the MFMAs
  auto fetch = [&](int slot) {
    const char* st = dgb_smem + slot * STAGE_BYTES;
    // every fragment of the stage is requested up front — the x fragments first, then the W fragments — and the MFMAs run W
    // fragment by W fragment: the LayerNorm statistics need only the x fragments and the first four MFMAs only the first W
    // fragment, so the compiler's counted LDS waits let the matrix pipe start while the later W fragments are still on their
    // way (round 6: with W first and the loops the other way round the stream opened with `s_waitcnt lgkmcnt(0)` — the
    // whole 8 KB read burst of the wave exposed in front of its first MFMA: profiles/r06_dec_big_timeline.txt).  The order
    // of MFMAs on DIFFERENT accumulators does not enter any result.
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
        xv[ks][a] = *reinterpret_cast<const half8_t*>(st + (((wm * 4 + a) * KC + ks) * 64 + lane) * 16);
#pragma unroll
      for (int b = 0; b < FB; ++b)
        wv[ks][b] = *reinterpret_cast<const half8_t*>(st + (((PX + wn * FB + b) * KC + ks) * 64 + lane) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma = [&]() {                      // the MFMAs of the fetched stage
    __builtin_amdgcn_sched_barrier(0);
    DGB_TL(4);
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) {
      if (LNF) {
        // The two waves of a row block (same wm, wn = 0 / 1) read the same four x fragments: each accumulates the
        // statistics of TWO of the four row tiles — wn = 0 tiles 0, 1; wn = 1 tiles 2, 3 — and they swap the sums before the
        // epilogue (round 6: 16 instead of 32 v_dot2c per k-step and wave; a row's sums are made by one wave with exactly
        // the instructions both made before: the same bits).  The wave's two tiles are picked with wave-uniform selects on
        // the fragment registers (a branch, or an index, would put the sums into scratch memory).
        const half2_t one2 = {(half_t)1.f, (half_t)1.f};
        const half8_t xs0 = wn ? xv[ks][2] : xv[ks][0], xs1 = wn ? xv[ks][3] : xv[ks][1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half2_t h0 = {xs0[2 * e], xs0[2 * e + 1]}, h1 = {xs1[2 * e], xs1[2 * e + 1]};
          rs[0] = __builtin_amdgcn_fdot2(h0, one2, rs[0], false);
          rq[0] = __builtin_amdgcn_fdot2(h0, h0, rq[0], false);
          rs[1] = __builtin_amdgcn_fdot2(h1, one2, rs[1], false);
          rq[1] = __builtin_amdgcn_fdot2(h1, h1, rq[1], false);
        }
      }
#pragma unroll
      for (int b = 0; b < FB; ++b) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv[ks][b], xv[ks][a], acc[a][b], 0, 0, 0);
      }
    }
  };
  auto compute = [&](int slot) { fetch(slot); mma(); };
  auto slice_end = [&]() {                // the skinny kernel's fixed-order reduction, one term at a time
    if (LNF) {                            // (the wave's two row tiles: wn * 2 + 0 / 1)
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        float pa = rs[aa], pb = rq[aa];
        pa += __shfl_xor(pa, 16, 64); pa += __shfl_xor(pa, 32, 64);
        pb += __shfl_xor(pb, 16, 64); pb += __shfl_xor(pb, 32, 64);
        sal[aa] += pa; sbl[aa] += pb;
        rs[aa] = 0.f; rq[aa] = 0.f;
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int b = 0; b < FB; ++b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tot[a][b][e] += acc[a][b][e];
        acc[a][b] = floatx4{0, 0, 0, 0};
      }
    }
  };

  // (Round 6 built the encoder GEMM's recipe here too — two wave groups staggered by half a k-step, one issuing the DMA of the
  //  stage four k-steps ahead while the other multiplies, ring of 6 / 5 — bit-identical and 10 % SLOWER (39.9 vs 35.9 us for qkv at
  //  1 280 rows): the stamps of profiles/ubench/dec_big_timeline.hip show a group's DMA issue at 160-190 cycles of a k-step and
  //  its multiply segment at ~1 050, and the stagger doubles the barriers.  Removed; the code is in the history: commit 1d52bf7,
  //  profiles/r06_dec_linear_bench_call2_stagger.txt.)
#pragma unroll
  for (int c0 = 0; c0 < NST - 1; ++c0) issue(c0, c0);
  int slot = 0, fill = NST - 1;           // slot of stage c; slot refilled in iteration c (= slot of stage c - 1)
  int to_slice = ch_per_slice;
  const int n_steady = nch - (NST - 1);
  // steady state: stages c .. c + NST - 2 are in flight; wait until only the NST - 2 younger ones are
  for (int c = 0; c < n_steady; ++c) {
    DGB_TL(0);
    wait_vmcnt<(NST - 2) * PPW>();
    DGB_TL(1);
    DGB_BARRIER();   // stage c has landed for every wave, and every wave has finished reading stage c - 1 ...
    DGB_TL(2);
    // the LDS reads of stage c leave BEFORE the DMA of stage c + NST - 1 is issued (round 6): the ~270 cycles a wave spends
    // issuing its global_load_lds pieces now run under the latency of its own 8 KB read burst instead of in front of it
    // (profiles/r06_dec_big_timeline.txt: barrier | issue 267-478 | reads + lgkmcnt(0) wait | MFMAs, all in series)
    fetch(slot);
    DGB_TL(3);
    issue(c + NST - 1, fill);             // ... whose slot (stage c - 1's) is the one refilled now
    mma();
    DGB_TL(5);
    if (--to_slice == 0) { to_slice = ch_per_slice; slice_end(); }
    fill = slot;
    slot = (slot + 1 == NST) ? 0 : slot + 1;
  }
  // tail: nothing left to issue; the queue is drained once
  wait_vmcnt<0>();
#pragma unroll 1
  for (int c = n_steady; c < nch; ++c) {
    DGB_BARRIER();
    compute(slot);
    if (--to_slice == 0) { to_slice = ch_per_slice; slice_end(); }
    slot = (slot + 1 == NST) ? 0 : slot + 1;
  }
  DGB_BARRIER();     // every wave is done with the ring: it becomes the epilogue's staging area

  if (LNF) {
    // every wave publishes the sums of its two row tiles; row tile a of this wave's row block was accumulated by the wave
    // with wn = a >> 1 as its tile a & 1
    float* xch = reinterpret_cast<float*>(dgb_smem);       // [wave][tile 0 / 1][row 16][sum, sum of squares]
    if (g == 0) {
#pragma unroll
      for (int aa = 0; aa < 2; ++aa) {
        xch[((wave * 2 + aa) * 16 + i) * 2] = sal[aa];
        xch[((wave * 2 + aa) * 16 + i) * 2 + 1] = sbl[aa];
      }
    }
    DGB_BARRIER_LDS();
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int owner = (wave & ~1) | (a >> 1);
      sa[a] = xch[((owner * 2 + (a & 1)) * 16 + i) * 2];
      sb[a] = xch[((owner * 2 + (a & 1)) * 16 + i) * 2 + 1];
    }
    DGB_BARRIER_LDS();                                     // the staging area below overwrites the exchange words
  }
  // ---------------------------------- epilogue ----------------------------------
  char* ep = dgb_smem + wave * (64 * DGB_EP_STRIDE);       // this wave's 64-row x (16 FB)-column patch, fp16
  const int row0 = (rt0 + wm * 4) * 16;                    // first row / column of the wave's patch
  const int col0 = (ct0 + wn * FB) * 16;
  constexpr int CPR = FB * 2;                              // 16-byte chunks per patch row
  constexpr int RPI = 64 / CPR;                            // patch rows one wave instruction covers
  const int cr = lane / CPR, cc = lane % CPR;              // row-segment pass: this lane's row within the group, chunk
  if (res) {   // residual patch -> LDS in whole row segments (a wave's LDS operations execute in order)
#pragma unroll
    for (int j = 0; j < 64 / RPI; ++j) {
      const int r = j * RPI + cr;
      const int row = row0 + r, n = col0 + cc * 8;
      intx4 v = {0, 0, 0, 0};
      if (row < R && n < N) v = *reinterpret_cast<const intx4*>(res + (size_t)row * ldr + n);
      *reinterpret_cast<intx4*>(ep + r * DGB_EP_STRIDE + cc * 16) = v;
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    float mu = 0.f, rstd = 1.f;
    if (LNF) dec_ln_stats(sa[a], sb[a], K, mu, rstd);
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      int n = col0 + b * 16 + 4 * g;                       // this lane: D[n + e][row], e = 0..3
      if (n > N - 4) n = N - 4;                            // clamped columns are never stored
      char* cell = ep + (a * 16 + i) * DGB_EP_STRIDE + (b * 16 + 4 * g) * 2;
      half4_t r4 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
      if (res) r4 = *reinterpret_cast<const half4_t*>(cell);
      const half4_t o = dec_epilogue4<LNF>(tot[a][b], mu, rstd, s1, cf, bias, res != nullptr, r4, n, act);
      *reinterpret_cast<half4_t*>(cell) = o;
    }
  }
  if (out) {   // row-major copy: whole row segments, 16 bytes per lane
#pragma unroll
    for (int j = 0; j < 64 / RPI; ++j) {
      const int r = j * RPI + cr;
      const int row = row0 + r, n = col0 + cc * 8;
      const intx4 v = *reinterpret_cast<const intx4*>(ep + r * DGB_EP_STRIDE + cc * 16);
      if (row < R && n < N) *reinterpret_cast<intx4*>(out + (size_t)row * ldo + n) = v;
    }
  }
  if (out_frag) {   // fragment-major copy: one whole 1 KB fragment (16 rows x 32 columns) per wave instruction
    const int KSo = N >> 5;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int kb = 0; kb < FB / 2; ++kb) {
        const int row = row0 + a * 16 + i, n = col0 + kb * 32 + g * 8;   // lane (i, g): row i, column octet g
        const intx4 v = *reinterpret_cast<const intx4*>(ep + (a * 16 + i) * DGB_EP_STRIDE + (kb * 32 + g * 8) * 2);
        if (row < R && n < N)
          *reinterpret_cast<intx4*>(out_frag + ((size_t)((row >> 4) * KSo + (n >> 5)) * 64 + lane) * 8) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// int8 form of the fragment-major skinny GEMM (compute_type int8_float16).  Same structure as
// dec_gemm_frag_kernel with v_mfma_i32_16x16x64_i8: a 16-byte fragment holds 16 int8 (k-step = 64), weights are
// permuted at pack time, activations are written fragment-major by quant_rows_kernel(frag = 1); the epilogue
// de-quantises with the per-row activation scale and the per-row weight scale.  Output fp16 row-major.
// ------------------------------------------------------------------------------------
template <int WAVES, int RT, int NT, int CH_ = 0>
__global__ __launch_bounds__(WAVES * 64) void dec_gemm_frag_i8_kernel(
    const int8_t* __restrict__ xq, const float* __restrict__ x_scale, const int8_t* __restrict__ Wq,
    const float* __restrict__ w_scale, const half_t* __restrict__ bias, const half_t* __restrict__ res, int ldr,
    half_t* __restrict__ out, int ldo, int R, int N, int K, int act) {
  __shared__ int red[WAVES][RT * NT][64][4];
  constexpr int CH = CH_ ? CH_ : 20 / (RT + NT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int ct0 = blockIdx.x * NT, rt0 = blockIdx.y * RT;
  const int n_rt = (R + 15) >> 4;
  const int KS = K >> 6;
  const int per = (KS + WAVES - 1) / WAVES;
  const int ks0 = wave * per;
  int nks = KS - ks0;
  if (nks > per) nks = per;
  intx4 acc[RT][NT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = intx4{0, 0, 0, 0};
  if (nks > 0) {
    const intx4* wp[NT];
    const intx4* xp[RT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
      wp[b] = reinterpret_cast<const intx4*>(Wq) + ((size_t)(ct0 + b) * KS + ks0) * 64 + lane;
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      int rt = rt0 + a;
      if (rt > n_rt - 1) rt = n_rt - 1;
      xp[a] = reinterpret_cast<const intx4*>(xq) + ((size_t)rt * KS + ks0) * 64 + lane;
    }
    for (int c = 0; c < nks; c += CH) {
      intx4 wv[NT][CH], xv[RT][CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const int jj = (c + j < nks) ? c + j : nks - 1;
#pragma unroll
        for (int b = 0; b < NT; ++b) wv[b][j] = wp[b][(size_t)jj * 64];
#pragma unroll
        for (int a = 0; a < RT; ++a) xv[a][j] = xp[a][(size_t)jj * 64];
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (c + j < nks) {
#pragma unroll
          for (int a = 0; a < RT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[b][j], xv[a][j], acc[a][b], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][a * NT + b][lane][e] = acc[a][b][e];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < RT; ++a) {
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      if ((a * NT + b) % WAVES != wave) continue;
      const int row = (rt0 + a) * 16 + i;
      if (row >= R) continue;
      int v[4] = {0, 0, 0, 0};
#pragma unroll
      for (int w = 0; w < WAVES; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += red[w][a * NT + b][lane][e];      // exact: integer accumulation
      const float sx = x_scale[row];
      const int n = (ct0 + b) * 16 + 4 * g;
      const half4_t o = dec_epilogue4_i8(v, sx, w_scale, bias, res, ldr, row, n, act);
      *reinterpret_cast<half4_t*>(out + (size_t)row * ldo + n) = o;
    }
  }
}

// ------------------------------------------------------------------------------------
// K16: the vocabulary projection (and any linear with MANY column tiles): full K per wave.
//
// Same fragment-major operands as above, but here a wave owns RT row tiles x NT column tiles over the WHOLE K
// (no split, no reduction, no LDS, no barrier): N = 51 866 gives 3 242 column tiles, so the chip is filled by
// columns alone.  The K loop is software-pipelined by hand over two register sets of CH k-steps each: the loads
// of the next set are issued before the MFMAs of the current one.  A workgroup is WM x WN such waves; its waves
// share the x fragments through the CU's L1.  RT = 5 covers the 80 rows of a 16-chunk x beam-5 step in one wave,
// so every weight byte is requested exactly once per 80 rows.
// I8: int8 operands (v_mfma_i32_16x16x64_i8, 64 elements per 16-byte fragment) with per-row scales.
// ------------------------------------------------------------------------------------
// (the folded form with RT = 5 sits at 172 registers — one allocation granule above the 168 that let a third wave per SIMD in;
//  __launch_bounds__(256, 3) gets there with 8 spilled values and is 11 % SLOWER: 2.87 vs 2.53 ms per batch in one A/B of
//  three builds on one box, profiles/r06_ab_logits_split.jsonl)
template <bool I8, bool LNF, bool F32, int RT, int NT, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void dec_gemm_wave_kernel(
    const void* __restrict__ xfv, const float* __restrict__ x_scale, const void* __restrict__ Wfv,
    const float* __restrict__ w_scale, const half_t* __restrict__ bias, const float* __restrict__ s1,
    const float* __restrict__ cf, void* __restrict__ outv, int ldo, int R, int N, int K, int n_rg) {
  constexpr int CH = 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int i = lane & 15, g = lane >> 4;
  const int n_rt = (R + 15) >> 4, n_ct = (N + 15) >> 4;
  // 1-D grid: blocks 8k .. 8k+7 are eight column groups of ONE row group, the next eight blocks the same columns
  // of the next row group — the hardware places block b on XCD b % 8, so the n_rg workgroups that stream the
  // same weight columns run back to back on the same XCD and all but the first find them in its L2
  const int b8 = blockIdx.x >> 3;
  const int cgrp = (b8 / n_rg) * 8 + (blockIdx.x & 7), rgrp = b8 % n_rg;
  const int rt0 = (rgrp * WM + wm) * RT, ct0 = (cgrp * WN + wn) * NT;
  // SPLIT (round 6): the four waves of a workgroup (WM = 1: the same row tiles, four column groups) each accumulate the
  // folded LayerNorm's statistics of ONE or TWO of the row tiles — wave wn: tile wn, wave 0 also tile 4 — instead of all
  // RT of them each, and swap the sums through LDS before the epilogue: 8-16 instead of 40 v_dot2c per k-step and wave
  // beside 10 MFMAs (measured: 2.60 -> 2.53 ms per batch — the kernel waits on its weight stream, not on its vector
  // instructions; kept because it is free).  A row's sums are made by one
  // wave with the instructions every wave used before: the same bits.  The swap needs all four waves at ONE barrier, so the
  // waves of the padded last column groups stay (clamped loads, nothing stored) instead of leaving.
  constexpr bool SPLIT = LNF && WM == 1 && WN == 4 && RT >= 4;
  __shared__ float xst[SPLIT ? RT : 1][16][2];
  if (rt0 >= n_rt) return;                    // (the whole workgroup: WM = 1 when SPLIT)
  if (!SPLIT && ct0 >= n_ct) return;          // whole waves leave: there is no barrier in the un-split kernel
  const int KS = K / (I8 ? 64 : 32);
  const intx4* wp[NT];
  const intx4* xp[RT];
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    int ct = ct0 + b; if (ct > n_ct - 1) ct = n_ct - 1;       // a missing tile re-reads the last one; result dropped
    wp[b] = reinterpret_cast<const intx4*>(Wfv) + (size_t)ct * KS * 64 + lane;
  }
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    int rt = rt0 + a; if (rt > n_rt - 1) rt = n_rt - 1;
    xp[a] = reinterpret_cast<const intx4*>(xfv) + (size_t)rt * KS * 64 + lane;
  }
  floatx4 accf[RT][NT];
  intx4 acci[RT][NT];
  float rs[RT], rq[RT];
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    rs[a] = 0.f; rq[a] = 0.f;
#pragma unroll
    for (int b = 0; b < NT; ++b) { accf[a][b] = floatx4{0, 0, 0, 0}; acci[a][b] = intx4{0, 0, 0, 0}; }
  }
  struct Set { intx4 w[CH][NT]; intx4 x[CH][RT]; };
  auto load = [&](Set& f, int ks0) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      int ks = ks0 + j; if (ks > KS - 1) ks = KS - 1;        // clamped tail slot: loaded, never multiplied
#pragma unroll
      for (int b = 0; b < NT; ++b) f.w[j][b] = wp[b][(size_t)ks * 64];
#pragma unroll
      for (int a = 0; a < RT; ++a) f.x[j][a] = xp[a][(size_t)ks * 64];
    }
  };
  auto compute = [&](const Set& f, int ks0) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      if (ks0 + j < KS) {
#pragma unroll
        for (int a = 0; a < RT; ++a) {
#pragma unroll
          for (int b = 0; b < NT; ++b) {
            if (I8) acci[a][b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(f.w[j][b], f.x[j][a], acci[a][b], 0, 0, 0);
            else accf[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, f.w[j][b]),
                                                                    __builtin_bit_cast(half8_t, f.x[j][a]), accf[a][b],
                                                                    0, 0, 0);
          }
          if (LNF && !SPLIT) {
            const half8_t xv = __builtin_bit_cast(half8_t, f.x[j][a]);
            const half2_t one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const half2_t h2 = {xv[2 * e], xv[2 * e + 1]};
              rs[a] = __builtin_amdgcn_fdot2(h2, one2, rs[a], false);
              rq[a] = __builtin_amdgcn_fdot2(h2, h2, rq[a], false);
            }
          }
        }
        if (SPLIT) {
          // this wave's tile wn, picked with wave-uniform selects on the fragment registers (an index or a branch around an
          // array element would put the sums into scratch memory); sums in rs[0] / rq[0]; wave 0: also tile 4 in rs[1] / rq[1]
          const half2_t one2 = {(half_t)1.f, (half_t)1.f};
          const intx4 xsel = wn == 0 ? f.x[j][0] : (wn == 1 ? f.x[j][1] : (wn == 2 ? f.x[j][2] : f.x[j][RT > 3 ? 3 : 0]));
          const half8_t xv = __builtin_bit_cast(half8_t, xsel);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const half2_t h2 = {xv[2 * e], xv[2 * e + 1]};
            rs[0] = __builtin_amdgcn_fdot2(h2, one2, rs[0], false);
            rq[0] = __builtin_amdgcn_fdot2(h2, h2, rq[0], false);
          }
          if (RT > 4 && wn == 0) {
            const half8_t xw = __builtin_bit_cast(half8_t, f.x[j][RT > 4 ? 4 : 0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const half2_t h2 = {xw[2 * e], xw[2 * e + 1]};
              rs[1] = __builtin_amdgcn_fdot2(h2, one2, rs[1], false);
              rq[1] = __builtin_amdgcn_fdot2(h2, h2, rq[1], false);
            }
          }
        }
      }
    }
  };
  Set fa, fb;
  load(fa, 0);
  for (int ks0 = 0; ks0 < KS; ks0 += 2 * CH) {
    load(fb, ks0 + CH);
    compute(fa, ks0);
    load(fa, ks0 + 2 * CH);
    compute(fb, ks0 + CH);
  }
  if (SPLIT) {   // publish this wave's tile(s): row i of a tile, the 4 k-octet lanes hold the partial sums over the whole K
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (q == 1 && !(RT > 4 && wn == 0)) break;
      float sa = rs[q], sb = rq[q];
      sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
      sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
      if (g == 0) { xst[q == 0 ? wn : 4][i][0] = sa; xst[q == 0 ? wn : 4][i][1] = sb; }
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    float mu = 0.f, rstd = 1.f;
    if (LNF) {   // row i of tile a: the 4 k-octet lanes hold the partial sums over the whole K
      float sa, sb;
      if (SPLIT) {
        sa = xst[a][i][0]; sb = xst[a][i][1];
      } else {
        sa = rs[a]; sb = rq[a];
        sa += __shfl_xor(sa, 16, 64); sa += __shfl_xor(sa, 32, 64);
        sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
      }
      mu = sa / (float)K;
      rstd = rsqrtf(fmaxf(sb / (float)K - mu * mu, 0.f) + 1e-5f);
    }
    const int row = (rt0 + a) * 16 + i;
    if (rt0 + a >= n_rt || row >= R) continue;
    const float sx = I8 ? x_scale[row] : 1.f;
    const bool pair_ok = (ldo & 1) == 0;      // (wave-uniform) every row starts 8-byte aligned
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      if (ct0 + b >= n_ct) continue;
      const int n = (ct0 + b) * 16 + 4 * g;   // this lane: D[n + e][row], e = 0..3
      float t4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ne = n + e < N ? n + e : N - 1;   // (clamped: computed, never stored)
        float tv = I8 ? (float)acci[a][b][e] * sx * w_scale[ne] : accf[a][b][e];
        if (LNF) tv = rstd * (tv - mu * s1[ne]) + cf[ne];
        else if (bias) tv += (float)bias[ne];
        t4[e] = tv;
      }
      if (F32 && pair_ok) {
        // a lane's four logits are consecutive columns of one row: two 8-byte stores (n and ldo are even: aligned)
        // instead of four 4-byte ones — 265 MB of logits leave a 1 280-row step through 40 store instructions per wave
        float* o = reinterpret_cast<float*>(outv) + (size_t)row * ldo + n;
        if (n + 1 < N) *reinterpret_cast<floatx2*>(o) = floatx2{t4[0], t4[1]};
        else if (n < N) o[0] = t4[0];
        if (n + 3 < N) *reinterpret_cast<floatx2*>(o + 2) = floatx2{t4[2], t4[3]};
        else if (n + 2 < N) o[2] = t4[2];
        continue;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= N) continue;
        if (F32) reinterpret_cast<float*>(outv)[(size_t)row * ldo + n + e] = t4[e];
        else reinterpret_cast<half_t*>(outv)[(size_t)row * ldo + n + e] = (half_t)t4[e];
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// K13: decoder self-attention, KV cache with slot indirection.  cache layout [slot][H][cache_ctx][64] (cache_ctx =
// the positions the RUN can reach, <= n_ctx: the cache is sized per run, decoder.hip).  The new K/V
// (position pos) are written to the row's own slot; older positions are read from kvidx[row][p] (slot inside the
// chunk).  One workgroup = (head, chunk), one wave per beam row of the chunk: the beams of a chunk share most of
// their history (kvidx points them at the same slots), so the rows a wave fetches are in the CU's L1 / the XCD's L2
// when its sibling asks for them.  QK and PV both read a position's 128-byte row with 8 lanes x 16 bytes (every
// load instruction = 8 whole rows, 8 position groups per wave, 4 loads in flight per lane), reduced by shuffles.
// ------------------------------------------------------------------------------------
#define SA_MAX_CTX 448
// sum over the 8 lanes that share a position (lanes 8g .. 8g+7), result in all of them: two quad permutes and a
// half-row mirror on the DPP path of the VALU instead of three ds_bpermute round trips through the LDS crossbar (what
// __shfl_xor compiles to: the dependent chain of a score was 3 x (bpermute + lgkmcnt wait + add), ~250 cycles; the
// kernel is issue / latency bound, not bandwidth bound).  Same additions as the xor-1 / xor-2 / xor-4 butterfly — fp
// addition commutes — so every lane gets the bits it got before.
static __device__ __forceinline__ float sum8_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return v;
}
// (8 waves per SIMD: the kernel waits on memory 70 % of its wave cycles (profiles/r03_pmc_sq.json) with 70 registers = 7
//  waves; at 55 registers it spills nothing and a CU holds six (head, chunk) workgroups of five beams instead of five)
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void dec_self_attn_kernel(const half_t* __restrict__ qkv, int d, half_t* __restrict__ kc,
                                                             half_t* __restrict__ vc, int n_ctx, int cache_ctx,
                                                             int H, const uint8_t* __restrict__ kvidx2, int Kbeam, int kmul,
                                                             half_t* __restrict__ out, const int* __restrict__ d_step,
                                                             int pos_fixed, int P, int R_total, int frag) {
  extern __shared__ float sa_smem[];            // per wave: sp[n_ctx] floats, ssrc[n_ctx] ints
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* sp = sa_smem + (size_t)wave * 2 * n_ctx;
  int* ssrc = reinterpret_cast<int*>(sp + n_ctx);
  const int h = blockIdx.x, c = blockIdx.y;
  const int r = c * kmul + wave, kb = wave;
  const int step = *d_step;
  const int pos = pos_fixed >= 0 ? pos_fixed : P - 1 + step;
  const int slot = c * Kbeam + kb;
  const int cur = (pos_fixed >= 0) ? 0 : (step & 1);
  const uint8_t* kvidx = kvidx2 + ((size_t)cur * R_total + slot) * n_ctx;
  const half_t* qr = qkv + (size_t)r * 3 * d + h * 64;
  const size_t head_stride = (size_t)cache_ctx * 64;
  const size_t slot_stride = (size_t)H * head_stride;
  const int pg = lane >> 3, cc = lane & 7;     // position group, 16-byte chunk of the 128-byte row
  // this lane's 8 dims of q (scaled), k_new, v_new
  const half8_t q8 = *reinterpret_cast<const half8_t*>(qr + cc * 8);
  const half8_t kn8 = *reinterpret_cast<const half8_t*>(qr + d + cc * 8);
  const half8_t vn8 = *reinterpret_cast<const half8_t*>(qr + 2 * d + cc * 8);
  float q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] = (float)q8[e] * 0.125f;
  if (pg == 0) {
    *reinterpret_cast<half8_t*>(kc + slot * slot_stride + h * head_stride + (size_t)pos * 64 + cc * 8) = kn8;
    *reinterpret_cast<half8_t*>(vc + slot * slot_stride + h * head_stride + (size_t)pos * 64 + cc * 8) = vn8;
  }
  auto dot8 = [&](const half8_t& k) {
    float sacc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sacc += q[e] * (float)k[e];
    return sum8_dpp(sacc);                     // the 8 lanes of a position all hold its score
  };
  const float s_new = dot8(kn8);
  float mx = s_new;
  const half_t* kbase = kc + h * head_stride + cc * 8;
  for (int p0 = 0; p0 < pos; p0 += 32) {
    half8_t kv[4];
    int src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + 8 * j + pg;
      const int pc = p < pos ? p : pos - 1;    // clamped: a tail slot re-reads the last position, unused
      src[j] = c * Kbeam + kvidx[pc];
      kv[j] = *reinterpret_cast<const half8_t*>(kbase + src[j] * slot_stride + (size_t)pc * 64);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + 8 * j + pg;
      const float sc = dot8(kv[j]);
      if (p < pos) {
        if (cc == 0) { sp[p] = sc; ssrc[p] = src[j]; }
        mx = fmaxf(mx, sc);
      }
    }
  }
  mx = wave_max(mx);
  // sp / ssrc are private to the wave and a wave's LDS operations execute in order: no workgroup barrier, only a
  // compiler-level one so that the cross-lane reads below are not moved above the writes
  __builtin_amdgcn_wave_barrier();
  float sum = 0.f;
  for (int p = lane; p < pos; p += 64) {
    const float e = __expf(sp[p] - mx);
    sp[p] = e;
    sum += e;
  }
  const float e_new = __expf(s_new - mx);
  sum = wave_sum(sum) + e_new;
  __builtin_amdgcn_wave_barrier();
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  const half_t* vb = vc + h * head_stride + cc * 8;
#pragma unroll 4
  for (int p = pg; p < pos; p += 8) {
    const half8_t vv = *reinterpret_cast<const half8_t*>(vb + ssrc[p] * slot_stride + (size_t)p * 64);
    const float w = sp[p];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += w * (float)vv[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float a = acc[e];
    a += __shfl_xor(a, 8, 64);
    a += __shfl_xor(a, 16, 64);
    a += __shfl_xor(a, 32, 64);
    acc[e] = a;
  }
  if (pg == 0) {
    const float inv = 1.f / sum;
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((acc[e] + e_new * (float)vn8[e]) * inv);
    half_t* dst = frag ? out + frag_off(r, h * 64 + cc * 8, d >> 5) : out + (size_t)r * d + h * 64 + cc * 8;
    *reinterpret_cast<half8_t*>(dst) = o;
  }
}

// ------------------------------------------------------------------------------------
// K13, second form (round 5): the same arithmetic per row — same dot products, same DPP reductions, the same
// ascending-position accumulation, every fused multiply-add written as one — with the memory rounds restructured.
// What the first form above costs is DEPENDENT ROUND TRIPS, not bytes: per 32 positions one round of slot-table byte
// loads, then one of K rows; in the PV phase every V row load was followed by its own vmcnt(0) (13 rounds at 104
// positions): ~21 rounds x 0.4-0.5 us = the 10.5 us per layer of a solo step (profiles/r04_kernel_stats_w1.csv), and in
// merged runs 67 % of the wave cycles waiting (profiles/r05_pmc_sq_w32.json).  Here:
//   A  the row's slot table goes to LDS in ONE round of 4-byte loads (4 positions per lane and load);
//   B  K rows in batches of 4 U per lane, addresses from LDS; PIPE: the next batch is issued before the current one is
//      reduced;
//   C  the first V batch is issued BEFORE the softmax reductions (its addresses do not depend on them);
//   D  V rows in the same batches.
// U = 4, no PIPE (latency form, solo runs): 128 positions per round — a 104-position step is one slot-table round, one K
// round and one V round that overlaps the softmax.  U = 1, PIPE (throughput form, merged runs): 64 registers, 8 waves
// per SIMD, two batches in flight.  Both return the same bits (the row's arithmetic does not depend on U), so a merged
// run still returns what each caller gets alone.
// ------------------------------------------------------------------------------------
// PB (position block: prompt forward, align): the kmul rows of a chunk are kmul CONSECUTIVE POSITIONS pos_fixed + wave
// of the chunk's beam slot 0 instead of kmul beams at one position.  Position p of the same block is a sibling row of
// this launch: its K / V are read from the qkv buffer (the bytes the sibling writes to the cache), earlier positions
// from the cache — the same values in the same order as position-by-position launches: the same bits.
template <int U, bool PIPE, int MAXT, bool PB = false>
__global__ __launch_bounds__(MAXT) void dec_self_attn2_kernel(const half_t* __restrict__ qkv, int d, half_t* __restrict__ kc,
                                                             half_t* __restrict__ vc, int n_ctx, int cache_ctx,
                                                             int H, const uint8_t* __restrict__ kvidx2, int Kbeam, int kmul,
                                                             half_t* __restrict__ out, const int* __restrict__ d_step,
                                                             int pos_fixed, int P, int R_total, int frag) {
  extern __shared__ float sa_smem[];            // per wave: sp[n_ctx] floats, ssrc[n_ctx] ints
  constexpr int NB = 4 * U;                     // rows per lane and batch; a batch covers 8 * NB positions
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* sp = sa_smem + (size_t)wave * 2 * n_ctx;
  int* ssrc = reinterpret_cast<int*>(sp + n_ctx);
  const int h = blockIdx.x, c = blockIdx.y;
  const int r = c * kmul + wave, kb = PB ? 0 : wave;
  const int step = PB ? 0 : *d_step;
  const int pos = PB ? pos_fixed + wave : (pos_fixed >= 0 ? pos_fixed : P - 1 + step);
  const int slot = c * Kbeam + kb;
  const int cur = (pos_fixed >= 0) ? 0 : (step & 1);
  const uint8_t* kvidx = kvidx2 + ((size_t)cur * R_total + slot) * n_ctx;
  const half_t* qr = qkv + (size_t)r * 3 * d + h * 64;
  const size_t head_stride = (size_t)cache_ctx * 64;
  const size_t slot_stride = (size_t)H * head_stride;
  const int pg = lane >> 3, cc = lane & 7;     // position group, 16-byte chunk of the 128-byte row
  // ---- A: slot table -> LDS (n_ctx is a multiple of 4: launch_self_attn checks; the row starts 4-byte aligned) ----
  for (int p4 = 4 * lane; p4 < pos; p4 += 256) {
    const unsigned w = PB ? 0u : *reinterpret_cast<const unsigned*>(kvidx + p4);   // (prompt positions: beam slot 0)
    intx4 s4;
    s4[0] = c * Kbeam + (int)(w & 0xffu); s4[1] = c * Kbeam + (int)((w >> 8) & 0xffu);
    s4[2] = c * Kbeam + (int)((w >> 16) & 0xffu); s4[3] = c * Kbeam + (int)(w >> 24);
    *reinterpret_cast<intx4*>(ssrc + p4) = s4;
  }
  const half8_t q8 = *reinterpret_cast<const half8_t*>(qr + cc * 8);
  const half8_t kn8 = *reinterpret_cast<const half8_t*>(qr + d + cc * 8);
  const half8_t vn8 = *reinterpret_cast<const half8_t*>(qr + 2 * d + cc * 8);
  float q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] = (float)q8[e] * 0.125f;
  if (pg == 0) {
    *reinterpret_cast<half8_t*>(kc + slot * slot_stride + h * head_stride + (size_t)pos * 64 + cc * 8) = kn8;
    *reinterpret_cast<half8_t*>(vc + slot * slot_stride + h * head_stride + (size_t)pos * 64 + cc * 8) = vn8;
  }
  auto dot8 = [&](const half8_t& k) {
    float sacc = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) sacc = __builtin_fmaf(q[e], (float)k[e], sacc);
    return sum8_dpp(sacc);                     // the 8 lanes of a position all hold its score
  };
  const float s_new = dot8(kn8);
  float mx = s_new;
  // sp / ssrc are private to the wave and a wave's LDS operations execute in order: no workgroup barrier, only a
  // compiler-level one so that reads are not moved above the writes
  __builtin_amdgcn_wave_barrier();
  // wave-uniform base + 32-bit byte offset per lane (a layer's cache is < 4 GB: launch_self_attn checks): the loads
  // take the scalar-base form, one address register per row in flight instead of two
  const char* kbase = reinterpret_cast<const char*>(kc + h * head_stride);
  const char* vbase = reinterpret_cast<const char*>(vc + h * head_stride);
  const unsigned slot_bytes = (unsigned)(slot_stride * sizeof(half_t));
  struct Batch { half8_t row[NB]; };
  // PB: sib = K (d) / V (2 d) of the chunk's first row in the qkv buffer, this head; a sibling row is 3 d halves further
  auto fetch = [&](Batch& b, const char* base, int p0, const half_t* sib) {   // rows of positions p0 + 8 j + pg (clamped into [0, pos))
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int p = p0 + 8 * j + pg;
      const int pc = p < pos ? p : pos - 1;
      const unsigned off = (unsigned)ssrc[pc] * slot_bytes + (unsigned)(pc * 128 + cc * 16);
      if (PB && pc >= pos_fixed)
        b.row[j] = *reinterpret_cast<const half8_t*>(sib + (size_t)(pc - pos_fixed) * 3 * d + cc * 8);
      else
        b.row[j] = *reinterpret_cast<const half8_t*>(base + off);
    }
  };
  const half_t* ksib = qkv + (size_t)c * kmul * 3 * d + d + h * 64;
  const half_t* vsib = ksib + d;
  auto score = [&](const Batch& b, int p0) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int p = p0 + 8 * j + pg;
      const float sc = dot8(b.row[j]);
      if (p < pos) {
        if (cc == 0) sp[p] = sc;
        mx = fmaxf(mx, sc);
      }
    }
  };
  Batch b0, b1;
  // ---- B: scores ----
  if (pos > 0) {
    fetch(b0, kbase, 0, ksib);
    for (int p0 = 0; p0 < pos; p0 += 16 * NB) {
      if (PIPE) {
        // (unconditional prefetches — past the end they re-read the last position: a fetch under a branch makes the
        //  wait-count pass assume the worst at the join and wait for the batch just issued)
        fetch(b1, kbase, p0 + 8 * NB, ksib);
        score(b0, p0);
        fetch(b0, kbase, p0 + 16 * NB, ksib);
        score(b1, p0 + 8 * NB);
      } else {
        score(b0, p0);
        if (p0 + 8 * NB < pos) {
          fetch(b0, kbase, p0 + 8 * NB, ksib);
          score(b0, p0 + 8 * NB);
          if (p0 + 16 * NB < pos) fetch(b0, kbase, p0 + 16 * NB, ksib);
        }
      }
    }
    // ---- C: the first V batch leaves before the reductions ----
    fetch(b0, vbase, 0, vsib);
  }
  mx = wave_max_v(mx);
  __builtin_amdgcn_wave_barrier();
  float sum = 0.f;
  for (int p = lane; p < pos; p += 64) {
    const float e = __expf(sp[p] - mx);
    sp[p] = e;
    sum += e;
  }
  const float e_new = __expf(s_new - mx);
  sum = wave_sum_v(sum) + e_new;
  __builtin_amdgcn_wave_barrier();
  // ---- D: P V, positions ascending per lane (p = pg, pg + 8, ...) ----
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  auto accum = [&](const Batch& b, int p0) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int p = p0 + 8 * j + pg;
      if (p < pos) {
        const float w = sp[p];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(w, (float)b.row[j][e], acc[e]);
      }
    }
  };
  for (int p0 = 0; p0 < pos; p0 += 16 * NB) {
    if (PIPE) {
      fetch(b1, vbase, p0 + 8 * NB, vsib);
      accum(b0, p0);
      fetch(b0, vbase, p0 + 16 * NB, vsib);
      accum(b1, p0 + 8 * NB);
    } else {
      accum(b0, p0);
      if (p0 + 8 * NB < pos) {
        fetch(b0, vbase, p0 + 8 * NB, vsib);
        accum(b0, p0 + 8 * NB);
        if (p0 + 16 * NB < pos) fetch(b0, vbase, p0 + 16 * NB, vsib);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = sum_x8_x16_x32_v(acc[e]);   // the position groups: lanes 8, 16, 32 away
  if (pg == 0) {
    const float inv = 1.f / sum;
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((acc[e] + e_new * (float)vn8[e]) * inv);
    half_t* dst = frag ? out + frag_off(r, h * 64 + cc * 8, d >> 5) : out + (size_t)r * d + h * 64 + cc * 8;
    *reinterpret_cast<half8_t*>(dst) = o;
  }
}

// ------------------------------------------------------------------------------------
// K14: decoder cross-attention.  One workgroup = (chunk, head); the kmul (<= 16) beam
// queries of the chunk are the 16 columns of the MFMA tiles, so the chunk's K / V^T
// (the dominant HBM stream of the whole decode: 2*1500*64*2 B per (chunk, head, layer))
// is read ONCE for all beams, straight from HBM into MFMA operands (no LDS staging:
// every byte is used exactly once per workgroup).
//   S^T tile = K rows x Q^T     (A row i of sub-tile t holds key base+8*(i>>2)+4t+(i&3), so a
//                                lane ends with 8 CONSECUTIVE keys of its query)
//   O^T tile = V^T x P^T        (A = 8 consecutive keys of one V^T row)
// K and V^T are stored FRAGMENT-MAJOR by the projection GEMM's epilogue (gemm.hip): the 16 bytes
// lane l needs for operand run q of key group gi sit at ((gi*4 + q)*64 + l)*16 B, so every load
// instruction of a wave is one contiguous 1 KB run and a (chunk, head) is two linear 192 KB streams
// (row-shaped fragment loads of 64 B out of 16 different lines were texture-address bound).
// The waves stride over 32-key groups with an online softmax each; merged through LDS.
// ------------------------------------------------------------------------------------
// MINW: minimum waves per SIMD the register allocation must allow (__launch_bounds__' second argument).  1 = whatever
// the kernel wants (110 registers: 4 waves per SIMD = two workgroups per CU and nothing beside them); 5 caps it at 96
// (4 values spilled) so that two workgroups leave a fifth wave slot of up to 128 registers per SIMD to a 4-wave workgroup of
// the OTHER decode lane's linears / self-attention (knob 7, profiles/r06_ab_cross_regs.jsonl).
template <int CA_WAVES, bool NTL, int MINW = 1>
__global__ __launch_bounds__(CA_WAVES * 64, MINW) void dec_cross_attn_kernel(const half_t* __restrict__ qx, int d,
                                                             const half_t* __restrict__ ck,
                                                             const half_t* __restrict__ cvt, int T, int kvp,
                                                             int kmul, half_t* __restrict__ out,
                                                             const int* __restrict__ done, int kv_div, int frag,
                                                             const int* __restrict__ slot_map) {
  __shared__ float sm[CA_WAVES][16], sl[CA_WAVES][16];
  __shared__ float so[CA_WAVES][16][65];
  const int h = blockIdx.x, c = blockIdx.y;
  if (done && done[c]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  // encoder chunk whose K / V^T this decode chunk attends to, and the slot of the cross-attention pool that holds it
  // (the pool is shared by the decode lanes of the device: a run's chunks are wherever their encoder outputs were given
  //  a block, decoder.hip: CrossPool)
  const int ce = slot_map[c / kv_div];
  const half_t* kbase = ck + ((size_t)ce * (d >> 6) + h) * kvp * 64 + lane * 8;   // [chunk][head][group][run][lane][8]
  const half_t* vbase = cvt + ((size_t)ce * (d >> 6) + h) * kvp * 64 + lane * 8;
  half8_t qf[2];
  {
    if (j < kmul) {
      const half_t* qp = qx + (size_t)(c * kmul + j) * d + h * 64 + g * 8;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        half8_t v = *reinterpret_cast<const half8_t*>(qp + s * 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * (half_t)0.125f;
        qf[s] = v;
      }
    } else {
      qf[0] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      qf[1] = qf[0];
    }
  }
  floatx4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = floatx4{0, 0, 0, 0};
  float m_run = -1.0e30f, l_run = 0.f;
  const float LOG2E = 1.4426950408889634f;
  const int ngroups = (T + 31) >> 5;
  // K / V^T fragments of the NEXT key group are requested before the current group is processed
  // (8 x 16 B per lane always in flight per wave; 8 waves per workgroup): the kernel is a pure HBM stream.
  auto load_kv = [&](int gi, half8_t (&kf)[4], half8_t (&vf)[4]) {
    const half_t* kp = kbase + (size_t)gi * 2048;
    const half_t* vp = vbase + (size_t)gi * 2048;
    // K runs: q = 2*sub + s (sub: keys +0 / +4 of the interleaved A rows, s: dims 0-31 / 32-63)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      kf[q] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(kp + q * 512))
                  : *reinterpret_cast<const half8_t*>(kp + q * 512);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      vf[dt] = NTL ? __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(vp + dt * 512))
                   : *reinterpret_cast<const half8_t*>(vp + dt * 512);
  };
  half8_t kcur[4], vcur[4], knxt[4], vnxt[4];
  if (wave < ngroups) load_kv(wave, kcur, vcur);
  for (int gi = wave; gi < ngroups; gi += CA_WAVES) {
    const int base = gi * 32;
    const bool has_next = (gi + CA_WAVES) < ngroups;
    if (has_next) load_kv(gi + CA_WAVES, knxt, vnxt);
    floatx4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kcur[0], qf[0], s0, 0, 0, 0);
    s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kcur[1], qf[1], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kcur[2], qf[0], s1, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kcur[3], qf[1], s1, 0, 0, 0);
    // lane (query j, g) now holds keys base + 8g + {0..3} (s0) and + {4..7} (s1)
    float sc[8];
    float mx = -1.0e30f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = (e < 4 ? s0[e] : s1[e - 4]) * LOG2E;
      if (base + 8 * g + e >= T) v = -1.0e30f;
      sc[e] = v;
      mx = fmaxf(mx, v);
    }
    mx = pair16_max(mx);
    mx = pair32_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    m_run = m_new;
    half8_t pf;
    float psum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pv = __builtin_amdgcn_exp2f(sc[e] - m_new);
      psum += pv;
      pf[e] = (half_t)pv;
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[dt][e] *= alpha;
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vcur[dt], pf, o[dt], 0, 0, 0);
    }
    if (has_next) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { kcur[q] = knxt[q]; vcur[q] = vnxt[q]; }
    }
  }
  // combine the 4 key-group lanes of a query, then the 4 waves
  l_run = pair16_sum(l_run);
  l_run = pair32_sum(l_run);
  if (g == 0) { sm[wave][j] = m_run; sl[wave][j] = l_run; }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 4; ++e) so[wave][j][dt * 16 + 4 * g + e] = o[dt][e];
  __syncthreads();
  for (int idx = tid; idx < kmul * 64; idx += CA_WAVES * 64) {
    const int qq = idx >> 6, dh = idx & 63;
    float M = sm[0][qq];
#pragma unroll
    for (int w = 1; w < CA_WAVES; ++w) M = fmaxf(M, sm[w][qq]);
    float Lsum = 0.f, O = 0.f;
#pragma unroll
    for (int w = 0; w < CA_WAVES; ++w) {
      const float f = __builtin_amdgcn_exp2f(sm[w][qq] - M);
      Lsum += sl[w][qq] * f;
      O += so[w][qq][dh] * f;
    }
    const int orow = c * kmul + qq;
    half_t* dst = frag ? out + frag_off(orow, h * 64 + dh, d >> 5) : out + (size_t)orow * d + h * 64 + dh;
    *dst = (half_t)(O / Lsum);
  }
}

// ------------------------------------------------------------------------------------
// K22: no-speech probability = softmax(logits at the <sot> position)[no_speech]
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dec_nospeech_kernel(const float* __restrict__ logits, int V, int row_mul,
                                                            int no_speech_id, float* __restrict__ out) {
  __shared__ float red[32];
  const int b = blockIdx.x;
  const float* lg = logits + (size_t)b * row_mul * V;
  float mx = -3.0e38f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, lg[v]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  float m2 = -3.0e38f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m2 = fmaxf(m2, red[i]);
  __syncthreads();
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += __expf(lg[v] - m2);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
    out[b] = __expf(lg[no_speech_id] - m2) / tot;
  }
}

// ------------------------------------------------------------------------------------
// K17/K18 (+ the row part of K19): logits rules, fp32 log-softmax and the row's top-C
// candidates of cum + logp, one workgroup per row.
// Rule order: repetition penalty, no-repeat-ngram, suppress-blank (first
// step), suppress list, [min_new_tokens], timestamp rules (a)-(e), log-softmax.
// ------------------------------------------------------------------------------------
#define LP_THREADS 1024
#define LP_WAVES (LP_THREADS / 64)
#define LP_NV 56 /* values per thread kept in registers: V <= 56*1024 */
// counter-based Gumbel noise: murmur3 finaliser of (seed, row, step, token) -> u strictly in (0,1) -> -log(-log u).
// oracle/whisper.py::_gumbel restates the same integer hash.
static __device__ __forceinline__ float gumbel_noise(unsigned seed_lo, unsigned seed_hi, unsigned row, unsigned step,
                                                     unsigned v) {
  unsigned h = seed_lo ^ (row * 0x9E3779B9u) ^ (step * 0x85EBCA6Bu) ^ (v * 0xC2B2AE35u) ^ (seed_hi * 0x27D4EB2Fu);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  // 23 bits: (h >> 9) + 0.5 is exact in fp32 for every h, so u stays strictly inside (0, 1) (with 24 bits the
  // top value rounds to 1.0 and the noise becomes +inf: that token would win whatever its probability)
  const float u = ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
  return -logf(-logf(u));
}

// The row lives in registers (56 values per thread): all of its loads are issued back to back before the first use,
// the suppress list is one bit per token (one scalar 8-byte load per 64 tokens of a wave), every element-wise rule
// is a range test on uniform scalars.  Top-C: each thread keeps its own best (key, slot); a round is one wave
// arg-max + 16 partials through LDS, and only the winning thread rescans its 56 values.
// TXI: the first TXI values of every thread are text ids whatever the thread (ts_begin >= TXI * 1024: 48 for every
// Whisper vocabulary, 0 for the synthetic test vocabularies) — a compile-time class split, no per-lane test there.
template <bool SMP, int TXI>
__global__ __launch_bounds__(LP_THREADS) void dec_logits_process_kernel(fwd::GenDev gp, float* __restrict__ logits,
                                                                        const unsigned long long* __restrict__ sup_bits,
                                                                        const int* __restrict__ hist2,
                                                                        const float* __restrict__ cum2,
                                                                        const int* __restrict__ d_step,
                                                                        const int* __restrict__ done,
                                                                        float* __restrict__ cand_val,
                                                                        int* __restrict__ cand_tok) {
  __shared__ float red_mt[LP_WAVES], red_ms[LP_WAVES], red_st[LP_WAVES], red_ss[LP_WAVES];
  __shared__ float bkey_s[2][LP_WAVES];
  __shared__ int btok_s[2][LP_WAVES];
  __shared__ int sh_last_ts;
  const int r = blockIdx.x;
  const int c = r / gp.K;
  if (done[c]) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int step = *d_step;
  const int cur = step & 1;
  const int n = step;  // tokens generated so far on this row
  const int* hist = hist2 + ((size_t)cur * gp.R + r) * gp.n_text_ctx;
  float* lg = logits + (size_t)r * gp.V;
  const int V = gp.V, tb = gp.ts_begin;
  const float NEG = -INFINITY;
  const int NONE = 0x7fffffff;

  // ---- sparse rules that touch a few ids (in HBM, before the row is pulled into registers) ----
  if (gp.rep_pen != 1.0f && n > 0) {
    for (int i = tid; i < n; i += LP_THREADS) {
      const int t = hist[i];
      bool first = true;
      for (int q = 0; q < i; ++q)
        if (hist[q] == t) { first = false; break; }
      if (first) {
        const float v = lg[t];
        lg[t] = v < 0.f ? v * gp.rep_pen : v / gp.rep_pen;
      }
    }
    __syncthreads();
  }
  if (gp.ngram > 0 && n + 1 >= gp.ngram) {
    const int k = gp.ngram;
    for (int s = tid; s + k <= n; s += LP_THREADS) {
      bool match = true;
      for (int q = 0; q < k - 1; ++q)
        if (hist[s + q] != hist[n - (k - 1) + q]) { match = false; break; }
      if (match) lg[hist[s + k - 1]] = NEG;
    }
    __syncthreads();
  }
  // ---- the row: every load in flight at once ----
  float val[LP_NV];
#pragma unroll
  for (int i = 0; i < LP_NV; ++i) {
    const int v = tid + i * LP_THREADS;
    // (unconditional, clamped address: a load under `if (v < V)` becomes a branch around the load and the wait-count
    //  pass then drains vmcnt(0) after every second one — 28 dependent round trips, 71 us per row, instead of one)
    const float x = lg[v < V ? v : V - 1];
    val[i] = (v < V) ? x : NEG;
  }
  // ---- timestamp-rule scalars (uniform): the position of the last timestamp token by a block arg-max ----
  int a_hi = 0;        // ids in [0, a_hi) are forbidden
  int b_hi = tb;       // timestamps in [tb, b_hi) are forbidden
  int c_lo = NONE;     // ids >= c_lo are forbidden
  if (gp.with_ts) {
    if (tid == 0) sh_last_ts = -1;
    __syncthreads();
    if (tid < n && hist[tid] >= tb) atomicMax(&sh_last_ts, tid);   // n <= n_text_ctx <= LP_THREADS
    __syncthreads();
    const int lp = sh_last_ts;
    const bool last_ts = n >= 1 && lp == n - 1;
    const bool penult_ts = n < 2 || hist[n - 2] >= tb;
    if (lp >= 0) {
      const int last_seen = hist[lp];
      b_hi = (last_ts && !penult_ts) ? last_seen : last_seen + 1;
    }
    if (last_ts) {
      if (penult_ts) b_hi = NONE;   // a closed pair: no timestamp at all
      else a_hi = gp.eot;           // an open one: text is forbidden, the pair has to close
    }
    if (n == 0) {
      a_hi = tb;                    // the first token is a timestamp
      if (gp.mits >= 0) c_lo = tb + gp.mits + 1;
    }
  }
  const int kill_a = (n < gp.min_new) ? gp.eot : -1;
  const int kill_b = gp.with_ts ? gp.no_ts : -1;
  // ---- element-wise masks, class maxima (text / timestamp ids) ----
  // The 64 tokens a wave holds in val[i] are CONSECUTIVE ids (v0 = 64 * (16 i + wave) .. v0 + 63), so every rule — the
  // suppress list, the id ranges of the timestamp rules, the single ids — is one 64-bit lane mask per i.  Lane l of
  // the wave builds the mask of i = l (56 lanes, once); applying mask i is then two v_readlane + ONE v_cndmask (the
  // mask as the select operand) per value instead of ~12 vector compares / selects per value.  The class split (text
  // ids below ts_begin, timestamps from there) is uniform for every i but tb / 1024, where it is a per-lane test.
  auto below = [](int nb) -> unsigned long long {   // lanes [0, nb) of a wave's 64 ids
    return nb <= 0 ? 0ull : (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull));
  };
  auto span = [&](int lo, int hi, int v0) -> unsigned long long {   // ids in [lo, hi) among v0 .. v0 + 63
    return below(hi - v0) & ~below(lo - v0);
  };
  unsigned km_lo, km_hi;
  {
    const int li = lane < LP_NV ? lane : LP_NV - 1;
    const int v0 = (li * LP_WAVES + wv) * 64;
    unsigned long long km = sup_bits[li * LP_WAVES + wv];
    km |= below(a_hi - v0) | span(tb, b_hi, v0) | ~below(c_lo - v0) | span(kill_a, kill_a + 1, v0) |
          span(kill_b, kill_b + 1, v0);
    if (n == 0 && gp.suppress_blank) {   // first step only
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q < gp.n_sup_begin) km |= span(gp.sup_begin[q], gp.sup_begin[q] + 1, v0);
    }
    km_lo = (unsigned)km; km_hi = (unsigned)(km >> 32);
  }
  float mt = NEG, ms = NEG;
#pragma unroll
  for (int i = 0; i < LP_NV; ++i) {
    const unsigned long long km = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)km_hi, i) << 32) |
                                  (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)km_lo, i);
    const float x = __builtin_amdgcn_inverse_ballot_w64(km) ? NEG : val[i];
    val[i] = x;
    if (i < TXI) mt = fmaxf(mt, x);
    else {
      const bool text = tid + i * LP_THREADS < tb;
      mt = fmaxf(mt, text ? x : NEG);
      ms = fmaxf(ms, text ? NEG : x);
    }
  }
  mt = wave_max_v(mt);
  ms = wave_max_v(ms);
  if (lane == 0) { red_mt[wv] = mt; red_ms[wv] = ms; }
  __syncthreads();
  float max_t = red_mt[0], max_s = red_ms[0];
#pragma unroll
  for (int i = 1; i < LP_WAVES; ++i) { max_t = fmaxf(max_t, red_mt[i]); max_s = fmaxf(max_s, red_ms[i]); }
  // ---- class sums of exp(x - class max): exp(-inf) = 0 drops the masked ids ----
  const float off_t = (max_t == NEG) ? 0.f : max_t, off_s = (max_s == NEG) ? 0.f : max_s;
  float st = 0.f, ss = 0.f;
#pragma unroll
  for (int i = 0; i < LP_NV; ++i) {
    if (i < TXI) st += __expf(val[i] - off_t);
    else {
      const bool text = tid + i * LP_THREADS < tb;
      const float e = __expf(val[i] - (text ? off_t : off_s));
      st += text ? e : 0.f;     // (x + 0 == x: the bits of `if (text) st += e; else ss += e;`)
      ss += text ? 0.f : e;
    }
  }
  st = wave_sum_v(st);   // the butterfly of __shfl_xor 32 .. 1 on the VALU: same partners, same bits (common.h)
  ss = wave_sum_v(ss);
  if (lane == 0) { red_st[wv] = st; red_ss[wv] = ss; }
  __syncthreads();
  float sum_t = 0.f, sum_s = 0.f;
#pragma unroll
  for (int i = 0; i < LP_WAVES; ++i) { sum_t += red_st[i]; sum_s += red_ss[i]; }   // fixed order: every thread agrees
  const float lse_t = (max_t == NEG) ? NEG : max_t + logf(sum_t);
  const float lse_s = (max_s == NEG) ? NEG : max_s + logf(sum_s);
  float lse;
  if (lse_t == NEG) lse = lse_s;
  else if (lse_s == NEG) lse = lse_t;
  else {
    const float hi = fmaxf(lse_t, lse_s), lo = fminf(lse_t, lse_s);
    lse = hi + log1pf(expf(lo - hi));
  }
  // rule (e): if logsumexp(timestamps) > max(text) (in log-prob space the common lse cancels): text is masked
  const bool mask_text = __builtin_amdgcn_readfirstlane((gp.with_ts && lse_s > max_t) ? 1 : 0) != 0;   // uniform
  if (mask_text) lse = lse_s;
  // (-inf - finite = -inf: the masked ids need no test; lse is -inf only when every id is masked)
  const float lse_sub = (lse == NEG) ? 0.f : lse;
  // ---- log-probs; this thread's best key ----
  const float cum = cum2[(size_t)cur * gp.R + r];
  constexpr bool smp = SMP;   // a separate instantiation: 2 x 56 inlined logf stay out of the beam / greedy kernel
  const int C = smp ? 1 : 2 * gp.K;
  float bkey = NEG;
  int bq = -1;
#pragma unroll
  for (int i = 0; i < LP_NV; ++i) {
    const int v = tid + i * LP_THREADS;
    float x = val[i];
    if (mask_text) x = (v < tb) ? NEG : x;
    x = x - lse_sub;
    val[i] = x;
    float key = x;
    if (smp && x != NEG) key = x * gp.inv_temp + gumbel_noise(gp.seed_lo, gp.seed_hi, (unsigned)r, (unsigned)step, (unsigned)v);
    if (key > bkey) { bkey = key; bq = i; }   // ascending index: ties keep the lowest
  }
  // ---- top-C of cum + logp: C rounds (value desc, index asc).  Sampling mode (Gumbel-max): one round on
  //      logp/T + Gumbel noise; the recorded score stays cum + logp. ----
  for (int cidx = 0; cidx < C; ++cidx) {
    float k = bkey;
    int t = (bq < 0) ? NONE : tid + bq * LP_THREADS;
    wave_argmax_v(k, t);        // DPP / permlane exchanges (the ds_bpermute butterfly was ~1 500 cycles per round)
    const int par = cidx & 1;   // partials double-buffered: one barrier per round
    if (lane == 0) { bkey_s[par][wv] = k; btok_s[par][wv] = t; }
    __syncthreads();
    // the 16 wave partials: lane l takes partial l % 16, four exchanges inside the 16-lane row leave the winner in
    // every lane (the relation is a total order on (key, token): the result does not depend on the exchange pattern)
    float wk = bkey_s[par][lane & (LP_WAVES - 1)];
    int wt = btok_s[par][lane & (LP_WAVES - 1)];
    argmax_pair_step(wk, wt, dpp_f<FW_DPP_XOR1>(wk), dpp_i<FW_DPP_XOR1>(wt));
    argmax_pair_step(wk, wt, dpp_f<FW_DPP_XOR2>(wk), dpp_i<FW_DPP_XOR2>(wt));
    argmax_pair_step(wk, wt, dpp_f<FW_DPP_ROR4>(wk), dpp_i<FW_DPP_ROR4>(wt));
    argmax_pair_step(wk, wt, dpp_f<FW_DPP_ROR8>(wk), dpp_i<FW_DPP_ROR8>(wt));
    if (wt == NONE) {   // nothing left on this row
      if (tid == 0) {
        cand_val[(size_t)r * 32 + cidx] = NEG;
        cand_tok[(size_t)r * 32 + cidx] = 0;
        if (smp) { cand_val[(size_t)r * 32 + 1] = NEG; cand_tok[(size_t)r * 32 + 1] = 0; }
      }
      continue;
    }
    if ((wt & (LP_THREADS - 1)) == tid) {   // the winning thread: record, retire the value, find its next best
      const int wq = wt / LP_THREADS;
      float raw = NEG;
      bkey = NEG; bq = -1;
#pragma unroll
      for (int i = 0; i < LP_NV; ++i) {
        if (i == wq) { raw = val[i]; val[i] = NEG; }
        if (!smp && val[i] > bkey) { bkey = val[i]; bq = i; }
      }
      cand_val[(size_t)r * 32 + cidx] = cum + raw;
      cand_tok[(size_t)r * 32 + cidx] = wt;
      if (smp) { cand_val[(size_t)r * 32 + 1] = NEG; cand_tok[(size_t)r * 32 + 1] = 0; }
    }
  }
}

// ------------------------------------------------------------------------------------
// K19/K20: per-chunk beam update [CT2-ext: BeamSearch::search].  Merges the K rows' top-2K
// candidates (stable: value desc, flat index asc), walks the first K slots (EOS / last step ->
// finished hypothesis, replaced by the next non-EOS secondary candidate), then rewrites
// the per-row state (token history, KV-slot table, cum, next input token).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dec_beam_update_kernel(fwd::GenDev gp, const float* __restrict__ cand_val,
                                                             const int* __restrict__ cand_tok, int* __restrict__ hist2,
                                                             float* __restrict__ cum2, uint8_t* __restrict__ kvidx2,
                                                             int* __restrict__ cur_tok, const int* __restrict__ d_step,
                                                             int* __restrict__ done, int* __restrict__ n_done,
                                                             int* __restrict__ n_fin, int* __restrict__ fin_tok,
                                                             int* __restrict__ fin_len, float* __restrict__ fin_score,
                                                             float* __restrict__ fin_cum) {
  __shared__ float ov[32];
  __shared__ int ok[32], ot[32];
  __shared__ int s_parent[16], s_tok[16];
  __shared__ float s_cum[16];
  __shared__ int s_done;
  const int c = blockIdx.x;
  if (done[c]) return;
  const int tid = threadIdx.x;
  const int K = gp.K, C = 2 * K, NT = gp.n_text_ctx;
  const int step = *d_step;
  const int cur = step & 1, nxt = cur ^ 1;
  const int pos = gp.P - 1 + step;
  const bool last_step = (step + 1) >= gp.budget;
  if (tid == 0) {
    // ---- merge: top-C of the (live rows x C) candidates ----
    const int nsrc = (step == 0) ? 1 : K;
    int ptr[16];
    for (int k = 0; k < nsrc; ++k) ptr[k] = 0;
    int nsel = 0;
    for (; nsel < C; ++nsel) {
      int bk = -1;
      float bvv = -INFINITY;
      int btok = 0;
      for (int k = 0; k < nsrc; ++k) {
        if (ptr[k] >= C) continue;
        const float v = cand_val[(size_t)(c * K + k) * 32 + ptr[k]];
        const int t = cand_tok[(size_t)(c * K + k) * 32 + ptr[k]];
        if (v == -INFINITY) continue;
        if (bk < 0 || v > bvv) { bk = k; bvv = v; btok = t; }  // ascending k: ties keep lowest flat index
      }
      if (bk < 0) break;
      ov[nsel] = bvv; ok[nsel] = bk; ot[nsel] = btok;
      ptr[bk]++;
    }
    // ---- walk ----
    int nf = n_fin[c];
    int sec = K, nlive = 0;
    for (int slot = 0; slot < K; ++slot) {
      int jdx = slot;
      if (jdx >= nsel) break;
      if (ot[jdx] == gp.eot || last_step) {
        if (nf < fwd::FIN_CAP) {
          const int kk = ok[jdx];
          const int* hsrc = hist2 + ((size_t)cur * gp.R + c * K + kk) * NT;
          int* dst = fin_tok + ((size_t)c * fwd::FIN_CAP + nf) * NT;
          int len = step;
          for (int q = 0; q < step; ++q) dst[q] = hsrc[q];
          if (ot[jdx] != gp.eot) { dst[len] = ot[jdx]; len++; }
          fin_len[c * fwd::FIN_CAP + nf] = len;
          fin_cum[c * fwd::FIN_CAP + nf] = ov[jdx];
          const float denom = gp.lp_pow != 0.f ? powf((float)(len > 0 ? len : 1), gp.lp_pow) : 1.f;
          fin_score[c * fwd::FIN_CAP + nf] = ov[jdx] / denom;
          nf++;
        }
        if (last_step) continue;
        while (sec < nsel && ot[sec] == gp.eot) sec++;
        jdx = sec++;
        if (jdx >= nsel) continue;
      }
      s_parent[nlive] = ok[jdx]; s_tok[nlive] = ot[jdx]; s_cum[nlive] = ov[jdx];
      nlive++;
    }
    n_fin[c] = nf;
    const bool fin = last_step || nf >= gp.max_fin || nlive == 0;
    if (fin) {
      done[c] = 1;
      atomicAdd(n_done, 1);
    } else {
      for (int k = nlive; k < K; ++k) { s_parent[k] = s_parent[0]; s_tok[k] = s_tok[0]; s_cum[k] = -INFINITY; }
    }
    s_done = fin ? 1 : 0;
  }
  __syncthreads();
  if (s_done) return;
  // ---- rewrite row state for the next step ----
  for (int k = 0; k < K; ++k) {
    const int pr = c * K + s_parent[k], nr = c * K + k;
    const int* hs = hist2 + ((size_t)cur * gp.R + pr) * NT;
    int* hd = hist2 + ((size_t)nxt * gp.R + nr) * NT;
    const uint8_t* is = kvidx2 + ((size_t)cur * gp.R + pr) * NT;
    uint8_t* id = kvidx2 + ((size_t)nxt * gp.R + nr) * NT;
    for (int q = tid; q < step; q += 64) hd[q] = hs[q];
    for (int q = tid; q < pos; q += 64) id[q] = is[q];
    if (tid == 0) {
      hd[step] = s_tok[k];
      id[pos] = (uint8_t)s_parent[k];
      cum2[(size_t)nxt * gp.R + nr] = s_cum[k];
      cur_tok[nr] = s_tok[k];
    }
  }
}

__global__ void dec_step_advance_kernel(int* d_step) { *d_step += 1; }

// per-token probability for align: p[r] = softmax(logits[r])[target[r]]
__global__ __launch_bounds__(1024) void dec_token_prob_kernel(const float* __restrict__ logits, int V,
                                                              const int* __restrict__ target, float* __restrict__ out,
                                                              int out_stride, int out_off, int row_mul) {
  __shared__ float red[32];
  const int b = blockIdx.x;
  const float* lg = logits + (size_t)b * row_mul * V;   // (position blocks: chunk b's row of this position)
  float mx = -3.0e38f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) mx = fmaxf(mx, lg[v]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  float m2 = -3.0e38f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) m2 = fmaxf(m2, red[i]);
  __syncthreads();
  float s = 0.f;
  for (int v = threadIdx.x; v < V; v += blockDim.x) s += __expf(lg[v] - m2);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) tot += red[i];
    const int t = target[b];
    out[(size_t)b * out_stride + out_off] = (t >= 0 && t < V) ? __expf(lg[t] - m2) / tot : 0.f;
  }
}

// cross-attention probabilities of selected heads for align:
// probs[b][hsel][tok][t] = softmax_t( q[b][head] . K[b][t][head] / 8 ),  one workgroup per (hsel, b)
__global__ __launch_bounds__(256) void dec_cross_probs_kernel(const half_t* __restrict__ qx, int d,
                                                              const half_t* __restrict__ ck, int T, int kvp,
                                                              const int* __restrict__ heads, int n_sel,
                                                              float* __restrict__ probs, int n_tok, int tok_idx,
                                                              int blk_n) {
  __shared__ float sq[64];
  __shared__ float red[4];
  // blk_n > 1: query row y = chunk * blk_n + j is token tok_idx + j of its chunk (position blocks)
  const int hs = blockIdx.x, row = blockIdx.y, b = row / blk_n;
  tok_idx += row - b * blk_n;
  const int h = heads[hs];
  const int tid = threadIdx.x;
  if (tid < 64) sq[tid] = (float)qx[(size_t)row * d + h * 64 + tid] * 0.125f;
  __syncthreads();
  float* pr = probs + (((size_t)b * n_sel + hs) * n_tok + tok_idx) * T;
  float mx = -3.0e38f;
  for (int t = tid; t < T; t += 256) {
    // fragment-major K (see K14): dims 8j..8j+7 of key t live in run 2*sub + (j>>2), lane 16*(j&3) + jj
    const int r = t & 31;
    const half_t* kb = ck + ((size_t)b * (d >> 6) + h) * kvp * 64 +
                       ((size_t)((t >> 5) * 4 + 2 * ((r >> 2) & 1)) * 64 + (((r >> 3) << 2) | (r & 3))) * 8;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const half8_t kv = *reinterpret_cast<const half8_t*>(kb + ((size_t)(j >> 2) * 64 + (j & 3) * 16) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += sq[j * 8 + e] * (float)kv[e];
    }
    pr[t] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int t = tid; t < T; t += 256) {
    const float e = __expf(pr[t] - mx);
    pr[t] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
  for (int t = tid; t < T; t += 256) pr[t] *= inv;
}

namespace fwd {

// row count from which the decoder linears of a decode run take the LDS-staged GEMM-shaped kernel
// (profiles/r03_dec_linear_bench.txt)
// (per linear: DEC_BIG_RULES below)

void launch_embed(hipStream_t st, const int* tok, const half_t* emb, const half_t* pos_emb, half_t* x, half_t* xfrag,
                  int rows, int d, const int* d_step, int pos_fixed, int P, int blk_n) {
  dec_embed_kernel<<<rows, 128, 0, st>>>(tok, emb, pos_emb, x, xfrag, d, d_step, pos_fixed, P, blk_n);
}

template <bool LNF, int RT, int NT, int CH = 0>
static void frag_go(hipStream_t st, int waves, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1,
                    const float* cf, const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R, int N,
                    int K, int act) {
  const dim3 grid((N / 16 + NT - 1) / NT, ((R + 15) / 16 + RT - 1) / RT);
  if (waves == 8)
    dec_gemm_frag_kernel<8, LNF, RT, NT, CH><<<grid, 512, 0, st>>>(xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R,
                                                                   N, K, act);
  else
    dec_gemm_frag_kernel<4, LNF, RT, NT, CH><<<grid, 256, 0, st>>>(xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R,
                                                                   N, K, act);
}

// GEMM-shaped kernel of merged runs (dec_gemm_big_kernel), workgroup shape `cfg`; -1 when the shape does not fit.
//   cfg 0: 4 x 2 waves, 256 rows x 128 columns, 2 k-steps per stage, 3 stages (144 KB)  — the wide linears (qkv, ffn1)
//   cfg 1: 2 x 2 waves, 128 rows x 64 columns (2 column tiles per wave), 1 k-step per stage, 5 stages (60 KB, two
//          workgroups per CU) — the linears with 1280 columns (out / cross-q / cross-out, ffn2), which a wider tile
//          cannot spread over the chip
//   cfg 2: 2 x 2 waves, 128 x 128, 1 k-step per stage, 4 stages (64 KB)
//   (round 6 measured and removed: cfg 3 / 4, cfg 0's tile with two staggered wave groups (10 % slower:
//    profiles/r06_dec_linear_bench_call2_stagger.txt); cfg 5 / 6, cfg 1's tile with two k-steps per stage and a ring of 3 / 4 — cfg 6
//    was 4 % FASTER for ffn2 in the isolated table (_call15_kc2.txt) and 18 % SLOWER in the pipeline (96 KB of LDS: one workgroup
//    per CU and nothing of the other decode lane beside it; profiles/r06_ab_ffn2_cfg.jsonl, two builds alternating on one box:
//    ffn2 8.8-9.0 vs 10.3-10.9 ms per step) — the isolated table ranks forms of EQUAL LDS footprint only; cfg 7, cfg 2's tile
//    with two k-steps per stage (_call16_cfg7.txt))
// (measured next to 256 x 64, 128 x 256, 8 waves on 128 x 128, deeper rings, two k-steps per barrier, LDS reads
//  software-pipelined under the MFMAs, L2 touch-ahead: profiles/r03_dec_linear_bench.txt — none better)
template <bool LNF, int S, int WM, int WN, int FB, int KC, int NST>
static int big_go(hipStream_t st, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1, const float* cf,
                   const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R, int N, int K, int act) {
  constexpr int lds = (4 * WM + FB * WN) * KC * 1024 * NST;
  // the dynamic-LDS limit is a per-device attribute of the function; decode lanes launch from several host threads:
  // one bit per device, set after the attribute call returned (two threads may both make the call: it is idempotent)
  static std::atomic<unsigned long long> attr_done{0};   // (per instantiation)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm_big_kernel<LNF, S, WM, WN, FB, KC, NST>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return -1;
    attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
  }
  const int nMt = ((R + 15) / 16 + 4 * WM - 1) / (4 * WM), nNt = (N / 16 + FB * WN - 1) / (FB * WN);
  dec_gemm_big_kernel<LNF, S, WM, WN, FB, KC, NST><<<nMt * nNt, WM * WN * 64, lds, st>>>(xf, Wf, bias, s1, cf, res, ldr, out,
                                                                                           ldo, out_frag, R, N, K, act, nNt);
  return hipPeekAtLastError() == hipSuccess ? 0 : -1;
}
template <int WM, int WN, int FB, int KC, int NST>
static int big_cfg(hipStream_t st, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1, const float* cf,
                   const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R, int N, int K, int act) {
  if (K % 32 != 0 || N % 32 != 0 || R < 1) return -1;
  const int S = K >= 2560 ? 8 : 4;   // the slice count of launch_dec_gemm_skinny's instantiation for this K
  const int KS = K / 32;
  if (KS % S != 0 || (KS / S) % KC != 0 || KS / KC < NST) return -1;
  if ((ldo % 8) || (res && (ldr % 8))) return -1;   // 16-byte row segments
#define DGB(LNF_, S_) big_go<LNF_, S_, WM, WN, FB, KC, NST>(st, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act)
  if (s1) return S == 8 ? DGB(true, 8) : DGB(true, 4);
  return S == 8 ? DGB(false, 8) : DGB(false, 4);
#undef DGB
}
int launch_dec_gemm_big(hipStream_t st, int cfg, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1,
                        const float* cf, const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R,
                        int N, int K, int act) {
  switch (cfg) {
    case 0: return big_cfg<4, 2, 4, 2, 3>(st, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
    case 1: return big_cfg<2, 2, 2, 1, 5>(st, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
    case 2: return big_cfg<2, 2, 4, 1, 4>(st, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
    default: return -1;
  }
}

// tile-shape experiments of profiles/dec_linear_bench.py (fw_bench_dec_linear): variant -> instantiation
template <int WAVES, int RT, int NT, int CH>
static void frag_variant(hipStream_t st, bool lnf, const half_t* xf, const half_t* Wf, const half_t* bias,
                         const float* s1, const float* cf, half_t* out, int R, int N, int K) {
  const dim3 grid(N / 16 / NT, ((R + 15) / 16 + RT - 1) / RT);
  if (lnf)
    dec_gemm_frag_kernel<WAVES, true, RT, NT, CH><<<grid, WAVES * 64, 0, st>>>(xf, Wf, bias, s1, cf, nullptr, 0, out, N,
                                                                              nullptr, R, N, K, 0);
  else
    dec_gemm_frag_kernel<WAVES, false, RT, NT, CH><<<grid, WAVES * 64, 0, st>>>(xf, Wf, bias, nullptr, nullptr, nullptr,
                                                                               0, out, N, nullptr, R, N, K, 0);
}
int launch_dec_gemm_frag_variant(hipStream_t st, int variant, bool lnf, const half_t* xf, const half_t* Wf,
                                 const half_t* bias, const float* s1, const float* cf, half_t* out, int R, int N,
                                 int K) {
  if (K % 32 != 0 || N % 64 != 0 || R < 1) return -1;
  switch (variant) {
    case 0: frag_variant<4, 2, 2, 5>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;   // product, K < 2560
    case 1: frag_variant<8, 2, 2, 5>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;   // product, K >= 2560
    case 2: frag_variant<8, 4, 2, 3>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 3: frag_variant<8, 4, 2, 4>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 4: frag_variant<8, 4, 4, 2>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 5: frag_variant<8, 4, 4, 3>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 6: frag_variant<4, 4, 2, 3>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 7: frag_variant<8, 2, 4, 3>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 8: frag_variant<8, 8, 2, 2>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 9: frag_variant<4, 4, 4, 2>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 23: frag_variant<4, 1, 1, 10>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 24: frag_variant<8, 1, 1, 10>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 25: frag_variant<4, 1, 2, 6>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 26: frag_variant<8, 1, 2, 6>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 21: frag_variant<4, 4, 4, 5>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 22: frag_variant<8, 4, 4, 5>(st, lnf, xf, Wf, bias, s1, cf, out, R, N, K); break;
    case 10: case 11: case 12:
      return launch_dec_gemm_big(st, variant - 10, xf, Wf, lnf ? nullptr : bias, lnf ? s1 : nullptr, lnf ? cf : nullptr,
                                 nullptr, 0, out, N, nullptr, R, N, K, 0);
    default: return -1;
  }
  return 0;
}

// Which linear of a decoder layer a shape is: 0 qkv (N = 3 K), 1 d x d (out / cross-q / cross-out), 2 ffn1, 3 ffn2.
static int dec_linear_role(int N, int K) { return N == 3 * K ? 0 : (N >= 2560 ? 2 : (K >= 2560 ? 3 : 1)); }
// Row counts from which each linear of a decode run takes the LDS-staged GEMM-shaped kernel, and the workgroup shape it
// takes there (launch_dec_gemm_big's cfg) — ONE table: the launcher below, fw_dec_big_min_rows_of (the benchmark prices a
// linear against the MFMA roof only from ITS row count on) and the tests read it.  Measured crossovers:
// profiles/r05_dec_linear_bench_call1.txt (round 5), profiles/r06_dec_linear_bench_*.txt (round 6: staggered 256 x 128).
struct DecBigRule { int rows; int cfg; };
static const DecBigRule DEC_BIG_RULES[4][2] = {
    // (round 6, after the LayerNorm-statistics split, profiles/r06_dec_linear_bench_call11_crossovers.txt, us per launch
    //  register-streaming vs LDS-staged: qkv 576 / 640 / 768 rows 20.9 / 23.0 / 26.6 vs 22.1 / 21.9 / 22.4 (128 x 128); ffn1 448 /
    //  512 / 768 rows 21.8 / 24.3 / 33.3 vs 22.0 / 22.2 / 23.6 (128 x 128; 256 x 128 is the same from ~900 rows); ffn2 768 / 832
    //  rows 32.3 / 37.0 vs 34.2 / 34.0 (128 x 64); d x d 1 024 / 1 120 rows 12.3 / 13.2 vs 12.8 / 12.9 (128 x 64))
    /* qkv  */ {{640, 2}, {1280, 0}},
    /* dxd  */ {{1088, 1}, {1 << 30, 1}},
    /* ffn1 */ {{512, 2}, {1280, 0}},
    /* ffn2 */ {{832, 1}, {1 << 30, 1}},   // (NOT the two-k-steps form: 4 % faster isolated, 18 % slower in the pipeline, r06_ab_ffn2_cfg.jsonl)
};
static int dec_big_cfg_for(int R, int N, int K) {
  const DecBigRule* r = DEC_BIG_RULES[dec_linear_role(N, K)];
  if (R >= r[1].rows) return r[1].cfg;
  if (R >= r[0].rows) return r[0].cfg;
  return -1;
}
// role 0..3 as above; compute_type 0 float16, 1 int8_float16 (launch_dec_gemm_frag_i8: the 4 x 4-tile form from
// DEC_BIG_MIN_ROWS_I8 rows for every linear); role < 0: the lowest row count of any linear
#define DEC_BIG_MIN_ROWS_I8 1024
int dec_big_min_rows_of(int role, int compute_type) {
  if (compute_type == 1) return DEC_BIG_MIN_ROWS_I8;
  if (role >= 0 && role < 4) return DEC_BIG_RULES[role][0].rows;
  int lo = DEC_BIG_RULES[0][0].rows;
  for (int k = 1; k < 4; ++k) lo = DEC_BIG_RULES[k][0].rows < lo ? DEC_BIG_RULES[k][0].rows : lo;
  return lo;
}

static bool skinny_one_tile(int R, int N) { return R <= 16 || (R <= 96 && N <= 1280); }

// The per-layer decoder linears: K split over the waves of a workgroup, 2 x 2 tiles of 16 x 16 per workgroup
// (measured best of {1,2} x {1,2}: profiles/r01_sweep_dec_gemm_frag_tiles.jsonl), row groups on grid.y so any
// number of rows works (a merged decode run carries up to 128 chunks x 5 beams).  out (row-major) and out_frag
// (fragment-major, for the next GEMM) are both optional; res is row-major.
int launch_dec_gemm_frag(hipStream_t st, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1,
                         const float* cf, const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R,
                         int N, int K, int act) {
  // Merged runs of >= DEC_BIG_MIN_ROWS rows: the LDS-staged GEMM-shaped kernel, 256 x 128 tiles for the wide linears and
  // 128 x 64 for those with 1280 columns (measured per layer at 1 520 rows: 216 -> 157 us; the register-streaming kernel
  // with 4 x 4 tiles reaches 174: profiles/r03_dec_linear_bench.txt).  Same K slices, same reduction order, same pinned
  // epilogue as the register-streaming kernel: the same bits.
  // Round 5: the crossover is per linear (profiles/r05_dec_linear_bench_call1.txt, us per launch, register-streaming vs
  // LDS-staged at 800 / 960 / 1 120 rows): qkv 27.3 / 31.8 / 36.1 vs 25.0 / 25.8 / 35.3 (128 x 128 tiles; 256 x 128 from
  // 1 280 rows), ffn1 34.4 / 40.0 / 45.0 vs 34.9 / 35.3 / 37.3 (256 x 128), ffn2 33.0 / 39.1 / 45.1 vs 35.8 / 35.9 / 36.6
  // (128 x 64), d x d 10.1 / 11.9 / 13.2 vs 12.8 / 13.0 / 13.1 (128 x 64): the fixed ~12-35 us of an LDS-staged launch
  // (its K loop's latency) is reached at a different row count by each shape.  Same bits from every form.
  const int cfg = dec_big_cfg_for(R, N, K);
  if (cfg >= 0 &&
      launch_dec_gemm_big(st, cfg, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act) == 0)
    return 0;
  return launch_dec_gemm_skinny(st, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
}

int dec_big_min_rows() { return dec_big_min_rows_of(-1, 0); }

int launch_dec_gemm_skinny(hipStream_t st, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1,
                           const float* cf, const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R,
                           int N, int K, int act) {
  if (K % 32 != 0 || N % 32 != 0 || R < 1) return -1;
  const int waves = K >= 2560 ? 8 : 4;   // keeps a wave's share at <= 20 k-steps = 2 chunks of loads
  // Tile grouping by row count (the arithmetic of an output does not depend on it: same bits).  One 16 x 16 tile per
  // workgroup when there are few rows — twice to four times the workgroups streaming the weights, a wave's whole K share
  // in flight at once: single utterance (5 rows) 41.8 -> 31.4 us per layer, and at a solo batch (80 rows) for the
  // linears with 1280 columns (5.6 -> 4.4, 12.1 -> 10.0 us); 2 x 2 tiles otherwise, also for merged runs below
  // DEC_BIG_MIN_ROWS (4 x 2, 2 x 4, 4 x 4 tiles measured no better there: profiles/r03_dec_linear_bench.txt, README.md)
  const bool one_tile = skinny_one_tile(R, N);
  if (one_tile) {
    if (s1) frag_go<true, 1, 1, 10>(st, waves, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
    else frag_go<false, 1, 1, 10>(st, waves, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
    return 0;
  }
  if (s1) frag_go<true, 2, 2>(st, waves, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
  else frag_go<false, 2, 2>(st, waves, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
  return 0;
}

// the register-streaming kernel with an explicit tile grouping (tests: every grouping returns the same bits)
int launch_dec_gemm_skinny_tiles(hipStream_t st, int tiles, const half_t* xf, const half_t* Wf, const half_t* bias,
                                 const float* s1, const float* cf, const half_t* res, int ldr, half_t* out, int ldo,
                                 half_t* out_frag, int R, int N, int K, int act) {
  if (K % 32 != 0 || N % 32 != 0 || R < 1) return -1;
  const int waves = K >= 2560 ? 8 : 4;
  if (tiles == 1) {
    if (s1) frag_go<true, 1, 1, 10>(st, waves, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
    else frag_go<false, 1, 1, 10>(st, waves, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
  } else if (tiles == 2) {
    if (s1) frag_go<true, 2, 2>(st, waves, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
    else frag_go<false, 2, 2>(st, waves, xf, Wf, bias, s1, cf, res, ldr, out, ldo, out_frag, R, N, K, act);
  } else {
    return -1;
  }
  return 0;
}

// int8 form (see dec_gemm_frag_i8_kernel); xq / Wq fragment-major, out row-major fp16
int launch_dec_gemm_frag_i8(hipStream_t st, const int8_t* xq, const float* x_scale, const int8_t* Wq,
                            const float* w_scale, const half_t* bias, const half_t* res, int ldr, half_t* out, int ldo,
                            int R, int N, int K, int act) {
  if (K % 64 != 0 || N % 32 != 0 || R < 1 || !x_scale || !w_scale) return -1;
  if (R >= DEC_BIG_MIN_ROWS_I8 && N % 64 == 0) {   // (measured crossover of the int8 forms)
    // merged runs: 4 x 4 tiles per workgroup, a quarter of the operand traffic per output (the fp16 form of this grouping
    // measured 216 -> 174 us per layer at 1 520 rows); integer accumulation: the result does not depend on the grouping
    const dim3 g4(N / 64, ((R + 15) / 16 + 3) / 4);
    if (K >= 2560)
      dec_gemm_frag_i8_kernel<8, 4, 4, 3><<<g4, 512, 0, st>>>(xq, x_scale, Wq, w_scale, bias, res, ldr, out, ldo, R, N, K, act);
    else
      dec_gemm_frag_i8_kernel<4, 4, 4, 5><<<g4, 256, 0, st>>>(xq, x_scale, Wq, w_scale, bias, res, ldr, out, ldo, R, N, K, act);
    return 0;
  }
  const dim3 grid(N / 32, ((R + 15) / 16 + 1) / 2);
  if (K >= 2560)
    dec_gemm_frag_i8_kernel<8, 2, 2><<<grid, 512, 0, st>>>(xq, x_scale, Wq, w_scale, bias, res, ldr, out, ldo, R, N, K, act);
  else
    dec_gemm_frag_i8_kernel<4, 2, 2><<<grid, 256, 0, st>>>(xq, x_scale, Wq, w_scale, bias, res, ldr, out, ldo, R, N, K, act);
  return 0;
}

template <bool I8, bool LNF, int RT>
static void wave_go(hipStream_t st, const void* xf, const float* x_scale, const void* Wf, const float* w_scale,
                    const float* s1, const float* cf, float* out, int ldo, int R, int N, int K) {
  constexpr int NT = 2, WM = 1, WN = 4;
  const int n_rt = (R + 15) / 16, n_ct = (N + 15) / 16;
  const int n_cg = ((n_ct + NT * WN - 1) / (NT * WN) + 7) & ~7;   // padded to whole groups of 8 (extra waves leave)
  const int n_rg = (n_rt + RT * WM - 1) / (RT * WM);
  dec_gemm_wave_kernel<I8, LNF, true, RT, NT, WM, WN><<<n_cg * n_rg, WM * WN * 64, 0, st>>>(
      xf, x_scale, Wf, w_scale, nullptr, s1, cf, out, ldo, R, N, K, n_rg);
}

// Vocabulary projection -> float32 logits [R][ldo].  fp16: xf = the raw residual stream, fragment-major, with the
// final LayerNorm folded into Wf / s1 / cf.  int8: xf = the LayerNorm'ed rows quantised per row (x_scale),
// Wf int8 + w_scale.  Wf holds ceil(N / 16) fragment-major column tiles (the last one zero-padded).
int launch_dec_logits(hipStream_t st, bool i8, const void* xf, const float* x_scale, const void* Wf,
                      const float* w_scale, const float* s1, const float* cf, float* out, int ldo, int R, int N, int K) {
  if (K % (i8 ? 64 : 32) != 0 || R < 1) return -1;
  if (i8 ? (!x_scale || !w_scale) : ((s1 == nullptr) != (cf == nullptr))) return -1;
  const int n_rt = (R + 15) / 16;
  // fp16 with s1 == cf == null: xf already holds fp16(LayerNorm(x)) and Wf the tied embedding itself (no bias)
#define WG(RT)                                                                                   \
  do {                                                                                           \
    if (i8) wave_go<true, false, RT>(st, xf, x_scale, Wf, w_scale, s1, cf, out, ldo, R, N, K);   \
    else if (s1) wave_go<false, true, RT>(st, xf, x_scale, Wf, w_scale, s1, cf, out, ldo, R, N, K); \
    else wave_go<false, false, RT>(st, xf, x_scale, Wf, w_scale, s1, cf, out, ldo, R, N, K);     \
  } while (0)
  if (n_rt == 1) WG(1);
  else if (n_rt == 2) WG(2);
  else if (n_rt == 3) WG(3);
  else if (n_rt == 4) WG(4);
  else WG(5);   // 5 row tiles = the 80 rows of a 16-chunk beam-5 step per wave; more rows: further row groups
#undef WG
  return 0;
}

bool self_attn_block_ok(int n_ctx, int cache_ctx, int d, int R_total) {
  return (n_ctx & 3) == 0 && (size_t)R_total * cache_ctx * d * sizeof(half_t) < ((size_t)1 << 32);
}

void launch_self_attn(hipStream_t st, const half_t* qkv, int d, half_t* kc, half_t* vc, int n_ctx, int cache_ctx, int H,
                      const uint8_t* kvidx2, int Kbeam, int kmul, half_t* out, int rows, const int* d_step,
                      int pos_fixed, int P, int R_total, int frag, int blk_n) {
  if (blk_n > 0) {   // a block of blk_n consecutive positions per chunk (self_attn_block_ok holds: the caller checked)
    dec_self_attn2_kernel<1, true, 1024, true><<<dim3(H, rows / blk_n), blk_n * 64, (size_t)blk_n * 2 * n_ctx * sizeof(float), st>>>(
        qkv, d, kc, vc, n_ctx, cache_ctx, H, kvidx2, Kbeam, blk_n, out, d_step, pos_fixed, P, R_total, frag);
    return;
  }
  // one workgroup per (head, chunk), one wave per row of the chunk (kmul <= 16); LDS: scores + source slots per wave
  const dim3 grid(H, rows / kmul);
  const size_t lds = (size_t)kmul * 2 * n_ctx * sizeof(float);
  int form = g_self_attn_form.load(std::memory_order_relaxed);   // 0: by size (product); 1: first form; 2 / 3: forced
  // the second form needs n_ctx % 4 == 0 (4-byte slot-table loads) and a layer's cache below 4 GB (32-bit offsets)
  if ((n_ctx & 3) != 0 || (size_t)R_total * cache_ctx * d * sizeof(half_t) >= ((size_t)1 << 32)) form = 1;
  if (form == 0) form = (kmul <= 8 && (int)(grid.x * grid.y) <= 768) ? 2 : 3;
  if (form == 2 && kmul > 8) form = 3;
#define SA_ARGS qkv, d, kc, vc, n_ctx, cache_ctx, H, kvidx2, Kbeam, kmul, out, d_step, pos_fixed, P, R_total, frag
  if (form == 1) dec_self_attn_kernel<<<grid, kmul * 64, lds, st>>>(SA_ARGS);
  else if (form == 2) dec_self_attn2_kernel<4, false, 512><<<grid, kmul * 64, lds, st>>>(SA_ARGS);   // latency form
  else dec_self_attn2_kernel<1, true, 1024><<<grid, kmul * 64, lds, st>>>(SA_ARGS);                  // throughput form
#undef SA_ARGS
}
void set_self_attn_form(int form) { g_self_attn_form.store(form); g_forms_epoch.fetch_add(1); }
int kernel_forms_epoch() { return g_forms_epoch.load(); }
void bump_kernel_forms_epoch() { g_forms_epoch.fetch_add(1); }

void launch_cross_attn(hipStream_t st, const half_t* qx, int d, const half_t* ck, const half_t* cvt, int T, int kvp,
                       int kmul, half_t* out, int B, int H, const int* done, int kv_div, int frag, const int* slot_map) {
  // 8 waves per (chunk, head) keep 64 KB of loads in flight per workgroup (4 waves measured slower)
  // the K / V^T stream is read once per step and never again before 15 GB of other chunks have passed: non-temporal
  // loads (measured 75.3 -> 68.3 ms per batch, 5.4 -> 5.9 TB/s)
  const int cap = g_cross_regs.load(std::memory_order_relaxed);
  if (cap == 1)
    dec_cross_attn_kernel<8, true, 5><<<dim3(H, B), 512, 0, st>>>(qx, d, ck, cvt, T, kvp, kmul, out, done, kv_div, frag, slot_map);
  else if (cap == 2)
    dec_cross_attn_kernel<8, true, 6><<<dim3(H, B), 512, 0, st>>>(qx, d, ck, cvt, T, kvp, kmul, out, done, kv_div, frag, slot_map);
  else
    dec_cross_attn_kernel<8, true><<<dim3(H, B), 512, 0, st>>>(qx, d, ck, cvt, T, kvp, kmul, out, done, kv_div, frag, slot_map);
}
void set_cross_attn_regs(int cap) { g_cross_regs.store(cap); g_forms_epoch.fetch_add(1); }

void launch_nospeech(hipStream_t st, const float* logits, int V, int row_mul, int no_speech_id, float* out, int B) {
  dec_nospeech_kernel<<<B, 1024, 0, st>>>(logits, V, row_mul, no_speech_id, out);
}

void launch_logits_process(hipStream_t st, const GenDev& gp, float* logits, const unsigned long long* sup_bits,
                           const int* hist2, const float* cum2, const int* d_step, const int* done, float* cand_val,
                           int* cand_tok) {
  // every Whisper vocabulary keeps its timestamp ids above 48 * 1024 (ts_begin 50 363 .. 50 365); the synthetic
  // test vocabularies do not: they take the instantiation with a per-lane class test everywhere
  const bool wide = gp.ts_begin >= 48 * LP_THREADS;
#define LP_GO(SMP, TXI) \
  dec_logits_process_kernel<SMP, TXI><<<gp.R, LP_THREADS, 0, st>>>(gp, logits, sup_bits, hist2, cum2, d_step, done, cand_val, cand_tok)
  if (gp.sample) { if (wide) LP_GO(true, 48); else LP_GO(true, 0); }
  else { if (wide) LP_GO(false, 48); else LP_GO(false, 0); }
#undef LP_GO
}

void launch_beam_update(hipStream_t st, const GenDev& gp, const float* cand_val, const int* cand_tok, int* hist2,
                        float* cum2, uint8_t* kvidx2, int* cur_tok, const int* d_step, int* done, int* n_done,
                        int* n_fin, int* fin_tok, int* fin_len, float* fin_score, float* fin_cum) {
  dec_beam_update_kernel<<<gp.B, 64, 0, st>>>(gp, cand_val, cand_tok, hist2, cum2, kvidx2, cur_tok, d_step, done,
                                              n_done, n_fin, fin_tok, fin_len, fin_score, fin_cum);
}

void launch_step_advance(hipStream_t st, int* d_step) { dec_step_advance_kernel<<<1, 1, 0, st>>>(d_step); }

void launch_token_prob(hipStream_t st, const float* logits, int V, const int* target, float* out, int out_stride,
                       int out_off, int rows, int row_mul) {
  dec_token_prob_kernel<<<rows, 1024, 0, st>>>(logits, V, target, out, out_stride, out_off, row_mul);
}

void launch_cross_probs(hipStream_t st, const half_t* qx, int d, const half_t* ck, int T, int kvp, const int* heads,
                        int n_layer_heads, int n_sel, float* probs, int n_tok, int tok_idx, int B, int blk_n) {
  const int bn = blk_n > 0 ? blk_n : 1;
  dec_cross_probs_kernel<<<dim3(n_layer_heads, B * bn), 256, 0, st>>>(qx, d, ck, T, kvp, heads, n_sel, probs, n_tok,
                                                                      tok_idx, bn);
}

}  // namespace fwd
