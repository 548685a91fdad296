// K7 — encoder self-attention, flash style: softmax(Q K^T / 8) V, non-causal,
// T = 1500 keys, head dim 64.  One workgroup = (batch b, head h, 128 queries);
// each of the 4 waves owns 32 queries.
//
// gfx950 mapping
//  * S^T = K Q^T with v_mfma_f32_32x32x16_f16 (A = K tile rows from LDS, B = the wave's
//    Q fragment held in registers): the accumulator puts ONE query per lane (column
//    j = lane&31) and 16 keys per 32-key block in its registers, so the row max / sum
//    need a single cross-lane exchange (lane ^ 32) instead of a shuffle tree.
//  * O^T = V^T P^T: the contraction index of an MFMA may be any permutation as long as
//    both operands agree, so the P fragment is just the lane's own 8 accumulator values
//    (keys {0..3, 8..11} + 4*(lane>>5) + 16*ks) and the V^T fragment reads the same keys
//    from a time-contiguous V^T tile — no P shuffle, no LDS round trip for P.
//  * V is consumed transposed ([dh][time]); the QKV GEMM's TRANS epilogue emits it
//    in that layout, so no transpose pass exists anywhere.
//  * K tile: 128-byte rows, XOR-swizzled 16-byte chunks (conflict-free ds_read_b128);
//    V^T tile: rows padded to 136 bytes (conflict-free ds_read_b64).
//  * register-staged double buffering: next tile's global loads are issued before
//    the MFMA/softmax block, written to LDS after it; one barrier per 64-key tile.
//  * XCD-aware 1-D grid: the 12 query-tile workgroups of a (chunk, head) all stream the same K / V^T (384 KB); the
//    hardware places block b on XCD b % 8, so with the query tile as the fastest grid index they were spread over all
//    eight L2s and every L2 fetched the pair's K / V^T from HBM (measured round 3: 1 070 MB fetched per launch against
//    246 MB algorithmic).  Now the (chunk, head) pairs are dealt round-robin to the XCDs and the query tiles of a pair
//    are consecutive blocks OF ONE XCD: ~8 pairs are resident per XCD at a time (3 MB of K / V^T in a 4 MB L2), the
//    first workgroup of a pair misses and the other eleven hit.
//  * online softmax in fp32 with exp2.  The vector pipe, not the matrix pipe, is what bounds this kernel (per 64-key
//    tile and wave: 16 MFMAs = 512 matrix cycles against, originally, ~210 vector instructions), so the softmax is
//    written for the fewest vector instructions: log2 e / 8 is folded into the Q fragment once (no per-score scaling),
//    the tile maximum uses 3-input max, the running accumulator is rescaled only when the maximum actually moved (a
//    wave-uniform branch; for most tiles after the first few it does not), and P is converted to fp16 two at a time.
#include "common.h"
#include "kernels.h"

#define AE_Q 128
#define AE_KV 64
#define AE_VSTRIDE 68 /* halves per V^T tile row (136 B) */
#define AE_LAG 5.0f   /* how far (log2 domain) a query's scores may run ahead of its softmax reference: P <= 32 */

__global__ __launch_bounds__(256) void attn_enc_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                       int64_t ld, int64_t qk_bstride,
                                                       const half_t* __restrict__ vt, int64_t ldvt,
                                                       int64_t vt_bstride, half_t* __restrict__ out, int64_t ldo,
                                                       int64_t o_bstride, int T, int H, int n_pairs, int n_qt,
                                                       int order) {
  __shared__ __attribute__((aligned(16))) half_t sK[2][AE_KV * 64];
  __shared__ __attribute__((aligned(16))) half_t sV[2][64 * AE_VSTRIDE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  // block -> (pair, query tile): XCD x = block % 8 owns the pairs x, x + 8, ...; its blocks walk them tile by tile
  // (order 1 = the round-3 mapping, query tile fastest over ALL XCDs: kept for the A/B of profiles/attn_bench.py)
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int pair = order ? (int)blockIdx.x / n_qt : (idx / n_qt) * 8 + xcd;
  if (pair >= n_pairs) return;               // (whole workgroups: before any barrier)
  const int b = pair / H, h = pair - b * H, q0 = (order ? (int)blockIdx.x % n_qt : idx % n_qt) * AE_Q;

  const half_t* qb = q + (size_t)b * qk_bstride + h * 64;
  const half_t* kb = k + (size_t)b * qk_bstride + h * 64;
  const half_t* vb = vt + (size_t)b * vt_bstride + (size_t)(h * 64) * ldvt;

  // Q fragment (B operand): query row l31 of this wave, d = ks*16 + hi*8 .. +7, pre-scaled by log2(e) / 8 (fp32 product,
  // one rounding to fp16): the scores come out of the MFMA in the exp2 domain
  half8_t qf[4];
  {
    int qr = q0 + wave * 32 + l31;
    if (qr > T - 1) qr = T - 1;
    const half_t* qp = qb + (size_t)qr * ld + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      half8_t v = *reinterpret_cast<const half8_t*>(qp + ks * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * (0.125f * 1.4426950408889634f));
      qf[ks] = v;
    }
  }

  // staging map: rows r0, r0+32; 16-byte chunk c
  const int c = tid & 7, r0 = tid >> 3;
  const int nkt = (T + AE_KV - 1) / AE_KV;
  intx4 rk[2], rv[2];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      int key = kt * AE_KV + row;
      if (key > T - 1) key = T - 1;
      rk[i] = *reinterpret_cast<const intx4*>(kb + (size_t)key * ld + c * 8);
      rv[i] = *reinterpret_cast<const intx4*>(vb + (size_t)row * ldvt + kt * AE_KV + c * 8);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<intx4*>(&sK[buf][row * 64 + ((c ^ ((row >> 1) & 7)) << 3)]) = rk[i];
      intx2* dv = reinterpret_cast<intx2*>(&sV[buf][row * AE_VSTRIDE + c * 8]);
      dv[0] = intx2{rv[i][0], rv[i][1]};
      dv[1] = intx2{rv[i][2], rv[i][3]};
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();

  floatx16 o[2] = {floatx16{0}, floatx16{0}};
  // The softmax denominator rides on the matrix pipe: lsum += ONES x P^T with the same fp16 P fragments the PV product
  // consumes (every row of the tile = sum over the 16 keys of a fragment, per query = per lane: both lane halves already
  // folded in).  The kernel is bound by its vector instructions (SQ counters) with the MFMA pipe 27 %
  // busy: 4 MFMAs per tile replace 32 v_add per lane and the final cross-half exchange; and the denominator is the sum
  // of exactly the P values the numerator used.
  floatx16 lsum = floatx16{0};
  const half8_t ones8 = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
  // The softmax reference rides on the matrix pipe too: the scores leave the QK^T MFMAs already relative to a per-query
  // reference m_ref (the C operand of the first MFMA of a tile is a tile of -m_ref), so a tile whose scores stay within
  // AE_LAG of the reference needs no subtraction at all (32 v_sub per lane and tile otherwise).  m_ref is the running
  // maximum as of its last update: it is set from the first tile, and raised (accumulators rescaled, as before) only
  // when some query of the wave sees a score more than AE_LAG above it — so the largest P of a query stays within
  // [1, 2^AE_LAG]: no overflow of the fp16 P fragments, no underflow of the terms that matter.
  float m_ref = 0.f;
  floatx16 cneg = floatx16{0};               // -m_ref in every register (a lane's 16 rows belong to one query)

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nkt;
    if (more) gload(kt + 1);

    // ---- S^T[key][query] for the 64 keys of this tile ----
    floatx16 s[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = ks * 2 + hi;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        const int row = kb2 * 32 + l31;
        const half8_t kf = *reinterpret_cast<const half8_t*>(&sK[cur][row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3)]);
        s[kb2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? cneg : s[kb2], 0, 0, 0);
      }
    }
    // ---- online softmax (one query per lane; partner lane^32 holds the other keys) ----
    const int kbase = kt * AE_KV + 4 * hi;
    const bool tail = (kt * AE_KV + AE_KV) > T;
    if (tail) {
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + kb2 * 32 + (r & 3) + 8 * (r >> 2);
          if (key >= T) s[kb2][r] = -1.0e30f;
        }
    }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = __builtin_fmaxf(__builtin_fmaxf(mx, s[0][r]), s[1][r]);   // -> v_max3_f32
    mx = pair32_max(mx);                     // the partner lane holds the query's other 32 keys of the tile
    // mx is relative to m_ref.  Move the reference only for the first tile (it fixes the reference: any sign) or when
    // some query of the wave ran more than AE_LAG ahead of it (wave-uniform branch: no divergence; then every lane
    // raises its own reference to its own maximum, which is always allowed)
    if (kt == 0 || __builtin_amdgcn_ballot_w64(mx > AE_LAG) != 0) {
      const float delta = kt == 0 ? mx : fmaxf(mx, 0.f);
      // (first tile: lsum and o are still zero and delta may have any sign — a first tile whose scores all lie below
      // -128 would make exp2(-delta) infinite and 0 x inf = NaN; nothing is rescaled there)
      const float alpha = kt == 0 ? 1.f : __builtin_amdgcn_exp2f(-delta);
      m_ref += delta;
#pragma unroll
      for (int r = 0; r < 16; ++r) { cneg[r] = -m_ref; lsum[r] *= alpha; }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb2][r] -= delta;
    }
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb2][r] = __builtin_amdgcn_exp2f(s[kb2][r]);

    // ---- O^T[dh][query] += V^T[dh][key] * P^T[key][query] ----
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        half8_t pf;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {   // v_cvt_pkrtz would truncate: round-to-nearest pairs
          const half2_t h2 = {(half_t)s[kb2][ks2 * 8 + e], (half_t)s[kb2][ks2 * 8 + e + 1]};
          pf[e] = h2[0]; pf[e + 1] = h2[1];
        }
        lsum = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones8, pf, lsum, 0, 0, 0);
        const int koff = kb2 * 32 + ks2 * 16 + 4 * hi;  // keys koff+{0..3}, koff+8+{0..3}
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const half_t* vr = &sV[cur][(dt * 32 + l31) * AE_VSTRIDE + koff];
          const half4_t v0 = *reinterpret_cast<const half4_t*>(vr);
          const half4_t v1 = *reinterpret_cast<const half4_t*>(vr + 8);
          half8_t vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[dt], 0, 0, 0);
        }
      }

    if (more) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- normalise and store: lane = query, 4 consecutive dh per register group ----
  const float inv = 1.0f / lsum[0];             // (every row of lsum holds the query's whole sum)
  const int qr = q0 + wave * 32 + l31;
  if (qr < T) {
    half_t* op = out + (size_t)b * o_bstride + (size_t)qr * ldo + h * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        half4_t ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = (half_t)(o[dt][g * 4 + e] * inv);
        *reinterpret_cast<half4_t*>(op + dt * 32 + 8 * g + 4 * hi) = ov;
      }
  }
}

namespace fwk {
void launch_attn_enc(hipStream_t st, const half_t* q, const half_t* k, int64_t ld, int64_t qk_bstride,
                     const half_t* vt, int64_t ldvt, int64_t vt_bstride, half_t* out, int64_t ldo,
                     int64_t o_bstride, int B, int H, int T, int order) {
  const int n_qt = (T + AE_Q - 1) / AE_Q, n_pairs = B * H;
  const int grid = ((n_pairs + 7) / 8) * 8 * n_qt;
  // (132 registers: 3 waves per SIMD.  Forcing 4 with __launch_bounds__(256, 4) spills and measured 6 % slower.)
  attn_enc_kernel<<<grid, 256, 0, st>>>(q, k, ld, qk_bstride, vt, ldvt, vt_bstride, out, ldo, o_bstride, T, H, n_pairs, n_qt, order);
}
}  // namespace fwk
