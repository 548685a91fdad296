// K2/K3/K6/K8-K11 — the dense fp16 / int8 MFMA GEMM of the encoder (and of every
// "many rows" linear: conv1d-as-GEMM, fused QKV, out-proj, FFN, cross-K/V projection).
//
//   C[z][m][n] = epi( sum_k A[z][m][k] * W[n][k] )      A, W, C fp16; fp32 accumulate
//   epi(v)     = act(v + bias[n]) + res[z][m][n]         act = identity | exact-erf GELU
//
// Both operands are K-contiguous ("B^T" form), which is the natural MFMA fragment
// order: lane (row = l&31, k-octet = l>>5) reads 16 contiguous bytes.
//
// gfx950 mapping (MI355X_MICROARCH.md / cdna_hip_programming.md section 5: what lifts a GEMM past the 128x128
// two-barrier structure is the depth of the staging pipeline, not the tile alone)
//  * 256 x 256 x 128-byte block tile (64 halves or 128 int8 of K), 8 waves as 2 (M) x 4 (N), each wave
//    128 x 64 = 4 x 2 tiles of v_mfma_f32_32x32x16_f16 / v_mfma_i32_32x32x32_i8 (128 accumulator registers);
//    one workgroup per CU, 128 KB of LDS.
//  * HBM/L2 -> LDS by direct DMA (global_load_lds, 16 B per lane, no VGPR round trip).  A K tile is staged as FOUR
//    UNITS of 16 KB — A0 / A1 = the first / second 64 rows of every wave row-half, B0 / B1 = the first / second 32
//    columns of every wave column-quarter — because that is the granularity at which the workgroup consumes it:
//    a K tile is multiplied in four PHASES of 8 MFMAs per wave,
//         phase 1: A0 x B0   (reads A0, B0 from LDS into registers)
//         phase 2: A0 x B1   (reads B1; A0 stays in registers)
//         phase 3: A1 x B0   (reads A1; B0 stays in registers)
//         phase 4: A1 x B1   (reads nothing)
//    so a unit's LDS bytes are dead one to three phases after they were read and can be refilled with the tile
//    two steps ahead while the current tile is still being multiplied.  One unit (2 DMA instructions per wave)
//    is issued per phase:   phase 1: B1(t+1)   phase 2: A1(t+1)   phase 3: B0(t+2)   phase 4: A0(t+2)
//    which puts every unit 5-6 phases (>= 1300 cycles of MFMA work) ahead of its first read, and ONE counted wait,
//    s_waitcnt vmcnt(8) at the end of every load segment, retires exactly the unit that the NEXT phase reads (the
//    four younger units stay in flight across the barriers; the queue is never drained inside the K loop).
//  * The 8 waves run as two groups (wave row 0 / wave row 1 = the two waves of every SIMD) staggered by half a
//    phase: each phase is  [load segment: ds_read fragments, issue DMA, counted wait] barrier [compute segment: 8
//    MFMAs] barrier,  and group 1 enters the loop one barrier late, so on every SIMD one wave multiplies while
//    the other loads.  RAW / WAR of the DMA ring under that stagger: a unit is read one phase after the wait that
//    retires it (all 8 waves have passed their wait and a barrier), and refilled two phases after its last read.
//  * LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row >> 1) & 7 — applied to the DMA SOURCE
//    address, the LDS image itself is lane-linear — which makes every ds_read_b128 lane group conflict-free.
//  * normal mode computes D = W_tile * A_tile^T so that each lane ends up with four consecutive n of one row m ->
//    8-byte stores; TRANS mode computes D = A_tile * W_tile^T, each lane holds four consecutive m of one column
//    n, and the tile is stored transposed (Ct[z][n][m]) — used to emit V^T for the attention kernels without a
//    separate transpose pass.
//  * conv1d (k=3, stride s) over a channel-last, zero-padded image is this GEMM with lda = s*C and K = 3*C: the
//    three taps of an output row are contiguous in memory.
//  * 1-D grid with a bijective XCD remap (an XCD gets a contiguous range of logical tile ids), and the logical ids walk
//    the (m panel, n tile) plane in BLOCKS of bm x bn tiles (~32 = the workgroups an XCD runs at a time; the m panels
//    of all chunks of the batch count as one axis: they share W).  What an XCD's L2 has to supply to the 32 workgroups
//    it runs in near lockstep is then bm A panels + bn W tiles per K tile (4 + 8 = 12 for the 20-tile-wide FFN-up GEMM)
//    instead of 2 + 20 with n fastest across the whole width — and every byte the L2 misses is HBM bandwidth taken
//    from the decode runs' cross-attention stream, which is what the chip is short of.
#include "common.h"
#include "kernels.h"
#include <atomic>
#include <type_traits>

#define GB_M 256
#define GB_N 256
#define GB_UNIT_BYTES (128 * 128)          // one staging unit: 128 rows x 128 B
#define GB_SLOT_BYTES (4 * GB_UNIT_BYTES)  // one K tile: A0, A1, B0, B1
#define GB_LDS_BYTES (8 * 64 * 272)        // the ring (2 K tiles = 128 KB); the epilogue stages 8 x 17 KB through it

typedef int intx16 __attribute__((ext_vector_type(16)));

// profiles/ubench/gemm_timeline.hip compiles this file with GB_TL defined to stamp s_memtime at the segment
// boundaries of one workgroup; in the product build the probes are empty
#ifndef GB_TL
#define GB_TL(slot)
#endif

// workgroup barrier that the compiler may not move LDS reads or DMA issues across (the raw builtin carries no fence;
// the counted vmcnt protocol below is what orders the DMA ring, so no vmcnt(0) drain is wanted here)
#define GB_BARRIER()                      \
  do {                                    \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_barrier();         \
    asm volatile("" ::: "memory");         \
  } while (0)

// I8: the int8_float16 path (K25): A and W are int8 (per-row dequant scales a_scale[m], w_scale[n]),
// v_mfma_i32_32x32x32_i8 accumulates in int32, the epilogue de-quantises.  The tile is defined in BYTES
// (128-byte rows = 64 halves or 128 int8), so staging, swizzle and fragment reads are shared.
template <bool TRANS, bool I8, bool LAYERED = false>
__global__ __launch_bounds__(512) void gemm_f16_kernel(fwk::GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int ES = I8 ? 1 : 2;        // element size

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 2, wn = wave & 3;

  int bid = xcd_remap(blockIdx.x, gridDim.x);
  int mp, nt;                                // m panel over all chunks of the batch, n tile
  if (p.blk_n > 0) {
    // bands of blk_m panels, inside a band column blocks of blk_n tiles, inside a block n fastest (blk_n divides nNt)
    const int per_band = p.blk_m * p.nNt;
    const int band = bid / per_band, r = bid - band * per_band;
    int bm = p.n_mp - band * p.blk_m; if (bm > p.blk_m) bm = p.blk_m;     // the last band may be short
    const int per_blk = bm * p.blk_n;
    const int cb = r / per_blk, rr = r - cb * per_blk;
    const int mi = rr / p.blk_n;
    mp = band * p.blk_m + mi;
    nt = cb * p.blk_n + (rr - mi * p.blk_n);
  } else {
    mp = bid / p.nNt;
    nt = bid - mp * p.nNt;
  }
  const int z = mp / p.nMt;
  const int mt = mp - z * p.nMt;
  // layered launch: the n tile runs over the layers' columns; everything below sees ONE layer (p.N columns, its W / bias / C)
  // (LAYERED is its own instantiation: the code of the plain launches is what it was)
  int layer = 0;
  if (LAYERED) { layer = nt / p.nNt_layer; nt -= layer * p.nNt_layer; }
  const int m0 = mt * GB_M, n0 = nt * GB_N;

  const char* Ab = reinterpret_cast<const char*>(p.A) + (size_t)z * p.a_bstride * ES;
  const char* Wb = reinterpret_cast<const char*>(p.W) + (LAYERED ? (size_t)layer * p.w_lstride * ES : 0);
  const half_t* const biasp = (LAYERED && p.bias) ? p.bias + (size_t)layer * p.bias_lstride : p.bias;
  half_t* const Cp = LAYERED ? p.C + (size_t)layer * p.c_lstride : p.C;

  // ---- staging addresses.  A wave DMA instruction fills 64 consecutive 16-byte LDS slots = 8 unit rows; wave w
  // issues pieces w and w + 8 of a unit (unit rows 8*piece .. +7).  Unit row u of A0 is tile row (u>>6)*128 + (u&63)
  // (+64 for A1); of B0 tile column (u>>5)*64 + (u&31) (+32 for B1).  LDS slot (u, c') receives global chunk
  // c = c' ^ ((u >> 1) & 7); the fragment reads below apply the same involution. ----
  // (32-bit offsets from the wave-uniform operand bases: the DMA instruction takes SGPR base + VGPR offset — half the
  //  address registers and no 64-bit add per piece in the load segment: 890 -> 960 TFLOP/s encoder-weighted)
  unsigned goff[4][2];   // [unit: A0, A1, B0, B1][piece]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = (i * 8 + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((u >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int am = m0 + (u >> 6) * 128 + (u & 63) + h * 64; if (am > p.M - 1) am = p.M - 1;
      int wr = n0 + (u >> 5) * 64 + (u & 31) + h * 32; if (wr > p.N - 1) wr = p.N - 1;
      goff[h][i] = (unsigned)((size_t)am * p.lda * ES + c * 16);
      goff[2 + h][i] = (unsigned)((size_t)wr * p.ldw * ES + c * 16);
    }
  }
  const int nk = p.K * ES / 128;
  auto issue = [&](int unit, int kt) {   // 2 DMA instructions per wave
    char* dst = smem_raw + (kt & 1) * GB_SLOT_BYTES + unit * GB_UNIT_BYTES;
    const char* base = unit < 2 ? Ab : Wb;
    const unsigned koff = (unsigned)kt * 128u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (size_t)(goff[unit][i] + koff)),
                                       (__attribute__((address_space(3))) void*)(dst + (i * 8 + wave) * 1024), 16, 0, 0);
  };

  floatx16 accf[4][2];
  intx16 acci[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { accf[i][j] = floatx16{0}; acci[i][j] = intx16{0}; }

  // fragment read offsets inside a unit (bytes, without the k-chunk): row * 128, and the row's swizzle key
  int arow[2], akey[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = wm * 64 + i * 32 + l31;
    arow[i] = u * 128;
    akey[i] = (u >> 1) & 7;
  }
  const int bu = wn * 32 + l31;
  const int brow = bu * 128, bkey = (bu >> 1) & 7;

  intx4 fa[2][4], fb0[4], fb1[4];
  auto read_a = [&](const char* unit) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        fa[i][ks] = *reinterpret_cast<const intx4*>(unit + arow[i] + (((ks * 2 + hi) ^ akey[i]) << 4));
  };
  auto read_b = [&](const char* unit, intx4 (&fb)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const intx4*>(unit + brow + (((ks * 2 + hi) ^ bkey) << 4));
  };
  // 8 MFMAs: the two row tiles of A half `ah` (registers fa) x column tile `ni`
  auto mma = [&](int ah, int ni, const intx4 (&fb)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int mi = ah * 2 + i;
        // TRANS: D[m][n];  else D[n][m]  (both accumulate into acc[mi][ni])
        const intx4 opa = TRANS ? fa[i][ks] : fb[ks];
        const intx4 opb = TRANS ? fb[ks] : fa[i][ks];
        if (I8) acci[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(opa, opb, acci[mi][ni], 0, 0, 0);
        else accf[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, opa),
                                                                 __builtin_bit_cast(half8_t, opb), accf[mi][ni], 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
  };
  // end of a load segment: retire the unit the next phase reads (everything but the four youngest units); when
  // this phase had nothing left to issue the queue is shorter than the count assumes, so drain it
  auto seg_wait = [&](bool issued) {
    if (issued) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---- prologue: tile 0 whole, tile 1's B0 / A0 (the issue order of the steady state) ----
  issue(2, 0); issue(0, 0); issue(3, 0); issue(1, 0);
  if (nk > 1) { issue(2, 1); issue(0, 1); }
  seg_wait(nk > 1);
  GB_BARRIER();
  if (wm == 1) GB_BARRIER();   // group 1 runs half a phase behind group 0

  // one K tile; n1 / n2 = tiles kt + 1 / kt + 2 exist — compile-time facts, so that the steady-state loop carries no
  // branch and no queue-length test in its load segments (the last two tiles run the same code with them false)
  auto tile = [&](int kt, auto N1, auto N2) {
    constexpr bool n1 = decltype(N1)::value, n2 = decltype(N2)::value;
    const char* slot = smem_raw + (kt & 1) * GB_SLOT_BYTES;
    // phase 1: A0 x B0
    GB_TL(0);
    read_a(slot);
    read_b(slot + 2 * GB_UNIT_BYTES, fb0);
    if (n1) issue(3, kt + 1);
    seg_wait(n1);
    GB_TL(1);
    GB_BARRIER();
    GB_TL(2);
    mma(0, 0, fb0);
    GB_TL(3);
    GB_BARRIER();
    // phase 2: A0 x B1
    GB_TL(4);
    read_b(slot + 3 * GB_UNIT_BYTES, fb1);
    if (n1) issue(1, kt + 1);
    seg_wait(n1);
    GB_TL(5);
    GB_BARRIER();
    GB_TL(6);
    mma(0, 1, fb1);
    GB_TL(7);
    GB_BARRIER();
    // phase 3: A1 x B0
    GB_TL(8);
    read_a(slot + GB_UNIT_BYTES);
    if (n2) issue(2, kt + 2);
    seg_wait(n2);
    GB_TL(9);
    GB_BARRIER();
    GB_TL(10);
    mma(1, 0, fb0);
    GB_TL(11);
    GB_BARRIER();
    // phase 4: A1 x B1
    GB_TL(12);
    if (n2) issue(0, kt + 2);
    seg_wait(n2);
    GB_TL(13);
    GB_BARRIER();
    GB_TL(14);
    mma(1, 1, fb1);
    GB_TL(15);
    GB_BARRIER();
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  int kt = 0;
  for (; kt + 2 < nk; ++kt) tile(kt, T_{}, T_{});
  if (kt + 1 < nk) { tile(kt, T_{}, F_{}); ++kt; }
  if (kt < nk) tile(kt, F_{}, F_{});
  if (wm == 0) GB_BARRIER();   // the barrier group 1 spent on the stagger

  // ---------------------------------- epilogue ----------------------------------
  const float* sa = I8 ? p.a_scale + (size_t)z * p.as_bstride : nullptr;
  if (!TRANS && p.head_rows == 0) {
    // Row-major output: the accumulator layout gives a lane 4 consecutive n of ONE row, so direct stores would be
    // 32 eight-byte stores per lane into 32 different cache lines per instruction — store-issue bound, and with one
    // workgroup per CU nothing hides that tail.  Instead the wave's 128 x 64 sub-tile goes through its own 17 KB
    // of the (now idle) LDS ring, in two halves of 64 rows, as float32 — bias and activation applied on the way
    // in — and leaves as whole 128-byte row segments, 16 B per lane, the residual (read with the same coalesced
    // pattern) added BEFORE the one rounding to fp16: y = fp16(act(acc + bias) + res), as the oracle states it.
    constexpr int EP_STRIDE = 272;                       // bytes per staged row: 64 floats + 16 (16-B aligned)
    char* ep = smem_raw + wave * (64 * EP_STRIDE);
    half_t* Cb = Cp + (size_t)z * p.c_bstride;
    const half_t* Rb = p.res ? p.res + (size_t)z * p.r_bstride : nullptr;
    const int mw = m0 + wm * 128, nw = n0 + wn * 64;     // this wave's sub-tile
    const int rr = lane >> 3, cc = (lane & 7) * 8;       // store pass: 8 lanes per row segment, 8 columns each
    const bool vec_ok = (p.ldc % 8 == 0) && (p.c_bstride % 8 == 0) && (!Rb || ((p.ldr % 8 == 0) && (p.r_bstride % 8 == 0)));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
          const int mi = half * 2 + mh;
          const int r = mh * 32 + l31;                   // row inside the staged half
          int mc = mw + half * 64 + r; if (mc > p.M - 1) mc = p.M - 1;
          const float sam = I8 ? sa[mc] : 1.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c = ni * 32 + 8 * g + 4 * hi;
            int n = nw + c; if (n > p.N - 4) n = p.N - 4;  // clamped columns are never stored
            floatx4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              v[e] = I8 ? (float)acci[mi][ni][g * 4 + e] * sam * p.w_scale[n + e] : accf[mi][ni][g * 4 + e];
            if (biasp) {
              const half4_t bv = *reinterpret_cast<const half4_t*>(biasp + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)bv[e];
            }
            if (p.act == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            *reinterpret_cast<floatx4*>(ep + r * EP_STRIDE + c * 4) = v;
          }
        }
      // (a wave's LDS operations execute in order: its reads below see its writes above, and the next half's
      //  writes come after these reads)
#pragma unroll 4
      for (int j = 0; j < 8; ++j) {
        const int r = j * 8 + rr;
        const int m = mw + half * 64 + r, n = nw + cc;
        const floatx4 v0 = *reinterpret_cast<const floatx4*>(ep + r * EP_STRIDE + cc * 4);
        const floatx4 v1 = *reinterpret_cast<const floatx4*>(ep + r * EP_STRIDE + cc * 4 + 16);
        if (m >= p.M || n >= p.N) continue;
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (vec_ok && n + 8 <= p.N) {
          if (Rb) {
            const half8_t rv = *reinterpret_cast<const half8_t*>(Rb + (size_t)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
          }
          half8_t o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
          *reinterpret_cast<half8_t*>(Cb + (size_t)m * p.ldc + n) = o;
        } else {
          for (int e = 0; e < 8 && n + e < p.N; ++e) {
            float t = v[e];
            if (Rb) t += (float)Rb[(size_t)m * p.ldr + n + e];
            Cb[(size_t)m * p.ldc + n + e] = (half_t)t;
          }
        }
      }
    }
  } else if (p.head_rows > 0) {
    // Cross-attention K (plain) / V^T (TRANS), MFMA-fragment-major per (chunk, head) — the layout dec_cross_attn_kernel
    // streams: a 32-key group of a head is 4 runs of 64 lanes x 16 B,
    //   K:    run q = 2*sub + s, lane = 16*g + j  <->  key 32*gi + 8*(j>>2) + 4*sub + (j&3), dims 32*s + 8*g + [0,8)
    //   V^T:  run dt,            lane = 16*g + j  <->  dim 16*dt + j,                      keys 32*gi + 8*g + [0,8)
    // A wave's 128 x 64 sub-tile is 128 keys of ONE head = 16 consecutive runs = 16 KB contiguous in the destination,
    // but the accumulator hands a lane 4 values of one row: stored directly that is 32 eight-byte stores per lane into
    // 32 different lines per instruction (store-issue bound: these projections ran at 530 TFLOP/s against 950 for the
    // row-major epilogue).  So the sub-tile goes through the wave's patch of the idle ring, in two halves of 64 keys,
    // as fp16 [key][dim] (K) or [dim][key] (V^T), and leaves as whole 1 KB runs, 16 B per lane.
    constexpr int FS = 144;                                  // bytes per staged row: 64 halves + 16
    char* ep = smem_raw + wave * (64 * FS);
    half_t* Cb = Cp + (size_t)z * p.c_bstride;
    const int mw = m0 + wm * 128, nw = n0 + wn * 64;         // this wave's keys / its head's 64 dims
    if (nw < p.N) {
      half_t* Hb = Cb + (size_t)(nw >> 6) * p.head_rows * 64;
      const int g4 = lane >> 4, j = lane & 15;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mh = 0; mh < 2; ++mh) {
            const int mi = half * 2 + mh;
            if (!TRANS) {
              const int r = mh * 32 + l31;                   // key inside the half
              int mc = mw + half * 64 + r; if (mc > p.M - 1) mc = p.M - 1;
              const float sam = I8 ? sa[mc] : 1.f;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int c = ni * 32 + 8 * g + 4 * hi;
                int n = nw + c; if (n > p.N - 4) n = p.N - 4;
                half4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  float v = I8 ? (float)acci[mi][ni][g * 4 + e] * sam * p.w_scale[n + e] : accf[mi][ni][g * 4 + e];
                  if (biasp) v += (float)biasp[n + e];
                  if (p.act == 1) v = gelu_erf(v);
                  o[e] = (half_t)v;
                }
                *reinterpret_cast<half4_t*>(ep + r * FS + c * 2) = o;
              }
            } else {
              const int c = ni * 32 + l31;                   // dim inside the head
              int n = nw + c; if (n > p.N - 1) n = p.N - 1;
              const float bv = biasp ? (float)biasp[n] : 0.f;
              const float swn = I8 ? p.w_scale[n] : 1.f;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int r = mh * 32 + 8 * g + 4 * hi;      // 4 consecutive keys inside the half
                half4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  int mm = mw + half * 64 + r + e; if (mm > p.M - 1) mm = p.M - 1;
                  float v = (I8 ? (float)acci[mi][ni][g * 4 + e] * swn * sa[mm] : accf[mi][ni][g * 4 + e]) + bv;
                  if (p.act == 1) v = gelu_erf(v);
                  o[e] = (half_t)v;
                }
                *reinterpret_cast<half4_t*>(ep + c * FS + r * 2) = o;
              }
            }
          }
        // (a wave's LDS operations execute in order: the reads below see the writes above)
#pragma unroll
        for (int gl = 0; gl < 2; ++gl) {
          const int key0 = mw + half * 64 + gl * 32;          // first key of the group
          if (key0 >= p.M) continue;                          // (wave-uniform)
          half_t* Gb = Hb + (size_t)(key0 >> 5) * 4 * 512;    // 4 runs of 512 halves
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            intx4 v;
            if (!TRANS) {
              const int kl = gl * 32 + 8 * (j >> 2) + 4 * (q >> 1) + (j & 3);
              v = *reinterpret_cast<const intx4*>(ep + kl * FS + (32 * (q & 1) + 8 * g4) * 2);
              if (mw + half * 64 + kl < p.M) *reinterpret_cast<intx4*>(Gb + ((size_t)q * 64 + lane) * 8) = v;
            } else {
              const int kl = gl * 32 + 8 * g4;                // this lane's 8 consecutive keys
              v = *reinterpret_cast<const intx4*>(ep + (16 * q + j) * FS + kl * 2);
              const int left = p.M - (mw + half * 64 + kl);   // keys of the 8 that exist
              if (left <= 0) continue;
              if (left < 8) {                                 // the padded keys of the last group stay zero
                half8_t h = __builtin_bit_cast(half8_t, v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                  if (e >= left) h[e] = (half_t)0.f;
                v = __builtin_bit_cast(intx4, h);
              }
              *reinterpret_cast<intx4*>(Gb + ((size_t)q * 64 + lane) * 8) = v;
            }
          }
        }
      }
    }
  } else if (TRANS && !I8 && p.vt_stage) {
    // Plain transposed output Ct[z][n][m] (the encoder's V^T, [d][t_pad] per chunk: what attn_enc_kernel multiplies P
    // with).  The accumulator hands a lane 4 consecutive m of ONE column n: stored directly that is 32 eight-byte stores
    // per lane, each instruction into 32 different rows of Ct (the store-issue-bound pattern the other two epilogues left
    // in rounds 2 and 4).  The wave's 128 (m) x 64 (n) sub-tile goes through its 17 KB patch of the idle ring as fp16
    // [n][m] — same arithmetic per element as the direct form below, so the same bits — and leaves as whole 256-byte
    // row segments of Ct, 16 B per lane.  Columns m >= M of Ct (the padding up to t_pad) are never written, as before.
    // (fp16 only: in the int8 instantiation the staged form next to the direct one spills 38 registers — the per-row
    //  de-quantisation factors are loads of their own — so int8_float16 keeps the direct stores.)
    constexpr int TS = 272;                                  // bytes per staged row: 128 halves + 16
    char* ep = smem_raw + wave * (64 * TS);
    half_t* Cb = Cp + (size_t)z * p.c_bstride;
    const int mw = m0 + wm * 128, nw = n0 + wn * 64;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int c = ni * 32 + l31;                           // column of the sub-tile = row of the staged patch
      int n = nw + c; if (n > p.N - 1) n = p.N - 1;          // clamped columns are never stored
      const float bv = biasp ? (float)biasp[n] : 0.f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r = mi * 32 + 8 * g + 4 * hi;            // 4 consecutive m inside the sub-tile
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = accf[mi][ni][g * 4 + e] + bv;
            if (p.act == 1) v = gelu_erf(v);
            o[e] = (half_t)v;
          }
          *reinterpret_cast<half4_t*>(ep + c * TS + r * 2) = o;
        }
    }
    // (a wave's LDS operations execute in order: the reads below see the writes above)
    const int rr = lane >> 4, cc = (lane & 15) * 8;          // 16 lanes per staged row, 8 m each
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const int c = j * 4 + rr;
      const int n = nw + c, m = mw + cc;
      const intx4 v = *reinterpret_cast<const intx4*>(ep + c * TS + cc * 2);
      if (n >= p.N || m >= p.M) continue;
      half_t* dst = Cb + (size_t)n * p.ldc + m;
      if (m + 8 <= p.M) {
        *reinterpret_cast<intx4*>(dst) = v;
      } else {
        const half8_t h = __builtin_bit_cast(half8_t, v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (m + e < p.M) dst[e] = h[e];
      }
    }
  } else {
    half_t* Cb = Cp + (size_t)z * p.c_bstride;  // Ct[z][n][m], ldc = row stride of Ct
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + l31;
        if (n >= p.N) continue;
        const float bv = biasp ? (float)biasp[n] : 0.f;
        const float swn = I8 ? p.w_scale[n] : 1.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = m0 + wm * 128 + mi * 32 + 8 * g + 4 * hi;
          half4_t o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int mm = m + e; if (mm > p.M - 1) mm = p.M - 1;
            float v = (I8 ? (float)acci[mi][ni][g * 4 + e] * swn * sa[mm] : accf[mi][ni][g * 4 + e]) + bv;
            if (p.act == 1) v = gelu_erf(v);
            o[e] = (half_t)v;
          }
          half_t* dst = Cb + (size_t)n * p.ldc + m;
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

// tile order of launch_gemm: 1 = blocked (the product), 0 = n fastest across the whole width (rounds 1-3); a process-wide
// knob for the A/B of profiles/gemm_bench.py (fw_test_knob)
std::atomic<int> g_gemm_order{1};
// plain V^T epilogue: 1 = staged through LDS (the product), 0 = direct stores (rounds 1-4); knob 5 of fw_test_knob
std::atomic<int> g_gemm_vt_stage{1};

int launch_gemm(hipStream_t st, const GemmParams& pin, int batch, bool trans) {
  GemmParams p = pin;
  const bool i8 = p.a_scale != nullptr;
  const int es = i8 ? 1 : 2;
  if (p.K <= 0 || (p.K * es) % 128 != 0) return -1;
  if (((p.lda * es) % 16) || ((p.ldw * es) % 16) || ((p.a_bstride * es) % 16)) return -1;
  if (!trans && ((p.N % 4) || (p.ldc % 4) || (p.c_bstride % 4))) return -1;
  if (i8 && !p.w_scale) return -1;
  p.nMt = (p.M + GB_M - 1) / GB_M;
  p.nNt_layer = (p.N + GB_N - 1) / GB_N;
  if (p.n_layers > 1 && (i8 || p.res)) return -1;      // (the int8 path keeps one launch per layer; no layered residual)
  p.nNt = p.nNt_layer * (p.n_layers > 1 ? p.n_layers : 1);
  p.n_mp = p.nMt * batch;
  p.blk_m = p.blk_n = 0;
  // (16-byte row segments of Ct need ldc and the batch stride to be multiples of 8 halves)
  p.vt_stage = (trans && !i8 && p.head_rows == 0 && g_gemm_vt_stage.load(std::memory_order_relaxed) == 1 && p.ldc % 8 == 0 &&
                p.c_bstride % 8 == 0) ? 1 : 0;
  if (g_gemm_order.load(std::memory_order_relaxed) == 1 && p.nNt > 1 && p.n_mp > 1) {
    // L2 fill traffic of a block of bm x bn tiles that an XCD's 32 workgroups run in near lockstep is (bm + bn) operand
    // panels (measured: the out-projection, 6.4 + 5 panels per 32 tiles, fetches 119 MB = that model's 112; the FFN-up
    // GEMM with n fastest across its 20 tiles, 1.6 + 20 panels, fetched 0.85-0.9 GB per launch against 74 MB of
    // operands): the divisor of the width that makes the block squarest, bm = 32 / bn
    int bn = 1, best = 1 << 30;
    for (int c = 1; c <= p.nNt && c <= 16; ++c) {
      if (p.nNt % c) continue;
      const int cost = c + (32 + c - 1) / c;
      if (cost <= best) { best = cost; bn = c; }
    }
    p.blk_n = bn;
    p.blk_m = 32 / bn > 1 ? 32 / bn : 1;
  }
  const int lds = GB_LDS_BYTES;  // 128 KiB: one workgroup per CU
  static std::atomic<unsigned long long> attr_done{0};   // one bit per device (the limit is a per-device attribute)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<false, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<true, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<true, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
  }
  const int grid = p.nMt * p.nNt * batch;
  if (p.n_layers > 1) {
    static std::atomic<unsigned long long> attr_l{0};
    if (!((attr_l.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<false, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_kernel<true, false, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_l.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    if (trans) gemm_f16_kernel<true, false, true><<<grid, 512, lds, st>>>(p);
    else gemm_f16_kernel<false, false, true><<<grid, 512, lds, st>>>(p);
    return 0;
  }
  if (i8) {
    if (trans) gemm_f16_kernel<true, true><<<grid, 512, lds, st>>>(p);
    else gemm_f16_kernel<false, true><<<grid, 512, lds, st>>>(p);
  } else {
    if (trans) gemm_f16_kernel<true, false><<<grid, 512, lds, st>>>(p);
    else gemm_f16_kernel<false, false><<<grid, 512, lds, st>>>(p);
  }
  return 0;
}

}  // namespace fwk
