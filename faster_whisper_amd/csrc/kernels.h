// Host-side launch API of the gfx950 kernels (internal to libfwamd.so).
#pragma once
#include <atomic>
#include "common.h"

namespace fwk {

// ---- log-mel (logmel.hip) ---------------------------------------------------------
size_t logmel_lds_bytes();
void launch_logmel(hipStream_t st, const float* pcm_dev, const int64_t* offsets_dev, int B, int max_frames_total,
                   const float* consts, const float* filtT, int mel_pad, int n_mels, float* raw,
                   int64_t raw_bstride, int raw_stride, int* chunk_max, int drop_last, int out_frames,
                   float* out_f32, half_t* out_cl, int c_pad, int* n_frames_out);
void launch_features_to_cl(hipStream_t st, const float* feats_dev, int B, int n_mels, int frames, half_t* out_cl,
                           int c_pad);

// ---- dense GEMM (gemm.hip) --------------------------------------------------------
struct GemmParams {
  const half_t* A; int64_t lda; int64_t a_bstride;   // A[z][m][k], k contiguous
  const half_t* W; int64_t ldw;                       // W[n][k]
  const half_t* bias;                                 // [N] or null
  const half_t* res; int64_t ldr; int64_t r_bstride;  // residual [z][m][n] or null (added after act)
  half_t* C; int64_t ldc; int64_t c_bstride;          // C[z][m][n]   (TRANS: Ct[z][n][m], ldc = row stride)
  int M, N, K;
  int act;                                            // 0 none, 1 exact GELU
  int head_rows;                                      // > 0: cross-attention K (plain) / V^T (TRANS) in MFMA-fragment-major
                                                      // layout per 64-column head, head_rows = keys per head padded to 32
  // int8 path (a_scale != null): A and W point to int8 data; dequant scales per A row / per W row
  const float* a_scale; int64_t as_bstride; const float* w_scale;
  int nMt, nNt;                                       // filled by launch_gemm
  int n_mp, blk_m, blk_n;                             // ... m panels of the whole batch, tile-order block (gemm.hip)
  int vt_stage;                                       // ... TRANS, plain Ct: epilogue staged through LDS (gemm.hip), 0 = direct stores
  // LAYERED launch (n_layers > 1; fp16 only): the same A is multiplied with the W of n_layers layers in ONE launch — the
  // grid's n tiles run over all layers (n tile t belongs to layer t / nNt_layer), N / bias / C describe ONE layer and
  // layer l reads W + l * w_lstride, bias + l * bias_lstride and writes C + l * c_lstride (element strides).  Used for the
  // cross-attention K / V^T projections of all decoder layers (decoder.hip: 2 launches instead of 2 x 32)
  int n_layers; int64_t w_lstride, bias_lstride, c_lstride;
  int nNt_layer;                                      // filled by launch_gemm: n tiles of one layer
};
int launch_gemm(hipStream_t st, const GemmParams& p, int batch, bool trans);
extern std::atomic<int> g_gemm_order;                 // tile order knob (gemm.hip), for A/B measurements only
extern std::atomic<int> g_gemm_vt_stage;              // 1 (product): the plain V^T epilogue goes through LDS; 0: direct stores (A/B)

// ---- row kernels (rowops.hip) -----------------------------------------------------
// y[r] = LN(x[r]) * g + b, eps 1e-5, fp32 statistics (two-pass in registers)
// frag != 0: y in the MFMA-fragment-major layout of the decoder linears' activations (needs d % 32 == 0)
void launch_layernorm(hipStream_t st, const half_t* x, const half_t* g, const half_t* b, half_t* y, int rows, int d,
                      int frag = 0);
// per-row dynamic int8 quantisation (absmax / 127), optionally preceded by LayerNorm (g != null):
// xq[r][:] = rint(y * 127 / absmax(y)), scale[r] = absmax / 127 with y = LN(x[r]) rounded to fp16, or x[r]
// frag != 0: xq is written MFMA-fragment-major for the int8 register-streaming skinny GEMM (needs d % 64 == 0)
void launch_quant_rows(hipStream_t st, const half_t* x, int64_t ldx, const half_t* g, const half_t* b, int8_t* xq,
                       float* scale, int rows, int d, int frag = 0);
void launch_f32_to_f16(hipStream_t st, const float* x, half_t* y, int64_t n);
void launch_f16_to_f32(hipStream_t st, const half_t* x, float* y, int64_t n);

// ---- encoder attention (attn_enc.hip) ---------------------------------------------
// q,k: [B][T][ld] fp16 (head h at column h*64), vt: [B][H*64][ldvt] (V transposed, time contiguous)
// out: [B][T][ldo]. softmax(q k^T / 8) v, non-causal, T keys.
void launch_attn_enc(hipStream_t st, const half_t* q, const half_t* k, int64_t ld, int64_t qk_bstride,
                     const half_t* vt, int64_t ldvt, int64_t vt_bstride, half_t* out, int64_t ldo,
                     int64_t o_bstride, int B, int H, int T, int order = 0);

}  // namespace fwk
