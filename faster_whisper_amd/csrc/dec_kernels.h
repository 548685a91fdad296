// Launch API of the decoder-step kernels (dec_kernels.hip).
#pragma once
#include "common.h"

namespace fwd {

constexpr int FIN_CAP = 48;  // finished hypotheses kept per chunk (>= round(K*patience) + K)

// Per-generate constants handed to the kernels by value.
struct GenDev {
  int B, K, R;             // chunks, beams per chunk, rows = B*K
  int P;                   // prompt length (all prompts of a call have the same length)
  int budget;              // max new tokens
  int max_fin;             // round(K * patience)
  int V, n_text_ctx;
  int with_ts, suppress_blank, min_new, mits, ngram;
  float rep_pen, lp_pow;
  int eot, no_ts, ts_begin;
  int n_sup_begin, sup_begin[8];
  int sample;              // 1: draw the next token from softmax(logp / T) instead of arg-max / beam
  float inv_temp;
  unsigned seed_lo, seed_hi;
  int kv_div;              // decode chunks per encoder chunk (sampling: num_hypotheses), 1 otherwise
  int ctx, cache_rows;     // self-attention cache geometry of this run: positions per slot, slots per layer
};

// blk_n > 0: POSITION BLOCKS (prompt forward, align) — the rows of a chunk are blk_n consecutive positions pos_fixed ..
// of its beam slot 0 (row = chunk * blk_n + j) instead of beams at one position; same per-row arithmetic, same bits
void launch_embed(hipStream_t st, const int* tok, const half_t* emb, const half_t* pos_emb, half_t* x, half_t* xfrag,
                  int rows, int d,
                  const int* d_step, int pos_fixed, int P, int blk_n = 0);
// whether launch_self_attn can take position blocks for this cache geometry
bool self_attn_block_ok(int n_ctx, int cache_ctx, int d, int R_total);
// n_ctx: stride of the slot table (the text context); cache_ctx: positions per slot of the K/V cache of this run
void launch_self_attn(hipStream_t st, const half_t* qkv, int d, half_t* kc, half_t* vc, int n_ctx, int cache_ctx, int H,
                      const uint8_t* kvidx2, int Kbeam, int kmul, half_t* out, int rows, const int* d_step,
                      int pos_fixed, int P, int R_total, int frag, int blk_n = 0);
// 0: the product's choice by launch size; 1: the first form (rounds 1-4); 2: latency form; 3: throughput form (same bits)
void set_self_attn_form(int form);
void set_cross_attn_regs(int cap);   // fw_test_knob(7, ..): register cap of dec_cross_attn_kernel (0 none, 1: 96, 2: 80)
// changes whenever a measurement knob changed the kernels a decode step launches: cached step graphs carry it
int kernel_forms_epoch();
void bump_kernel_forms_epoch();
// slot_map: [B / kv_div] encoder chunk of the run -> chunk slot behind ck / cvt (the cross-attention pool)
void launch_cross_attn(hipStream_t st, const half_t* qx, int d, const half_t* ck, const half_t* cvt, int T, int kvp,
                       int kmul, half_t* out, int B, int H, const int* done, int kv_div, int frag, const int* slot_map);
// rows from which launch_dec_gemm_frag hands a decode run's linears to the GEMM-shaped kernel
int dec_big_min_rows();
int dec_big_min_rows_of(int role, int compute_type);   // role 0 qkv, 1 d x d, 2 ffn1, 3 ffn2, < 0: the lowest; 0 fp16 / 1 int8
// frag = 1: `out` is a fragment-major [rows/16][d/32][64][8] buffer (input of launch_dec_gemm_frag)
int launch_dec_gemm_frag(hipStream_t st, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1,
                         const float* cf, const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R,
                         int N, int K, int act);
// the skinny kernel whatever the row count (launch_dec_gemm_frag hands merged runs to the GEMM-shaped kernel)
int launch_dec_gemm_skinny(hipStream_t st, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1,
                           const float* cf, const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R,
                           int N, int K, int act);
// ... with an explicit tile grouping (1: one 16 x 16 tile per workgroup, 2: 2 x 2 tiles)
int launch_dec_gemm_skinny_tiles(hipStream_t st, int tiles, const half_t* xf, const half_t* Wf, const half_t* bias,
                                 const float* s1, const float* cf, const half_t* res, int ldr, half_t* out, int ldo,
                                 half_t* out_frag, int R, int N, int K, int act);
// the GEMM-shaped kernel of merged runs whatever the row count, workgroup shape cfg (dec_kernels.hip); same bits
int launch_dec_gemm_big(hipStream_t st, int cfg, const half_t* xf, const half_t* Wf, const half_t* bias, const float* s1,
                        const float* cf, const half_t* res, int ldr, half_t* out, int ldo, half_t* out_frag, int R,
                        int N, int K, int act);
int launch_dec_gemm_frag_variant(hipStream_t st, int variant, bool lnf, const half_t* xf, const half_t* Wf,
                                 const half_t* bias, const float* s1, const float* cf, half_t* out, int R, int N,
                                 int K);
int launch_dec_gemm_frag_i8(hipStream_t st, const int8_t* xq, const float* x_scale, const int8_t* Wq,
                            const float* w_scale, const half_t* bias, const half_t* res, int ldr, half_t* out, int ldo,
                            int R, int N, int K, int act);
// vocabulary projection -> float32 logits; operands fragment-major (Wf: ceil(N/16) column tiles, zero-padded)
int launch_dec_logits(hipStream_t st, bool i8, const void* xf, const float* x_scale, const void* Wf,
                      const float* w_scale, const float* s1, const float* cf, float* out, int ldo, int R, int N, int K);
void launch_nospeech(hipStream_t st, const float* logits, int V, int row_mul, int no_speech_id, float* out, int B);
// sup_bits: the suppress list, one bit per token id, LP_SUP_WORDS 64-bit words (zero-padded past the vocabulary)
#define LP_SUP_WORDS 896
void launch_logits_process(hipStream_t st, const GenDev& gp, float* logits, const unsigned long long* sup_bits,
                           const int* hist2, const float* cum2, const int* d_step, const int* done, float* cand_val,
                           int* cand_tok);
void launch_beam_update(hipStream_t st, const GenDev& gp, const float* cand_val, const int* cand_tok, int* hist2,
                        float* cum2, uint8_t* kvidx2, int* cur_tok, const int* d_step, int* done, int* n_done,
                        int* n_fin, int* fin_tok, int* fin_len, float* fin_score, float* fin_cum);
void launch_step_advance(hipStream_t st, int* d_step);
// row b reads logits row b * row_mul
void launch_token_prob(hipStream_t st, const float* logits, int V, const int* target, float* out, int out_stride,
                       int out_off, int rows, int row_mul = 1);
void launch_cross_probs(hipStream_t st, const half_t* qx, int d, const half_t* ck, int T, int kvp, const int* heads,
                        int n_layer_heads, int n_sel, float* probs, int n_tok, int tok_idx, int B, int blk_n = 0);

}  // namespace fwd
