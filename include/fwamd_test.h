/*
 * fwamd_test.h — test and measurement hooks of libfwamd.so.  NOT part of the drop-in boundary (include/fwamd.h):
 * thin wrappers over single kernels on host buffers, so that tests/ can parity-test each kernel in isolation and
 * profiles/ can time one kernel outside the pipeline.  Nothing in faster_whisper_amd/ (the product) calls them.
 */
#ifndef FWAMD_TEST_H
#define FWAMD_TEST_H

#include "fwamd.h"

#ifdef __cplusplus
extern "C" {
#endif

int32_t fw_test_gemm(fw_model* m, const float* A, const float* W, const float* bias, const float* residual,
                     int32_t M, int32_t N, int32_t K, int32_t act_gelu, int32_t use_int8, float* out);
/* one decoder linear exactly as a decode step runs it (fragment-major operands, LayerNorm folded when ln_g/ln_b are
 * given, GELU when act = 1, residual added last): x [R][K], W [N][K], bias [N] | NULL, res [R][N] | NULL ->
 * out [R][N] (row-major result) and out_from_frag [R][N] (the fragment-major copy the next linear reads, un-permuted
 * on the host).  use_int8 = 0: what a decode step launches for this row count; 1: the int8_float16 form (needs an
 * int8_float16 model; ln must be NULL); 5: the register-streaming (skinny) kernel whatever the row count, 6 / 7: that kernel with one tile / 2 x 2 tiles per
 * workgroup; 10 + cfg: the GEMM-shaped kernel of
 * merged runs (dec_gemm_big_kernel) with workgroup shape cfg, whatever the row count.  0, 5, 6, 7, 10.. return the same bits. */
int32_t fw_test_dec_linear(fw_model* m, const float* x, const float* W, const float* bias, const float* ln_g,
                           const float* ln_b, const float* res, int32_t R, int32_t N, int32_t K, int32_t act,
                           int32_t use_int8, float* out, float* out_from_frag);
/* the vocabulary projection of a decode step: x [R][d] raw residual rows -> float32 logits [R][n_vocab] (final
 * LayerNorm folded in fp16 mode, applied by the row quantiser in int8_float16 mode), with the model's own weights */
int32_t fw_test_dec_logits(fw_model* m, const float* x, int32_t R, float* out);
/* one launch of the logits-rules kernel (suppress lists, repetition penalty, no-repeat n-gram, timestamp rules,
 * log-softmax, top-2K candidates of cum + logp, or the Gumbel arg-max when opts selects sampling) on caller-provided
 * logits [R][n_vocab] and row state: hist [R][n] = the n tokens generated so far on each row, cum [R].  Outputs
 * cand_val / cand_tok [R][2 * beam_size] ([R][1] when sampling).  Replaces nothing in the reference: it exposes the
 * device form of CTranslate2's logits processors to tests/test_gpu_logits_rules.py. */
int32_t fw_test_logits_rules(fw_model* m, const float* logits, int32_t R, const int32_t* hist, int32_t n,
                             const float* cum, const fw_gen_opts* opts, int32_t with_timestamps, float* cand_val,
                             int32_t* cand_tok);
/* measurement hook (profiles/gemm_bench.py): average milliseconds of one launch of the "many rows" GEMM
 * C[batch][M][N] = A[batch][M][K] W[N][K]^T on device-resident pseudo-random operands (fp16, or int8 on an
 * int8_float16 model); lda = K + a_pad, ldw = K + w_pad elements; trans: the transposed-output form */
int32_t fw_bench_gemm(fw_model* m, int32_t M, int32_t N, int32_t K, int32_t batch, int32_t a_pad, int32_t w_pad,
                      int32_t trans, int32_t iters, float* ms_out);
/* micro-benchmark of the decoder linear kernel (dec_gemm_frag_kernel) for a tile-shape `variant` (dec_kernels.hip:
 * launch_dec_gemm_frag_variant; 0 / 1 = the product's skinny kernel, 10 + cfg = the GEMM-shaped kernel of merged runs) over a rotating weight set larger than the caches:
 * mean microseconds per launch of a [R] x [N][K] linear (lnf = LayerNorm-folded form). */
int32_t fw_bench_dec_linear(fw_model* m, int32_t R, int32_t N, int32_t K, int32_t lnf, int32_t variant, int32_t iters,
                            float* us_out);
int32_t fw_test_layernorm(fw_model* m, const float* x, const float* g, const float* b,
                          int32_t rows, int32_t d, float* out);
int32_t fw_test_attention(fw_model* m, const float* q, const float* k, const float* v,
                          int32_t B, int32_t H, int32_t T, float* out);
/* measurement hook (profiles/attn_bench.py): mean milliseconds of one launch of the encoder self-attention kernel for
 * B chunks x H heads x T positions on device-resident pseudo-random operands; variant = the workgroup mapping
 * (0: XCD-aware, the product's; 1: query tile fastest over all XCDs, round 3's) */
int32_t fw_bench_attention(fw_model* m, int32_t B, int32_t H, int32_t T, int32_t variant, int32_t iters, float* ms_out);

/* rows from which a decode run's per-layer linears take the GEMM-shaped kernel (dec_kernels.hip: DEC_BIG_MIN_ROWS);
 * bench.py prices the decoder linears against the MFMA roof from this row count on, against HBM below */
int32_t fw_dec_big_min_rows(void);
/* the same per linear: role 0 qkv, 1 d x d (out / cross-q / cross-out), 2 ffn1, 3 ffn2 (< 0: the lowest of the four);
 * compute_type 0 float16, 1 int8_float16 (one row count for every linear).  Each linear switches at its own measured
 * crossover, so a run between the lowest and the highest has some linears on either kernel: bench.py prices per role */
int32_t fw_dec_big_min_rows_of(int32_t role, int32_t compute_type);
/* process-wide measurement knob for A/B runs inside one process.  id 1: encoder GEMM tile order (1 = blocked, the
 * product's; 0 = n fastest across the whole width, rounds 1-3).  id 2: decoder self-attention form (0 = by launch size,
 * the product's; 1 = the first form of rounds 1-4; 2 = latency form; 3 = throughput form — all four return the same bits).  id 4 (3 was the weight prefetch of
 * solo runs, measured slower twice and removed: profiles/r05_ab_wprefetch_*.jsonl): position blocks for the prompt forward and align (1, the default:
 * up to 16 positions per decoder pass; 0: one position per pass, rounds 1-4 — the same bits).  id 5: the plain transposed
 * GEMM epilogue, i.e. the encoder's V^T (1, the default: staged through LDS, whole row segments of Ct; 0: direct 8-byte stores,
 * rounds 1-4 — the same bits).  id 6: the cross-attention K / V^T projections (1, the default: all decoder layers in two
 * launches of the encoder GEMM; 0: two launches per layer, rounds 1-4 — the same bits).  id 7: register cap of the decoder
 * cross-attention kernel (0, the default: none, 110 registers; 1: 96; 2: 80 — the same bits; profiles/r06_ab_cross_regs.jsonl) */
int32_t fw_test_knob(int32_t id, int32_t value);
/* host-only (no device needed): chunks an IDLE two-lane decode group wants queued before it leads a run — its even share of
 * the work it knows of (`queued` chunks in `n_queued` requests + one request of that average size per worker inside an encode
 * call) over the runs that work needs (>= 2; `want` = chunks one run takes), at least one batch.  The rule the leader of a
 * run applies when no run is in progress (fw_model_set_merge_wait, include/fwamd.h). */
int64_t fw_test_idle_lead_chunks(int64_t queued, int32_t n_queued, int32_t encoding, int64_t want, int32_t max_batch);

#ifdef __cplusplus
}
#endif
#endif /* FWAMD_TEST_H */
