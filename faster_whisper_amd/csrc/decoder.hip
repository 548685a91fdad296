// Decoder side of libfwamd.so: cross-K/V projection, KV-cached decode loop with on-device
// logits rules + beam search (hipGraph-replayed step), language detection and
// cross-attention alignment.
//
// Reference interfaces replaced (faster_whisper/transcribe.py):
//   ctranslate2.models.Whisper.generate         :222-236, :1446-1459
//   ctranslate2.models.Whisper.detect_language  :215, :1193, :1823
//   ctranslate2.models.Whisper.align            :1709-1715
// The decoding rules restate CTranslate2 4.x / openai-whisper behaviour (SURVEY.md
// Appendix A); oracle/whisper.py is the CPU statement of the same rules.
//
// Decode groups.  A decode step is HBM-bound: it streams every decoder weight once (1.47 GB for large-v3)
// whatever the number of rows.  CTranslate2 runs `inter_threads` replicas side by side, each streaming the weights
// for its own batch; here the worker replicas of a device share ONE decode workspace sized for all of their
// batches (288 GB of HBM make that cheap), and fw_generate calls that arrive while a run is in progress are
// merged into the next run: up to decode_batch chunks x beam rows share each weight byte and each launch.  The
// encoders of the other workers overlap with the running decode on their own streams (MFMA-bound next to
// HBM-bound).  Results are independent of the merge: every kernel works per row / per chunk.
#include "engine.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <numeric>

#include "dec_kernels.h"
#include "kernels.h"

namespace fw {

using fwd::FIN_CAP;
using fwd::GenDev;

// rows of a merged decode run above which a second, nearly empty round of workgroups starts (dec_kernels.hip)
#define DEC_RUN_MAX_ROWS 1600

// Cross-attention K / V^T pool of a decode group: n_blocks blocks of EB (= max_batch) chunk slots, layer-major
// [L][n_blocks * EB][H][kvp * 64] (a layer's launch sees every slot of that layer behind one base pointer; a run reaches
// its chunks through a slot table).  A block holds the projected K / V^T of ONE encoder output; fw_generate /
// fw_detect_language / fw_align take the block that already holds their encoder output (a hit: generate after
// detect_language, align after generate, the temperature ladder re-decoding a window) or the least recently used free
// one.  A decode run takes all its blocks at once (no hold-and-wait between the lanes).
struct CrossPool {
  half_t *ck = nullptr, *cvt = nullptr;
  int n_blocks = 0, EB = 0, kvp = 0;
  struct Block { uint64_t enc_id = 0; int n = 0; bool busy = false; uint64_t stamp = 0; };
  std::vector<Block> blocks;
  std::mutex mu;
  std::condition_variable cv;
  uint64_t clock = 0;
  int n_slots() const { return n_blocks * EB; }
};

struct GraphSlot {
  hipGraphExec_t exec = nullptr;
  GenDev key;
  uint64_t stamp = 0;
  int forms = 0;    // fwd::kernel_forms_epoch() at capture: a measurement knob that changes which kernels a step
                    // launches (fw_test_knob) must not be answered with a graph captured before it was set
};

struct GenWorkspace {
  int B = 0, K = 0, R = 0, NT = 0;       // capacity of a run: chunks, beams per chunk, rows = B * K
  int EB = 0;                            // chunks of one encoder output (the models' max_batch)
  int* slot_map = nullptr;               // [R] chunk of the run -> chunk slot of the cross-attention pool (device)
  // self-attention cache: ONE allocation of L x self_cap x d halves per K and V, self_cap = R x NTs row-positions.
  // Its geometry is the RUN's: [L][rows of the run][H][ctx of the run][64] with rows x ctx <= self_cap, so a run
  // whose calls ask for max_length 104 holds 4.3x the rows of one that asks for the whole text context.
  half_t *sk = nullptr, *sv = nullptr;
  int NTs = 0;                           // positions per row at full row capacity
  int64_t self_cap = 0;                  // rows x positions
  half_t *x = nullptr, *qkv = nullptr, *att = nullptr, *qc = nullptr, *ffn = nullptr;
  float* logits = nullptr;               // [R][V]
  int* prompt_dev = nullptr;             // [NT][max(R, B)]
  int* prompt_blk = nullptr;             // the same tokens in position-block order: [block][chunk][position of the block]
  int* cur_tok = nullptr;                // [R]
  int* hist2 = nullptr;                  // [2][R][NT]
  float* cum2 = nullptr;                 // [2][R]
  uint8_t* kvidx2 = nullptr;             // [2][R][NT]
  float* cand_val = nullptr;             // [R][32]
  int* cand_tok = nullptr;
  int *done = nullptr, *n_done = nullptr, *n_fin = nullptr, *fin_tok = nullptr, *fin_len = nullptr;
  float *fin_score = nullptr, *fin_cum = nullptr;
  int* d_step = nullptr;
  float* no_speech = nullptr;
  unsigned long long* sup_bits = nullptr;   // suppress list, one bit per token id
  int* zero_done = nullptr;              // [R] zeros (kernels that take a `done` pointer outside generate)
  half_t *x_frag = nullptr, *att_frag = nullptr, *ffn_frag = nullptr;   // fragment-major GEMM inputs (fp16)
  half_t* xn_frag = nullptr;             // fp16(LayerNorm(x)), fragment-major (explicit-LayerNorm order, Model::ln_unfold)
  int8_t* xq = nullptr;                  // int8_float16: quantised linear input, fragment-major [R16][4d]
  float* xs = nullptr;                   //               per-row de-quantisation scale [R]
  int8_t* ekq = nullptr;                 // int8_float16: one encoder output quantised for the cross-K/V projection
  float* eks = nullptr;
  // hipGraphs of the decode step, one per distinct GenDev (merged runs differ in their chunk count)
  std::vector<GraphSlot> graphs;
  uint64_t graph_clock = 0;
  bool graphs_enabled = true;
};

static std::atomic<uint64_t> g_tensor_id{1};
uint64_t next_tensor_id() { return g_tensor_id.fetch_add(1); }

// positions per cache row at full row capacity: the whole text context when one encoder batch is all the workspace
// holds, otherwise what set_decode_batch chose (>= the share that lets ONE full-context encoder batch run)
static int self_positions(const Model* m, int decode_batch, int nts) {
  const int NT = m->cfg.n_text_ctx;
  const int need = (int)(((int64_t)NT * m->max_batch + decode_batch - 1) / decode_batch);   // EB*K*NT <= B*K*NTs
  return std::min(NT, std::max(std::max(nts, need), 8));
}

int64_t gen_workspace_bytes(const Model* m, int lane_chunks, int nts) {
  const fw_config& c = m->cfg;
  const int64_t B = lane_chunks, R = B * m->max_beam, d = c.d_model, L = c.n_dec_layers, NT = c.n_text_ctx;
  const int64_t NTs = self_positions(m, lane_chunks, nts > 0 ? nts : (int)NT);
  int64_t n = 2 * (L * R * NTs * d) * 2;                               // self K/V (fp16)
  n += R * (int64_t)c.n_vocab * 4 + R * 12 * d * 2 * 2;                // logits, activations
  n += R * FIN_CAP * NT * 4 + 3 * R * NT * 4;                          // finished hypotheses, histories
  return n + (64 << 20);
}

int64_t cross_pool_bytes(const Model* m, int pool_chunks) {
  const fw_config& c = m->cfg;
  const int64_t kvp = ((c.n_audio_ctx + 31) / 32) * 32;
  return 2 * ((int64_t)c.n_dec_layers * pool_chunks * c.d_model * kvp) * 2;   // K and V^T, fp16
}

// the pool of the group m decodes for: m's own (primary) or the primary's (a lane)
static CrossPool* pool_of(Model* m) { return (m->pool_owner ? m->pool_owner : m)->xpool; }

static int cross_pool_ensure(Model* m) {
  Model* owner = m->pool_owner ? m->pool_owner : m;
  if (owner->xpool) return FW_OK;
  const fw_config& c = owner->cfg;
  CrossPool* p = new CrossPool();
  p->EB = owner->max_batch;
  p->n_blocks = std::max(1, std::max(owner->decode_batch, owner->max_batch) / owner->max_batch);
  p->kvp = ((c.n_audio_ctx + 31) / 32) * 32;
  p->blocks.assign(p->n_blocks, CrossPool::Block());
  const size_t n = (size_t)c.n_dec_layers * p->n_slots() * c.d_model * p->kvp;
  int rc;
  if ((rc = dev_alloc_t(&p->ck, n)) || (rc = dev_alloc_t(&p->cvt, n))) {
    if (p->ck) (void)hipFree(p->ck);
    delete p;
    return rc;
  }
  // the padded keys (>= T) are never written: K garbage is masked, V^T must be 0 (0 * NaN)
  hipError_t he = hipMemset(p->ck, 0, n * sizeof(half_t));
  if (he == hipSuccess) he = hipMemset(p->cvt, 0, n * sizeof(half_t));
  if (he == hipSuccess) he = hipDeviceSynchronize();
  if (he != hipSuccess) {
    (void)hipFree(p->ck); (void)hipFree(p->cvt);
    delete p;
    set_error("cross-attention pool setup failed: %s", hipGetErrorString(he));
    return FW_ENODEV;
  }
  owner->xpool = p;
  return FW_OK;
}

void cross_pool_free(Model* m) {
  CrossPool* p = m->xpool;
  if (!p) return;
  if (p->ck) (void)hipFree(p->ck);
  if (p->cvt) (void)hipFree(p->cvt);
  delete p;
  m->xpool = nullptr;
}

// Blocks for the encoder outputs `ids` (chunk counts `ns`), ALL of them or none: waits until enough blocks are free.
// blk[i] = block index, hit[i] = the block already holds that encoder output's K / V^T.
static void pool_acquire(CrossPool* p, const std::vector<uint64_t>& ids, const std::vector<int>& ns, std::vector<int>& blk,
                         std::vector<char>& hit) {
  std::unique_lock<std::mutex> lk(p->mu);
  const int n = (int)ids.size();
  blk.assign(n, -1);
  hit.assign(n, 0);
  for (;;) {
    int n_free = 0;
    for (const CrossPool::Block& b : p->blocks) n_free += b.busy ? 0 : 1;
    if (n_free >= n) break;
    p->cv.wait(lk);
  }
  for (int i = 0; i < n; ++i)          // first the hits ...
    for (int k = 0; k < p->n_blocks; ++k) {
      CrossPool::Block& b = p->blocks[k];
      if (!b.busy && b.enc_id == ids[i] && b.n == ns[i] && ids[i] != 0) { b.busy = true; blk[i] = k; hit[i] = 1; break; }
    }
  for (int i = 0; i < n; ++i) {        // ... then the least recently used free blocks
    if (blk[i] >= 0) continue;
    int best = -1;
    for (int k = 0; k < p->n_blocks; ++k)
      if (!p->blocks[k].busy && (best < 0 || p->blocks[k].stamp < p->blocks[best].stamp)) best = k;
    CrossPool::Block& b = p->blocks[best];
    b.busy = true; b.enc_id = 0; b.n = 0;   // marked empty until the projection has been launched
    blk[i] = best;
  }
  for (int i = 0; i < n; ++i) p->blocks[blk[i]].stamp = ++p->clock;
}

static void pool_release(CrossPool* p, const std::vector<int>& blk) {
  {
    std::lock_guard<std::mutex> lk(p->mu);
    for (int k : blk)
      if (k >= 0) p->blocks[k].busy = false;
  }
  p->cv.notify_all();
}

struct PoolHold {   // releases its blocks when the decode run / detect_language / align call ends, however it ends
  CrossPool* p;
  std::vector<int> blk;
  hipStream_t st = nullptr;   // the stream the blocks' projections / readers were launched on
  // A block carries its encoder output's id from the moment the projection is LAUNCHED; the other lane (another stream)
  // may take it as a hit, or recycle it, as soon as it is released.  Every normal return path has synchronised the stream
  // already (this wait then costs nothing); an error return taken earlier must not free blocks that kernels of this
  // stream may still be writing.
  ~PoolHold() {
    if (!p) return;
    if (st && !blk.empty()) (void)hipStreamSynchronize(st);
    pool_release(p, blk);
  }
};

static int gen_workspace_build(Model* m) {
  const fw_config& c = m->cfg;
  GenWorkspace* g = new GenWorkspace();
  m->gen = g;   // gen_workspace_ensure frees it again when anything below fails
  if (m->decode_batch < m->max_batch) m->decode_batch = m->max_batch;
  g->B = lane_chunks_of(m);
  g->EB = m->max_batch;
  g->K = m->max_beam;
  g->R = g->B * g->K;
  g->NT = c.n_text_ctx;
  g->NTs = self_positions(m, g->B, m->decode_self_ctx > 0 ? m->decode_self_ctx : c.n_text_ctx);
  g->self_cap = (int64_t)g->R * g->NTs;
  const size_t B = g->B, R = g->R, d = c.d_model, L = c.n_dec_layers, NT = g->NT;
  (void)B;
  FW_CHECK_ARG(R <= 2048, "decode_batch * max_beam must be <= 2048 (got %zu)", R);
  FW_HIP(hipSetDevice(m->device));
  if (!m->dec_stream) FW_HIP(create_stream(&m->dec_stream, "DEC"));
  int rc;
#define A(p, n) do { if ((rc = dev_alloc_t(&(p), (n)))) return rc; } while (0)
  A(g->slot_map, R);
  A(g->sk, L * (size_t)g->self_cap * d);
  A(g->sv, L * (size_t)g->self_cap * d);
  A(g->x, R * d); A(g->qkv, R * 3 * d); A(g->att, R * d); A(g->qc, R * d);
  A(g->ffn, R * 4 * d);
  A(g->logits, R * c.n_vocab);
  A(g->prompt_dev, NT * R); A(g->prompt_blk, NT * R);
  A(g->cur_tok, R);
  A(g->hist2, 2 * R * NT);
  A(g->cum2, 2 * R);
  A(g->kvidx2, 2 * R * NT);
  A(g->cand_val, R * 32);
  A(g->cand_tok, R * 32);
  // per-chunk state is sized by ROWS: random sampling runs every hypothesis as its own beam-1 chunk
  A(g->done, R); A(g->n_done, 1); A(g->n_fin, R);
  A(g->fin_tok, R * FIN_CAP * NT); A(g->fin_len, R * FIN_CAP);
  A(g->fin_score, R * FIN_CAP); A(g->fin_cum, R * FIN_CAP);
  A(g->d_step, 1);
  A(g->no_speech, R);
  A(g->sup_bits, (size_t)LP_SUP_WORDS);
  A(g->zero_done, R);
  const size_t R16 = (R + 15) / 16 * 16;   // whole 16-row tiles
  if (m->compute_type == FW_COMPUTE_INT8_FLOAT16) {
    A(g->xq, R16 * 4 * d);
    A(g->xs, R16);
    FW_HIP(hipMemset(g->xq, 0, R16 * 4 * d));
    FW_HIP(hipMemset(g->xs, 0, R16 * sizeof(float)));
    A(g->ekq, (size_t)g->EB * c.n_audio_ctx * d);
    A(g->eks, (size_t)g->EB * c.n_audio_ctx);
  } else {
    A(g->x_frag, R16 * d); A(g->att_frag, R16 * d); A(g->ffn_frag, R16 * 4 * d); A(g->xn_frag, R16 * d);
    FW_HIP(hipMemset(g->xn_frag, 0, R16 * d * sizeof(half_t)));
    FW_HIP(hipMemset(g->x_frag, 0, R16 * d * sizeof(half_t)));
    FW_HIP(hipMemset(g->att_frag, 0, R16 * d * sizeof(half_t)));
    FW_HIP(hipMemset(g->ffn_frag, 0, R16 * 4 * d * sizeof(half_t)));
  }
#undef A
  FW_HIP(hipMemset(g->slot_map, 0, R * sizeof(int)));
  FW_HIP(hipMemset(g->zero_done, 0, R * sizeof(int)));
  FW_HIP(hipMemset(g->d_step, 0, sizeof(int)));
  FW_HIP(hipDeviceSynchronize());
  const char* ng = getenv("FWAMD_NO_GRAPH");   // rocprofv3 7.2 crashes on replayed graphs: profile eagerly
  g->graphs_enabled = !(ng && ng[0] == '1');
  return FW_OK;
}

// A workspace is installed whole or not at all: after a failed build (argument check, out of HBM) the next call
// starts from scratch instead of launching kernels on a half-allocated one.
int gen_workspace_ensure(Model* m) {
  int rc = cross_pool_ensure(m);
  if (rc) return rc;
  if (m->gen) return FW_OK;
  rc = gen_workspace_build(m);
  if (rc) gen_workspace_free(m);
  return rc;
}

void gen_workspace_free(Model* m) {
  GenWorkspace* g = m->gen;
  if (!g) return;
  for (GraphSlot& s : g->graphs)
    if (s.exec) (void)hipGraphExecDestroy(s.exec);
  void* ptrs[] = {g->slot_map, g->sk, g->sv, g->x, g->qkv, g->att, g->qc, g->ffn, g->logits, g->prompt_dev,
                  g->cur_tok, g->hist2, g->cum2, g->kvidx2, g->cand_val, g->cand_tok, g->done, g->n_done, g->n_fin,
                  g->fin_tok, g->fin_len, g->fin_score, g->fin_cum, g->d_step, g->no_speech, g->sup_bits,
                  g->zero_done, g->xq, g->xs, g->ekq, g->eks, g->x_frag, g->att_frag, g->ffn_frag, g->xn_frag, g->prompt_blk};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete g;
  m->gen = nullptr;
}

// K11: cross-attention K / V^T of every decoder layer for the chunks of one encoder output, written to the chunk
// slots of pool block `blk` (held by the caller; once per encoder output as long as the block is not recycled)
// cross-K/V projections of ALL decoder layers as two launches (K, V^T) instead of two per layer (fp16; knob 6 /
// FWAMD_CROSS_KV_LAYERED=0: per layer, rounds 1-4).  The 64 per-layer GEMMs are 24 000 x 1 280 x 1 280 each: 480 tiles =
// 1.875 rounds of 256 workgroups (the last round 7/8 full) with a prologue / epilogue ramp per launch; as one launch the K
// (or V^T) projection of a 16-chunk batch is 15 360 tiles = 60 full rounds.  Same tiles, same arithmetic: the same bits.
static std::atomic<int> g_cross_kv_layered{-1};
static bool cross_kv_layered() {
  int v = g_cross_kv_layered.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("FWAMD_CROSS_KV_LAYERED");
    v = (e && atoi(e) == 0) ? 0 : 1;
    g_cross_kv_layered.store(v);
  }
  return v != 0;
}
void set_cross_kv_layered(int on) { g_cross_kv_layered.store(on ? 1 : 0); }
// element strides between consecutive decoder layers' cross.kv weights / biases when they are uniform (they are for a blob
// packed by pack_blob: every layer holds the same items at the same sizes), else 0
static bool cross_kv_strides(const Model* m, int64_t* ws, int64_t* bs) {
  const int L = m->cfg.n_dec_layers;
  if (L < 2) return false;
  const int64_t w1 = m->dec[1].ck.w - m->dec[0].ck.w, b1 = m->dec[1].ck.b - m->dec[0].ck.b;
  for (int l = 1; l < L; ++l) {
    if (m->dec[l].ck.w - m->dec[0].ck.w != l * w1 || m->dec[l].cv.w - m->dec[0].cv.w != l * w1) return false;
    if (m->dec[l].ck.b - m->dec[0].ck.b != l * b1 || m->dec[l].cv.b - m->dec[0].cv.b != l * b1) return false;
  }
  *ws = w1; *bs = b1;
  return w1 > 0;
}

static int ensure_cross_kv(Model* m, const Tensor* enc, int blk, bool hit) {
  if (hit) return FW_OK;
  GenWorkspace* g = m->gen;
  CrossPool* pool = pool_of(m);
  const int B = enc->B;
  const int b0 = blk * pool->EB;
  // (a failure half way leaves the block marked empty — pool_acquire did that — never pointing at partly overwritten K/V)
  const fw_config& c = m->cfg;
  const int d = c.d_model, T = c.n_audio_ctx;
  const int64_t xs = (int64_t)T * d;
  hipStream_t st = m->dec_stream;
  int rc;
  ProfScope ps(m, PF_CROSS_KV_GEMM, 2.0 * c.n_dec_layers * B * (double)T * d * (2.0 * d), 0, st);
  const bool i8 = m->compute_type == FW_COMPUTE_INT8_FLOAT16;
  const int kvp = pool->kvp;
  const int64_t kvs = (int64_t)d * kvp;   // one chunk's K (or V^T): H heads x kvp keys x 64
  int64_t ws = 0, bs = 0;
  if (!i8 && cross_kv_layered() && cross_kv_strides(m, &ws, &bs)) {
    const int64_t cls = (int64_t)pool->n_slots() * kvs;          // a layer's region of the pool
    half_t* kd = pool->ck + (size_t)b0 * kvs;
    half_t* vd = pool->cvt + (size_t)b0 * kvs;
    if ((rc = run_linear_layers(m, m->dec[0].ck, c.n_dec_layers, ws, bs, enc->data, d, xs, kd, d, kvs, cls, T, B, false, kvp, st)))
      return rc;
    if ((rc = run_linear_layers(m, m->dec[0].cv, c.n_dec_layers, ws, bs, enc->data, d, xs, vd, kvp, kvs, cls, T, B, true, kvp, st)))
      return rc;
    std::lock_guard<std::mutex> lk(pool->mu);
    pool->blocks[blk].enc_id = enc->id;
    pool->blocks[blk].n = B;
    return FW_OK;
  }
  for (int l = 0; l < c.n_dec_layers; ++l) {
    const DecLayerW& L = m->dec[l];
    half_t* kd = pool->ck + ((size_t)l * pool->n_slots() + b0) * kvs;
    half_t* vd = pool->cvt + ((size_t)l * pool->n_slots() + b0) * kvs;
    // both land in the MFMA-fragment-major layout of the decode kernel (head_rows = kvp selects it in the
    // GEMM epilogue): K through the plain epilogue, V^T through the transposed one
    if (i8) {
      // the encoder output is quantised once (layer 0) and shared by all 2L projections
      if ((rc = run_linear_i8(m, L.ck, l == 0 ? enc->data : nullptr, nullptr, kd, d, kvs, nullptr, 0, 0, T, B, 0,
                              false, kvp, st, g->ekq, g->eks)))
        return rc;
      if ((rc = run_linear_i8(m, L.cv, nullptr, nullptr, vd, kvp, kvs, nullptr, 0, 0, T, B, 0, true, kvp, st, g->ekq,
                              g->eks)))
        return rc;
      continue;
    }
    if ((rc = run_linear(m, L.ck, enc->data, d, xs, kd, d, kvs, nullptr, 0, 0, T, B, 0, false, kvp, st))) return rc;
    if ((rc = run_linear(m, L.cv, enc->data, d, xs, vd, kvp, kvs, nullptr, 0, 0, T, B, 0, true, kvp, st))) return rc;
  }
  {
    std::lock_guard<std::mutex> lk(pool->mu);
    pool->blocks[blk].enc_id = enc->id;
    pool->blocks[blk].n = B;
  }
  return FW_OK;
}

struct StepCfg {
  int rows, kmul;      // rows processed, queries per chunk (1 for prefill, K for beam steps)
  int B;
  int pos_fixed;       // >= 0: explicit position (prefill / detect / align); -1: P-1+*d_step
  int P;
  const int* tok;      // [rows] device
  bool need_logits;
  int nospeech_rowmul; // > 0: run the no-speech kernel on rows b*rowmul after the logits GEMM
  bool beam_tail;      // logits rules + beam update + step advance
  const int* done;     // per-chunk done flags for cross-attn early exit
  int kv_slot0 = 0;    // align: first pool slot of the call's block (its chunks are contiguous there)
  // POSITION BLOCK (prompt forward, align): blk_n > 0 — rows = B * blk_n, row = chunk * blk_n + j is position
  // pos_fixed + j of the chunk's beam slot 0, kmul = blk_n.  One pass over the decoder for blk_n positions: the weights,
  // and the chunk's cross-attention K / V^T (the dominant HBM stream), are read once for all of them.
  int blk_n = 0;
  int nospeech_off = 0; // row of the chunk's block whose logits feed the no-speech kernel
  // align extras
  const int* sel_heads_dev = nullptr;   // [n_sel_total] head ids, grouped per layer
  const int* sel_layer_off = nullptr;   // host: [L+1] offsets into sel_heads
  float* probs = nullptr; int n_sel_total = 0, n_tok = 0, tok_idx = 0;
};

#define DG(call)                                                        \
  do {                                                                  \
    if ((call) != 0) {                                                  \
      set_error("decoder gemm: unsupported shape (%s)", #call);        \
      return FW_ERUNTIME;                                               \
    }                                                                   \
  } while (0)

// position blocks for the prompt forward and align (knob 4 / FWAMD_POS_BLOCKS, default ON; 0: one position per pass,
// rounds 1-4): the same bits either way (tests/test_gpu_model.py::test_position_blocks_same_bits)
static std::atomic<int> g_pos_blocks{-1};
static bool pos_blocks_on() {
  int v = g_pos_blocks.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("FWAMD_POS_BLOCKS");
    v = (e && atoi(e) == 0) ? 0 : 1;
    g_pos_blocks.store(v);
  }
  return v != 0;
}
void set_pos_blocks(int on) { g_pos_blocks.store(on ? 1 : 0); }
// positions per block for a pass over `chunks` chunks: the workspace holds g->R rows, the cross-attention kernel takes
// <= 16 queries per chunk
static int pos_block_size(const Model* m, const GenWorkspace* g, const GenDev& gp, int chunks) {
  if (!pos_blocks_on() || chunks < 1) return 1;
  if (!fwd::self_attn_block_ok(g->NT, gp.ctx, m->cfg.d_model, gp.R)) return 1;
  return std::max(1, std::min(16, g->R / chunks));
}

// One decoder forward over `rows` rows.  Everything that varies from step to step lives in HBM (d_step, beam
// tables), and everything a captured graph bakes in by value is part of GenDev (the graph key).
static int run_step(Model* m, const GenDev& gp, const StepCfg& s) {
  GenWorkspace* g = m->gen;
  const fw_config& c = m->cfg;
  const int d = c.d_model, H = c.n_heads, T = c.n_audio_ctx, NT = g->NT;
  CrossPool* pool = pool_of(m);
  const int kvp = pool->kvp;
  hipStream_t st = m->dec_stream;
  const int rows = s.rows;
  const bool i8 = m->compute_type == FW_COMPUTE_INT8_FLOAT16;
  {
    ProfScope ps(m, PF_DEC_MISC, 0, 0, st);
    fwd::launch_embed(st, s.tok, m->tok_emb, m->dec_pos, g->x, i8 ? nullptr : g->x_frag, rows, d, g->d_step,
                      s.pos_fixed, s.P, s.blk_n);
  }
  // int8_float16 (K25): the row quantiser (fused with the LayerNorm where one feeds the linear) writes the int8
  // rows fragment-major, then the int8 skinny GEMM de-quantises in its epilogue
  auto lin_q = [&](const half_t* xin, const LNW* ln, const LinearW& L, const half_t* res, half_t* outp, int act) -> int {
    fwk::launch_quant_rows(st, xin, L.K, ln ? ln->g : nullptr, ln ? ln->b : nullptr, g->xq, g->xs, rows, L.K, 1);
    return fwd::launch_dec_gemm_frag_i8(st, g->xq, g->xs, L.wq, L.wscale, L.b, res, L.N, outp, L.N, rows, L.N, L.K, act);
  };
  // fp16: xin is the fragment-major copy of the input; the residual stream is kept row-major too (residual
  // adds), the FFN hidden only fragment-major; LayerNorms are folded into qkv / cross-q / ffn1
  auto lin_f = [&](const half_t* xin_frag, const LinearW& L, const half_t* res, half_t* outp, half_t* outp_frag,
                   int act) -> int {
    return fwd::launch_dec_gemm_frag(st, xin_frag, L.w, L.b, L.s1, L.cf, res, L.N, outp, L.N, outp_frag, rows, L.N,
                                     L.K, act);
  };
  const int frag = i8 ? 0 : 1;
  // fp16, explicit-LayerNorm order (Model::ln_unfold == 2): the LayerNorm is its own kernel, its fp16 output (fragment-
  // major) feeds the plain weight — the rounding points of an fp16 LayerNorm followed by an fp16 GEMM
  const bool unf = !i8 && m->ln_unfold >= 2;
  auto lin_u = [&](const LNW& ln, const LinearW& L, half_t* outp, half_t* outp_frag, int act) -> int {
    fwk::launch_layernorm(st, g->x, ln.g, ln.b, g->xn_frag, rows, d, 1);
    return fwd::launch_dec_gemm_frag(st, g->xn_frag, L.w, L.b, nullptr, nullptr, nullptr, L.N, outp, L.N, outp_frag, rows,
                                     L.N, L.K, act);
  };
  for (int l = 0; l < c.n_dec_layers; ++l) {
    const DecLayerW& L = m->dec[l];
    // the run's own cache geometry: [layer][cache_rows slots][H][ctx][64]
    half_t* kc = g->sk + (size_t)l * gp.cache_rows * gp.ctx * d;
    half_t* vc = g->sv + (size_t)l * gp.cache_rows * gp.ctx * d;
    const half_t* ck = pool->ck + (size_t)l * pool->n_slots() * d * kvp;
    const half_t* cvt = pool->cvt + (size_t)l * pool->n_slots() * d * kvp;
    {
      ProfScope ps(m, PF_DEC_GEMM_QKV, 2.0 * rows * 3.0 * d * d, 2.0 * 3.0 * d * d, st);
      if (i8) DG(lin_q(g->x, &L.ln1, L.qkv, nullptr, g->qkv, 0));
      else if (unf) DG(lin_u(L.ln1, L.qkv_p, g->qkv, nullptr, 0));
      else DG(lin_f(g->x_frag, L.qkv, nullptr, g->qkv, nullptr, 0));
    }
    {
      ProfScope ps(m, PF_DEC_SELF_ATTN, 0, 0, st);
      fwd::launch_self_attn(st, g->qkv, d, kc, vc, NT, gp.ctx, H, g->kvidx2, gp.K, s.kmul, frag ? g->att_frag : g->att, rows,
                            g->d_step, s.pos_fixed, s.P, gp.R, frag, s.blk_n);
    }
    {
      ProfScope ps(m, PF_DEC_GEMM_DXD, 2.0 * rows * 2.0 * d * d, 2.0 * 2.0 * d * d, st);
      if (i8) {
        DG(lin_q(g->att, nullptr, L.out, g->x, g->x, 0));
        DG(lin_q(g->x, &L.ln2, L.cq, nullptr, g->qc, 0));
      } else {
        DG(lin_f(g->att_frag, L.out, g->x, g->x, g->x_frag, 0));
        if (unf) DG(lin_u(L.ln2, L.cq_p, g->qc, nullptr, 0));
        else DG(lin_f(g->x_frag, L.cq, nullptr, g->qc, nullptr, 0));
      }
    }
    if (s.probs && s.sel_layer_off[l + 1] > s.sel_layer_off[l]) {
      ProfScope ps(m, PF_DEC_MISC, 0, 0, st);
      const int off = s.sel_layer_off[l], n = s.sel_layer_off[l + 1] - off;
      fwd::launch_cross_probs(st, g->qc, d, ck + (size_t)s.kv_slot0 * d * kvp, T, kvp, s.sel_heads_dev + off, n, s.n_sel_total,
                              s.probs + (size_t)off * s.n_tok * T, s.n_tok, s.tok_idx, s.B, s.blk_n);
    }
    {
      ProfScope ps(m, PF_DEC_CROSS_ATTN, 4.0 * rows * (double)T * d, 4.0 * (s.B / gp.kv_div) * (double)T * d, st);
      fwd::launch_cross_attn(st, g->qc, d, ck, cvt, T, kvp, s.kmul, frag ? g->att_frag : g->att, s.B, H, s.done,
                             gp.kv_div, frag, g->slot_map);
    }
    {
      ProfScope ps(m, PF_DEC_GEMM_DXD, 2.0 * rows * 1.0 * d * d, 2.0 * 1.0 * d * d, st);
      if (i8) DG(lin_q(g->att, nullptr, L.cout, g->x, g->x, 0));
      else DG(lin_f(g->att_frag, L.cout, g->x, g->x, g->x_frag, 0));
    }
    {
      ProfScope ps(m, PF_DEC_GEMM_FFN1, 2.0 * rows * 4.0 * d * d, 2.0 * 4.0 * d * d, st);
      if (i8) DG(lin_q(g->x, &L.ln3, L.ffn1, nullptr, g->ffn, 1));
      else if (unf) DG(lin_u(L.ln3, L.ffn1_p, nullptr, g->ffn_frag, 1));
      else DG(lin_f(g->x_frag, L.ffn1, nullptr, nullptr, g->ffn_frag, 1));
    }
    {
      ProfScope ps(m, PF_DEC_GEMM_FFN2, 2.0 * rows * 4.0 * d * d, 2.0 * 4.0 * d * d, st);
      if (i8) DG(lin_q(g->ffn, nullptr, L.ffn2, g->x, g->x, 0));
      else DG(lin_f(g->ffn_frag, L.ffn2, g->x, g->x, g->x_frag, 0));
    }
  }
  if (s.need_logits || s.beam_tail) {
    ProfScope ps(m, PF_DEC_LOGITS, 2.0 * rows * (double)c.n_vocab * d, 2.0 * c.n_vocab * d, st);
    if (i8) {
      fwk::launch_quant_rows(st, g->x, d, m->dec_ln.g, m->dec_ln.b, g->xq, g->xs, rows, d, 1);
      DG(fwd::launch_dec_logits(st, true, g->xq, g->xs, m->logits.wq, m->logits.wscale, nullptr, nullptr, g->logits,
                                c.n_vocab, rows, c.n_vocab, d));
    } else if (m->ln_unfold >= 1) {
      fwk::launch_layernorm(st, g->x, m->dec_ln.g, m->dec_ln.b, g->xn_frag, rows, d, 1);
      DG(fwd::launch_dec_logits(st, false, g->xn_frag, nullptr, m->logits_p.w, nullptr, nullptr, nullptr, g->logits,
                                c.n_vocab, rows, c.n_vocab, d));
    } else {
      DG(fwd::launch_dec_logits(st, false, g->x_frag, nullptr, m->logits.w, nullptr, m->logits.s1, m->logits.cf,
                                g->logits, c.n_vocab, rows, c.n_vocab, d));
    }
  }
  if (s.nospeech_rowmul > 0) {
    ProfScope ps(m, PF_DEC_MISC, 0, 0, st);
    fwd::launch_nospeech(st, g->logits + (size_t)s.nospeech_off * c.n_vocab, c.n_vocab, s.nospeech_rowmul,
                         c.tok_no_speech, g->no_speech, s.B);
  }
  if (s.beam_tail) {
    ProfScope ps(m, PF_DEC_SAMPLE, 0, 8.0 * rows * c.n_vocab, st);
    fwd::launch_logits_process(st, gp, g->logits, g->sup_bits, g->hist2, g->cum2, g->d_step, g->done, g->cand_val,
                               g->cand_tok);
    fwd::launch_beam_update(st, gp, g->cand_val, g->cand_tok, g->hist2, g->cum2, g->kvidx2, g->cur_tok, g->d_step,
                            g->done, g->n_done, g->n_fin, g->fin_tok, g->fin_len, g->fin_score, g->fin_cum);
    fwd::launch_step_advance(st, g->d_step);
  }
  return FW_OK;
}

// generated-token budget: the reference treats max_length as prompt + new tokens
// (transcribe.py:193-207). [CT2-ext] single place that decides; mirrors oracle.whisper.max_new_tokens.
static int max_new_tokens(int max_length, int P) { return std::max(0, max_length - P); }
// positions a run can reach (prompt + budget), rounded up to 8: the per-slot extent of its self-attention cache
static int run_ctx(const fw_config& c, int max_length, int P) {
  const int reach = P + max_new_tokens(std::min(max_length, c.n_text_ctx), P);
  return std::min(c.n_text_ctx, (std::max(reach, 1) + 7) / 8 * 8);
}

static int check_launch(const char* what) {
  hipError_t he = hipGetLastError();
  if (he != hipSuccess) {
    set_error("%s: kernel launch failed: %s", what, hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  return FW_OK;
}

// ---- one fw_generate call ------------------------------------------------------------------------------------
struct GenRequest {
  const Tensor* enc;
  const int32_t* prompts;
  const int32_t* prompt_offsets;
  int B, P;
  const fw_gen_opts* o;
  int32_t* out_ids; int32_t* out_lens; float* out_scores; float* out_no_speech;
  bool sampling;
  int with_ts = 1, sot_pos = -1;   // from the prompt: no <|notimestamps|> -> timestamp rules; position of <sot>
  int rc = FW_OK;
  std::string err;
  bool taken = false;   // part of a run that is being decoded (its caller waits for `done`, it must not lead another run)
  bool done = false;
};

// calls that may share a decode run: same scalar options, same prompt length, same suppress list (beam / greedy
// only: the hypotheses of a sampling call are rows of their own and run alone)
static bool mergeable(const GenRequest& a, const GenRequest& b) {
  if (a.sampling || b.sampling) return false;
  const fw_gen_opts &x = *a.o, &y = *b.o;
  if (a.P != b.P || x.beam_size != y.beam_size || x.patience != y.patience || x.num_hypotheses != y.num_hypotheses ||
      x.length_penalty != y.length_penalty || x.repetition_penalty != y.repetition_penalty ||
      x.no_repeat_ngram_size != y.no_repeat_ngram_size || x.max_length != y.max_length ||
      x.max_initial_timestamp_index != y.max_initial_timestamp_index || x.suppress_blank != y.suppress_blank ||
      x.min_new_tokens != y.min_new_tokens || x.n_suppress_tokens != y.n_suppress_tokens)
    return false;
  if (x.n_suppress_tokens && memcmp(x.suppress_tokens, y.suppress_tokens, x.n_suppress_tokens * sizeof(int32_t)))
    return false;
  // the timestamp rules are switched by <|notimestamps|> in the prompt, the <sot> position selects the no-speech
  // row: both must agree; the prompts themselves may differ (language token per chunk, transcribe.py:212-220)
  return a.with_ts == b.with_ts && a.sot_pos == b.sot_pos;
}

// chunks of a run led by `r` that fit the self-attention cache (before the workspace exists: its planned size)
// share (percent) of a run's capacity an IDLE decode group (no run in progress) waits for; FWAMD_IDLE_FILL_PCT overrides
static int idle_fill_pct() {
  static const int v = [] {
    const char* e = getenv("FWAMD_IDLE_FILL_PCT");
    const int x = e ? atoi(e) : 100;
    return x < 1 ? 1 : (x > 100 ? 100 : x);
  }();
  return v;
}
// an idle two-lane group splits the known work evenly over its runs (see the gather loop); FWAMD_IDLE_BALANCE=0 turns it
// off (the A/B of profiles/r05_ab_idle_balance.jsonl: burst 2 904 -> 2 976x, steady state 3 105 -> 3 105x)
static bool idle_balance() {
  static const bool v = [] { const char* e = getenv("FWAMD_IDLE_BALANCE"); return !e || atoi(e) != 0; }();
  return v;
}
// chunks an IDLE two-lane group wants queued before it leads a run: its even share of the work it knows of — `queued`
// chunks in `n_queued` requests plus one request of that average size per worker inside an encode call — over the runs
// that work needs (at least two: one per lane; more when it exceeds two runs' capacity `want`), never less than one batch
// (round 6: splitting the known work of an idle group into at least THREE runs instead of two — the first run starting after a
//  third of a burst's encoder passes — measured +-0.4 % on the 20-batch burst and on the steady state: profiles/r06_ab_idle_runs.jsonl)
int64_t idle_lead_chunks(int64_t queued, int n_queued, int encoding, int64_t want, int max_batch, int lanes) {
  const int64_t per_req = std::max<int64_t>(1, queued / std::max<int64_t>(1, (int64_t)n_queued));
  const int64_t outstanding = queued + (int64_t)std::max(0, encoding) * per_req;
  const int64_t n_runs = std::max<int64_t>(std::max(2, lanes), (outstanding + std::max<int64_t>(1, want) - 1) / std::max<int64_t>(1, want));
  return std::max<int64_t>(max_batch, (outstanding + n_runs - 1) / n_runs);
}
static int64_t planned_self_cap(const Model* dm) {   // rows x positions of a lane's self-attention cache
  const int B = lane_chunks_of(dm);
  const int nts = self_positions(dm, B, dm->decode_self_ctx > 0 ? dm->decode_self_ctx : dm->cfg.n_text_ctx);
  return (int64_t)B * dm->max_beam * nts;
}
static int64_t self_chunk_capacity(Model* dm, const GenRequest& r) {
  // rows of one encoder chunk in the run: beam_size, or num_hypotheses beam-1 rows when sampling (kv_div)
  const int64_t rows_per_chunk = r.sampling ? std::max(1, r.o->num_hypotheses) : r.o->beam_size;
  return planned_self_cap(dm) / (rows_per_chunk * run_ctx(dm->cfg, r.o->max_length, r.P));
}

// Decode run over the concatenated chunks of `reqs` (all mergeable with reqs[0]).  Caller holds m->dec_mu.
static int generate_run(Model* m, const std::vector<GenRequest*>& reqs) {
  int rc = gen_workspace_ensure(m);
  if (rc) return rc;
  GenWorkspace* g = m->gen;
  const fw_config& c = m->cfg;
  const GenRequest& r0 = *reqs[0];
  const fw_gen_opts* o = r0.o;
  const int K = o->beam_size, P = r0.P;
  const bool sampling = r0.sampling;
  const int nh = o->num_hypotheses, ml = o->max_length;
  const int kv_div = sampling ? nh : 1;   // decode chunks per encoder chunk
  int B = 0;
  for (const GenRequest* r : reqs) B += r->B;
  const int Bx = B * kv_div;
  FW_CHECK_ARG(Bx * K <= g->R && Bx <= g->R && B <= g->B, "decode run of %d chunks x %d rows exceeds the workspace", B,
               kv_div * K);
  const int budget = max_new_tokens(std::min(o->max_length, c.n_text_ctx), P);
  const int ctx = run_ctx(c, o->max_length, P);
  FW_CHECK_ARG((int64_t)Bx * K * ctx <= g->self_cap, "decode run of %d rows x %d positions exceeds the self-attention cache", Bx * K, ctx);
  FW_HIP(hipSetDevice(m->device));
  hipStream_t st = m->dec_stream;
  // ---- the run's blocks of the cross-attention pool (all at once), the projection of those that are not there yet,
  //      and the run's chunk -> pool slot table ----
  CrossPool* pool = pool_of(m);
  PoolHold hold{pool, {}, st};
  std::vector<int> slot_host((size_t)Bx);
  {
    std::vector<uint64_t> ids;
    std::vector<int> ns;
    std::vector<char> hit;
    for (const GenRequest* r : reqs) { ids.push_back(r->enc->id); ns.push_back(r->enc->B); }
    FW_CHECK_ARG((int)reqs.size() <= pool->n_blocks, "decode run of %zu calls exceeds the %d blocks of the cross-attention pool",
                 reqs.size(), pool->n_blocks);
    pool_acquire(pool, ids, ns, hold.blk, hit);
    int bx = 0;
    for (size_t i = 0; i < reqs.size(); ++i) {
      if ((rc = ensure_cross_kv(m, reqs[i]->enc, hold.blk[i], hit[i] != 0))) return rc;
      for (int b = 0; b < reqs[i]->B; ++b)
        for (int j = 0; j < kv_div; ++j) slot_host[bx++] = hold.blk[i] * pool->EB + b;
    }
  }
  GenDev gp;
  memset(&gp, 0, sizeof(gp));
  gp.B = Bx; gp.K = K; gp.R = Bx * K; gp.P = P; gp.budget = budget; gp.kv_div = kv_div;
  gp.ctx = ctx; gp.cache_rows = Bx * K;
  gp.sample = sampling ? 1 : 0;
  gp.inv_temp = sampling ? 1.0f / o->sampling_temperature : 1.0f;
  gp.seed_lo = (unsigned)(o->seed & 0xffffffffu);
  gp.seed_hi = (unsigned)(o->seed >> 32);
  gp.max_fin = std::max(1, (int)lroundf((float)K * o->patience));
  if (gp.max_fin > FIN_CAP - K) gp.max_fin = FIN_CAP - K;
  gp.V = c.n_vocab; gp.n_text_ctx = g->NT;
  const int sot_pos = r0.sot_pos;
  gp.with_ts = r0.with_ts;
  gp.suppress_blank = o->suppress_blank ? 1 : 0;
  gp.min_new = o->min_new_tokens;
  gp.mits = o->max_initial_timestamp_index;
  gp.ngram = o->no_repeat_ngram_size;
  gp.rep_pen = o->repetition_penalty;
  gp.lp_pow = o->length_penalty;
  gp.eot = c.tok_eot; gp.no_ts = c.tok_no_timestamps; gp.ts_begin = c.tok_timestamp_begin;
  gp.n_sup_begin = c.n_suppress_begin;
  for (int i = 0; i < c.n_suppress_begin; ++i) gp.sup_begin[i] = c.suppress_begin[i];
  bool want_nsp = false;
  for (const GenRequest* r : reqs) want_nsp = want_nsp || r->o->return_no_speech_prob;

  // ---- state init (only the rows of this run) ----
  // (the ping-pong halves of hist2 / kvidx2 / cum2 are gp.R rows apart: the kernels index them with the run's R)
  const size_t NT = g->NT, Rr = (size_t)gp.R;
  FW_HIP(hipMemsetAsync(g->hist2, 0, 2 * Rr * NT * sizeof(int), st));
  FW_HIP(hipMemsetAsync(g->kvidx2, 0, 2 * Rr * NT, st));
  FW_HIP(hipMemsetAsync(g->cum2, 0, 2 * Rr * sizeof(float), st));
  FW_HIP(hipMemsetAsync(g->done, 0, Rr * sizeof(int), st));
  FW_HIP(hipMemsetAsync(g->n_done, 0, sizeof(int), st));
  FW_HIP(hipMemsetAsync(g->n_fin, 0, Rr * sizeof(int), st));
  FW_HIP(hipMemsetAsync(g->d_step, 0, sizeof(int), st));
  FW_HIP(hipMemsetAsync(g->no_speech, 0, Rr * sizeof(float), st));
  std::vector<unsigned long long> mask(LP_SUP_WORDS, 0ull);
  for (int i = 0; i < o->n_suppress_tokens; ++i) {
    const int t = o->suppress_tokens[i];
    if (t >= 0 && t < c.n_vocab) mask[t >> 6] |= 1ull << (t & 63);
  }
  FW_HIP(hipMemcpyAsync(g->sup_bits, mask.data(), mask.size() * sizeof(mask[0]), hipMemcpyHostToDevice, st));
  // prompt tokens transposed to [pos][bx]; first-step tokens replicated per beam
  std::vector<int> ptok((size_t)P * Bx), first((size_t)Bx * K);
  {
    int bx = 0;
    for (const GenRequest* r : reqs)
      for (int b = 0; b < r->B; ++b)
        for (int j = 0; j < kv_div; ++j, ++bx) {
          const int32_t* pr = r->prompts + r->prompt_offsets[b];
          for (int p = 0; p < P; ++p) ptok[(size_t)p * Bx + bx] = pr[p];
          for (int k = 0; k < K; ++k) first[(size_t)bx * K + k] = pr[P - 1];
        }
  }
  FW_HIP(hipMemcpyAsync(g->prompt_dev, ptok.data(), ptok.size() * sizeof(int), hipMemcpyHostToDevice, st));
  FW_HIP(hipMemcpyAsync(g->cur_tok, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice, st));
  // (the cross-attention kernel divides the decode chunk by kv_div before it looks the slot up: one entry per ENCODER chunk)
  {
    std::vector<int> enc_slots((size_t)B);
    for (int b = 0; b < B; ++b) enc_slots[b] = slot_host[(size_t)b * kv_div];
    slot_host.swap(enc_slots);
  }
  FW_HIP(hipMemcpyAsync(g->slot_map, slot_host.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, st));
  FW_HIP(hipStreamSynchronize(st));   // the host buffers above are locals

  // ---- prompt forward (all but the last token): Bx rows, beam slot 0 of every chunk ----
  //      in blocks of up to 16 positions per pass (one pass for the 3 positions of the usual 4-token prompt; a prompt that
  //      carries 223 tokens of previous text takes 14 passes instead of 223)
  {
    const int nbmax = pos_block_size(m, g, gp, Bx);
    std::vector<int> blk_tok;
    if (nbmax > 1 && P - 1 > 1) {
      blk_tok.reserve((size_t)(P - 1) * Bx);
      for (int pos = 0; pos < P - 1;) {
        const int nb = std::min(nbmax, P - 1 - pos);
        for (int b = 0; b < Bx; ++b)
          for (int j = 0; j < nb; ++j) blk_tok.push_back(ptok[(size_t)(pos + j) * Bx + b]);
        pos += nb;
      }
      FW_HIP(hipMemcpyAsync(g->prompt_blk, blk_tok.data(), blk_tok.size() * sizeof(int), hipMemcpyHostToDevice, st));
      FW_HIP(hipStreamSynchronize(st));
    }
    for (int pos = 0; pos < P - 1;) {
      const int nb = blk_tok.empty() ? 1 : std::min(nbmax, P - 1 - pos);
      StepCfg s;
      s.B = Bx; s.pos_fixed = pos; s.P = P;
      s.need_logits = sot_pos >= pos && sot_pos < pos + nb && want_nsp;
      s.beam_tail = false;
      s.done = g->done;
      if (nb > 1) {
        s.rows = Bx * nb; s.kmul = nb; s.blk_n = nb; s.tok = g->prompt_blk + (size_t)pos * Bx;
        s.nospeech_rowmul = s.need_logits ? nb : 0;
        s.nospeech_off = s.need_logits ? sot_pos - pos : 0;
      } else {
        s.rows = Bx; s.kmul = 1; s.tok = blk_tok.empty() ? g->prompt_dev + (size_t)pos * Bx : g->prompt_blk + (size_t)pos * Bx;
        s.nospeech_rowmul = s.need_logits ? 1 : 0;
      }
      if ((rc = run_step(m, gp, s))) return rc;
      pos += nb;
    }
  }
  if ((rc = check_launch("prompt forward"))) return rc;

  int steps_done = 0;
  if (budget > 0) {
    // ---- step 0 (eager): last prompt token on all rows; beams are identical copies ----
    StepCfg s;
    s.rows = Bx * K; s.kmul = K; s.B = Bx; s.pos_fixed = -1; s.P = P; s.tok = g->cur_tok;
    s.need_logits = true;
    s.nospeech_rowmul = (sot_pos == P - 1 && want_nsp) ? K : 0;
    s.beam_tail = true;
    s.done = g->done;
    if ((rc = run_step(m, gp, s))) return rc;
    steps_done = 1;
    s.nospeech_rowmul = 0;

    // ---- steps 1.. : one hipGraph replay per step; graphs are cached per GenDev (everything the captured
    //      kernels take by value, kv_div included) ----
    hipGraphExec_t exec = nullptr;
    if (g->graphs_enabled && !m->prof_on) {
      for (GraphSlot& gs : g->graphs)
        if (gs.exec && gs.forms == fwd::kernel_forms_epoch() && memcmp(&gs.key, &gp, sizeof(gp)) == 0) {
          exec = gs.exec; gs.stamp = ++g->graph_clock;
        }
      if (!exec) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          const int rc2 = run_step(m, gp, s);
          const hipError_t e2 = hipStreamEndCapture(st, &graph);
          hipGraphExec_t ne = nullptr;
          if (rc2 == FW_OK && e2 == hipSuccess && graph && hipGraphInstantiate(&ne, graph, nullptr, nullptr, 0) == hipSuccess) {
            GraphSlot* slot = nullptr;
            // (merged runs come in every multiple of a batch up to the lane's capacity: 20 sizes with the bench's workers;
            //  a cache of 12 re-captured and re-instantiated 261-node graphs all the time)
            if (g->graphs.size() < 40) { g->graphs.emplace_back(); slot = &g->graphs.back(); }
            else {   // evict the least recently used
              slot = &g->graphs[0];
              for (GraphSlot& gs : g->graphs) if (gs.stamp < slot->stamp) slot = &gs;
              if (slot->exec) (void)hipGraphExecDestroy(slot->exec);
            }
            slot->exec = ne; slot->key = gp; slot->stamp = ++g->graph_clock; slot->forms = fwd::kernel_forms_epoch();
            exec = ne;
          }
          if (graph) (void)hipGraphDestroy(graph);
        }
        (void)hipGetLastError();
      }
    }
    int n_done_host = 0;
    while (steps_done < budget) {
      const int burst = std::min(4, budget - steps_done);
      for (int i = 0; i < burst; ++i) {
        if (exec) {
          hipError_t he = hipGraphLaunch(exec, st);
          if (he != hipSuccess) {
            set_error("hipGraphLaunch failed: %s", hipGetErrorString(he));
            return FW_ERUNTIME;
          }
        } else if ((rc = run_step(m, gp, s))) {
          return rc;
        }
      }
      steps_done += burst;
      FW_HIP(hipMemcpyAsync(&n_done_host, g->n_done, sizeof(int), hipMemcpyDeviceToHost, st));
      FW_HIP(hipStreamSynchronize(st));
      if (n_done_host >= Bx) break;
    }
  }
  FW_HIP(hipStreamSynchronize(st));
  if ((rc = check_launch("decode loop"))) return rc;
  prof_collect(m);

  // ---- finalize on the host: best num_hypotheses by normalised score (stable) ----
  std::vector<int> n_fin(Bx), fin_len((size_t)Bx * FIN_CAP);
  std::vector<float> fin_score((size_t)Bx * FIN_CAP), nsp(Bx);
  std::vector<int> fin_tok((size_t)Bx * FIN_CAP * NT);
  // (on the decode stream, not the legacy default stream: a default-stream copy would wait for every blocking
  //  stream of the device)
  FW_HIP(hipMemcpyAsync(n_fin.data(), g->n_fin, Bx * sizeof(int), hipMemcpyDeviceToHost, st));
  FW_HIP(hipMemcpyAsync(fin_len.data(), g->fin_len, fin_len.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  FW_HIP(hipMemcpyAsync(fin_score.data(), g->fin_score, fin_score.size() * sizeof(float), hipMemcpyDeviceToHost, st));
  FW_HIP(hipMemcpyAsync(fin_tok.data(), g->fin_tok, fin_tok.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  FW_HIP(hipMemcpyAsync(nsp.data(), g->no_speech, Bx * sizeof(float), hipMemcpyDeviceToHost, st));
  FW_HIP(hipStreamSynchronize(st));
  int cb = 0;   // first chunk of the request inside the run
  for (GenRequest* r : reqs) {
    for (int b = 0; b < r->B; ++b) {
      const int gb = cb + b;
      if (r->o->return_no_speech_prob) r->out_no_speech[b] = nsp[(size_t)gb * kv_div];
      // candidate hypotheses of encoder chunk gb: (decode chunk, finished index)
      std::vector<std::pair<int, int>> hyps;
      for (int j = 0; j < kv_div; ++j) {
        const int bx = gb * kv_div + j;
        for (int f = 0; f < n_fin[bx]; ++f) hyps.push_back({bx, f});
      }
      std::stable_sort(hyps.begin(), hyps.end(), [&](const std::pair<int, int>& a, const std::pair<int, int>& bb) {
        return fin_score[(size_t)a.first * FIN_CAP + a.second] > fin_score[(size_t)bb.first * FIN_CAP + bb.second];
      });
      for (int h = 0; h < nh && h < (int)hyps.size(); ++h) {
        const size_t f = (size_t)hyps[h].first * FIN_CAP + hyps[h].second;
        int len = fin_len[f];
        if (len > ml) len = ml;
        r->out_lens[b * nh + h] = len;
        if (r->o->return_scores) r->out_scores[b * nh + h] = fin_score[f];
        memcpy(r->out_ids + ((size_t)b * nh + h) * ml, &fin_tok[f * NT], (size_t)len * sizeof(int));
      }
    }
    cb += r->B;
  }
  return FW_OK;
}

}  // namespace fw

using namespace fw;

extern "C" {

int32_t fw_generate(fw_model* fm, const fw_tensor* enc_t, const int32_t* prompts, const int32_t* prompt_offsets,
                    int32_t B, const fw_gen_opts* o, int32_t* out_ids, int32_t* out_lens, float* out_scores,
                    float* out_no_speech) {
  FW_CHECK_ARG(fm && enc_t && prompts && prompt_offsets && o && out_ids && out_lens && out_scores && out_no_speech,
               "null argument");
  Model* m = &fm->impl;
  Model* dm = decoder_of(m);
  const Tensor* enc = &enc_t->impl;
  const fw_config& c = m->cfg;
  FW_CHECK_ARG(enc->owner && decoder_of(enc->owner) == dm, "encoder output belongs to a different model");
  FW_CHECK_ARG(B == enc->B, "batch %d does not match the encoder output batch %d", B, enc->B);
  FW_CHECK_ARG(B >= 1 && B <= dm->max_batch, "batch %d exceeds max_batch %d", B, dm->max_batch);
  const int K = o->beam_size;
  FW_CHECK_ARG(K >= 1 && K <= dm->max_beam, "beam_size %d not in [1, max_beam=%d]", K, dm->max_beam);
  // random sampling (the sequential path's temperature fallback, transcribe.py:1433-1439): beam_size 1,
  // sampling_topk 0 (whole distribution), num_hypotheses = best_of independent samples per chunk
  const bool sampling = (K == 1 && o->sampling_topk != 1);
  if (sampling) {
    FW_CHECK_ARG(o->sampling_topk == 0, "sampling_topk must be 1 (greedy) or 0 (sample the whole distribution)");
    FW_CHECK_ARG(o->sampling_temperature > 0.f, "sampling_temperature must be positive");
    FW_CHECK_ARG(o->num_hypotheses >= 1, "num_hypotheses must be positive");
  } else {
    FW_CHECK_ARG(o->num_hypotheses >= 1 && o->num_hypotheses <= std::max(K, 1),
                 "num_hypotheses %d must be in [1, beam_size]", o->num_hypotheses);
  }
  FW_CHECK_ARG(o->patience > 0.f, "patience must be positive");
  FW_CHECK_ARG(o->max_length >= 1, "max_length must be positive");
  const int P = prompt_offsets[1] - prompt_offsets[0];
  FW_CHECK_ARG(P >= 1, "prompts must not be empty");
  for (int b = 0; b < B; ++b)
    FW_CHECK_ARG(prompt_offsets[b + 1] - prompt_offsets[b] == P,
                 "all prompts of a generate() call must have the same length (prompt %d has %d tokens, expected %d)",
                 b, prompt_offsets[b + 1] - prompt_offsets[b], P);
  FW_CHECK_ARG(P <= c.n_text_ctx, "prompt length %d exceeds the text context %d", P, c.n_text_ctx);
  for (int i = 0; i < B * P; ++i)
    FW_CHECK_ARG(prompts[prompt_offsets[0] + i] >= 0 && prompts[prompt_offsets[0] + i] < c.n_vocab,
                 "prompt token %d out of range", prompts[prompt_offsets[0] + i]);
  const int nh = o->num_hypotheses;
  for (int i = 0; i < B * nh; ++i) { out_lens[i] = 0; out_scores[i] = 0.f; }
  for (int b = 0; b < B; ++b) out_no_speech[b] = 0.f;

  GenRequest req;
  req.enc = enc; req.prompts = prompts; req.prompt_offsets = prompt_offsets; req.B = B; req.P = P; req.o = o;
  req.out_ids = out_ids; req.out_lens = out_lens; req.out_scores = out_scores; req.out_no_speech = out_no_speech;
  req.sampling = sampling;
  for (int b = 0; b < B; ++b) {
    // every prompt of the call must agree on what switches the decoding rules (true for every call site of the
    // reference: one prompt per batch, language token swapped in place)
    int wt = 1, sp = -1;
    for (int i = 0; i < P; ++i) {
      const int t = prompts[prompt_offsets[b] + i];
      if (t == c.tok_no_timestamps) wt = 0;
      if (t == c.tok_sot) sp = i;
    }
    if (b == 0) { req.with_ts = wt; req.sot_pos = sp; }
    FW_CHECK_ARG(wt == req.with_ts && sp == req.sot_pos,
                 "the prompts of a generate() call must agree on <|notimestamps|> and on the <|startoftranscript|> position");
  }

  DecodeGroup& grp = dm->grp;
  std::unique_lock<std::mutex> lk(grp.mu);
  // fw_model_set_decode_batch rebuilds the workspaces and the second lane under grp.resizing: nothing is read from
  // them (capacities, dm->lane1) and nothing is queued until it is done
  while (grp.resizing) grp.cv.wait(lk);
  {
    // this call alone must fit a lane (rows and self-attention cache), so that a call that cannot is refused here,
    // before it is merged with others whose run it would fail
    const int64_t rows = sampling ? (int64_t)B * o->num_hypotheses : (int64_t)B * K;
    const int64_t row_cap = (int64_t)lane_chunks_of(dm) * dm->max_beam;
    const int ctx = run_ctx(c, o->max_length, P);
    if (rows > row_cap || rows * ctx > planned_self_cap(dm)) {
      lk.unlock();
      set_error("this call needs %lld decoder rows x %d positions; a decode run of this model holds %lld rows and %lld "
                "row-positions (batch x num_hypotheses / beam_size too large for max_length %d)",
                (long long)rows, ctx, (long long)row_cap, (long long)planned_self_cap(dm), o->max_length);
      return FW_EINVAL;
    }
  }
  grp.queue.push_back(&req);
  grp.last_arrival = std::chrono::steady_clock::now();
  while (!req.done) {
    const int n_lanes = n_lanes_of(dm);        // (read under grp.mu: stable while a request is queued, see above)
    if (req.taken || grp.gathering || grp.active_runs >= std::min(n_lanes, grp.lanes_enabled.load())) {
      grp.cv.wait(lk);
      continue;
    }
    // lead the next run: wait for the requests of workers that are still encoding (each arrives within one
    // encoder pass) unless most of a run's capacity is already claimed, then take every queued request that can share
    // a run with the oldest one.  The wait trades this caller's latency for rows per run; it is bounded by the
    // merge-wait knob (fw_model_set_merge_wait: default 2.5 measured encoder passes after the last arrival, at most
    // 250 ms — a pass next to a decode run takes up to twice the pass the average was taken from, and a leader that
    // gives up between two arrivals starts a run of one or two batches: the driver's 20-step burst 2 976 -> 3 030x with
    // 150 or 250 ms, the steady state unchanged, profiles/r05_ab_idle_balance.jsonl; 0 = never) and skipped when no
    // member encode is in flight.  One caller gathers at a time; with two lanes
    // the next one starts gathering as soon as this one has taken its requests and gone off to run them.
    grp.gathering = true;
    const int cap = lane_chunks_of(dm);
    int wait_ms = grp.merge_wait_ms.load();
    if (wait_ms < 0) wait_ms = std::min(250, std::max(5, (grp.enc_pass_us.load() * 5 / 2 + 999) / 1000));
    const int fill_pct = grp.merge_fill_pct.load();
    if (cap > dm->max_batch && wait_ms > 0 && !grp.queue.front()->sampling) {
      for (;;) {
        int queued = 0;
        for (const GenRequest* r : grp.queue) queued += r->B;
        const int64_t want = std::min<int64_t>(cap, std::max<int64_t>(dm->max_batch, DEC_RUN_MAX_ROWS / std::max(1, grp.queue.front()->o->beam_size)));
        // no run in progress at all: every further request waited for is decode time the chip spends idle on the HBM
        // side (the encoders are MFMA-bound), so an idle group leads with a smaller share of a run's capacity
        // (idle_fill_pct(): the driver's 20-step burst was ONE run of 288 chunks after 18 serial encoder passes plus two
        // runs of 16 chunks; with 50 % it is two runs of 160 chunks, the second gathered under the first)
        const int pct = grp.active_runs == 0 ? std::min(fill_pct, idle_fill_pct()) : fill_pct;
        if ((int64_t)queued * 100 >= want * pct || grp.encoding.load() <= 0) break;
        // ... and with two lanes it splits the work it KNOWS of evenly over the runs that work needs: what is queued plus
        // one request per worker that is inside an encode call (running or waiting for the encoder; counted at the size of
        // the queued requests).  20 batches in flight -> two runs of 10, not one of 18 and leftovers.
        if (idle_balance() && grp.active_runs == 0 && n_lanes >= 2 && grp.lanes_enabled.load() >= 2 &&
            queued >= idle_lead_chunks(queued, (int)grp.queue.size(), grp.encoding.load(), want, dm->max_batch,
                                       std::min(n_lanes, grp.lanes_enabled.load())))
          break;
        if (std::chrono::steady_clock::now() - grp.last_arrival > std::chrono::milliseconds(wait_ms)) break;
        grp.cv.wait_for(lk, std::chrono::microseconds(200));
      }
    }
    std::vector<GenRequest*> batch;
    GenRequest* first = grp.queue.front();
    int chunks = 0;
    // chunks the self-attention cache holds at this call's context (mergeable calls share max_length and P)
    // ... and the rows above which the 4 x 4-tile decoder linears no longer fit the chip in one round of workgroups
    // (d x d: 20 column groups x 25 row groups of 64 rows = 500 of 512 workgroup slots)
    int64_t run_cap = std::min<int64_t>(cap, std::max<int64_t>(first->B, self_chunk_capacity(dm, *first)));
    if (!first->sampling)
      run_cap = std::min<int64_t>(run_cap, std::max<int64_t>(dm->max_batch, DEC_RUN_MAX_ROWS / std::max(1, first->o->beam_size)));
    // (one pool block per call: a run never takes more calls than the pool has blocks)
    const int max_calls = std::max(1, std::max(dm->decode_batch, dm->max_batch) / dm->max_batch);
    for (auto it = grp.queue.begin(); it != grp.queue.end();) {
      GenRequest* r = *it;
      if (r == first || (mergeable(*first, *r) && chunks + r->B <= run_cap && (int)batch.size() < max_calls)) {
        batch.push_back(r);
        r->taken = true;
        chunks += r->B;
        it = grp.queue.erase(it);
      } else {
        ++it;
      }
    }
    int lane = 0;
    while (lane + 1 < n_lanes && grp.lane_busy[lane]) ++lane;    // the first free lane (active_runs < lanes: there is one)
    grp.lane_busy[lane] = true;
    grp.active_runs += 1;
    grp.gathering = false;
    grp.cv.notify_all();                    // (the next leader may gather while this run decodes)
    lk.unlock();
    grp.n_runs.fetch_add(1);
    grp.n_requests.fetch_add((int64_t)batch.size());
    grp.n_chunks.fetch_add(chunks);
    if (chunks > grp.max_run_chunks.load()) grp.max_run_chunks.store(chunks);
    int rc;
    // FWAMD_RUN_LOG=1: one stderr line per decode run (start / end on the host clock, lane, calls, chunks) — the timeline of a
    // burst (profiles/r06_run_log_burst.txt)
    static const bool run_log = [] { const char* e = getenv("FWAMD_RUN_LOG"); return e && atoi(e) != 0; }();
    const auto t_run0 = std::chrono::steady_clock::now();
    {
      Model* lm = lane_model(dm, lane);     // the lane's model: own workspace, own stream, same weights
      std::lock_guard<std::mutex> dl(lm->dec_mu);
      rc = generate_run(lm, batch);
    }
    if (run_log) {
      const auto t_run1 = std::chrono::steady_clock::now();
      const double s0 = std::chrono::duration<double>(t_run0.time_since_epoch()).count();
      const double s1 = std::chrono::duration<double>(t_run1.time_since_epoch()).count();
      fprintf(stderr, "[fwamd run] lane %d calls %d chunks %d start %.4f end %.4f (%.1f ms)\n", lane, (int)batch.size(), chunks,
              s0, s1, 1e3 * (s1 - s0));
    }
    const std::string err = rc ? fw_last_error() : "";
    lk.lock();
    for (GenRequest* r : batch) { r->rc = rc; r->err = err; r->done = true; }
    grp.lane_busy[lane] = false;
    grp.active_runs -= 1;
    grp.cv.notify_all();
  }
  lk.unlock();
  if (req.rc) set_error("%s", req.err.c_str());
  return req.rc;
}

// ---------------------------------------------------------------------------------------------------
// Test hook: ONE launch of the logits-rules kernel (rules, log-softmax, top-2K / Gumbel arg-max) on caller-provided
// logits and row state, outside any decode run: tests/test_gpu_logits_rules.py compares the candidates with the
// oracle's rule restatement id for id.  rows = R (row r belongs to chunk r / beam_size), n = tokens generated so far
// (the same for every row), hist [R][n], cum [R]; out: cand_val / cand_tok [R][2 * beam_size] (sampling: [R][1]).
// ---------------------------------------------------------------------------------------------------
int32_t fw_test_logits_rules(fw_model* fm, const float* logits, int32_t R, const int32_t* hist, int32_t n,
                             const float* cum, const fw_gen_opts* o, int32_t with_timestamps, float* cand_val,
                             int32_t* cand_tok) {
  FW_CHECK_ARG(fm && logits && cum && o && cand_val && cand_tok && R >= 1 && n >= 0, "bad arguments");
  Model* m = &fm->impl;
  const fw_config& c = m->cfg;
  const int K = o->beam_size;
  const bool sampling = K == 1 && o->sampling_topk != 1;
  FW_CHECK_ARG(K >= 1 && K <= 16 && R % K == 0 && n < c.n_text_ctx && (n == 0 || hist), "bad geometry");
  FW_CHECK_ARG(c.n_vocab <= LP_SUP_WORDS * 64, "vocabulary too large for the rules kernel");
  FW_HIP(hipSetDevice(m->device));
  GenDev gp;
  memset(&gp, 0, sizeof(gp));
  gp.B = R / K; gp.K = K; gp.R = R; gp.V = c.n_vocab; gp.n_text_ctx = c.n_text_ctx;
  gp.sample = sampling ? 1 : 0;
  gp.inv_temp = sampling ? 1.0f / o->sampling_temperature : 1.0f;
  gp.seed_lo = (unsigned)(o->seed & 0xffffffffu);
  gp.seed_hi = (unsigned)(o->seed >> 32);
  gp.with_ts = with_timestamps ? 1 : 0;
  gp.suppress_blank = o->suppress_blank ? 1 : 0;
  gp.min_new = o->min_new_tokens;
  gp.mits = o->max_initial_timestamp_index;
  gp.ngram = o->no_repeat_ngram_size;
  gp.rep_pen = o->repetition_penalty;
  gp.eot = c.tok_eot; gp.no_ts = c.tok_no_timestamps; gp.ts_begin = c.tok_timestamp_begin;
  gp.n_sup_begin = c.n_suppress_begin;
  for (int i = 0; i < c.n_suppress_begin; ++i) gp.sup_begin[i] = c.suppress_begin[i];
  std::vector<unsigned long long> mask(LP_SUP_WORDS, 0ull);
  for (int i = 0; i < o->n_suppress_tokens; ++i) {
    const int t = o->suppress_tokens[i];
    if (t >= 0 && t < c.n_vocab) mask[t >> 6] |= 1ull << (t & 63);
  }
  const size_t NT = (size_t)c.n_text_ctx;
  std::vector<int> h2(2 * (size_t)R * NT, 0);
  std::vector<float> c2(2 * (size_t)R, 0.f);
  const int cur = n & 1;   // the kernel reads the half selected by the step's parity
  for (int r = 0; r < R; ++r) {
    for (int i = 0; i < n; ++i) h2[((size_t)cur * R + r) * NT + i] = hist[(size_t)r * n + i];
    c2[(size_t)cur * R + r] = cum[r];
  }
  float *d_lg = nullptr, *d_cum = nullptr, *d_cv = nullptr;
  int *d_hist = nullptr, *d_step = nullptr, *d_done = nullptr, *d_ct = nullptr;
  unsigned long long* d_bits = nullptr;
  int rc = FW_OK;
  auto cleanup = [&]() {
    (void)hipFree(d_lg); (void)hipFree(d_cum); (void)hipFree(d_cv); (void)hipFree(d_hist); (void)hipFree(d_step);
    (void)hipFree(d_done); (void)hipFree(d_ct); (void)hipFree(d_bits);
  };
#define TH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { set_error("%s: %s", #call, hipGetErrorString(e_)); cleanup(); return FW_ENODEV; } } while (0)
  const size_t lg_bytes = (size_t)R * c.n_vocab * sizeof(float);
  TH(hipMalloc(&d_lg, lg_bytes));
  TH(hipMalloc(&d_cum, c2.size() * sizeof(float)));
  TH(hipMalloc(&d_cv, (size_t)R * 32 * sizeof(float)));
  TH(hipMalloc(&d_ct, (size_t)R * 32 * sizeof(int)));
  TH(hipMalloc(&d_hist, h2.size() * sizeof(int)));
  TH(hipMalloc(&d_step, sizeof(int)));
  TH(hipMalloc(&d_done, (size_t)R * sizeof(int)));
  TH(hipMalloc(&d_bits, mask.size() * sizeof(mask[0])));
  TH(hipMemcpy(d_lg, logits, lg_bytes, hipMemcpyHostToDevice));
  TH(hipMemcpy(d_cum, c2.data(), c2.size() * sizeof(float), hipMemcpyHostToDevice));
  TH(hipMemcpy(d_hist, h2.data(), h2.size() * sizeof(int), hipMemcpyHostToDevice));
  TH(hipMemcpy(d_step, &n, sizeof(int), hipMemcpyHostToDevice));
  TH(hipMemset(d_done, 0, (size_t)R * sizeof(int)));
  TH(hipMemset(d_cv, 0, (size_t)R * 32 * sizeof(float)));
  TH(hipMemset(d_ct, 0, (size_t)R * 32 * sizeof(int)));
  TH(hipMemcpy(d_bits, mask.data(), mask.size() * sizeof(mask[0]), hipMemcpyHostToDevice));
  fwd::launch_logits_process(nullptr, gp, d_lg, d_bits, d_hist, d_cum, d_step, d_done, d_cv, d_ct);
  TH(hipGetLastError());
  TH(hipDeviceSynchronize());
  const int C = sampling ? 1 : 2 * K;
  std::vector<float> hv((size_t)R * 32);
  std::vector<int> ht((size_t)R * 32);
  TH(hipMemcpy(hv.data(), d_cv, hv.size() * sizeof(float), hipMemcpyDeviceToHost));
  TH(hipMemcpy(ht.data(), d_ct, ht.size() * sizeof(int), hipMemcpyDeviceToHost));
#undef TH
  for (int r = 0; r < R; ++r)
    for (int j = 0; j < C; ++j) {
      cand_val[(size_t)r * C + j] = hv[(size_t)r * 32 + j];
      cand_tok[(size_t)r * C + j] = ht[(size_t)r * 32 + j];
    }
  cleanup();
  return rc;
}

int32_t fw_detect_language(fw_model* fm, const fw_tensor* enc_t, int32_t B, int32_t* out_lang_ids,
                           float* out_probs) {
  FW_CHECK_ARG(fm && enc_t && out_lang_ids && out_probs, "null argument");
  Model* m = decoder_of(&fm->impl);
  const Tensor* enc = &enc_t->impl;
  const fw_config& c = m->cfg;
  FW_CHECK_ARG(c.is_multilingual && c.n_langs > 0, "detect_language needs a multilingual model");
  FW_CHECK_ARG(enc->owner && decoder_of(enc->owner) == m && B == enc->B && B <= m->max_batch,
               "bad encoder output / batch");
  std::lock_guard<std::mutex> lk(m->dec_mu);
  FW_HIP(hipSetDevice(m->device));
  int rc;
  if ((rc = gen_workspace_ensure(m))) return rc;
  GenWorkspace* g = m->gen;
  hipStream_t st = m->dec_stream;
  CrossPool* pool = pool_of(m);
  PoolHold hold{pool, {}, st};
  std::vector<int> slots(B);
  {
    std::vector<char> hit;
    pool_acquire(pool, {enc->id}, {enc->B}, hold.blk, hit);
    if ((rc = ensure_cross_kv(m, enc, hold.blk[0], hit[0] != 0))) return rc;
    for (int b = 0; b < B; ++b) slots[b] = hold.blk[0] * pool->EB + b;
  }
  FW_HIP(hipMemcpyAsync(g->slot_map, slots.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  GenDev gp;
  memset(&gp, 0, sizeof(gp));
  gp.B = B; gp.K = m->max_beam; gp.R = g->R; gp.P = 1; gp.V = c.n_vocab; gp.n_text_ctx = g->NT; gp.kv_div = 1;
  gp.ctx = 8; gp.cache_rows = B * m->max_beam;
  std::vector<int> tok(B, c.tok_sot);
  FW_HIP(hipMemsetAsync(g->kvidx2, 0, 2 * (size_t)g->R * g->NT, st));
  FW_HIP(hipMemsetAsync(g->d_step, 0, sizeof(int), st));
  FW_HIP(hipMemcpyAsync(g->prompt_dev, tok.data(), B * sizeof(int), hipMemcpyHostToDevice, st));
  FW_HIP(hipStreamSynchronize(st));
  StepCfg s;
  s.rows = B; s.kmul = 1; s.B = B; s.pos_fixed = 0; s.P = 1; s.tok = g->prompt_dev;
  s.need_logits = true; s.nospeech_rowmul = 0; s.beam_tail = false; s.done = g->zero_done;
  // rows of a prefill-style step use beam slot 0 of chunk b: slot stride is max_beam
  if ((rc = run_step(m, gp, s))) return rc;
  std::vector<float> lg((size_t)B * c.n_langs);
  FW_HIP(hipMemcpy2DAsync(lg.data(), c.n_langs * sizeof(float), g->logits + c.tok_lang_begin,
                          (size_t)c.n_vocab * sizeof(float), c.n_langs * sizeof(float), B, hipMemcpyDeviceToHost, st));
  FW_HIP(hipStreamSynchronize(st));
  if ((rc = check_launch("detect_language"))) return rc;
  prof_collect(m);
  for (int b = 0; b < B; ++b) {
    const float* r = &lg[(size_t)b * c.n_langs];
    float mx = r[0];
    for (int i = 1; i < c.n_langs; ++i) mx = std::max(mx, r[i]);
    std::vector<float> p(c.n_langs);
    float sum = 0.f;
    for (int i = 0; i < c.n_langs; ++i) { p[i] = expf(r[i] - mx); sum += p[i]; }
    for (int i = 0; i < c.n_langs; ++i) p[i] /= sum;
    std::vector<int> order(c.n_langs);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int bb) { return p[a] > p[bb]; });
    for (int i = 0; i < c.n_langs; ++i) {
      out_lang_ids[(size_t)b * c.n_langs + i] = c.tok_lang_begin + order[i];
      out_probs[(size_t)b * c.n_langs + i] = p[order[i]];
    }
  }
  return FW_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// align: standardise over tokens, median filter over frames, mean over heads (device),
// DTW (host, like CTranslate2).
// ------------------------------------------------------------------------------------
__global__ void align_stats_kernel(const float* __restrict__ probs, int n_sel, int n_tok_cap, int T,
                                   const int* __restrict__ n_tok, const int* __restrict__ nfr,
                                   float* __restrict__ stats) {
  const int b = blockIdx.z, hs = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nfr[b]) return;
  const int n = n_tok[b];
  const float* p = probs + (((size_t)b * n_sel + hs) * n_tok_cap) * T + t;
  float mean = 0.f;
  for (int i = 0; i < n; ++i) mean += p[(size_t)i * T];
  mean /= (float)n;
  float var = 0.f;
  for (int i = 0; i < n; ++i) {
    const float dlt = p[(size_t)i * T] - mean;
    var += dlt * dlt;
  }
  var /= (float)n;
  float* so = stats + (((size_t)b * n_sel + hs) * T + t) * 2;
  so[0] = mean;
  so[1] = 1.0f / sqrtf(var);
}

__global__ void align_filter_kernel(const float* __restrict__ probs, const float* __restrict__ stats, int n_sel,
                                    int n_tok_cap, int T, const int* __restrict__ n_tok, const int* __restrict__ nfr,
                                    int width, float* __restrict__ out) {
  const int b = blockIdx.z, tok = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int F = nfr[b];
  if (t >= F || tok >= n_tok[b]) return;
  const int pad = width / 2;
  float acc = 0.f;
  for (int hs = 0; hs < n_sel; ++hs) {
    const float* p = probs + (((size_t)b * n_sel + hs) * n_tok_cap + tok) * T;
    const float* st = stats + ((size_t)b * n_sel + hs) * T * 2;
    float w[15];
    const bool filt = pad > 0 && F > pad;
    const int wd = filt ? width : 1;
    for (int i = 0; i < wd; ++i) {
      int tt = filt ? t - pad + i : t;
      if (tt < 0) tt = -tt;                    // reflect padding
      if (tt >= F) tt = 2 * (F - 1) - tt;
      w[i] = (p[tt] - st[2 * tt]) * st[2 * tt + 1];
    }
    for (int i = 1; i < wd; ++i) {             // insertion sort (width <= 15)
      const float v = w[i];
      int q = i - 1;
      while (q >= 0 && w[q] > v) { w[q + 1] = w[q]; --q; }
      w[q + 1] = v;
    }
    acc += w[filt ? pad : 0];
  }
  out[((size_t)b * n_tok_cap + tok) * T + t] = acc / (float)n_sel;
}

namespace fw {
// openai-whisper timing.dtw_cpu on cost = -matrix (N text rows x M frames)
static void dtw_path(const float* mat, int ld, int N, int M, std::vector<int>& ti, std::vector<int>& fi) {
  std::vector<double> D((size_t)(N + 1) * (M + 1), INFINITY);
  std::vector<int8_t> tr((size_t)(N + 1) * (M + 1), -1);
  auto at = [&](int i, int j) -> size_t { return (size_t)i * (M + 1) + j; };
  D[at(0, 0)] = 0;
  for (int j = 1; j <= M; ++j)
    for (int i = 1; i <= N; ++i) {
      const double c0 = D[at(i - 1, j - 1)], c1 = D[at(i - 1, j)], c2 = D[at(i, j - 1)];
      double cc; int8_t t;
      if (c0 < c1 && c0 < c2) { cc = c0; t = 0; }
      else if (c1 < c0 && c1 < c2) { cc = c1; t = 1; }
      else { cc = c2; t = 2; }
      D[at(i, j)] = -(double)mat[(size_t)(i - 1) * ld + (j - 1)] + cc;
      tr[at(i, j)] = t;
    }
  for (int j = 0; j <= M; ++j) tr[at(0, j)] = 2;
  for (int i = 0; i <= N; ++i) tr[at(i, 0)] = 1;
  int i = N, j = M;
  ti.clear(); fi.clear();
  while (i > 0 || j > 0) {
    ti.push_back(i - 1); fi.push_back(j - 1);
    const int8_t t = tr[at(i, j)];
    if (t == 0) { --i; --j; }
    else if (t == 1) { --i; }
    else { --j; }
  }
  std::reverse(ti.begin(), ti.end());
  std::reverse(fi.begin(), fi.end());
}
}  // namespace fw

extern "C" int32_t fw_align(fw_model* fm, const fw_tensor* enc_t, const int32_t* start_seq, int32_t n_start,
                            const int32_t* text_tokens, const int32_t* text_offsets, const int32_t* num_frames,
                            int32_t B, int32_t median_filter_width, int32_t max_pairs, int32_t* out_pairs,
                            int32_t* out_n_pairs, float* out_probs) {
  FW_CHECK_ARG(fm && enc_t && start_seq && text_tokens && text_offsets && num_frames && out_pairs && out_n_pairs &&
                   out_probs, "null argument");
  Model* m = decoder_of(&fm->impl);
  const Tensor* enc = &enc_t->impl;
  const fw_config& c = m->cfg;
  FW_CHECK_ARG(enc->owner && decoder_of(enc->owner) == m && B == enc->B && B <= m->max_batch,
               "bad encoder output / batch");
  FW_CHECK_ARG(n_start >= 1, "start_sequence must not be empty");
  FW_CHECK_ARG(median_filter_width >= 1 && median_filter_width <= 15 && (median_filter_width & 1),
               "median_filter_width must be odd and <= 15");
  const int T = c.n_audio_ctx, d = c.d_model;
  // teacher-forced sequences: start + no_timestamps + text + eot
  std::vector<std::vector<int>> seq(B);
  int max_tok = 0;
  for (int b = 0; b < B; ++b) {
    const int nt = text_offsets[b + 1] - text_offsets[b];
    FW_CHECK_ARG(nt >= 0, "bad text offsets");
    for (int i = 0; i < n_start; ++i) seq[b].push_back(start_seq[i]);
    seq[b].push_back(c.tok_no_timestamps);
    for (int i = 0; i < nt; ++i) {
      const int t = text_tokens[text_offsets[b] + i];
      FW_CHECK_ARG(t >= 0 && t < c.n_vocab, "text token %d out of range", t);
      seq[b].push_back(t);
    }
    seq[b].push_back(c.tok_eot);
    FW_CHECK_ARG((int)seq[b].size() <= c.n_text_ctx, "align sequence of %zu tokens exceeds the text context", seq[b].size());
    max_tok = std::max(max_tok, (int)seq[b].size());
  }
  // alignment heads grouped per layer
  std::vector<std::vector<int>> per_layer(c.n_dec_layers);
  if (c.n_align_heads > 0) {
    for (int i = 0; i < c.n_align_heads; ++i) {
      const int l = c.align_heads[2 * i], h = c.align_heads[2 * i + 1];
      FW_CHECK_ARG(l >= 0 && l < c.n_dec_layers && h >= 0 && h < c.n_heads, "alignment head (%d,%d) out of range", l, h);
      per_layer[l].push_back(h);
    }
  } else {  // [CT2-ext] default: all heads of the upper half of the decoder
    for (int l = c.n_dec_layers / 2; l < c.n_dec_layers; ++l)
      for (int h = 0; h < c.n_heads; ++h) per_layer[l].push_back(h);
  }
  std::vector<int> sel_heads, layer_off(c.n_dec_layers + 1, 0);
  for (int l = 0; l < c.n_dec_layers; ++l) {
    layer_off[l] = (int)sel_heads.size();
    for (int h : per_layer[l]) sel_heads.push_back(h);
  }
  layer_off[c.n_dec_layers] = (int)sel_heads.size();
  const int n_sel = (int)sel_heads.size();
  FW_CHECK_ARG(n_sel > 0, "no alignment heads");

  std::lock_guard<std::mutex> lk(m->dec_mu);
  FW_HIP(hipSetDevice(m->device));
  int rc;
  if ((rc = gen_workspace_ensure(m))) return rc;
  GenWorkspace* g = m->gen;
  hipStream_t st = m->dec_stream;
  CrossPool* pool = pool_of(m);
  PoolHold hold{pool, {}, st};
  std::vector<int> slots(B);
  {
    std::vector<char> hit;
    pool_acquire(pool, {enc->id}, {enc->B}, hold.blk, hit);
    if ((rc = ensure_cross_kv(m, enc, hold.blk[0], hit[0] != 0))) return rc;
    for (int b = 0; b < B; ++b) slots[b] = hold.blk[0] * pool->EB + b;
  }
  const int kv_slot0 = slots[0];

  float *probs = nullptr, *stats = nullptr, *mat = nullptr, *tprob = nullptr;
  int *heads_dev = nullptr, *ntok_dev = nullptr, *nfr_dev = nullptr, *target_dev = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)probs, (void*)stats, (void*)mat, (void*)tprob, (void*)heads_dev, (void*)ntok_dev,
                    (void*)nfr_dev, (void*)target_dev})
      if (p) (void)hipFree(p);
  };
#define AL(p, n) do { if ((rc = dev_alloc_t(&(p), (n)))) { cleanup(); return rc; } } while (0)
  AL(probs, (size_t)B * n_sel * max_tok * T);
  AL(stats, (size_t)B * n_sel * T * 2);
  AL(mat, (size_t)B * max_tok * T);
  AL(tprob, (size_t)B * max_tok);
  AL(heads_dev, (size_t)n_sel);
  AL(ntok_dev, (size_t)B); AL(nfr_dev, (size_t)B);
  AL(target_dev, (size_t)max_tok * B);
#undef AL
  std::vector<int> ntok(B), nfr(B), ptok((size_t)max_tok * B), target((size_t)max_tok * B, -1);
  for (int b = 0; b < B; ++b) {
    ntok[b] = (int)seq[b].size();
    nfr[b] = std::min(T, std::max(1, num_frames[b] / 2));
    for (int p = 0; p < max_tok; ++p) {
      ptok[(size_t)p * B + b] = p < ntok[b] ? seq[b][p] : c.tok_eot;
      // the logits at position p predict token p+1; we want probs of the text tokens only
      const int n0 = n_start + 1;
      const int nt = ntok[b] - n0 - 1;
      if (p >= n0 - 1 && p < n0 - 1 + nt) target[(size_t)p * B + b] = seq[b][p + 1];
    }
  }
  hipError_t he = hipMemcpyAsync(heads_dev, sel_heads.data(), n_sel * sizeof(int), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(ntok_dev, ntok.data(), B * sizeof(int), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(nfr_dev, nfr.data(), B * sizeof(int), hipMemcpyHostToDevice, st);
  if (he == hipSuccess)
    he = hipMemcpyAsync(g->prompt_dev, ptok.data(), ptok.size() * sizeof(int), hipMemcpyHostToDevice, st);
  if (he == hipSuccess)
    he = hipMemcpyAsync(target_dev, target.data(), target.size() * sizeof(int), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemcpyAsync(g->slot_map, slots.data(), B * sizeof(int), hipMemcpyHostToDevice, st);
  if (he == hipSuccess) he = hipMemsetAsync(g->kvidx2, 0, 2 * (size_t)g->R * g->NT, st);
  if (he == hipSuccess) he = hipMemsetAsync(g->d_step, 0, sizeof(int), st);
  if (he == hipSuccess) he = hipMemsetAsync(tprob, 0, (size_t)B * max_tok * sizeof(float), st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  if (he != hipSuccess) {
    cleanup();
    set_error("align setup failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  GenDev gp;
  memset(&gp, 0, sizeof(gp));
  gp.B = B; gp.K = m->max_beam; gp.R = g->R; gp.P = max_tok; gp.V = c.n_vocab; gp.n_text_ctx = g->NT; gp.kv_div = 1;
  gp.ctx = std::min(g->NT, (max_tok + 7) / 8 * 8); gp.cache_rows = B * m->max_beam;   // <= EB * K * NT <= self_cap
  // teacher forcing: every position is known up front, so the text goes through the decoder in blocks of up to 16
  // positions per pass (a 100-token segment: 7 passes instead of 100)
  const int nbmax = pos_block_size(m, g, gp, B);
  if (nbmax > 1) {
    std::vector<int> blk_tok;
    blk_tok.reserve((size_t)max_tok * B);
    for (int pos = 0; pos < max_tok;) {
      const int nb = std::min(nbmax, max_tok - pos);
      for (int b = 0; b < B; ++b)
        for (int j = 0; j < nb; ++j) blk_tok.push_back(ptok[(size_t)(pos + j) * B + b]);
      pos += nb;
    }
    he = hipMemcpyAsync(g->prompt_blk, blk_tok.data(), blk_tok.size() * sizeof(int), hipMemcpyHostToDevice, st);
    if (he == hipSuccess) he = hipStreamSynchronize(st);
    if (he != hipSuccess) {
      cleanup();
      set_error("align setup failed: %s", hipGetErrorString(he));
      return FW_ERUNTIME;
    }
  }
  for (int pos = 0; pos < max_tok;) {
    const int nb = std::min(nbmax, max_tok - pos);
    StepCfg s;
    s.B = B; s.pos_fixed = pos; s.P = max_tok;
    if (nbmax > 1) {   // (a last block of one position has the same layout either way)
      s.rows = B * nb; s.kmul = nb; s.blk_n = nb > 1 ? nb : 0; s.tok = g->prompt_blk + (size_t)pos * B;
    } else {
      s.rows = B; s.kmul = 1; s.tok = g->prompt_dev + (size_t)pos * B;
    }
    bool any_target = false;
    for (int j = 0; j < nb; ++j)
      for (int b = 0; b < B; ++b) any_target |= target[(size_t)(pos + j) * B + b] >= 0;
    s.need_logits = any_target;
    s.nospeech_rowmul = 0; s.beam_tail = false; s.done = g->zero_done;
    s.sel_heads_dev = heads_dev; s.sel_layer_off = layer_off.data(); s.probs = probs; s.n_sel_total = n_sel;
    s.kv_slot0 = kv_slot0;
    s.n_tok = max_tok; s.tok_idx = pos;
    if ((rc = run_step(m, gp, s))) { cleanup(); return rc; }
    for (int j = 0; j < nb && any_target; ++j) {
      bool tj = false;
      for (int b = 0; b < B; ++b) tj |= target[(size_t)(pos + j) * B + b] >= 0;
      if (tj)
        fwd::launch_token_prob(st, g->logits + (size_t)j * c.n_vocab, c.n_vocab, target_dev + (size_t)(pos + j) * B, tprob,
                               max_tok, pos + j, B, nb);
    }
    pos += nb;
  }
  {
    dim3 g1((T + 127) / 128, n_sel, B);
    align_stats_kernel<<<g1, 128, 0, st>>>(probs, n_sel, max_tok, T, ntok_dev, nfr_dev, stats);
    dim3 g2((T + 127) / 128, max_tok, B);
    align_filter_kernel<<<g2, 128, 0, st>>>(probs, stats, n_sel, max_tok, T, ntok_dev, nfr_dev, median_filter_width,
                                             mat);
  }
  std::vector<float> hmat((size_t)B * max_tok * T), htprob((size_t)B * max_tok);
  he = hipMemcpyAsync(hmat.data(), mat, hmat.size() * sizeof(float), hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipMemcpyAsync(htprob.data(), tprob, htprob.size() * sizeof(float), hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  if (he == hipSuccess) he = hipGetLastError();
  cleanup();
  if (he != hipSuccess) {
    set_error("align failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  prof_collect(m);
  const int n0 = n_start + 1;
  for (int b = 0; b < B; ++b) {
    const int nt = ntok[b] - n0 - 1;
    out_n_pairs[b] = 0;
    for (int i = 0; i < nt; ++i) out_probs[text_offsets[b] + i] = htprob[(size_t)b * max_tok + (n0 - 1 + i)];
    if (nt < 0) continue;
    // rows of the cost matrix: positions n_start .. n_start + nt (the <|notimestamps|> position, whose
    // attention predicts the first text token, up to the last text token; the <|endoftext|> row is dropped)
    // = nt + 1 rows, like openai-whisper's matrix[len(sot_sequence):-1].  The reference indexes the path's
    // token jumps with word boundaries up to nt (transcribe.py:1741-1745), so it needs exactly these rows.
    std::vector<int> ti, fi;
    fw::dtw_path(&hmat[((size_t)b * max_tok + n_start) * T], T, nt + 1, nfr[b], ti, fi);
    const int np = std::min((int)ti.size(), max_pairs);
    for (int i = 0; i < np; ++i) {
      out_pairs[((size_t)b * max_pairs + i) * 2] = ti[i];
      out_pairs[((size_t)b * max_pairs + i) * 2 + 1] = fi[i];
    }
    out_n_pairs[b] = np;
  }
  return FW_OK;
}
