// Engine-internal structures of libfwamd.so (host side, C++17).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fwamd.h"
#include "common.h"

namespace fw {

void set_error(const char* fmt, ...);
#define FW_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fw::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return FW_ENODEV;                                                                \
    }                                                                                  \
  } while (0)
#define FW_CHECK_ARG(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      fw::set_error(__VA_ARGS__);      \
      return FW_EINVAL;                \
    }                                  \
  } while (0)

// ---- weight blob (what travels over RCCL at load time) --------------------------------
#define FW_BLOB_MAGIC "FWAMDBL1"
struct BlobHeader {
  char magic[8];
  int32_t version;
  int32_t n_tensors;
  int64_t total_bytes;
  int32_t compute_type;
  int32_t reserved;
  fw_config cfg;
};
struct BlobEntry {
  char name[64];
  int32_t dtype;  // 1 = f16, 2 = i8, 0 = f32
  int32_t ndim;
  int64_t dims[4];
  int64_t offset;  // from blob start, 256-byte aligned
  int64_t nbytes;
};

struct DevTensor {
  void* ptr = nullptr;
  int dtype = 1;
  int ndim = 0;
  int64_t dims[4] = {0, 0, 0, 0};
};

struct LinearW {           // y = x W^T + b ; W [N][K] fp16 (or int8 + per-row scale)
  const half_t* w = nullptr;
  const half_t* b = nullptr;
  const int8_t* wq = nullptr;    // int8 weights [N][K]           (int8_float16 only)
  const float* wscale = nullptr; // dequant scale per output row  (int8_float16 only)
  // LayerNorm-folded form (decoder): w = W.g, y = rstd*(w x - mu*s1) + cf   (engine.hip add_folded)
  const float* s1 = nullptr;
  const float* cf = nullptr;
  int N = 0, K = 0;
};
struct LNW { const half_t* g = nullptr; const half_t* b = nullptr; };

struct EncLayerW { LNW ln1, ln2; LinearW qk, v, out, ffn1, ffn2; };
// fp16: qkv, cq, ffn1 are LN-folded; a blob packed with FWAMD_LN_UNFOLD also carries the plain weights (qkv_p, cq_p, ffn1_p) and ln1/2/3 for the
// explicit-LayerNorm evaluation (Model::ln_unfold); int8_float16: explicit LayerNorms feeding the quantiser
struct DecLayerW { LNW ln1, ln2, ln3; LinearW qkv, out, cq, ck, cv, cout, ffn1, ffn2; LinearW qkv_p, cq_p, ffn1_p; };

enum ProfFamily {
  PF_LOGMEL = 0, PF_ENC_GEMM, PF_ENC_ATTN, PF_ENC_LN, PF_CROSS_KV_GEMM,
  PF_DEC_GEMM_QKV, PF_DEC_GEMM_DXD, PF_DEC_GEMM_FFN1, PF_DEC_GEMM_FFN2,
  PF_DEC_SELF_ATTN, PF_DEC_CROSS_ATTN, PF_DEC_LOGITS, PF_DEC_SAMPLE, PF_DEC_MISC, PF_COUNT
};

struct ProfAcc {
  double ms = 0; int64_t launches = 0; double flops = 0; double bytes = 0;
};

struct Model;
struct Tensor {  // fw_tensor
  Model* owner = nullptr;
  half_t* data = nullptr;  // [B][T][D] fp16, device
  int B = 0, T = 0, D = 0;
  uint64_t id = 0;         // unique per encoder output (cross-K/V cache key)
};

struct GenWorkspace;  // decoder-side buffers of one decode lane (decoder.hip)
struct GenRequest;    // one fw_generate call waiting to be decoded (decoder.hip)
struct CrossPool;     // cross-attention K / V^T of the encoder outputs in flight, shared by the lanes (decoder.hip)

// Decode group of a device: the worker replicas that share one decode workspace.  Concurrent fw_generate calls
// with identical options are merged into ONE decode run (their rows share every weight byte streamed per step);
// the first caller that finds no run in progress leads it, the others wait for their results.
struct DecodeGroup {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<GenRequest*> queue;
  bool gathering = false;         // a caller is collecting the requests of the next run (one at a time)
  bool resizing = false;          // fw_model_set_decode_batch is rebuilding the workspaces / the second lane: callers wait
  int active_runs = 0;            // runs in flight (<= lanes of the group)
  bool lane_busy[4] = {false, false, false, false};
  std::atomic<int> lanes_enabled{4};   // fw_model_set_decode_lanes: runs allowed in flight (1 .. the lanes the group has)
  std::atomic<int> encoding{0};   // member encodes in flight: requests that are about to arrive
  std::chrono::steady_clock::time_point last_arrival{};   // when the newest request was queued
  std::mutex enc_mu;              // one encoder pass at a time per device
  std::atomic<int> enc_pass_us{0};   // running mean of the time a member holds enc_mu (one encoder pass)
  std::atomic<int> merge_wait_ms{-1};   // fw_model_set_merge_wait: -1 = 2.5 encoder passes (<= 250 ms), 0 = never wait
  std::atomic<int> merge_fill_pct{90};  // ... and the share of a run's chunk capacity at which the leader stops waiting
  // statistics (fw_model_decode_stats): decode runs, fw_generate calls served, chunks decoded, largest run
  std::atomic<int64_t> n_runs{0}, n_requests{0}, n_chunks{0};
  std::atomic<int> max_run_chunks{0};
};

struct Model {
  fw_config cfg{};
  int compute_type = 0;
  int device = 0;
  int max_batch = 0, max_beam = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;

  // weights
  void* blob = nullptr;     // device blob (owned unless external)
  bool blob_owned = true;
  int64_t blob_bytes = 0;
  std::map<std::string, DevTensor> tensors;
  int c_pad = 0;  // mel channels padded to a multiple of 64 for the conv1 GEMM
  LinearW conv1, conv2;
  const half_t* enc_pos = nullptr;
  std::vector<EncLayerW> enc;
  LNW enc_ln_post;
  const half_t* tok_emb = nullptr;  // [V][d]
  const half_t* dec_pos = nullptr;  // [n_text_ctx][d]
  std::vector<DecLayerW> dec;
  LinearW logits;  // final LN folded into the tied-embedding projection (fp16) / int8 tied embedding
  LinearW logits_p;   // fp16: the tied embedding itself, fragment-major (explicit final LayerNorm, ln_unfold >= 1)
  LNW dec_ln;
  // fp16 evaluation order of the decoder LayerNorms.  0: folded into the consuming linear (one
  // launch fewer per LayerNorm; W o g is rounded to fp16 once more and the normalised row is never rounded); 1: the final
  // LayerNorm is its own kernel — fp16(LN(x)) times the tied embedding, the rounding points of the reference's fp16
  // path; 2: every decoder LayerNorm is.  FWAMD_LN_UNFOLD in the environment when the blob is PACKED (the plain weight
  // forms then travel in it: has_plain) and when the model is created.  A diagnostic; default off.
  int ln_unfold = 0;
  bool has_plain = false;

  // log-mel constants
  float* lm_consts = nullptr;  // cos table [400] + hann [400]
  float* lm_filtT = nullptr;   // [224][mel_pad]
  int lm_mel_pad = 0;

  // encoder workspaces (sized for max_batch)
  float* ws_pcm = nullptr; int64_t ws_pcm_cap = 0;
  int64_t* ws_offsets = nullptr;
  float* ws_raw = nullptr; int64_t ws_raw_cap = 0;   // raw log-mel
  int* ws_chunk_max = nullptr;
  int* ws_nframes = nullptr;
  float* ws_feat32 = nullptr;                         // [B][n_mels][3000]
  half_t* ws_mel_cl = nullptr;                        // [B][3002][c_pad]
  half_t* ws_conv1 = nullptr;                         // [B][3002][d]
  half_t *ws_x = nullptr, *ws_x2 = nullptr, *ws_xn = nullptr;  // [B][1500][d]
  half_t* ws_qk = nullptr;                            // [B][1500][2d]
  half_t* ws_vt = nullptr;                            // [B][d][t_pad]
  half_t* ws_att = nullptr;                           // [B][1500][d]
  half_t* ws_ffn = nullptr;                           // [B][1500][4d]
  int8_t* ws_xq = nullptr;                            // int8_float16: quantised GEMM input [B*1500][4d]
  float* ws_xs = nullptr;                             //               its per-row scales [B*1500]
  int t_pad = 0;

  // decode side.  A model either owns a decode workspace (gen, created on first use with room for decode_batch
  // chunks, run on dec_stream under dec_mu) or has joined another model of the same device (decoder != null).
  GenWorkspace* gen = nullptr;
  Model* decoder = nullptr;
  // Second decode LANE of a decode group (fw_model_set_decode_batch with room for >= 4 encoder batches): an internal,
  // decoder-only model on the same weight blob with a workspace and a stream of its own, so that TWO decode runs of the
  // group are in flight at once — the cross-attention stream of one (HBM-bound) beside the linears of the other:
  // measured +8 % (profiles/r03_two_groups_probe.txt).  Owned by the primary; never visible through the C ABI.
  // (FWAMD_DECODE_LANES = 3 / 4 builds further lanes of the same kind — measurement knob, profiles/r06_ab_lanes.jsonl)
  Model* lane1 = nullptr;        // = xlanes[0]
  Model* xlanes[3] = {nullptr, nullptr, nullptr};   // lanes 1 .. 3
  bool is_lane = false;
  // The cross-attention K / V^T cache is ONE pool per decode group (owned by the primary, borrowed by the second lane):
  // blocks of max_batch chunk slots, one block per encoder output in flight, handed to whichever lane decodes the
  // request — so both lanes can run full-size runs without each holding a cache of its own.
  CrossPool* xpool = nullptr;
  Model* pool_owner = nullptr;   // a lane: the primary whose pool it reads
  int decode_batch = 0;          // chunk slots of the pool (= encoder chunks the group keeps in flight)
  int lane_batch = 0;            // chunks ONE decode run (one lane's workspace) holds; 0: decode_batch
  int decode_self_ctx = 0;   // self-attention cache positions per row at full row capacity (0 = the text context)
  int dependents = 0;        // live models that use this one's weight blob or decode workspace (g_models_mu)
  bool free_deferred = false;   // fw_model_free was called while dependents > 0: freed with the last dependent
  void* self = nullptr;         // the fw_model this Model lives in
  Model* blob_owner = nullptr;  // the model whose blob this one borrows (fw_model_create_from_blob_dev on fw_model_blob)
  hipStream_t dec_stream = nullptr;
  std::mutex dec_mu;
  DecodeGroup grp;

  // pooled encoder-output buffers ([max_batch][T][d] each)
  std::mutex pool_mu;
  std::vector<half_t*> enc_pool;

  // profiling.  The encoder thread of the model (under mu) and the leader of a decode run (under dec_mu, any worker
  // thread) both record scopes on the same model, so the event lists have a mutex of their own.
  std::mutex prof_mu;
  bool prof_on = false;
  ProfAcc prof[PF_COUNT];
  struct PendingEv { hipEvent_t a, b; int fam; };
  std::vector<PendingEv> pending;
  std::vector<hipEvent_t> ev_pool;
};

// profiling scope: brackets a group of launches with HIP events on the model's stream
struct ProfScope {
  Model* m; int fam; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(Model* m_, int fam_, double flops, double bytes, hipStream_t st_ = nullptr);   // null: m->stream
  ~ProfScope();
};
void prof_collect(Model* m);

int dev_alloc(void** p, size_t bytes);
// a non-blocking stream at the priority the environment variable `env` names ("high" / "low"; anything else, or unset:
// the default priority).  ROCm gives every priority level its own hardware queues, so a decode stream created "high"
// neither shares a hardware queue with the encoder streams of the workers nor waits behind their workgroups when a CU
// frees up.  Measurement knob: FWAMD_DEC_STREAM_PRIO (decode lanes), FWAMD_ENC_STREAM_PRIO (encoder / replica streams).
hipError_t create_stream(hipStream_t* st, const char* role);   // role "ENC" / "DEC": engine.hip
// decode groups (decoder.hip): chunks an idle two-lane group waits for before it leads a run
int64_t idle_lead_chunks(int64_t queued, int n_queued, int encoding, int64_t want, int max_batch, int lanes = 2);
template <typename T>
inline int dev_alloc_t(T** p, size_t n) { return dev_alloc(reinterpret_cast<void**>(p), n * sizeof(T)); }

// linear layer on "many rows": C = act(A W^T + b) + res   (encoder / prefill / align)
// st == null: the model's encoder stream
int run_linear(Model* m, const LinearW& L, const half_t* A, int64_t lda, int64_t a_bs, half_t* C, int64_t ldc,
               int64_t c_bs, const half_t* res, int64_t ldr, int64_t r_bs, int M, int batch, int act, bool trans,
               int head_rows = 0, hipStream_t st = nullptr);

// xq / xs == null: the encoder's quantisation workspace (m->ws_xq / m->ws_xs)
int run_linear_i8(Model* m, const LinearW& L, const half_t* A, const LNW* ln, half_t* C, int64_t ldc, int64_t c_bs,
                  const half_t* res, int64_t ldr, int64_t r_bs, int M, int batch, int act, bool trans, int head_rows,
                  hipStream_t st = nullptr, int8_t* xq = nullptr, float* xs = nullptr);

// encoder forward on the channel-last mel image already in m->ws_mel_cl; result into out [B][1500][d]
// ONE launch for the same linear of `n_layers` consecutive layers on one input (fp16): layer l uses L0.w + l * w_lstride,
// L0.b + l * b_lstride and writes C + l * c_lstride (element strides); gemm.hip "LAYERED launch"
int run_linear_layers(Model* m, const LinearW& L0, int n_layers, int64_t w_lstride, int64_t b_lstride, const half_t* A,
                      int64_t lda, int64_t a_bs, half_t* C, int64_t ldc, int64_t c_bs, int64_t c_lstride, int M, int batch,
                      bool trans, int head_rows, hipStream_t st);
void set_cross_kv_layered(int on);                 // decoder.hip: knob 6
int run_encoder(Model* m, int B, half_t* out);

// decoder entry points (decoder.hip)
void set_pos_blocks(int on);                  // fw_test_knob(4, ..): position blocks for the prompt forward and align
uint64_t next_tensor_id();
int gen_workspace_ensure(Model* dm);          // creates dm's decode workspace on first use (caller holds dm->dec_mu)
void gen_workspace_free(Model* m);
// HBM of one lane's workspace for runs of up to lane_chunks chunks (self_ctx 0 = the text context) — without the
// cross-attention pool, which is cross_pool_bytes(m, pool_chunks) once per group
int64_t gen_workspace_bytes(const Model* m, int lane_chunks, int self_ctx);
int64_t cross_pool_bytes(const Model* m, int pool_chunks);
void cross_pool_free(Model* m);
inline Model* decoder_of(Model* m) { return m->decoder ? m->decoder : m; }
inline int n_lanes_of(const Model* m) {
  int n = 1;
  for (const Model* l : m->xlanes) n += l ? 1 : 0;
  return n;
}
inline Model* lane_model(Model* m, int lane) { return lane == 0 ? m : m->xlanes[lane - 1]; }
inline int lane_chunks_of(const Model* m) {
  const int b = m->lane_batch > 0 ? m->lane_batch : m->decode_batch;
  return b > m->max_batch ? b : m->max_batch;
}

}  // namespace fw

struct fw_model { fw::Model impl; };
struct fw_tensor { fw::Tensor impl; };
