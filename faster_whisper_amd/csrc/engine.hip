// libfwamd.so — host side of the MI355X Whisper engine: C ABI (include/fwamd.h), weight
// packing/upload, workspaces, the log-mel + encoder pipeline and the kernel test hooks.
// The decoder / generate / align side lives in decoder.hip.
//
// Reference interfaces replaced here (faster_whisper/transcribe.py): the
// ctranslate2.models.Whisper constructor (:689-698), .encode (:1400), the
// numpy FeatureExtractor call on the host (:463-467) and StorageView.from_array (:1875).
#include "engine.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "dec_kernels.h"
#include "kernels.h"

namespace fw {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int dev_alloc(void** p, size_t bytes) {
  if (bytes == 0) bytes = 256;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    *p = nullptr;
    return FW_ENOMEM;
  }
  return FW_OK;
}

// ---------------------------------------------------------------- profiling
static const char* kProfNames[PF_COUNT] = {
    "logmel", "enc_gemm", "enc_attn", "enc_layernorm", "cross_kv_gemm", "dec_gemm_qkv", "dec_gemm_dxd",
    "dec_gemm_ffn1", "dec_gemm_ffn2", "dec_self_attn", "dec_cross_attn", "dec_logits", "dec_sample", "dec_misc"};

static hipEvent_t ev_get(Model* m) {
  if (!m->ev_pool.empty()) {
    hipEvent_t e = m->ev_pool.back();
    m->ev_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}
ProfScope::ProfScope(Model* m_, int fam_, double flops, double bytes, hipStream_t st_)
    : m(m_), fam(fam_), st(st_ ? st_ : m_->stream) {
  if (!m->prof_on) return;
  {
    std::lock_guard<std::mutex> lk(m->prof_mu);
    a = ev_get(m);
    b = ev_get(m);
    m->prof[fam].flops += flops;
    m->prof[fam].bytes += bytes;
    m->prof[fam].launches += 1;
  }
  (void)hipEventRecord(a, st);
}
ProfScope::~ProfScope() {
  if (!a) return;
  (void)hipEventRecord(b, st);
  std::lock_guard<std::mutex> lk(m->prof_mu);
  m->pending.push_back({a, b, fam});
}
void prof_collect(Model* m) {
  // the events are waited for outside the lock (a scope of the other stream may be recorded meanwhile)
  std::vector<Model::PendingEv> todo;
  {
    std::lock_guard<std::mutex> lk(m->prof_mu);
    todo.swap(m->pending);
  }
  std::vector<std::pair<int, float>> got;
  for (auto& p : todo) {
    (void)hipEventSynchronize(p.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, p.a, p.b);
    got.push_back({p.fam, ms});
  }
  std::lock_guard<std::mutex> lk(m->prof_mu);
  for (auto& g : got) m->prof[g.first].ms += g.second;
  for (auto& p : todo) {
    m->ev_pool.push_back(p.a);
    m->ev_pool.push_back(p.b);
  }
}

// ---------------------------------------------------------------- fp16 helpers (host)
static inline uint16_t f32_to_f16_bits(float f) {
  half_t h = (half_t)f;  // round-to-nearest-even, same as the device cast
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}

// ---------------------------------------------------------------- blob packing
struct PackItem {
  std::string name;
  int ndim;
  int64_t dims[4];
  std::vector<uint16_t> data;  // fp16 payload (dtype 1)
  std::vector<float> fdata;    // fp32 payload (dtype 0)
  std::vector<int8_t> qdata;   // int8 payload (dtype 2)
  int dtype = 1;
  const void* bytes() const {
    return dtype == 1 ? (const void*)data.data() : dtype == 2 ? (const void*)qdata.data() : (const void*)fdata.data();
  }
  int64_t nbytes() const {
    return dtype == 1 ? (int64_t)data.size() * 2 : dtype == 2 ? (int64_t)qdata.size() : (int64_t)fdata.size() * 4;
  }
};

static inline float f16_bits_to_f32(uint16_t u) {
  half_t h;
  memcpy(&h, &u, 2);
  return (float)h;
}

static const fw_weight* find_w(const fw_weight* w, int n, const std::string& name) {
  for (int i = 0; i < n; ++i)
    if (name == w[i].name) return &w[i];
  return nullptr;
}
static int64_t numel(const fw_weight* w) {
  int64_t n = 1;
  for (int i = 0; i < w->ndim; ++i) n *= w->dims[i];
  return n;
}
static float w_at(const fw_weight* w, int64_t i) {
  if (w->dtype == FW_DT_F32) return reinterpret_cast<const float*>(w->data)[i];
  half_t h;
  memcpy(&h, reinterpret_cast<const uint16_t*>(w->data) + i, 2);
  return (float)h;
}
static void to_f16(const fw_weight* w, std::vector<uint16_t>& out) {
  const int64_t n = numel(w);
  out.resize(n);
  if (w->dtype == FW_DT_F16) {
    memcpy(out.data(), w->data, n * 2);
  } else {
    const float* s = reinterpret_cast<const float*>(w->data);
    for (int64_t i = 0; i < n; ++i) out[i] = f32_to_f16_bits(s[i]);
  }
}

static int expect_shape(const fw_weight* w, const std::string& name, std::initializer_list<int64_t> dims) {
  if (!w) {
    set_error("missing weight '%s'", name.c_str());
    return FW_EINVAL;
  }
  if ((size_t)w->ndim != dims.size()) {
    set_error("weight '%s': expected %zu dims, got %d", name.c_str(), dims.size(), w->ndim);
    return FW_EINVAL;
  }
  int i = 0;
  for (int64_t d : dims) {
    if (w->dims[i] != d) {
      set_error("weight '%s': dim %d is %lld, expected %lld", name.c_str(), i, (long long)w->dims[i], (long long)d);
      return FW_EINVAL;
    }
    ++i;
  }
  return FW_OK;
}

static int check_config(const fw_config* c) {
  FW_CHECK_ARG(c->n_mels > 0 && c->n_mels <= 128, "n_mels must be in (0,128], got %d", c->n_mels);
  FW_CHECK_ARG(c->d_model % 128 == 0 && c->d_model >= 128 && c->d_model <= 1280,
               "d_model must be a multiple of 128 in [128,1280], got %d", c->d_model);
  FW_CHECK_ARG(c->n_heads * 64 == c->d_model, "head dim must be 64 (n_heads=%d, d_model=%d)", c->n_heads, c->d_model);
  FW_CHECK_ARG(c->n_audio_ctx == 1500, "n_audio_ctx must be 1500, got %d", c->n_audio_ctx);
  FW_CHECK_ARG(c->n_text_ctx > 0 && c->n_text_ctx <= 448, "n_text_ctx must be in (0,448]");
  FW_CHECK_ARG(c->n_enc_layers > 0 && c->n_dec_layers > 0, "layer counts must be positive");
  FW_CHECK_ARG(c->n_vocab > c->tok_timestamp_begin && c->tok_timestamp_begin > 0, "bad vocabulary layout");
  FW_CHECK_ARG(c->n_align_heads >= 0 && c->n_align_heads <= FW_MAX_ALIGN_HEADS, "too many alignment heads");
  FW_CHECK_ARG(c->n_suppress_begin >= 0 && c->n_suppress_begin <= 8, "n_suppress_begin out of range");
  // the logits-rules kernel keeps a row's logits in registers: 56 values x 1024 threads
  FW_CHECK_ARG(c->n_vocab <= 57344, "n_vocab %d exceeds the 57344 ids the logits kernel holds per row", c->n_vocab);
  const int toks[] = {c->tok_eot, c->tok_sot, c->tok_translate, c->tok_transcribe, c->tok_sot_lm, c->tok_sot_prev,
                      c->tok_no_speech, c->tok_no_timestamps, c->tok_timestamp_begin};
  for (int t : toks) FW_CHECK_ARG(t >= 0 && t < c->n_vocab, "special token id %d outside the vocabulary of %d", t, c->n_vocab);
  FW_CHECK_ARG(c->n_langs >= 0 && c->tok_lang_begin >= 0 && c->tok_lang_begin + c->n_langs <= c->n_vocab,
               "language ids [%d, %d) outside the vocabulary of %d", c->tok_lang_begin, c->tok_lang_begin + c->n_langs,
               c->n_vocab);
  for (int i = 0; i < c->n_suppress_begin; ++i)
    FW_CHECK_ARG(c->suppress_begin[i] >= 0 && c->suppress_begin[i] < c->n_vocab, "suppress_begin id out of range");
  return FW_OK;
}

// Build the device blob image on the host: all tensors in their final kernel layouts.
static int pack_blob(const fw_config* cfg, const fw_weight* w, int nw, int compute_type,
                     std::vector<uint8_t>& blob) {
  const int d = cfg->d_model, nm = cfg->n_mels;
  const int c_pad = ((nm + 63) / 64) * 64;
  std::vector<PackItem> items;
  auto add_plain = [&](const std::string& name, std::initializer_list<int64_t> dims) -> int {
    const fw_weight* t = find_w(w, nw, name);
    int rc = expect_shape(t, name, dims);
    if (rc) return rc;
    PackItem it;
    it.name = name;
    it.ndim = (int)dims.size();
    int i = 0;
    for (int64_t x : dims) it.dims[i++] = x;
    for (; i < 4; ++i) it.dims[i] = 1;
    to_f16(t, it.data);
    items.push_back(std::move(it));
    return FW_OK;
  };
  // conv weights [out][in][3] -> GEMM form [out][tap*cin_pad + cin]
  auto add_conv = [&](const std::string& name, int cout, int cin, int cin_pad) -> int {
    const fw_weight* t = find_w(w, nw, name + ".w");
    int rc = expect_shape(t, name + ".w", {cout, cin, 3});
    if (rc) return rc;
    PackItem it;
    it.name = name + ".wg";
    it.ndim = 2;
    it.dims[0] = cout; it.dims[1] = 3 * cin_pad; it.dims[2] = it.dims[3] = 1;
    it.data.assign((size_t)cout * 3 * cin_pad, 0);
    for (int o = 0; o < cout; ++o)
      for (int c = 0; c < cin; ++c)
        for (int k = 0; k < 3; ++k)
          it.data[(size_t)o * 3 * cin_pad + (size_t)k * cin_pad + c] =
              f32_to_f16_bits(w_at(t, ((int64_t)o * cin + c) * 3 + k));
    items.push_back(std::move(it));
    return add_plain(name + ".b", {cout});
  };
  // LayerNorm folded into the consuming linear (decoder rows are few, so the LN kernel would be
  // pure launch overhead):   y = W (g*(x-mu)*rstd + b) + bias
  //                            = rstd * ( (W.g) x - mu * s1 ) + cf,   s1 = rowsum(W.g), cf = W b + bias
  // emits  <out>.wf [N][K] fp16 (W.g),  <out>.s1 [N] f32,  <out>.cf [N] f32
  auto add_folded = [&](const std::string& out, const std::string& wname, const std::string& bname,
                        const std::string& gname, const std::string& lbname, int N, int K) -> int {
    const fw_weight* W = find_w(w, nw, wname);
    const fw_weight* G = find_w(w, nw, gname);
    const fw_weight* LB = find_w(w, nw, lbname);
    const fw_weight* Bi = bname.empty() ? nullptr : find_w(w, nw, bname);
    int rc = expect_shape(W, wname, {N, K});
    if (rc) return rc;
    if ((rc = expect_shape(G, gname, {K}))) return rc;
    if ((rc = expect_shape(LB, lbname, {K}))) return rc;
    if (!bname.empty() && (rc = expect_shape(Bi, bname, {N}))) return rc;
    std::vector<float> g(K), lb(K);
    for (int k = 0; k < K; ++k) {
      g[k] = f16_bits_to_f32(f32_to_f16_bits(w_at(G, k)));
      lb[k] = f16_bits_to_f32(f32_to_f16_bits(w_at(LB, k)));
    }
    PackItem wf, s1, cf;
    wf.name = out + ".wf"; wf.ndim = 2; wf.dims[0] = N; wf.dims[1] = K; wf.dims[2] = wf.dims[3] = 1;
    wf.data.resize((size_t)N * K);
    s1.name = out + ".s1"; s1.ndim = 1; s1.dims[0] = N; s1.dims[1] = s1.dims[2] = s1.dims[3] = 1; s1.dtype = 0;
    cf.name = out + ".cf"; cf.ndim = 1; cf.dims[0] = N; cf.dims[1] = cf.dims[2] = cf.dims[3] = 1; cf.dtype = 0;
    s1.fdata.resize(N); cf.fdata.resize(N);
    for (int n = 0; n < N; ++n) {
      double a1 = 0.0, ac = 0.0;
      for (int k = 0; k < K; ++k) {
        const float wv = f16_bits_to_f32(f32_to_f16_bits(w_at(W, (int64_t)n * K + k)));
        const uint16_t wg = f32_to_f16_bits(wv * g[k]);
        wf.data[(size_t)n * K + k] = wg;
        a1 += (double)f16_bits_to_f32(wg);
        ac += (double)wv * (double)lb[k];
      }
      if (Bi) ac += (double)f16_bits_to_f32(f32_to_f16_bits(w_at(Bi, n)));
      s1.fdata[n] = (float)a1;
      cf.fdata[n] = (float)ac;
    }
    items.push_back(std::move(wf)); items.push_back(std::move(s1)); items.push_back(std::move(cf));
    return FW_OK;
  };
  // int8_float16 (K25, [CT2-ext] CTranslate2 convention): per output row n, scale = 127 / absmax(W[n,:]),
  // Wq = rint(W * scale) (round-half-even); the engine stores the DE-quantisation factor absmax / 127.
  // emits <out>.wq [N][K] int8 and <out>.ws [N] f32
  auto add_quant = [&](const std::string& out, const std::string& wname, int N, int K) -> int {
    const fw_weight* W = find_w(w, nw, wname);
    int rc = expect_shape(W, wname, {N, K});
    if (rc) return rc;
    PackItem wq, ws;
    wq.name = out + ".wq"; wq.ndim = 2; wq.dims[0] = N; wq.dims[1] = K; wq.dims[2] = wq.dims[3] = 1; wq.dtype = 2;
    ws.name = out + ".ws"; ws.ndim = 1; ws.dims[0] = N; ws.dims[1] = ws.dims[2] = ws.dims[3] = 1; ws.dtype = 0;
    wq.qdata.resize((size_t)N * K);
    ws.fdata.resize(N);
    std::vector<float> row(K);
    for (int n = 0; n < N; ++n) {
      float amax = 0.f;
      for (int k = 0; k < K; ++k) {
        row[k] = f16_bits_to_f32(f32_to_f16_bits(w_at(W, (int64_t)n * K + k)));
        amax = std::max(amax, fabsf(row[k]));
      }
      const float sc = amax > 0.f ? 127.0f / amax : 0.f;
      for (int k = 0; k < K; ++k) wq.qdata[(size_t)n * K + k] = (int8_t)lrintf(row[k] * sc);
      ws.fdata[n] = amax > 0.f ? amax / 127.0f : 1.0f;
    }
    items.push_back(std::move(wq)); items.push_back(std::move(ws));
    return FW_OK;
  };
  const bool i8 = compute_type == FW_COMPUTE_INT8_FLOAT16;
  // a linear layer: fp16 weight (+bias), or int8 weight + scale (+ fp16 bias) in int8_float16 mode
  auto add_linear = [&](const std::string& base, int N, int K) -> int {
    int rc = i8 ? add_quant(base, base + ".w", N, K) : add_plain(base + ".w", {N, K});
    if (rc) return rc;
    return add_plain(base + ".b", {N});
  };
  // a decoder linear fed by a LayerNorm: folded in fp16 mode; explicit LN (quantised output) in int8 mode
  // FWAMD_LN_UNFOLD (a diagnostic, default off: DESIGN.md section 5) set at PACK time: the fp16 blob then also carries
  // the plain weights + LayerNorms of qkv / cross.q / ffn1 / the vocabulary projection (+1 GB at large-v3) for the
  // explicit-LayerNorm evaluation order; without it only the folded forms travel
  const char* lu_env = getenv("FWAMD_LN_UNFOLD");
  const char* pp_env = getenv("FWAMD_PACK_PLAIN");   // (the probe packs once and creates models in all three orders)
  const bool with_plain = (lu_env && (lu_env[0] == '1' || lu_env[0] == '2') && !lu_env[1]) || (pp_env && pp_env[0] == '1');
  auto add_ln_linear = [&](const std::string& base, const std::string& ln, int N, int K) -> int {
    if (!i8) {
      int rc = add_folded(base, base + ".w", base + ".b", ln + ".g", ln + ".b", N, K);
      if (rc || !with_plain) return rc;
    }
    int rc = add_linear(base, N, K);
    if (rc) return rc;
    if ((rc = add_plain(ln + ".g", {K}))) return rc;
    return add_plain(ln + ".b", {K});
  };
  int rc;
#define TRY(x) do { rc = (x); if (rc) return rc; } while (0)
  TRY(add_conv("enc.conv1", d, nm, c_pad));
  TRY(add_conv("enc.conv2", d, d, d));
  TRY(add_plain("enc.pos", {cfg->n_audio_ctx, d}));
  char nb[96];
  for (int i = 0; i < cfg->n_enc_layers; ++i) {
    auto nmf = [&](const char* s) { snprintf(nb, sizeof(nb), "enc.%d.%s", i, s); return std::string(nb); };
    TRY(add_plain(nmf("ln1.g"), {d})); TRY(add_plain(nmf("ln1.b"), {d}));
    TRY(add_linear(nmf("attn.qkv"), 3 * d, d));
    TRY(add_linear(nmf("attn.out"), d, d));
    TRY(add_plain(nmf("ln2.g"), {d})); TRY(add_plain(nmf("ln2.b"), {d}));
    TRY(add_linear(nmf("ffn1"), 4 * d, d));
    TRY(add_linear(nmf("ffn2"), d, 4 * d));
  }
  TRY(add_plain("enc.ln_post.g", {d})); TRY(add_plain("enc.ln_post.b", {d}));
  TRY(add_plain("dec.tok_emb", {cfg->n_vocab, d}));
  if (i8) {
    // [CT2-ext] the token embedding shares the int8 projection weight: a lookup returns the de-quantised row
    PackItem& te = items.back();
    const int64_t K = d;
    for (int64_t n = 0; n < cfg->n_vocab; ++n) {
      float amax = 0.f;
      for (int64_t k = 0; k < K; ++k) amax = std::max(amax, fabsf(f16_bits_to_f32(te.data[n * K + k])));
      const float sc = amax > 0.f ? 127.0f / amax : 0.f, ds = amax > 0.f ? amax / 127.0f : 1.0f;
      for (int64_t k = 0; k < K; ++k)
        te.data[n * K + k] = f32_to_f16_bits((float)lrintf(f16_bits_to_f32(te.data[n * K + k]) * sc) * ds);
    }
  }
  TRY(add_plain("dec.pos", {cfg->n_text_ctx, d}));
  for (int i = 0; i < cfg->n_dec_layers; ++i) {
    auto nmf = [&](const char* s) { snprintf(nb, sizeof(nb), "dec.%d.%s", i, s); return std::string(nb); };
    // the three LayerNorms of a decoder block are folded into the linear that consumes them
    TRY(add_ln_linear(nmf("self.qkv"), nmf("ln1"), 3 * d, d));
    TRY(add_linear(nmf("self.out"), d, d));
    TRY(add_ln_linear(nmf("cross.q"), nmf("ln2"), d, d));
    TRY(add_linear(nmf("cross.kv"), 2 * d, d));
    TRY(add_linear(nmf("cross.out"), d, d));
    TRY(add_ln_linear(nmf("ffn1"), nmf("ln3"), 4 * d, d));
    TRY(add_linear(nmf("ffn2"), d, 4 * d));
  }
  if (i8) {
    TRY(add_quant("dec.logits", "dec.tok_emb", cfg->n_vocab, d));
    TRY(add_plain("dec.ln.g", {d})); TRY(add_plain("dec.ln.b", {d}));
  } else {
    TRY(add_folded("dec.logits", "dec.tok_emb", "", "dec.ln.g", "dec.ln.b", cfg->n_vocab, d));
    if (with_plain) {
      // the explicit order: final LayerNorm as a kernel, then the tied embedding itself (fragment-major copy)
      TRY(add_plain("dec.tok_emb", {cfg->n_vocab, d}));
      items.back().name = "dec.logits.wp";
      TRY(add_plain("dec.ln.g", {d})); TRY(add_plain("dec.ln.b", {d}));
    }
  }
#undef TRY

  // Fragment-major decoder linears (dec_kernels.hip): the 16 bytes lane l feeds to the MFMA for k-step ks of
  // column tile nt sit at ((nt*KS + ks)*64 + l)*16 B, i.e. element W[n][k] moves to
  //   fp16: ((n/16 * K/32 + k/32) * 64 + 16*((k/8)%4)  + n%16) * 8  + k%8     (32 elements per k-step)
  //   int8: ((n/16 * K/64 + k/64) * 64 + 16*((k/16)%4) + n%16) * 16 + k%16    (64 elements per k-step)
  // The six per-layer linears and the vocabulary projection are stored this way (the projection's N is padded
  // to a whole tile with zero rows); the big "many rows" GEMM (cross.kv) keeps [N][K].
  for (PackItem& it : items) {
    const std::string& nm = it.name;
    if (nm.compare(0, 4, "dec.") != 0 || it.ndim != 2 || it.dtype != (i8 ? 2 : 1)) continue;
    auto ends = [&](const char* suf) {
      const size_t n = strlen(suf);
      return nm.size() >= n && nm.compare(nm.size() - n, n, suf) == 0;
    };
    const bool hit = i8 ? (ends("self.qkv.wq") || ends("self.out.wq") || ends("cross.q.wq") || ends("cross.out.wq") ||
                           ends("ffn1.wq") || ends("ffn2.wq") || nm == "dec.logits.wq")
                        : (ends("self.qkv.wf") || ends("self.out.w") || ends("cross.q.wf") || ends("cross.out.w") ||
                           ends("ffn1.wf") || ends("ffn2.w") || nm == "dec.logits.wf" || ends("self.qkv.w") ||
                           ends("cross.q.w") || ends("ffn1.w") || nm == "dec.logits.wp");
    if (!hit) continue;
    const int64_t N = it.dims[0], K = it.dims[1], KE = i8 ? 64 : 32, OCT = KE / 4;
    const bool logits = nm.compare(0, 10, "dec.logits") == 0;
    if ((!logits && N % 32) || K % KE) {
      set_error("fragment-major packing needs N %% 32 == 0 and K %% %lld == 0 (%s is %lld x %lld)", (long long)KE,
                nm.c_str(), (long long)N, (long long)K);
      return FW_EINVAL;
    }
    const int64_t NP = (N + 15) / 16 * 16, KS = K / KE;
    it.dims[0] = NP;   // the stored extent (padding rows are zero)
    auto at = [&](int64_t n, int64_t k) {
      return (((n >> 4) * KS + k / KE) * 64 + ((k / OCT) & 3) * 16 + (n & 15)) * OCT + (k % OCT);
    };
    if (i8) {
      std::vector<int8_t> src;
      src.swap(it.qdata);
      it.qdata.assign((size_t)NP * K, 0);
      for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) it.qdata[at(n, k)] = src[n * K + k];
    } else {
      std::vector<uint16_t> src;
      src.swap(it.data);
      it.data.assign((size_t)NP * K, 0);
      for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) it.data[at(n, k)] = src[n * K + k];
    }
  }

  const int64_t hdr = (int64_t)sizeof(BlobHeader) + (int64_t)items.size() * sizeof(BlobEntry);
  int64_t off = (hdr + 255) / 256 * 256;
  std::vector<BlobEntry> entries(items.size());
  for (size_t i = 0; i < items.size(); ++i) {
    BlobEntry& e = entries[i];
    memset(&e, 0, sizeof(e));
    snprintf(e.name, sizeof(e.name), "%s", items[i].name.c_str());
    e.dtype = items[i].dtype;
    e.ndim = items[i].ndim;
    for (int k = 0; k < 4; ++k) e.dims[k] = items[i].dims[k];
    e.offset = off;
    e.nbytes = items[i].nbytes();
    off = (off + e.nbytes + 255) / 256 * 256;
  }
  blob.assign((size_t)off, 0);
  BlobHeader h;
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, FW_BLOB_MAGIC, 8);
  h.version = 1;
  h.n_tensors = (int32_t)items.size();
  h.total_bytes = off;
  h.compute_type = compute_type;
  h.reserved = 6;   // layout generation: 4 = decoder linears and the vocabulary projection fragment-major; 5 = fp16
                    // blobs carry the plain (LayerNorm-explicit) forms of qkv / cross.q / ffn1 / logits next to the folded
                    // ones; 6 = only when packed with FWAMD_LN_UNFOLD set
  h.cfg = *cfg;
  memcpy(blob.data(), &h, sizeof(h));
  memcpy(blob.data() + sizeof(h), entries.data(), entries.size() * sizeof(BlobEntry));
  for (size_t i = 0; i < items.size(); ++i)
    memcpy(blob.data() + entries[i].offset, items[i].bytes(), (size_t)entries[i].nbytes);
  return FW_OK;
}

// ---------------------------------------------------------------- mel constants
// get_mel_filters (feature_extractor.py:25-65), in double like numpy, stored float32.
static void compute_mel_filters(int n_mels, std::vector<float>& filt /* [n_mels][201] */) {
  const int n_fft = 400, nb = 201;
  const double sr = 16000.0;
  std::vector<double> fftfreqs(nb);
  const double val = 1.0 / (n_fft * (1.0 / sr));
  for (int k = 0; k < nb; ++k) fftfreqs[k] = k * val;
  const double max_mel = 45.245640471924965;
  const int n = n_mels + 2;
  std::vector<double> mels(n), freqs(n);
  const double step = (max_mel - 0.0) / (n - 1);
  for (int i = 0; i < n; ++i) mels[i] = i * step + 0.0;
  mels[n - 1] = max_mel;
  const double f_sp = 200.0 / 3;
  const double min_log_hz = 1000.0, min_log_mel = (min_log_hz - 0.0) / f_sp;
  const double logstep = log(6.4) / 27.0;
  for (int i = 0; i < n; ++i) {
    freqs[i] = 0.0 + f_sp * mels[i];
    if (mels[i] >= min_log_mel) freqs[i] = min_log_hz * exp(logstep * (mels[i] - min_log_mel));
  }
  filt.assign((size_t)n_mels * nb, 0.f);
  for (int i = 0; i < n_mels; ++i) {
    const double fd0 = freqs[i + 1] - freqs[i], fd1 = freqs[i + 2] - freqs[i + 1];
    const double enorm = 2.0 / (freqs[i + 2] - freqs[i]);
    for (int k = 0; k < nb; ++k) {
      const double lower = -(freqs[i] - fftfreqs[k]) / fd0;
      const double upper = (freqs[i + 2] - fftfreqs[k]) / fd1;
      double wv = std::max(0.0, std::min(lower, upper));
      wv *= enorm;
      filt[(size_t)i * nb + k] = (float)wv;
    }
  }
}

static int setup_logmel_consts(Model* m) {
  std::vector<float> consts(800);
  for (int j = 0; j < 400; ++j) consts[j] = (float)cos(2.0 * M_PI * (double)j / 400.0);
  // np.hanning(401)[:-1] : 0.5 - 0.5*cos(2*pi*n/(M-1)), M = 401
  for (int n = 0; n < 400; ++n) consts[400 + n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / 400.0));
  std::vector<float> filt;
  compute_mel_filters(m->cfg.n_mels, filt);
  m->lm_mel_pad = ((m->cfg.n_mels + 31) / 32) * 32;
  std::vector<float> filtT((size_t)224 * m->lm_mel_pad, 0.f);
  for (int i = 0; i < m->cfg.n_mels; ++i)
    for (int k = 0; k < 201; ++k) filtT[(size_t)k * m->lm_mel_pad + i] = filt[(size_t)i * 201 + k];
  int rc;
  if ((rc = dev_alloc_t(&m->lm_consts, 800))) return rc;
  if ((rc = dev_alloc_t(&m->lm_filtT, filtT.size()))) return rc;
  FW_HIP(hipMemcpy(m->lm_consts, consts.data(), 800 * sizeof(float), hipMemcpyHostToDevice));
  FW_HIP(hipMemcpy(m->lm_filtT, filtT.data(), filtT.size() * sizeof(float), hipMemcpyHostToDevice));
  return FW_OK;
}

// ---------------------------------------------------------------- model construction
std::mutex& g_models_mu_ref();
std::vector<Model*>& g_live_models_ref();
static const half_t* tptr(Model* m, const std::string& name) {
  auto it = m->tensors.find(name);
  return it == m->tensors.end() ? nullptr : reinterpret_cast<const half_t*>(it->second.ptr);
}

static int bind_weights(Model* m) {
  const fw_config& c = m->cfg;
  const int d = c.d_model;
  const bool i8 = m->compute_type == FW_COMPUTE_INT8_FLOAT16;
  m->c_pad = ((c.n_mels + 63) / 64) * 64;
  auto need = [&](const std::string& n, const half_t** out) -> int {
    *out = tptr(m, n);
    if (!*out) {
      set_error("weight blob lacks tensor '%s'", n.c_str());
      return FW_EINVAL;
    }
    return FW_OK;
  };
  auto need_f = [&](const std::string& n, const float** out) -> int {
    const half_t* p = nullptr;
    int r = need(n, &p);
    *out = reinterpret_cast<const float*>(p);
    return r;
  };
  // a linear: fp16 (<base>.w) or int8 (<base>.wq + <base>.ws); bias <base>.b.  rows [r0, r0+N) of the stored matrix
  auto linear = [&](const std::string& base, LinearW* L, int N, int K, int r0) -> int {
    int r;
    *L = LinearW();
    L->N = N; L->K = K;
    if (i8) {
      const half_t* q = nullptr;
      if ((r = need(base + ".wq", &q))) return r;
      L->wq = reinterpret_cast<const int8_t*>(q) + (size_t)r0 * K;
      if ((r = need_f(base + ".ws", &L->wscale))) return r;
      L->wscale += r0;
    } else {
      if ((r = need(base + ".w", &L->w))) return r;
      L->w += (size_t)r0 * K;
    }
    if ((r = need(base + ".b", &L->b))) return r;
    L->b += r0;
    return FW_OK;
  };
  auto folded = [&](const std::string& base, LinearW* L, int N, int K) -> int {
    int r;
    *L = LinearW();
    if ((r = need(base + ".wf", &L->w))) return r;
    if ((r = need_f(base + ".s1", &L->s1))) return r;
    if ((r = need_f(base + ".cf", &L->cf))) return r;
    L->N = N; L->K = K;
    return FW_OK;
  };
  auto ln = [&](const std::string& base, LNW* L) -> int {
    int r;
    if ((r = need(base + ".g", &L->g))) return r;
    return need(base + ".b", &L->b);
  };
  int rc;
#define TRY(x) do { if ((rc = (x))) return rc; } while (0)
  TRY(need("enc.conv1.wg", &m->conv1.w)); TRY(need("enc.conv1.b", &m->conv1.b));
  m->conv1.N = d; m->conv1.K = 3 * m->c_pad;
  TRY(need("enc.conv2.wg", &m->conv2.w)); TRY(need("enc.conv2.b", &m->conv2.b));
  m->conv2.N = d; m->conv2.K = 3 * d;
  TRY(need("enc.pos", &m->enc_pos));
  m->enc.resize(c.n_enc_layers);
  char nb[96];
  for (int i = 0; i < c.n_enc_layers; ++i) {
    EncLayerW& L = m->enc[i];
    auto nm = [&](const char* s) { snprintf(nb, sizeof(nb), "enc.%d.%s", i, s); return std::string(nb); };
    TRY(ln(nm("ln1"), &L.ln1));
    TRY(linear(nm("attn.qkv"), &L.qk, 2 * d, d, 0));
    TRY(linear(nm("attn.qkv"), &L.v, d, d, 2 * d));
    TRY(linear(nm("attn.out"), &L.out, d, d, 0));
    TRY(ln(nm("ln2"), &L.ln2));
    TRY(linear(nm("ffn1"), &L.ffn1, 4 * d, d, 0));
    TRY(linear(nm("ffn2"), &L.ffn2, d, 4 * d, 0));
  }
  TRY(ln("enc.ln_post", &m->enc_ln_post));
  TRY(need("dec.tok_emb", &m->tok_emb)); TRY(need("dec.pos", &m->dec_pos));
  m->dec.resize(c.n_dec_layers);
  m->has_plain = !i8 && tptr(m, "dec.logits.wp") != nullptr;
  for (int i = 0; i < c.n_dec_layers; ++i) {
    DecLayerW& L = m->dec[i];
    auto nm = [&](const char* s) { snprintf(nb, sizeof(nb), "dec.%d.%s", i, s); return std::string(nb); };
    if (i8) {
      TRY(ln(nm("ln1"), &L.ln1)); TRY(ln(nm("ln2"), &L.ln2)); TRY(ln(nm("ln3"), &L.ln3));
      TRY(linear(nm("self.qkv"), &L.qkv, 3 * d, d, 0));
      TRY(linear(nm("cross.q"), &L.cq, d, d, 0));
      TRY(linear(nm("ffn1"), &L.ffn1, 4 * d, d, 0));
    } else {
      TRY(folded(nm("self.qkv"), &L.qkv, 3 * d, d));
      TRY(folded(nm("cross.q"), &L.cq, d, d));
      TRY(folded(nm("ffn1"), &L.ffn1, 4 * d, d));
      if (m->has_plain) {   // packed with FWAMD_LN_UNFOLD: the explicit-LayerNorm forms travel too
        TRY(ln(nm("ln1"), &L.ln1)); TRY(ln(nm("ln2"), &L.ln2)); TRY(ln(nm("ln3"), &L.ln3));
        TRY(linear(nm("self.qkv"), &L.qkv_p, 3 * d, d, 0));
        TRY(linear(nm("cross.q"), &L.cq_p, d, d, 0));
        TRY(linear(nm("ffn1"), &L.ffn1_p, 4 * d, d, 0));
      }
    }
    TRY(linear(nm("self.out"), &L.out, d, d, 0));
    TRY(linear(nm("cross.kv"), &L.ck, d, d, 0));
    TRY(linear(nm("cross.kv"), &L.cv, d, d, d));
    TRY(linear(nm("cross.out"), &L.cout, d, d, 0));
    TRY(linear(nm("ffn2"), &L.ffn2, d, 4 * d, 0));
  }
  if (i8) {
    const half_t* q = nullptr;
    m->logits = LinearW();
    TRY(need("dec.logits.wq", &q));
    m->logits.wq = reinterpret_cast<const int8_t*>(q);
    TRY(need_f("dec.logits.ws", &m->logits.wscale));
    m->logits.N = c.n_vocab; m->logits.K = d;
    TRY(ln("dec.ln", &m->dec_ln));
  } else {
    TRY(folded("dec.logits", &m->logits, c.n_vocab, d));
    m->logits_p = LinearW();
    if (m->has_plain) {
      TRY(need("dec.logits.wp", &m->logits_p.w));
      m->logits_p.N = c.n_vocab; m->logits_p.K = d;
      TRY(ln("dec.ln", &m->dec_ln));
    }
  }
#undef TRY
  return FW_OK;
}

static int alloc_workspaces(Model* m) {
  const fw_config& c = m->cfg;
  const size_t B = m->max_batch, d = c.d_model, T = c.n_audio_ctx;
  m->t_pad = ((c.n_audio_ctx + 63) / 64) * 64;  // 1536
  int rc;
#define A(p, n) do { if ((rc = dev_alloc_t(&(p), (n)))) return rc; } while (0)
  A(m->ws_offsets, B + 1);
  A(m->ws_chunk_max, B);
  A(m->ws_nframes, B);
  m->ws_raw_cap = (int64_t)B * c.n_mels * 3008;
  A(m->ws_raw, (size_t)m->ws_raw_cap);
  A(m->ws_feat32, B * c.n_mels * 3000);
  A(m->ws_mel_cl, B * 3002 * m->c_pad);
  A(m->ws_conv1, B * 3002 * d);
  A(m->ws_x, B * T * d);
  A(m->ws_x2, B * T * d);
  A(m->ws_xn, B * T * d);
  A(m->ws_qk, B * T * 2 * d);
  A(m->ws_vt, B * d * m->t_pad);
  A(m->ws_att, B * T * d);
  A(m->ws_ffn, B * T * 4 * d);
  if (m->compute_type == FW_COMPUTE_INT8_FLOAT16) {
    A(m->ws_xq, B * T * 4 * d);
    A(m->ws_xs, B * T);
  }
#undef A
  // conv padding rows / V^T time padding must be zero and are never written again
  FW_HIP(hipMemset(m->ws_conv1, 0, B * 3002 * d * sizeof(half_t)));
  FW_HIP(hipMemset(m->ws_vt, 0, B * d * m->t_pad * sizeof(half_t)));
  FW_HIP(hipMemset(m->ws_mel_cl, 0, B * 3002 * m->c_pad * sizeof(half_t)));
  return FW_OK;
}

static int model_from_blob(const void* blob_dev, int64_t blob_bytes, bool owned, int device, int max_batch,
                           int max_beam, fw_model** out, bool decoder_lane = false) {
  FW_CHECK_ARG(max_batch >= 1 && max_batch <= 256, "max_batch must be in [1,256], got %d", max_batch);
  FW_CHECK_ARG(max_beam >= 1 && max_beam <= 16, "max_beam must be in [1,16], got %d", max_beam);
  FW_CHECK_ARG(max_batch * max_beam <= 2048, "max_batch * max_beam must be <= 2048 decoder rows, got %d", max_batch * max_beam);
  FW_CHECK_ARG(blob_bytes >= (int64_t)sizeof(BlobHeader), "weight blob too small");
  BlobHeader h;
  FW_HIP(hipMemcpy(&h, blob_dev, sizeof(h), hipMemcpyDeviceToHost));
  FW_CHECK_ARG(memcmp(h.magic, FW_BLOB_MAGIC, 8) == 0 && h.version == 1, "bad weight blob magic/version");
  FW_CHECK_ARG(h.total_bytes <= blob_bytes, "weight blob truncated (%lld > %lld)", (long long)h.total_bytes,
               (long long)blob_bytes);
  int rc = check_config(&h.cfg);
  if (rc) return rc;
  FW_CHECK_ARG(h.compute_type == FW_COMPUTE_FLOAT16 || h.compute_type == FW_COMPUTE_INT8_FLOAT16,
               "unsupported compute type %d", h.compute_type);
  std::vector<BlobEntry> entries(h.n_tensors);
  FW_HIP(hipMemcpy(entries.data(), reinterpret_cast<const char*>(blob_dev) + sizeof(h),
                   entries.size() * sizeof(BlobEntry), hipMemcpyDeviceToHost));
  fw_model* fm = new fw_model();
  Model* m = &fm->impl;
  m->cfg = h.cfg;
  m->compute_type = h.compute_type;
  if (h.reserved != 6) {
    delete fm;
    set_error("weight blob was packed by an older libfwamd (layout generation %d, expected 6): repack it", h.reserved);
    return FW_EINVAL;
  }
  m->device = device;
  m->max_batch = max_batch;
  m->max_beam = max_beam;
  m->blob = const_cast<void*>(blob_dev);
  m->blob_owned = owned;
  m->blob_bytes = blob_bytes;
  for (auto& e : entries) {
    DevTensor t;
    t.ptr = reinterpret_cast<char*>(m->blob) + e.offset;
    t.dtype = e.dtype;
    t.ndim = e.ndim;
    for (int k = 0; k < 4; ++k) t.dims[k] = e.dims[k];
    m->tensors[e.name] = t;
  }
  auto fail = [&](int code) { fw_model_free(fm); return code; };
  hipError_t he = create_stream(&m->stream, decoder_lane ? "DEC" : "ENC");
  if (he != hipSuccess) {
    set_error("hipStreamCreate failed: %s", hipGetErrorString(he));
    return fail(FW_ENODEV);
  }
  if ((rc = bind_weights(m))) return fail(rc);
  m->is_lane = decoder_lane;
  {
    const char* lu = getenv("FWAMD_LN_UNFOLD");   // fp16 evaluation order of the decoder LayerNorms (engine.h)
    if (lu && lu[0] >= '0' && lu[0] <= '2' && !lu[1]) m->ln_unfold = lu[0] - '0';
    if (m->ln_unfold && m->compute_type != FW_COMPUTE_INT8_FLOAT16 && !m->has_plain) {
      set_error("FWAMD_LN_UNFOLD=%d needs a weight blob packed with FWAMD_LN_UNFOLD set (this one carries the folded forms only)",
                m->ln_unfold);
      return fail(FW_EINVAL);
    }
  }
  if (!decoder_lane) {   // (a decode lane has no front end and no encoder: weights, a stream, a decode workspace)
    if ((rc = setup_logmel_consts(m))) return fail(rc);
    if ((rc = alloc_workspaces(m))) return fail(rc);
  }
  m->decode_batch = max_batch;   // the decode workspace itself is created on first use (decoder.hip)
  he = hipDeviceSynchronize();
  if (he != hipSuccess) {
    set_error("device sync after model setup failed: %s", hipGetErrorString(he));
    return fail(FW_ENODEV);
  }
  {
    // a model on a borrowed blob (fw_model_create_from_blob_dev on fw_model_blob's pointer) keeps the owner alive:
    // fw_model_free(owner) is deferred until its last dependent is gone
    std::lock_guard<std::mutex> lk(g_models_mu_ref());
    if (!owned && !decoder_lane)
      for (Model* o : g_live_models_ref())
        if (o->blob == m->blob && o->blob_owned) { m->blob_owner = o; o->dependents += 1; break; }
    if (!decoder_lane) g_live_models_ref().push_back(m);   // (a lane lives and dies with its primary)
  }
  m->self = fm;
  *out = fm;
  return FW_OK;
}

// CU mask of n CUs (a multiple of 32), the same number in every XCD.  Whether bit i of a HIP CU mask is CU (i / 8) of XCD
// (i % 8) — the order the KFD documents for multi-XCC parts — or CU (i % 32) of XCD (i / 32) does not matter here: with
// a = i % 8, b = i / 8 the rule (a + b) % 8 < n / 32 selects n / 8 CUs of every XCD under either reading (for a fixed a the
// 32 values of b hit every residue four times; for the four b of one XCD of the second reading every a-residue once each).
// invert: the complement (256 - n CUs, exactly the ones the plain mask leaves out) — the two sides of a partition.
static void balanced_cu_mask(int n_cus, bool invert, uint32_t mask[8]) {
  const int k = n_cus / 32;
  for (int w = 0; w < 8; ++w) mask[w] = 0;
  for (int i = 0; i < 256; ++i) {
    const bool in = ((i % 8) + (i / 8)) % 8 < k;
    if (in != invert) mask[i >> 5] |= 1u << (i & 31);
  }
}

// role "ENC" (a worker's encoder stream) or "DEC" (a decode lane's streams).  Environment, read per stream creation:
//   FWAMD_<role>_STREAM_PRIO = high | low       stream priority (measured: no effect, profiles/r05_ab_stream_priority.jsonl)
//   FWAMD_<role>_CUS = n | -n                   CU-masked stream: n CUs (multiple of 32, the same share of every XCD), or
//                                               with a minus sign the 256 - n CUs the mask of n leaves out — so
//                                               FWAMD_DEC_CUS=96 FWAMD_ENC_CUS=-96 is a disjoint 96 / 160 partition
//                                               (profiles/r06_ab_overlap.jsonl).  CU-mask streams are BLOCKING streams: the
//                                               hot path issues nothing on the default stream (results are copied on the decode stream), only set-up copies wait.
hipError_t create_stream(hipStream_t* st, const char* role) {
  char name[64];
  snprintf(name, sizeof(name), "FWAMD_%s_CUS", role);
  if (const char* c = getenv(name)) {
    const int v = atoi(c), n = v < 0 ? -v : v;
    if (n >= 32 && n <= 224 && n % 32 == 0) {
      uint32_t mask[8];
      balanced_cu_mask(n, v < 0, mask);
      return hipExtStreamCreateWithCUMask(st, 8, mask);
    }
  }
  snprintf(name, sizeof(name), "FWAMD_%s_STREAM_PRIO", role);
  const char* e = getenv(name);
  if (e && (!strcmp(e, "high") || !strcmp(e, "low"))) {
    int least = 0, greatest = 0;   // numerically: greatest priority <= least priority
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
      return hipStreamCreateWithPriority(st, hipStreamNonBlocking, e[0] == 'h' ? greatest : least);
  }
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

// ---------------------------------------------------------------- layers
int run_linear(Model* m, const LinearW& L, const half_t* A, int64_t lda, int64_t a_bs, half_t* C, int64_t ldc,
               int64_t c_bs, const half_t* res, int64_t ldr, int64_t r_bs, int M, int batch, int act, bool trans,
               int head_rows, hipStream_t st) {
  fwk::GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.a_bstride = a_bs;
  p.W = L.w; p.ldw = L.K;
  p.bias = L.b;
  p.res = res; p.ldr = ldr; p.r_bstride = r_bs;
  p.C = C; p.ldc = ldc; p.c_bstride = c_bs;
  p.M = M; p.N = L.N; p.K = L.K;
  p.act = act;
  p.head_rows = head_rows;
  if (fwk::launch_gemm(st ? st : m->stream, p, batch, trans) != 0) {
    set_error("gemm: unsupported shape M=%d N=%d K=%d lda=%lld", M, L.N, L.K, (long long)lda);
    return FW_ERUNTIME;
  }
  return FW_OK;
}

int run_linear_layers(Model* m, const LinearW& L0, int n_layers, int64_t w_lstride, int64_t b_lstride, const half_t* A,
                      int64_t lda, int64_t a_bs, half_t* C, int64_t ldc, int64_t c_bs, int64_t c_lstride, int M, int batch,
                      bool trans, int head_rows, hipStream_t st) {
  fwk::GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.lda = lda; p.a_bstride = a_bs;
  p.W = L0.w; p.ldw = L0.K;
  p.bias = L0.b;
  p.C = C; p.ldc = ldc; p.c_bstride = c_bs;
  p.M = M; p.N = L0.N; p.K = L0.K;
  p.head_rows = head_rows;
  p.n_layers = n_layers; p.w_lstride = w_lstride; p.bias_lstride = b_lstride; p.c_lstride = c_lstride;
  if (fwk::launch_gemm(st ? st : m->stream, p, batch, trans) != 0) {
    set_error("layered gemm: unsupported shape M=%d N=%d K=%d layers=%d", M, L0.N, L0.K, n_layers);
    return FW_ERUNTIME;
  }
  return FW_OK;
}

// int8_float16 linear on "many rows" (K25): per-row dynamic quantisation of A (optionally through the
// LayerNorm that feeds it), int8 x int8 -> int32 MFMA GEMM, de-quantising epilogue.  A == nullptr reuses the
// rows quantised by the previous call (fused Q|K and V projections share their input).
int run_linear_i8(Model* m, const LinearW& L, const half_t* A, const LNW* ln, half_t* C, int64_t ldc, int64_t c_bs,
                  const half_t* res, int64_t ldr, int64_t r_bs, int M, int batch, int act, bool trans,
                  int head_rows, hipStream_t st, int8_t* xq, float* xs) {
  if (!st) st = m->stream;
  if (!xq) { xq = m->ws_xq; xs = m->ws_xs; }
  if (A) fwk::launch_quant_rows(st, A, L.K, ln ? ln->g : nullptr, ln ? ln->b : nullptr, xq, xs, batch * M, L.K);
  fwk::GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = reinterpret_cast<const half_t*>(xq); p.lda = L.K; p.a_bstride = (int64_t)M * L.K;
  p.W = reinterpret_cast<const half_t*>(L.wq); p.ldw = L.K;
  p.bias = L.b;
  p.res = res; p.ldr = ldr; p.r_bstride = r_bs;
  p.C = C; p.ldc = ldc; p.c_bstride = c_bs;
  p.M = M; p.N = L.N; p.K = L.K;
  p.act = act;
  p.head_rows = head_rows;
  p.a_scale = xs; p.as_bstride = M; p.w_scale = L.wscale;
  if (fwk::launch_gemm(st, p, batch, trans) != 0) {
    set_error("int8 gemm: unsupported shape M=%d N=%d K=%d", M, L.N, L.K);
    return FW_ERUNTIME;
  }
  return FW_OK;
}

int run_encoder(Model* m, int B, half_t* out) {
  const fw_config& c = m->cfg;
  const int d = c.d_model, T = c.n_audio_ctx, H = c.n_heads;
  const int64_t xs = (int64_t)T * d;
  int rc;
  {
    // conv1: k=3, s=1 over the channel-last mel image; GELU; output rows 1..3000 of the padded conv1 image
    ProfScope ps(m, PF_ENC_GEMM, 2.0 * B * 3000.0 * d * (3.0 * c.n_mels), 0);
    if ((rc = run_linear(m, m->conv1, m->ws_mel_cl, m->c_pad, (int64_t)3002 * m->c_pad, m->ws_conv1 + d, d,
                         (int64_t)3002 * d, nullptr, 0, 0, 3000, B, 1, false)))
      return rc;
  }
  {
    // conv2: k=3, s=2: row t reads padded rows 2t..2t+2 (lda = 2d, K = 3d); GELU; + positional embedding
    ProfScope ps(m, PF_ENC_GEMM, 2.0 * B * (double)T * d * (3.0 * d), 0);
    if ((rc = run_linear(m, m->conv2, m->ws_conv1, 2 * d, (int64_t)3002 * d, m->ws_x, d, xs, m->enc_pos, d, 0, T, B,
                         1, false)))
      return rc;
  }
  half_t* x = m->ws_x;
  half_t* x2 = m->ws_x2;
  const bool i8 = m->compute_type == FW_COMPUTE_INT8_FLOAT16;
  for (int l = 0; l < c.n_enc_layers; ++l) {
    const EncLayerW& L = m->enc[l];
    if (i8) {
      // int8_float16: LayerNorm fused into the quantiser; every linear input is quantised per row
      {
        ProfScope ps(m, PF_ENC_GEMM, 2.0 * B * (double)T * d * (3.0 * d), 0);
        if ((rc = run_linear_i8(m, L.qk, x, &L.ln1, m->ws_qk, 2 * d, (int64_t)T * 2 * d, nullptr, 0, 0, T, B, 0, false,
                                0)))
          return rc;
        if ((rc = run_linear_i8(m, L.v, nullptr, nullptr, m->ws_vt, m->t_pad, (int64_t)d * m->t_pad, nullptr, 0, 0, T,
                                B, 0, true, 0)))
          return rc;
      }
      {
        ProfScope ps(m, PF_ENC_ATTN, 4.0 * B * (double)T * T * d, 0);
        fwk::launch_attn_enc(m->stream, m->ws_qk, m->ws_qk + d, 2 * d, (int64_t)T * 2 * d, m->ws_vt, m->t_pad,
                             (int64_t)d * m->t_pad, m->ws_att, d, xs, B, H, T);
      }
      {
        ProfScope ps(m, PF_ENC_GEMM, 2.0 * B * (double)T * d * (9.0 * d), 0);
        if ((rc = run_linear_i8(m, L.out, m->ws_att, nullptr, x2, d, xs, x, d, xs, T, B, 0, false, 0))) return rc;
        if ((rc = run_linear_i8(m, L.ffn1, x2, &L.ln2, m->ws_ffn, 4 * d, (int64_t)T * 4 * d, nullptr, 0, 0, T, B, 1,
                                false, 0)))
          return rc;
        if ((rc = run_linear_i8(m, L.ffn2, m->ws_ffn, nullptr, x, d, xs, x2, d, xs, T, B, 0, false, 0))) return rc;
      }
      continue;
    }
    {
      ProfScope ps(m, PF_ENC_LN, 0, 4.0 * B * T * d);
      fwk::launch_layernorm(m->stream, x, L.ln1.g, L.ln1.b, m->ws_xn, B * T, d);
    }
    {
      ProfScope ps(m, PF_ENC_GEMM, 2.0 * B * (double)T * d * (3.0 * d), 0);
      if ((rc = run_linear(m, L.qk, m->ws_xn, d, xs, m->ws_qk, 2 * d, (int64_t)T * 2 * d, nullptr, 0, 0, T, B, 0,
                           false)))
        return rc;
      if ((rc = run_linear(m, L.v, m->ws_xn, d, xs, m->ws_vt, m->t_pad, (int64_t)d * m->t_pad, nullptr, 0, 0, T, B, 0,
                           true)))
        return rc;
    }
    {
      ProfScope ps(m, PF_ENC_ATTN, 4.0 * B * (double)T * T * d, 0);
      fwk::launch_attn_enc(m->stream, m->ws_qk, m->ws_qk + d, 2 * d, (int64_t)T * 2 * d, m->ws_vt, m->t_pad,
                           (int64_t)d * m->t_pad, m->ws_att, d, xs, B, H, T);
    }
    {
      ProfScope ps(m, PF_ENC_GEMM, 2.0 * B * (double)T * d * d, 0);
      if ((rc = run_linear(m, L.out, m->ws_att, d, xs, x2, d, xs, x, d, xs, T, B, 0, false))) return rc;
    }
    {
      ProfScope ps(m, PF_ENC_LN, 0, 4.0 * B * T * d);
      fwk::launch_layernorm(m->stream, x2, L.ln2.g, L.ln2.b, m->ws_xn, B * T, d);
    }
    {
      ProfScope ps(m, PF_ENC_GEMM, 2.0 * B * (double)T * d * (8.0 * d), 0);
      if ((rc = run_linear(m, L.ffn1, m->ws_xn, d, xs, m->ws_ffn, 4 * d, (int64_t)T * 4 * d, nullptr, 0, 0, T, B, 1,
                           false)))
        return rc;
      if ((rc = run_linear(m, L.ffn2, m->ws_ffn, 4 * d, (int64_t)T * 4 * d, x, d, xs, x2, d, xs, T, B, 0, false)))
        return rc;
    }
  }
  {
    ProfScope ps(m, PF_ENC_LN, 0, 4.0 * B * T * d);
    fwk::launch_layernorm(m->stream, x, m->enc_ln_post.g, m->enc_ln_post.b, out, B * T, d);
  }
  hipError_t he = hipGetLastError();
  if (he != hipSuccess) {
    set_error("encoder launch failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  return FW_OK;
}

static int new_tensor(Model* m, int B, fw_tensor** out) {
  fw_tensor* t = new fw_tensor();
  t->impl.owner = m;
  t->impl.B = B;
  t->impl.T = m->cfg.n_audio_ctx;
  t->impl.D = m->cfg.d_model;
  t->impl.id = next_tensor_id();
  // encoder outputs come from a per-model pool (every buffer holds max_batch rows): hipMalloc/hipFree
  // synchronise the whole device and would serialise the replicas that share a GPU
  {
    std::lock_guard<std::mutex> lk(m->pool_mu);
    if (!m->enc_pool.empty()) {
      t->impl.data = m->enc_pool.back();
      m->enc_pool.pop_back();
    }
  }
  if (!t->impl.data) {
    int rc = dev_alloc_t(&t->impl.data, (size_t)m->max_batch * t->impl.T * t->impl.D);
    if (rc) {
      delete t;
      return rc;
    }
  }
  *out = t;
  return FW_OK;
}

static std::mutex g_models_mu;
static std::vector<Model*> g_live_models;
std::mutex& g_models_mu_ref() { return g_models_mu; }
std::vector<Model*>& g_live_models_ref() { return g_live_models; }

static int ensure_pcm(Model* m, int64_t n) {
  if (n <= m->ws_pcm_cap) return FW_OK;
  if (m->ws_pcm) (void)hipFree(m->ws_pcm);
  m->ws_pcm = nullptr;
  m->ws_pcm_cap = 0;
  int rc = dev_alloc_t(&m->ws_pcm, (size_t)n);
  if (rc) return rc;
  m->ws_pcm_cap = n;
  return FW_OK;
}

static int check_offsets(const int64_t* offsets, int B, int64_t* total, int* max_frames) {
  FW_CHECK_ARG(offsets && offsets[0] == 0, "offsets[0] must be 0");
  int mf = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t n = offsets[b + 1] - offsets[b];
    FW_CHECK_ARG(n >= 0 && n < (int64_t)1 << 30, "chunk %d has invalid length %lld", b, (long long)n);
    mf = std::max(mf, (int)((n + 160) / 160));
  }
  *total = offsets[B];
  *max_frames = mf;
  return FW_OK;
}

// raw buffer able to hold [B][n_mels][stride]
static int ensure_raw(Model* m, int B, int stride) {
  const int64_t need = (int64_t)B * m->cfg.n_mels * stride;
  if (need <= m->ws_raw_cap) return FW_OK;
  (void)hipFree(m->ws_raw);
  m->ws_raw = nullptr;
  m->ws_raw_cap = 0;
  int rc = dev_alloc_t(&m->ws_raw, (size_t)need);
  if (rc) return rc;
  m->ws_raw_cap = need;
  return FW_OK;
}

// log-mel for B ragged chunks whose PCM already sits at pcm_dev; writes ws_feat32 and/or ws_mel_cl
static int logmel_batch(Model* m, const float* pcm_dev, const int64_t* offsets_host, int B, bool want_f32,
                        bool want_cl) {
  int64_t total;
  int max_frames;
  int rc = check_offsets(offsets_host, B, &total, &max_frames);
  if (rc) return rc;
  const int stride = std::max(3008, (max_frames + 7) / 8 * 8);
  if ((rc = ensure_raw(m, B, stride))) return rc;
  FW_HIP(hipMemcpyAsync(m->ws_offsets, offsets_host, (B + 1) * sizeof(int64_t), hipMemcpyHostToDevice, m->stream));
  {
    ProfScope ps(m, PF_LOGMEL, 0, (double)total * 4 + (double)B * m->cfg.n_mels * 3000 * (want_f32 ? 4 : 2));
    fwk::launch_logmel(m->stream, pcm_dev, m->ws_offsets, B, max_frames, m->lm_consts, m->lm_filtT, m->lm_mel_pad,
                       m->cfg.n_mels, m->ws_raw, (int64_t)m->cfg.n_mels * stride, stride, m->ws_chunk_max, 1, 3000,
                       want_f32 ? m->ws_feat32 : nullptr, want_cl ? m->ws_mel_cl : nullptr, m->c_pad, m->ws_nframes);
  }
  hipError_t he = hipGetLastError();
  if (he != hipSuccess) {
    set_error("logmel launch failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  return FW_OK;
}

}  // namespace fw

using namespace fw;

// ================================================================== C ABI
extern "C" {

const char* fw_last_error(void) { return g_err; }
int32_t fw_abi_version(void) { return FW_ABI_VERSION; }

int32_t fw_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t fw_model_create(const fw_config* cfg, const fw_weight* weights, int32_t n_weights, int32_t compute_type,
                        int32_t device_index, int32_t max_batch, int32_t max_beam, fw_model** out) {
  FW_CHECK_ARG(cfg && weights && out, "null argument");
  FW_CHECK_ARG(compute_type == FW_COMPUTE_FLOAT16 || compute_type == FW_COMPUTE_INT8_FLOAT16,
               "unsupported compute_type %d", compute_type);
  int rc = check_config(cfg);
  if (rc) return rc;
  const int ndev = fw_device_count();
  if (ndev <= 0) {
    set_error("no HIP device visible: libfwamd has no CPU fallback");
    return FW_ENODEV;
  }
  FW_CHECK_ARG(device_index >= 0 && device_index < ndev, "device_index %d out of range (%d devices)", device_index,
               ndev);
  FW_HIP(hipSetDevice(device_index));
  std::vector<uint8_t> blob;
  if ((rc = pack_blob(cfg, weights, n_weights, compute_type, blob))) return rc;
  void* dblob = nullptr;
  if ((rc = dev_alloc(&dblob, blob.size()))) return rc;
  hipError_t he = hipMemcpy(dblob, blob.data(), blob.size(), hipMemcpyHostToDevice);
  if (he != hipSuccess) {
    (void)hipFree(dblob);
    set_error("weight upload failed: %s", hipGetErrorString(he));
    return FW_ENODEV;
  }
  rc = model_from_blob(dblob, (int64_t)blob.size(), true, device_index, max_batch, max_beam, out);
  return rc;
}

int32_t fw_pack_blob_size(const fw_config* cfg, const fw_weight* weights, int32_t n_weights, int32_t compute_type,
                          int64_t* size_out, void** handle_out) {
  FW_CHECK_ARG(cfg && weights && size_out && handle_out, "null argument");
  int rc = check_config(cfg);
  if (rc) return rc;
  auto* blob = new std::vector<uint8_t>();
  if ((rc = pack_blob(cfg, weights, n_weights, compute_type, *blob))) {
    delete blob;
    return rc;
  }
  *size_out = (int64_t)blob->size();
  *handle_out = blob;
  return FW_OK;
}
int32_t fw_pack_blob_copy(void* handle, void* dst, int64_t dst_bytes) {
  auto* blob = reinterpret_cast<std::vector<uint8_t>*>(handle);
  FW_CHECK_ARG(blob && dst && dst_bytes >= (int64_t)blob->size(), "bad blob copy arguments");
  memcpy(dst, blob->data(), blob->size());
  return FW_OK;
}
void fw_pack_blob_free(void* handle) { delete reinterpret_cast<std::vector<uint8_t>*>(handle); }

int32_t fw_model_create_from_blob_dev(const fw_config* cfg, const void* blob_dev, int64_t blob_bytes,
                                      int32_t compute_type, int32_t device_index, int32_t max_batch,
                                      int32_t max_beam, fw_model** out) {
  (void)cfg;
  (void)compute_type;
  FW_CHECK_ARG(blob_dev && out, "null argument");
  const int ndev = fw_device_count();
  if (ndev <= 0) {
    set_error("no HIP device visible: libfwamd has no CPU fallback");
    return FW_ENODEV;
  }
  FW_CHECK_ARG(device_index >= 0 && device_index < ndev, "device_index %d out of range", device_index);
  FW_HIP(hipSetDevice(device_index));
  return model_from_blob(blob_dev, blob_bytes, false, device_index, max_batch, max_beam, out);
}

void fw_model_free(fw_model* fm) {
  if (!fm) return;
  Model* m = &fm->impl;
  Model* release[2] = {nullptr, nullptr};
  {
    std::lock_guard<std::mutex> lk(g_models_mu);
    if (m->dependents > 0) {   // workers still use this model's weights / decode workspace: freed with the last one
      m->free_deferred = true;
      return;
    }
    g_live_models.erase(std::remove(g_live_models.begin(), g_live_models.end(), m), g_live_models.end());
    release[0] = m->blob_owner;
    release[1] = m->decoder;
  }
  (void)hipSetDevice(m->device);
  for (Model*& l : m->xlanes) {
    if (l) fw_model_free(static_cast<fw_model*>(l->self));
    l = nullptr;
  }
  m->lane1 = nullptr;
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  if (m->dec_stream) (void)hipStreamSynchronize(m->dec_stream);
  gen_workspace_free(m);
  cross_pool_free(m);       // (a lane has none of its own: it reads the primary's)
  if (m->dec_stream) (void)hipStreamDestroy(m->dec_stream);
  for (half_t* p : m->enc_pool) (void)hipFree(p);
  m->enc_pool.clear();
  void* ptrs[] = {m->lm_consts, m->lm_filtT, m->ws_pcm, m->ws_offsets, m->ws_raw, m->ws_chunk_max, m->ws_nframes,
                  m->ws_feat32, m->ws_mel_cl, m->ws_conv1, m->ws_x, m->ws_x2, m->ws_xn, m->ws_qk, m->ws_vt,
                  m->ws_att, m->ws_ffn, m->ws_xq, m->ws_xs};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (m->blob && m->blob_owned) (void)hipFree(m->blob);
  for (auto& p : m->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : m->ev_pool) (void)hipEventDestroy(e);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete fm;
  for (Model* t : release) {
    if (!t) continue;
    bool free_now = false;
    {
      std::lock_guard<std::mutex> lk(g_models_mu);
      t->dependents -= 1;
      free_now = t->free_deferred && t->dependents == 0;
    }
    if (free_now) fw_model_free(static_cast<fw_model*>(t->self));
  }
}

int32_t fw_model_blob(const fw_model* fm, void** blob_dev, int64_t* blob_bytes) {
  FW_CHECK_ARG(fm && blob_dev && blob_bytes, "null argument");
  *blob_dev = fm->impl.blob;
  *blob_bytes = fm->impl.blob_bytes;
  return FW_OK;
}

int32_t fw_model_info(const fw_model* fm, fw_config* cfg_out, int32_t* compute_type, int32_t* device_index,
                      int32_t* max_batch, int32_t* max_beam) {
  FW_CHECK_ARG(fm, "null model");
  const Model* m = &fm->impl;
  if (cfg_out) *cfg_out = m->cfg;
  if (compute_type) *compute_type = m->compute_type;
  if (device_index) *device_index = m->device;
  if (max_batch) *max_batch = m->max_batch;
  if (max_beam) *max_beam = m->max_beam;
  return FW_OK;
}

// rows of the largest decode run of a group (decoder.hip: DEC_RUN_MAX_ROWS)
#define DEC_GROUP_RUN_ROWS 1600

int32_t fw_model_set_decode_batch(fw_model* fm, int32_t decode_batch) {
  FW_CHECK_ARG(fm, "null model");
  Model* m = &fm->impl;
  FW_CHECK_ARG(!m->decoder, "this model has joined another model's decoder");
  FW_CHECK_ARG(decode_batch >= 1, "decode_batch must be positive");
  {
    // The workspaces, the pool and the second lane's model are rebuilt below: not while the group has work, and no
    // fw_generate call may read capacities / dm->lane1 or queue a request until the rebuild is done (grp.resizing:
    // callers wait on grp.cv).  fw_detect_language / fw_align hold dec_mu, which the rebuild takes.
    std::lock_guard<std::mutex> gl(m->grp.mu);
    FW_CHECK_ARG(m->grp.active_runs == 0 && !m->grp.gathering && m->grp.queue.empty() && !m->grp.resizing,
                 "decode runs are queued or in flight: set the decode batch before the first generate call or after the last one returned");
    m->grp.resizing = true;
  }
  struct Done {   // whatever the outcome: let the callers in again
    DecodeGroup& g;
    ~Done() {
      { std::lock_guard<std::mutex> gl(g.mu); g.resizing = false; }
      g.cv.notify_all();
    }
  } done{m->grp};
  std::lock_guard<std::mutex> lk(m->dec_mu);
  FW_HIP(hipSetDevice(m->device));
  // Sizes, in 70 % of the free HBM:
  //   pool   = the cross-attention K / V^T of `decode_batch` chunks (whole encoder batches, at least one): what the
  //            workers of the group keep in flight, ONE copy whatever the number of lanes;
  //   lanes  = two (two concurrent decode runs, engine.h: lane1) when the group holds at least four encoder batches,
  //            each with a workspace for the largest run there can be: the pool's chunks, at most DEC_GROUP_RUN_ROWS rows
  //            (decoder.hip caps runs there; one lane: 2048 rows).  The self-attention cache of a lane is rows x
  //            positions and a RUN lays it out for its own max_length, so when the budget is short the positions per
  //            row are lowered first (down to 160: a run that asks for more simply holds fewer rows), then the pool.
  int want = std::max(decode_batch, m->max_batch) / m->max_batch * m->max_batch;
  int lanes_wanted = 2;
  if (const char* e = getenv("FWAMD_DECODE_LANES")) lanes_wanted = std::min(4, std::max(1, atoi(e)));
  const int n_lanes = want >= 4 * m->max_batch ? lanes_wanted : 1;
  const int max_rows = n_lanes >= 2 ? DEC_GROUP_RUN_ROWS : 2048;
  auto lane_of = [&](int pool_chunks) {
    int lc = pool_chunks;
    while (lc > m->max_batch && (int64_t)lc * m->max_beam > max_rows) lc -= m->max_batch;
    return lc;
  };
  size_t free_b = 0, total_b = 0;
  FW_HIP(hipMemGetInfo(&free_b, &total_b));
  if (m->gen) free_b += (size_t)gen_workspace_bytes(m, lane_chunks_of(m), m->decode_self_ctx);
  for (const Model* l : m->xlanes)
    if (l && l->gen) free_b += (size_t)gen_workspace_bytes(l, lane_chunks_of(l), l->decode_self_ctx);
  if (m->xpool) free_b += (size_t)cross_pool_bytes(m, std::max(m->decode_batch, m->max_batch));
  const int64_t budget = (int64_t)(0.7 * (double)free_b);
  const int NT = m->cfg.n_text_ctx;
  const int ctx_steps[] = {NT, 320, 224, 160};
  int self_ctx = NT;
  for (;;) {
    bool fits = false;
    for (int cs : ctx_steps) {
      if (cs > NT) continue;
      if (cross_pool_bytes(m, want) + n_lanes * gen_workspace_bytes(m, lane_of(want), cs) <= budget) { self_ctx = cs; fits = true; break; }
    }
    if (fits || want <= m->max_batch) break;
    want -= m->max_batch;
  }
  const int lane_batch = lane_of(want);
  const bool lanes_ok = n_lanes == n_lanes_of(m);
  if (m->gen && m->xpool && lanes_ok && want == m->decode_batch && lane_batch == lane_chunks_of(m) &&
      self_ctx == m->decode_self_ctx)
    return FW_OK;
  if (m->dec_stream) FW_HIP(hipStreamSynchronize(m->dec_stream));
  gen_workspace_free(m);
  for (Model*& l : m->xlanes) {
    if (l) fw_model_free(static_cast<fw_model*>(l->self));
    l = nullptr;
  }
  m->lane1 = nullptr;
  cross_pool_free(m);
  m->decode_batch = want;
  m->lane_batch = lane_batch;
  m->decode_self_ctx = self_ctx;
  int rc = gen_workspace_ensure(m);     // (creates the pool too)
  if (rc) return rc;
  for (int k = 1; k < n_lanes; ++k) {
    fw_model* l = nullptr;
    if ((rc = model_from_blob(m->blob, m->blob_bytes, false, m->device, m->max_batch, m->max_beam, &l, true))) return rc;
    l->impl.decode_batch = want;
    l->impl.lane_batch = lane_batch;
    l->impl.decode_self_ctx = self_ctx;
    l->impl.pool_owner = m;
    l->impl.prof_on = m->prof_on;
    if ((rc = gen_workspace_ensure(&l->impl))) { fw_model_free(l); return rc; }
    m->xlanes[k - 1] = &l->impl;
  }
  m->lane1 = m->xlanes[0];
  return FW_OK;
}

int32_t fw_model_set_decode_lanes(fw_model* fm, int32_t lanes) {
  FW_CHECK_ARG(fm, "null model");
  Model* dm = decoder_of(&fm->impl);
  int have;
  {
    std::lock_guard<std::mutex> gl(dm->grp.mu);     // (the lanes are rebuilt under grp.resizing)
    have = std::max(2, n_lanes_of(dm));             // (a group that has not built its second lane yet still takes 2 ...
    if (const char* e = getenv("FWAMD_DECODE_LANES")) have = std::max(have, std::min(4, atoi(e)));   // ... or what it will build)
  }
  FW_CHECK_ARG(lanes >= 1 && lanes <= have, "decode lanes: 1 .. %d", have);
  dm->grp.lanes_enabled.store(lanes);
  return FW_OK;
}

int32_t fw_model_set_merge_wait(fw_model* fm, int32_t wait_ms, int32_t fill_percent) {
  FW_CHECK_ARG(fm, "null model");
  FW_CHECK_ARG(wait_ms >= -1 && wait_ms <= 10000, "merge wait: -1 (one encoder pass), 0 (never) or milliseconds <= 10000");
  FW_CHECK_ARG(fill_percent >= 1 && fill_percent <= 100, "merge fill: 1 .. 100 percent of a run's chunk capacity");
  DecodeGroup& g = decoder_of(&fm->impl)->grp;
  g.merge_wait_ms.store(wait_ms);
  g.merge_fill_pct.store(fill_percent);
  return FW_OK;
}

int32_t fw_model_decode_batch(const fw_model* fm) {
  if (!fm) return 0;
  const Model* m = fm->impl.decoder ? fm->impl.decoder : &fm->impl;
  return std::max(m->decode_batch, m->max_batch);
}

int32_t fw_model_run_capacity(const fw_model* fm) {
  if (!fm) return 0;
  return lane_chunks_of(fm->impl.decoder ? fm->impl.decoder : &fm->impl);
}

int32_t fw_model_decode_stats(const fw_model* fm, int64_t* runs, int64_t* requests, int64_t* chunks,
                              int32_t* max_run_chunks) {
  FW_CHECK_ARG(fm, "null model");
  const Model* m = fm->impl.decoder ? fm->impl.decoder : &fm->impl;
  if (runs) *runs = m->grp.n_runs.load();
  if (requests) *requests = m->grp.n_requests.load();
  if (chunks) *chunks = m->grp.n_chunks.load();
  if (max_run_chunks) *max_run_chunks = m->grp.max_run_chunks.load();
  return FW_OK;
}

int32_t fw_model_join_decoder(fw_model* fm, fw_model* decoder) {
  FW_CHECK_ARG(fm && decoder && fm != decoder, "need two distinct models");
  Model* m = &fm->impl;
  Model* d = &decoder->impl;
  FW_CHECK_ARG(!d->decoder, "the decoder model has itself joined another decoder");
  FW_CHECK_ARG(m->device == d->device && m->blob == d->blob && m->max_batch == d->max_batch && m->max_beam == d->max_beam,
               "models that share a decoder must share device, weight blob, max_batch and max_beam");
  FW_CHECK_ARG(!m->decoder, "this model has already joined a decoder");
  std::lock_guard<std::mutex> lk(m->dec_mu);
  FW_HIP(hipSetDevice(m->device));
  if (m->dec_stream) FW_HIP(hipStreamSynchronize(m->dec_stream));
  gen_workspace_free(m);
  cross_pool_free(m);
  {
    std::lock_guard<std::mutex> lk2(g_models_mu);   // the primary outlives its workers (fw_model_free defers)
    d->dependents += 1;
  }
  m->decoder = d;
  return FW_OK;
}

int32_t fw_logmel(fw_model* fm, const float* pcm, const int64_t* offsets, int32_t B, float* out,
                  int32_t* n_frames_out) {
  FW_CHECK_ARG(fm && offsets && out, "null argument");
  Model* m = &fm->impl;
  FW_CHECK_ARG(B >= 1 && B <= m->max_batch, "batch %d exceeds max_batch %d", B, m->max_batch);
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  int64_t total; int mf;
  int rc = check_offsets(offsets, B, &total, &mf);
  if (rc) return rc;
  FW_CHECK_ARG(total == 0 || pcm, "null pcm");
  if ((rc = ensure_pcm(m, std::max<int64_t>(total, 1)))) return rc;
  if (total) FW_HIP(hipMemcpyAsync(m->ws_pcm, pcm, total * sizeof(float), hipMemcpyHostToDevice, m->stream));
  if ((rc = logmel_batch(m, m->ws_pcm, offsets, B, true, false))) return rc;
  FW_HIP(hipMemcpyAsync(out, m->ws_feat32, (size_t)B * m->cfg.n_mels * 3000 * sizeof(float), hipMemcpyDeviceToHost,
                        m->stream));
  if (n_frames_out)
    FW_HIP(hipMemcpyAsync(n_frames_out, m->ws_nframes, B * sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
  FW_HIP(hipStreamSynchronize(m->stream));
  return FW_OK;
}

int32_t fw_logmel_full(fw_model* fm, const float* pcm, int64_t n_samples, float* out, int64_t out_frames) {
  FW_CHECK_ARG(fm && out, "null argument");
  Model* m = &fm->impl;
  FW_CHECK_ARG(n_samples >= 0 && n_samples < (int64_t)1 << 30, "invalid n_samples");
  const int64_t nf = (n_samples + 160) / 160;
  FW_CHECK_ARG(out_frames == nf, "out_frames must be n_samples/160 + 1 = %lld, got %lld", (long long)nf,
               (long long)out_frames);
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  int rc;
  if ((rc = ensure_pcm(m, std::max<int64_t>(n_samples, 1)))) return rc;
  if (n_samples) FW_HIP(hipMemcpyAsync(m->ws_pcm, pcm, n_samples * sizeof(float), hipMemcpyHostToDevice, m->stream));
  const int stride = (int)((nf + 7) / 8 * 8);
  if ((rc = ensure_raw(m, 1, std::max(stride, 3008)))) return rc;
  const int rstride = std::max(stride, 3008);
  int64_t offs[2] = {0, n_samples};
  FW_HIP(hipMemcpyAsync(m->ws_offsets, offs, 2 * sizeof(int64_t), hipMemcpyHostToDevice, m->stream));
  float* dout = nullptr;
  if ((rc = dev_alloc_t(&dout, (size_t)m->cfg.n_mels * nf))) return rc;
  {
    ProfScope ps(m, PF_LOGMEL, 0, (double)n_samples * 4 + (double)m->cfg.n_mels * nf * 4);
    fwk::launch_logmel(m->stream, m->ws_pcm, m->ws_offsets, 1, (int)nf, m->lm_consts, m->lm_filtT, m->lm_mel_pad,
                       m->cfg.n_mels, m->ws_raw, (int64_t)m->cfg.n_mels * rstride, rstride, m->ws_chunk_max, 0,
                       (int)nf, dout, nullptr, m->c_pad, nullptr);
  }
  hipError_t he = hipMemcpyAsync(out, dout, (size_t)m->cfg.n_mels * nf * sizeof(float), hipMemcpyDeviceToHost,
                                 m->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(m->stream);
  (void)hipFree(dout);
  if (he != hipSuccess) {
    set_error("logmel_full failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  return FW_OK;
}

namespace {
// tells the device's decode group that a request is on its way (the leader of the next decode run waits a moment
// for it instead of starting with a nearly empty workspace)
struct EncodingMark {
  fw::DecodeGroup& g;
  std::chrono::steady_clock::time_point t0;
  explicit EncodingMark(fw::Model* m) : g(fw::decoder_of(m)->grp) {
    g.encoding.fetch_add(1);
    // One encoder pass at a time per device: a pass fills the chip by itself, several at once only time-slice
    // (measured +1.5 % throughput with the lock, and per-kernel event timings stay meaningful).  The decode run of
    // the group keeps going on its own stream next to it.
    g.enc_mu.lock();
    t0 = std::chrono::steady_clock::now();
  }
  ~EncodingMark() {
    const int us = (int)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    const int old = g.enc_pass_us.load();
    g.enc_pass_us.store(old ? (3 * old + us) / 4 : us);
    g.enc_mu.unlock();
    g.encoding.fetch_sub(1);
    g.cv.notify_all();
  }
};
}  // namespace

int32_t fw_encode(fw_model* fm, const float* features, int32_t B, fw_tensor** out) {
  FW_CHECK_ARG(fm && features && out, "null argument");
  Model* m = &fm->impl;
  FW_CHECK_ARG(B >= 1 && B <= m->max_batch, "batch %d exceeds max_batch %d", B, m->max_batch);
  EncodingMark mark(m);
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  FW_HIP(hipMemcpyAsync(m->ws_feat32, features, (size_t)B * m->cfg.n_mels * 3000 * sizeof(float),
                        hipMemcpyHostToDevice, m->stream));
  fwk::launch_features_to_cl(m->stream, m->ws_feat32, B, m->cfg.n_mels, 3000, m->ws_mel_cl, m->c_pad);
  fw_tensor* t = nullptr;
  int rc = new_tensor(m, B, &t);
  if (rc) return rc;
  if ((rc = run_encoder(m, B, t->impl.data))) {
    fw_tensor_free(t);
    return rc;
  }
  hipError_t he = hipStreamSynchronize(m->stream);
  if (he != hipSuccess) {
    fw_tensor_free(t);
    set_error("encode failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  prof_collect(m);
  *out = t;
  return FW_OK;
}

static int encode_pcm_common(Model* m, const float* pcm_dev, const int64_t* offsets, int B, fw_tensor** out) {
  int rc;
  if ((rc = logmel_batch(m, pcm_dev, offsets, B, false, true))) return rc;
  fw_tensor* t = nullptr;
  if ((rc = new_tensor(m, B, &t))) return rc;
  if ((rc = run_encoder(m, B, t->impl.data))) {
    fw_tensor_free(t);
    return rc;
  }
  hipError_t he = hipStreamSynchronize(m->stream);
  if (he != hipSuccess) {
    fw_tensor_free(t);
    set_error("encode_pcm failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  prof_collect(m);
  *out = t;
  return FW_OK;
}

int32_t fw_encode_pcm(fw_model* fm, const float* pcm, const int64_t* offsets, int32_t B, fw_tensor** out) {
  FW_CHECK_ARG(fm && offsets && out, "null argument");
  Model* m = &fm->impl;
  FW_CHECK_ARG(B >= 1 && B <= m->max_batch, "batch %d exceeds max_batch %d", B, m->max_batch);
  EncodingMark mark(m);
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  int64_t total; int mf;
  int rc = check_offsets(offsets, B, &total, &mf);
  if (rc) return rc;
  FW_CHECK_ARG(total == 0 || pcm, "null pcm");
  if ((rc = ensure_pcm(m, std::max<int64_t>(total, 1)))) return rc;
  if (total) FW_HIP(hipMemcpyAsync(m->ws_pcm, pcm, total * sizeof(float), hipMemcpyHostToDevice, m->stream));
  return encode_pcm_common(m, m->ws_pcm, offsets, B, out);
}

int32_t fw_encode_pcm_dev(fw_model* fm, const float* pcm_dev, const int64_t* offsets, int32_t B, fw_tensor** out) {
  FW_CHECK_ARG(fm && pcm_dev && offsets && out, "null argument");
  Model* m = &fm->impl;
  FW_CHECK_ARG(B >= 1 && B <= m->max_batch, "batch %d exceeds max_batch %d", B, m->max_batch);
  EncodingMark mark(m);
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  return encode_pcm_common(m, pcm_dev, offsets, B, out);
}

int32_t fw_tensor_shape(const fw_tensor* t, int32_t* B, int32_t* T, int32_t* D) {
  FW_CHECK_ARG(t, "null tensor");
  if (B) *B = t->impl.B;
  if (T) *T = t->impl.T;
  if (D) *D = t->impl.D;
  return FW_OK;
}

int32_t fw_tensor_to_host(fw_model* fm, const fw_tensor* t, float* out) {
  FW_CHECK_ARG(fm && t && out, "null argument");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  const size_t n = (size_t)t->impl.B * t->impl.T * t->impl.D;
  float* tmp = nullptr;
  int rc = dev_alloc_t(&tmp, n);
  if (rc) return rc;
  fwk::launch_f16_to_f32(m->stream, t->impl.data, tmp, (int64_t)n);
  hipError_t he = hipMemcpyAsync(out, tmp, n * sizeof(float), hipMemcpyDeviceToHost, m->stream);
  if (he == hipSuccess) he = hipStreamSynchronize(m->stream);
  (void)hipFree(tmp);
  if (he != hipSuccess) {
    set_error("tensor_to_host failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  return FW_OK;
}

int32_t fw_tensor_from_host(fw_model* fm, const float* data, int32_t B, fw_tensor** out) {
  FW_CHECK_ARG(fm && data && out, "null argument");
  Model* m = &fm->impl;
  FW_CHECK_ARG(B >= 1 && B <= m->max_batch, "batch %d exceeds max_batch %d", B, m->max_batch);
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  fw_tensor* t = nullptr;
  int rc = new_tensor(m, B, &t);
  if (rc) return rc;
  const size_t n = (size_t)B * t->impl.T * t->impl.D;
  float* tmp = nullptr;
  if ((rc = dev_alloc_t(&tmp, n))) { fw_tensor_free(t); return rc; }
  hipError_t he = hipMemcpyAsync(tmp, data, n * sizeof(float), hipMemcpyHostToDevice, m->stream);
  fwk::launch_f32_to_f16(m->stream, tmp, t->impl.data, (int64_t)n);
  if (he == hipSuccess) he = hipStreamSynchronize(m->stream);
  (void)hipFree(tmp);
  if (he != hipSuccess) {
    fw_tensor_free(t);
    set_error("tensor_from_host failed: %s", hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  *out = t;
  return FW_OK;
}

void fw_tensor_free(fw_tensor* t) {
  if (!t) return;
  if (t->impl.data) {
    std::lock_guard<std::mutex> lk(g_models_mu);
    Model* m = t->impl.owner;
    const bool alive = std::find(g_live_models.begin(), g_live_models.end(), m) != g_live_models.end();
    if (alive) {
      std::lock_guard<std::mutex> lk2(m->pool_mu);
      m->enc_pool.push_back(t->impl.data);
    } else {
      (void)hipFree(t->impl.data);
    }
  }
  delete t;
}

// ---------------------------------------------------------------- measurement hooks
void fw_prof_enable(fw_model* fm, int32_t on) {
  if (!fm) return;
  fm->impl.prof_on = on != 0;
  for (Model* l : fm->impl.xlanes)
    if (l) l->prof_on = on != 0;
}
void fw_prof_reset(fw_model* fm) {
  if (!fm) return;
  prof_collect(&fm->impl);
  {
    std::lock_guard<std::mutex> lk(fm->impl.prof_mu);
    for (auto& p : fm->impl.prof) p = ProfAcc();
  }
  for (Model* l : fm->impl.xlanes) {
    if (!l) continue;
    prof_collect(l);
    std::lock_guard<std::mutex> lk(l->prof_mu);
    for (auto& p : l->prof) p = ProfAcc();
  }
}
int32_t fw_prof_count(void) { return PF_COUNT; }
const char* fw_prof_name(int32_t i) { return (i >= 0 && i < PF_COUNT) ? kProfNames[i] : ""; }
int32_t fw_prof_get(fw_model* fm, int32_t i, double* ms, int64_t* launches, double* flops, double* bytes) {
  FW_CHECK_ARG(fm && i >= 0 && i < PF_COUNT, "bad profile index");
  prof_collect(&fm->impl);
  ProfAcc p;
  {
    std::lock_guard<std::mutex> lk(fm->impl.prof_mu);
    p = fm->impl.prof[i];
  }
  for (Model* l : fm->impl.xlanes) {   // the further decode lanes of the group report through their primary
    if (!l) continue;
    prof_collect(l);
    std::lock_guard<std::mutex> lk(l->prof_mu);
    p.ms += l->prof[i].ms; p.launches += l->prof[i].launches; p.flops += l->prof[i].flops; p.bytes += l->prof[i].bytes;
  }
  if (ms) *ms = p.ms;
  if (launches) *launches = p.launches;
  if (flops) *flops = p.flops;
  if (bytes) *bytes = p.bytes;
  return FW_OK;
}
int32_t fw_synchronize(fw_model* fm) {
  FW_CHECK_ARG(fm, "null model");
  FW_HIP(hipSetDevice(fm->impl.device));
  FW_HIP(hipStreamSynchronize(fm->impl.stream));
  if (fm->impl.dec_stream) FW_HIP(hipStreamSynchronize(fm->impl.dec_stream));
  for (Model* l : fm->impl.xlanes)
    if (l && l->dec_stream) FW_HIP(hipStreamSynchronize(l->dec_stream));
  return FW_OK;
}
int32_t fw_dev_alloc(fw_model* fm, int64_t bytes, void** out_dev) {
  FW_CHECK_ARG(fm && out_dev && bytes >= 0, "bad argument");
  FW_HIP(hipSetDevice(fm->impl.device));
  return dev_alloc(out_dev, (size_t)bytes);
}
int32_t fw_dev_free(fw_model* fm, void* dev) {
  FW_CHECK_ARG(fm, "null model");
  FW_HIP(hipSetDevice(fm->impl.device));
  if (dev) FW_HIP(hipFree(dev));
  return FW_OK;
}
int32_t fw_dev_upload(fw_model* fm, void* dst_dev, const void* src_host, int64_t bytes) {
  FW_CHECK_ARG(fm && dst_dev && src_host && bytes >= 0, "bad argument");
  FW_HIP(hipSetDevice(fm->impl.device));
  FW_HIP(hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
  return FW_OK;
}

// ---------------------------------------------------------------- kernel test hooks
static int upload_f16(Model* m, const float* src, size_t n, half_t** dst) {
  int rc = dev_alloc_t(dst, n);
  if (rc) return rc;
  std::vector<uint16_t> tmp(n);
  for (size_t i = 0; i < n; ++i) tmp[i] = f32_to_f16_bits(src[i]);
  FW_HIP(hipMemcpy(*dst, tmp.data(), n * 2, hipMemcpyHostToDevice));
  return FW_OK;
}
static int download_f16(Model* m, const half_t* src, size_t n, float* dst) {
  std::vector<uint16_t> tmp(n);
  FW_HIP(hipStreamSynchronize(m->stream));
  FW_HIP(hipMemcpy(tmp.data(), src, n * 2, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < n; ++i) {
    half_t h;
    memcpy(&h, &tmp[i], 2);
    dst[i] = (float)h;
  }
  return FW_OK;
}

int32_t fw_test_gemm(fw_model* fm, const float* A, const float* W, const float* bias, const float* residual,
                     int32_t M, int32_t N, int32_t K, int32_t act_gelu, int32_t use_int8, float* out) {
  FW_CHECK_ARG(fm && A && W && out, "null argument");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  half_t *dA = nullptr, *dW = nullptr, *dB = nullptr, *dR = nullptr, *dC = nullptr;
  int rc;
  if ((rc = upload_f16(m, A, (size_t)M * K, &dA))) return rc;
  if ((rc = upload_f16(m, W, (size_t)N * K, &dW))) return rc;
  if (bias && (rc = upload_f16(m, bias, N, &dB))) return rc;
  if (residual && (rc = upload_f16(m, residual, (size_t)M * N, &dR))) return rc;
  if ((rc = dev_alloc_t(&dC, (size_t)M * N))) return rc;
  LinearW L{dW, dB, nullptr, nullptr, nullptr, nullptr, N, K};
  int8_t* dWq = nullptr;
  float* dWs = nullptr;
  if (use_int8) {
    // quantise W per output row exactly like the weight packer, run the int8 path
    if (m->compute_type != FW_COMPUTE_INT8_FLOAT16 || (int64_t)M * K > (int64_t)m->max_batch * 1500 * 4 * m->cfg.d_model ||
        M > m->max_batch * 1500) {
      set_error("int8 gemm test needs an int8_float16 model and M*K within its quantisation workspace");
      return FW_EINVAL;
    }
    std::vector<int8_t> wq((size_t)N * K);
    std::vector<float> ws(N);
    for (int n = 0; n < N; ++n) {
      float amax = 0.f;
      for (int k = 0; k < K; ++k) amax = std::max(amax, fabsf(f16_bits_to_f32(f32_to_f16_bits(W[(size_t)n * K + k]))));
      const float sc = amax > 0.f ? 127.0f / amax : 0.f;
      for (int k = 0; k < K; ++k)
        wq[(size_t)n * K + k] = (int8_t)lrintf(f16_bits_to_f32(f32_to_f16_bits(W[(size_t)n * K + k])) * sc);
      ws[n] = amax > 0.f ? amax / 127.0f : 1.0f;
    }
    if ((rc = dev_alloc_t(&dWq, wq.size()))) return rc;
    if ((rc = dev_alloc_t(&dWs, ws.size()))) return rc;
    FW_HIP(hipMemcpy(dWq, wq.data(), wq.size(), hipMemcpyHostToDevice));
    FW_HIP(hipMemcpy(dWs, ws.data(), ws.size() * sizeof(float), hipMemcpyHostToDevice));
    L.wq = dWq; L.wscale = dWs;
    if (act_gelu >= 2)
      rc = run_linear_i8(m, L, dA, nullptr, dC, M, 0, nullptr, 0, 0, M, 1, act_gelu - 2, true, 0);
    else
      rc = run_linear_i8(m, L, dA, nullptr, dC, N, 0, dR, N, 0, M, 1, act_gelu, false, 0);
  } else if (act_gelu == 8 || act_gelu == 9) {
    // the cross-attention K (8) / V^T (9) projection epilogues: output MFMA-fragment-major per 64-column head
    // (gemm.hip), un-permuted here into out [M][N]; the padded keys of the last 32-key group must stay zero
    const bool vt = act_gelu == 9;
    if (N % 64 || residual) { set_error("fragment-major gemm test: N %% 64 == 0, no residual"); return FW_EINVAL; }
    const int kvp = (M + 31) / 32 * 32, H = N / 64;
    half_t* dF = nullptr;
    if ((rc = dev_alloc_t(&dF, (size_t)H * kvp * 64))) return rc;
    FW_HIP(hipMemset(dF, 0, (size_t)H * kvp * 64 * sizeof(half_t)));
    rc = vt ? run_linear(m, L, dA, K, 0, dF, kvp, 0, nullptr, 0, 0, M, 1, 0, true, kvp)
            : run_linear(m, L, dA, K, 0, dF, N, 0, nullptr, 0, 0, M, 1, 0, false, kvp);
    std::vector<float> hf((size_t)H * kvp * 64);
    if (!rc) rc = download_f16(m, dF, hf.size(), hf.data());
    (void)hipFree(dF);
    if (!rc) {
      for (int mm = 0; mm < kvp && !rc; ++mm)
        for (int n = 0; n < N; ++n) {
          const int c = n & 63, r = mm & 31;
          const size_t off = vt ? (size_t)(n >> 6) * kvp * 64 + ((size_t)((mm >> 5) * 4 + (c >> 4)) * 64 + ((mm >> 3) & 3) * 16 + (c & 15)) * 8 + (mm & 7)
                                : (size_t)(n >> 6) * kvp * 64 + ((size_t)((mm >> 5) * 4 + 2 * ((r >> 2) & 1) + (c >> 5)) * 64 + ((c >> 3) & 3) * 16 + (((r >> 3) << 2) | (r & 3))) * 8 + (c & 7);
          if (mm < M) out[(size_t)mm * N + n] = hf[off];
          else if (hf[off] != 0.f) { set_error("fragment-major epilogue wrote the padded key %d", mm); rc = FW_ERUNTIME; break; }
        }
    }
    for (half_t* p : {dA, dW, dB, dR, dC})
      if (p) (void)hipFree(p);
    return rc;
  } else if (act_gelu >= 2) {
    // transposed-output mode: out is [N][M]
    rc = run_linear(m, L, dA, K, 0, dC, M, 0, nullptr, 0, 0, M, 1, act_gelu - 2, true);
  } else {
    rc = run_linear(m, L, dA, K, 0, dC, N, 0, dR, N, 0, M, 1, act_gelu, false);
  }
  if (!rc) rc = download_f16(m, dC, (size_t)M * N, out);
  for (half_t* p : {dA, dW, dB, dR, dC})
    if (p) (void)hipFree(p);
  if (dWq) (void)hipFree(dWq);
  if (dWs) (void)hipFree(dWs);
  return rc;
}

// fragment-major position of element (row, k) of a [rows][K] operand: 16-row tiles x k-steps of KE elements, the
// 16 bytes of lane l = 16*((k / (KE/4)) % 4) + row % 16 contiguous (dec_kernels.hip)
static inline size_t frag_pos(int64_t row, int64_t k, int64_t K, int KE) {
  const int OCT = KE / 4;
  return (size_t)((((row >> 4) * (K / KE) + k / KE) * 64 + ((k / OCT) & 3) * 16 + (row & 15)) * OCT + (k % OCT));
}

int32_t fw_test_dec_linear(fw_model* fm, const float* x, const float* W, const float* bias, const float* ln_g,
                           const float* ln_b, const float* res, int32_t R, int32_t N, int32_t K, int32_t act,
                           int32_t use_int8, float* out, float* out_from_frag) {
  FW_CHECK_ARG(fm && x && W && out && out_from_frag, "null argument");
  FW_CHECK_ARG(R >= 1 && N % 32 == 0 && K % 64 == 0, "need R >= 1, N %% 32 == 0, K %% 64 == 0");
  FW_CHECK_ARG((ln_g == nullptr) == (ln_b == nullptr), "ln_g and ln_b go together");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  hipStream_t st = m->stream;
  const int R16 = (R + 15) / 16 * 16;
  auto h = [](float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); };
  std::vector<void*> owned;
  auto up = [&](const void* src, size_t bytes, void** dst) -> int {
    int rc = dev_alloc(dst, bytes);
    if (rc) return rc;
    owned.push_back(*dst);
    FW_HIP(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return FW_OK;
  };
  auto cleanup = [&]() { for (void* p : owned) (void)hipFree(p); };
  int rc = FW_OK;
  half_t *d_res = nullptr, *d_bias = nullptr, *d_out = nullptr, *d_of = nullptr;
  std::vector<uint16_t> tmp;
  if (res) {
    tmp.resize((size_t)R * N);
    for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = f32_to_f16_bits(res[i]);
    if ((rc = up(tmp.data(), tmp.size() * 2, (void**)&d_res))) { cleanup(); return rc; }
  }
  if ((rc = dev_alloc_t(&d_out, (size_t)R * N))) { cleanup(); return rc; }
  owned.push_back(d_out);
  if ((rc = dev_alloc_t(&d_of, (size_t)R16 * N))) { cleanup(); return rc; }
  owned.push_back(d_of);
  FW_HIP(hipMemset(d_of, 0, (size_t)R16 * N * 2));
  if (use_int8 == 1) {
    if (m->compute_type != FW_COMPUTE_INT8_FLOAT16 || ln_g) {
      cleanup();
      set_error("int8 decoder-linear test needs an int8_float16 model and no LayerNorm");
      return FW_EINVAL;
    }
    std::vector<int8_t> wq((size_t)N * K);
    std::vector<float> ws(N);
    for (int n = 0; n < N; ++n) {
      float amax = 0.f;
      for (int k = 0; k < K; ++k) amax = std::max(amax, fabsf(h(W[(size_t)n * K + k])));
      const float sc = amax > 0.f ? 127.0f / amax : 0.f;
      for (int k = 0; k < K; ++k) wq[frag_pos(n, k, K, 64)] = (int8_t)lrintf(h(W[(size_t)n * K + k]) * sc);
      ws[n] = amax > 0.f ? amax / 127.0f : 1.0f;
    }
    int8_t *d_wq = nullptr, *d_xq = nullptr;
    float *d_ws = nullptr, *d_xs = nullptr;
    half_t* d_x = nullptr;
    tmp.resize((size_t)R * K);
    for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = f32_to_f16_bits(x[i]);
    if ((rc = up(wq.data(), wq.size(), (void**)&d_wq)) || (rc = up(ws.data(), ws.size() * 4, (void**)&d_ws)) ||
        (rc = up(tmp.data(), tmp.size() * 2, (void**)&d_x))) { cleanup(); return rc; }
    if (bias) {
      tmp.resize(N);
      for (int n = 0; n < N; ++n) tmp[n] = f32_to_f16_bits(bias[n]);
      if ((rc = up(tmp.data(), (size_t)N * 2, (void**)&d_bias))) { cleanup(); return rc; }
    }
    if ((rc = dev_alloc_t(&d_xq, (size_t)R16 * K))) { cleanup(); return rc; }
    owned.push_back(d_xq);
    if ((rc = dev_alloc_t(&d_xs, (size_t)R16))) { cleanup(); return rc; }
    owned.push_back(d_xs);
    FW_HIP(hipMemset(d_xq, 0, (size_t)R16 * K));
    fwk::launch_quant_rows(st, d_x, K, nullptr, nullptr, d_xq, d_xs, R, K, 1);
    if (fwd::launch_dec_gemm_frag_i8(st, d_xq, d_xs, d_wq, d_ws, d_bias, d_res, N, d_out, N, R, N, K, act) != 0) {
      cleanup();
      set_error("int8 decoder linear: unsupported shape R=%d N=%d K=%d", R, N, K);
      return FW_ERUNTIME;
    }
    rc = download_f16(m, d_out, (size_t)R * N, out);
    if (!rc) memcpy(out_from_frag, out, (size_t)R * N * sizeof(float));
    cleanup();
    return rc;
  }
  // fp16: fold the LayerNorm exactly like the weight packer (add_folded)
  std::vector<uint16_t> wf((size_t)N * K), xf((size_t)R16 * K, 0);
  std::vector<float> s1(N, 0.f), cf(N, 0.f);
  for (int n = 0; n < N; ++n) {
    double a1 = 0.0, ac = 0.0;
    for (int k = 0; k < K; ++k) {
      const float wv = h(W[(size_t)n * K + k]);
      const uint16_t wg = ln_g ? f32_to_f16_bits(wv * h(ln_g[k])) : f32_to_f16_bits(wv);
      wf[frag_pos(n, k, K, 32)] = wg;
      a1 += (double)f16_bits_to_f32(wg);
      if (ln_g) ac += (double)wv * (double)h(ln_b[k]);
    }
    if (bias) ac += (double)h(bias[n]);
    s1[n] = (float)a1;
    cf[n] = (float)ac;
  }
  for (int r = 0; r < R; ++r)
    for (int k = 0; k < K; ++k) xf[frag_pos(r, k, K, 32)] = f32_to_f16_bits(x[(size_t)r * K + k]);
  half_t *d_wf = nullptr, *d_xf = nullptr;
  float *d_s1 = nullptr, *d_cf = nullptr;
  if ((rc = up(wf.data(), wf.size() * 2, (void**)&d_wf)) || (rc = up(xf.data(), xf.size() * 2, (void**)&d_xf))) {
    cleanup();
    return rc;
  }
  if (ln_g) {
    if ((rc = up(s1.data(), (size_t)N * 4, (void**)&d_s1)) || (rc = up(cf.data(), (size_t)N * 4, (void**)&d_cf))) {
      cleanup();
      return rc;
    }
  } else if (bias) {
    tmp.resize(N);
    for (int n = 0; n < N; ++n) tmp[n] = f32_to_f16_bits(bias[n]);
    if ((rc = up(tmp.data(), (size_t)N * 2, (void**)&d_bias))) { cleanup(); return rc; }
  }
  // use_int8 >= 10: the GEMM-shaped kernel of merged runs (dec_gemm_big_kernel), workgroup shape use_int8 - 10,
  // whatever the row count; 5: the skinny kernel whatever the row count (the reference of the bit-identity test).
  const int lr =
      use_int8 >= 10 ? fwd::launch_dec_gemm_big(st, use_int8 - 10, d_xf, d_wf, d_bias, d_s1, d_cf, d_res, N, d_out, N, d_of, R, N, K, act)
      : use_int8 == 5 ? fwd::launch_dec_gemm_skinny(st, d_xf, d_wf, d_bias, d_s1, d_cf, d_res, N, d_out, N, d_of, R, N, K, act)
      : (use_int8 == 6 || use_int8 == 7) ? fwd::launch_dec_gemm_skinny_tiles(st, use_int8 - 5, d_xf, d_wf, d_bias, d_s1, d_cf, d_res, N, d_out, N, d_of, R, N, K, act)
                      : fwd::launch_dec_gemm_frag(st, d_xf, d_wf, d_bias, d_s1, d_cf, d_res, N, d_out, N, d_of, R, N, K, act);
  if (lr != 0) {
    cleanup();
    set_error("decoder linear: unsupported shape R=%d N=%d K=%d", R, N, K);
    return FW_ERUNTIME;
  }
  rc = download_f16(m, d_out, (size_t)R * N, out);
  std::vector<float> of((size_t)R16 * N);
  if (!rc) rc = download_f16(m, d_of, (size_t)R16 * N, of.data());
  if (!rc)
    for (int r = 0; r < R; ++r)
      for (int n = 0; n < N; ++n) out_from_frag[(size_t)r * N + n] = of[frag_pos(r, n, N, 32)];
  cleanup();
  return rc;
}

int32_t fw_dec_big_min_rows(void) { return fwd::dec_big_min_rows(); }
int32_t fw_dec_big_min_rows_of(int32_t role, int32_t compute_type) { return fwd::dec_big_min_rows_of(role, compute_type); }

// process-wide measurement knobs (A/B inside one process: profiles/gemm_bench.py); 1: encoder GEMM tile order
int32_t fw_test_knob(int32_t id, int32_t value) {
  FW_CHECK_ARG(id == 1 || id == 2 || id == 4 || id == 5 || id == 6 || id == 7, "unknown knob %d", id);
  if (id == 6) { set_cross_kv_layered(value); return FW_OK; }
  if (id == 7) { fwd::set_cross_attn_regs(value); return FW_OK; }
  if (id == 1) fwk::g_gemm_order.store(value);
  else if (id == 5) fwk::g_gemm_vt_stage.store(value);
  else if (id == 2) fwd::set_self_attn_form(value);
  else set_pos_blocks(value);
  return FW_OK;
}

// host-only: the run size an idle two-lane decode group leads with (decoder.hip: idle_lead_chunks); needs no device
int64_t fw_test_idle_lead_chunks(int64_t queued, int32_t n_queued, int32_t encoding, int64_t want, int32_t max_batch) {
  return idle_lead_chunks(queued, n_queued, encoding, want, max_batch);
}

int32_t fw_test_dec_logits(fw_model* fm, const float* x, int32_t R, float* out) {
  FW_CHECK_ARG(fm && x && out && R >= 1, "bad argument");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  hipStream_t st = m->stream;
  const int d = m->cfg.d_model, V = m->cfg.n_vocab, R16 = (R + 15) / 16 * 16;
  const bool i8 = m->compute_type == FW_COMPUTE_INT8_FLOAT16;
  std::vector<uint16_t> xh((size_t)R16 * d, 0);
  for (int r = 0; r < R; ++r)
    for (int k = 0; k < d; ++k)
      xh[i8 ? (size_t)r * d + k : frag_pos(r, k, d, 32)] = f32_to_f16_bits(x[(size_t)r * d + k]);
  half_t* d_x = nullptr;
  int8_t* d_xq = nullptr;
  float *d_xs = nullptr, *d_out = nullptr;
  int rc;
  auto cleanup = [&]() { for (void* p : {(void*)d_x, (void*)d_xq, (void*)d_xs, (void*)d_out}) if (p) (void)hipFree(p); };
  if ((rc = dev_alloc_t(&d_x, xh.size())) || (rc = dev_alloc_t(&d_out, (size_t)R * V))) { cleanup(); return rc; }
  FW_HIP(hipMemcpy(d_x, xh.data(), xh.size() * 2, hipMemcpyHostToDevice));
  int lr;
  if (i8) {
    if ((rc = dev_alloc_t(&d_xq, (size_t)R16 * d)) || (rc = dev_alloc_t(&d_xs, (size_t)R16))) { cleanup(); return rc; }
    FW_HIP(hipMemset(d_xq, 0, (size_t)R16 * d));
    fwk::launch_quant_rows(st, d_x, d, m->dec_ln.g, m->dec_ln.b, d_xq, d_xs, R, d, 1);
    lr = fwd::launch_dec_logits(st, true, d_xq, d_xs, m->logits.wq, m->logits.wscale, nullptr, nullptr, d_out, V, R, V, d);
  } else {
    lr = fwd::launch_dec_logits(st, false, d_x, nullptr, m->logits.w, nullptr, m->logits.s1, m->logits.cf, d_out, V, R, V,
                                d);
  }
  hipError_t he = hipSuccess;
  if (lr == 0) he = hipMemcpyAsync(out, d_out, (size_t)R * V * sizeof(float), hipMemcpyDeviceToHost, st);
  if (he == hipSuccess) he = hipStreamSynchronize(st);
  cleanup();
  if (lr != 0 || he != hipSuccess) {
    set_error("logits projection test failed: %s", lr ? "unsupported shape" : hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  return FW_OK;
}

// Micro-benchmark of the decoder linear kernel's tile shapes (profiles/dec_linear_bench.py): `iters` back-to-back
// launches on one stream over a ROTATING set of weight matrices larger than L2 + MALL (as in a decode step, where
// 1.5 GB of weights pass between two uses of the same matrix); us_out = mean microseconds per launch.
int32_t fw_bench_dec_linear(fw_model* fm, int32_t R, int32_t N, int32_t K, int32_t lnf, int32_t variant, int32_t iters,
                            float* us_out) {
  FW_CHECK_ARG(fm && us_out && R > 0 && N > 0 && K > 0 && iters > 0, "bad argument");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  const size_t wn = (size_t)N * K;
  const int copies = (int)std::max<size_t>(2, ((size_t)640 << 20) / (wn * 2));
  const size_t rp = ((size_t)R + 15) / 16 * 16;
  half_t *dW = nullptr, *dX = nullptr, *dO = nullptr, *dB = nullptr;
  float *dS = nullptr, *dC = nullptr;
  int rc;
  auto cleanup = [&]() { for (void* p : {(void*)dW, (void*)dX, (void*)dO, (void*)dB, (void*)dS, (void*)dC}) if (p) (void)hipFree(p); };
  if ((rc = dev_alloc_t(&dW, wn * copies)) || (rc = dev_alloc_t(&dX, rp * K)) || (rc = dev_alloc_t(&dO, rp * N)) ||
      (rc = dev_alloc_t(&dB, (size_t)N)) || (rc = dev_alloc_t(&dS, (size_t)N)) || (rc = dev_alloc_t(&dC, (size_t)N))) {
    cleanup();
    return rc;
  }
  {   // pseudo-random operands in [-1, 1) (constant fills clock the chip up: MI355X_MICROARCH.md, DVFS)
    std::vector<uint16_t> h(std::max(wn, rp * (size_t)K));
    uint32_t sd = 2463534242u;
    for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = f32_to_f16_bits(((int)(sd >> 16) % 2001 - 1000) * 1e-3f); }
    for (int c = 0; c < copies; ++c) FW_HIP(hipMemcpy(dW + (size_t)c * wn, h.data(), wn * 2, hipMemcpyHostToDevice));
    FW_HIP(hipMemcpy(dX, h.data(), rp * K * 2, hipMemcpyHostToDevice));
  }
  FW_HIP(hipMemset(dB, 0, (size_t)N * 2));
  FW_HIP(hipMemset(dS, 0, (size_t)N * 4));
  FW_HIP(hipMemset(dC, 0, (size_t)N * 4));
  hipEvent_t e0, e1;
  FW_HIP(hipEventCreate(&e0));
  FW_HIP(hipEventCreate(&e1));
  hipStream_t st = m->stream;
  int lr = 0;
  for (int i = 0; i < 4 && lr == 0; ++i)
    lr = fwd::launch_dec_gemm_frag_variant(st, variant, lnf != 0, dX, dW + (size_t)(i % copies) * wn, dB, dS, dC, dO, R, N, K);
  FW_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters && lr == 0; ++i)
    lr = fwd::launch_dec_gemm_frag_variant(st, variant, lnf != 0, dX, dW + (size_t)(i % copies) * wn, dB, dS, dC, dO, R, N, K);
  FW_HIP(hipEventRecord(e1, st));
  hipError_t he = hipEventSynchronize(e1);
  float ms = 0.f;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cleanup();
  if (lr != 0) { set_error("fw_bench_dec_linear: unsupported shape / variant"); return FW_EINVAL; }
  if (he != hipSuccess) { set_error("fw_bench_dec_linear: %s", hipGetErrorString(he)); return FW_ENODEV; }
  *us_out = ms * 1000.f / (float)iters;
  return FW_OK;
}

// measurement hook (profiles/gemm_bench.py): the encoder GEMM on device-resident pseudo-random operands,
// `iters` launches between two events.  lda = K + a_pad, ldw = K + w_pad elements (stride experiments).
int32_t fw_bench_gemm(fw_model* fm, int32_t M, int32_t N, int32_t K, int32_t batch, int32_t a_pad, int32_t w_pad,
                      int32_t trans, int32_t iters, float* ms_out) {
  FW_CHECK_ARG(fm && ms_out && M > 0 && N > 0 && K > 0 && batch > 0 && iters > 0, "bad argument");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  const bool i8 = m->compute_type == FW_COMPUTE_INT8_FLOAT16;
  const int64_t lda = K + a_pad, ldw = K + w_pad;
  const size_t es = i8 ? 1 : 2;
  // transposed output: Ct[z][n][m] with the row stride the encoder's V^T has (keys padded to a multiple of 64: t_pad)
  const int64_t ldct = (M + 63) / 64 * 64;
  const size_t na = (size_t)batch * M * lda, nw = (size_t)N * ldw,
               nc = trans ? (size_t)batch * N * ldct : (size_t)batch * M * N;
  void *dA = nullptr, *dW = nullptr;
  half_t* dC = nullptr;
  float *dsa = nullptr, *dsw = nullptr;
  int rc;
  auto cleanup = [&]() { for (void* p : {dA, dW, (void*)dC, (void*)dsa, (void*)dsw}) if (p) (void)hipFree(p); };
  if ((rc = dev_alloc(&dA, na * es)) || (rc = dev_alloc(&dW, nw * es)) || (rc = dev_alloc_t(&dC, nc))) { cleanup(); return rc; }
  {
    std::vector<uint16_t> h(std::max(na, nw));
    uint32_t st = 12345u;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = f32_to_f16_bits(((int)(st >> 16) % 2001 - 1000) * 1e-3f); }
    FW_HIP(hipMemcpy(dA, h.data(), na * es, hipMemcpyHostToDevice));
    FW_HIP(hipMemcpy(dW, h.data(), nw * es, hipMemcpyHostToDevice));
  }
  if (i8) {
    if ((rc = dev_alloc_t(&dsa, (size_t)batch * M)) || (rc = dev_alloc_t(&dsw, (size_t)N))) { cleanup(); return rc; }
    FW_HIP(hipMemset(dsa, 0, (size_t)batch * M * 4));
    FW_HIP(hipMemset(dsw, 0, (size_t)N * 4));
  }
  fwk::GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = (const half_t*)dA; p.lda = lda; p.a_bstride = (int64_t)M * lda;
  p.W = (const half_t*)dW; p.ldw = ldw;
  p.C = dC; p.ldc = trans ? ldct : N; p.c_bstride = trans ? (int64_t)N * ldct : (int64_t)M * N;
  p.M = M; p.N = N; p.K = K;
  p.a_scale = dsa; p.as_bstride = M; p.w_scale = dsw;
  hipEvent_t e0, e1;
  FW_HIP(hipEventCreate(&e0));
  FW_HIP(hipEventCreate(&e1));
  int lr = fwk::launch_gemm(m->stream, p, batch, trans != 0);   // warm-up
  FW_HIP(hipEventRecord(e0, m->stream));
  for (int i = 0; i < iters && lr == 0; ++i) lr = fwk::launch_gemm(m->stream, p, batch, trans != 0);
  FW_HIP(hipEventRecord(e1, m->stream));
  hipError_t he = hipEventSynchronize(e1);
  float ms = 0.f;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cleanup();
  if (lr != 0 || he != hipSuccess) {
    set_error("gemm bench failed: %s", lr ? "unsupported shape" : hipGetErrorString(he));
    return FW_ERUNTIME;
  }
  *ms_out = ms / (float)iters;
  return FW_OK;
}

int32_t fw_test_layernorm(fw_model* fm, const float* x, const float* g, const float* b, int32_t rows, int32_t d,
                          float* out) {
  FW_CHECK_ARG(fm && x && g && b && out, "null argument");
  FW_CHECK_ARG(d % 128 == 0 && d <= 1280, "d must be a multiple of 128 and <= 1280");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  half_t *dx = nullptr, *dg = nullptr, *db = nullptr, *dy = nullptr;
  int rc;
  if ((rc = upload_f16(m, x, (size_t)rows * d, &dx))) return rc;
  if ((rc = upload_f16(m, g, d, &dg))) return rc;
  if ((rc = upload_f16(m, b, d, &db))) return rc;
  if ((rc = dev_alloc_t(&dy, (size_t)rows * d))) return rc;
  fwk::launch_layernorm(m->stream, dx, dg, db, dy, rows, d);
  rc = download_f16(m, dy, (size_t)rows * d, out);
  for (half_t* p : {dx, dg, db, dy})
    if (p) (void)hipFree(p);
  return rc;
}

// q,k,v,out: float32 [B][T][H*64]
int32_t fw_test_attention(fw_model* fm, const float* q, const float* k, const float* v, int32_t B, int32_t H,
                          int32_t T, float* out) {
  FW_CHECK_ARG(fm && q && k && v && out, "null argument");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  const int d = H * 64, tp = (T + 63) / 64 * 64;
  const size_t n = (size_t)B * T * d;
  half_t *dq = nullptr, *dk = nullptr, *dvt = nullptr, *dout = nullptr;
  int rc;
  if ((rc = upload_f16(m, q, n, &dq))) return rc;
  if ((rc = upload_f16(m, k, n, &dk))) return rc;
  std::vector<float> vt((size_t)B * d * tp, 0.f);
  for (int b = 0; b < B; ++b)
    for (int t = 0; t < T; ++t)
      for (int c = 0; c < d; ++c) vt[((size_t)b * d + c) * tp + t] = v[((size_t)b * T + t) * d + c];
  if ((rc = upload_f16(m, vt.data(), vt.size(), &dvt))) return rc;
  if ((rc = dev_alloc_t(&dout, n))) return rc;
  fwk::launch_attn_enc(m->stream, dq, dk, d, (int64_t)T * d, dvt, tp, (int64_t)d * tp, dout, d, (int64_t)T * d, B, H,
                       T);
  rc = download_f16(m, dout, n, out);
  for (half_t* p : {dq, dk, dvt, dout})
    if (p) (void)hipFree(p);
  return rc;
}

// measurement hook (profiles/attn_bench.py): mean milliseconds of one launch of the encoder self-attention on
// device-resident pseudo-random Q | K ([B][T][2d] as the fused projection leaves them) and V^T ([B][d][T padded]);
// variant: reserved (0)
int32_t fw_bench_attention(fw_model* fm, int32_t B, int32_t H, int32_t T, int32_t variant, int32_t iters, float* ms_out) {
  FW_CHECK_ARG(fm && ms_out && B > 0 && H > 0 && T > 0 && iters > 0, "bad argument");
  Model* m = &fm->impl;
  std::lock_guard<std::mutex> lk(m->mu);
  FW_HIP(hipSetDevice(m->device));
  const int d = H * 64, tp = (T + 63) / 64 * 64;
  const size_t nqk = (size_t)B * T * 2 * d, nvt = (size_t)B * d * tp, no = (size_t)B * T * d;
  half_t *dqk = nullptr, *dvt = nullptr, *dout = nullptr;
  int rc;
  auto cleanup = [&]() { for (half_t* p : {dqk, dvt, dout}) if (p) (void)hipFree(p); };
  if ((rc = dev_alloc_t(&dqk, nqk)) || (rc = dev_alloc_t(&dvt, nvt)) || (rc = dev_alloc_t(&dout, no))) { cleanup(); return rc; }
  {
    std::vector<uint16_t> h(std::max(nqk, nvt));
    uint32_t sd = 777u;
    for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = f32_to_f16_bits(((int)(sd >> 16) % 2001 - 1000) * 2e-3f); }
    FW_HIP(hipMemcpy(dqk, h.data(), nqk * 2, hipMemcpyHostToDevice));
    FW_HIP(hipMemcpy(dvt, h.data(), nvt * 2, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  FW_HIP(hipEventCreate(&e0));
  FW_HIP(hipEventCreate(&e1));
  auto go = [&]() {   // variant: the workgroup -> (chunk, head, query tile) mapping (attn_enc.hip: 0 = XCD-aware, 1 = round 3's)
    fwk::launch_attn_enc(m->stream, dqk, dqk + d, 2 * d, (int64_t)T * 2 * d, dvt, tp, (int64_t)d * tp, dout, d, (int64_t)T * d, B,
                         H, T, variant);
  };
  go();
  FW_HIP(hipEventRecord(e0, m->stream));
  for (int i = 0; i < iters; ++i) go();
  FW_HIP(hipEventRecord(e1, m->stream));
  hipError_t he = hipEventSynchronize(e1);
  float ms = 0.f;
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cleanup();
  if (he != hipSuccess) { set_error("attention bench failed: %s", hipGetErrorString(he)); return FW_ERUNTIME; }
  *ms_out = ms / (float)iters;
  return FW_OK;
}

}  // extern "C"
