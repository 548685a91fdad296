// Silero VAD v6 forward pass, host C++.
//
// The reference runs this network on the CPU through onnxruntime with one intra-op thread
// (faster_whisper/vad.py:295-351, asset silero_vad_v6.onnx); this is the native equivalent behind the C ABI:
// the window-parallel front end (reflect pad, STFT-as-convolution, magnitude, four small convolutions, the
// LSTM's input projection) runs on a pool of host threads, the LSTM recurrence over the windows is sequential.
// Network (ONNX node list, restated in oracle/silero.py):
//   [N,576] -> reflect pad 128|128 -> conv basis[258][256] stride 128, frame 0 dropped -> |.| [129][4]
//   -> conv 129->128 k3 s1 -> conv 128->64 k3 s2 -> conv 64->64 k3 s2 -> conv 64->128 k3 s1 (ReLU each, zero pad 1)
//   -> LSTM(128, hidden 128; ONNX gate order i,o,f,c; the N windows are the sequence) -> ReLU -> 128->1 -> sigmoid
// The device version (csrc/vad.hip, fw_vad_forward_dev: one workgroup per window + one persistent workgroup for the
// recurrence) exists and is tested against this one; the Python front uses this host path by default, as the reference
// runs the VAD on the CPU — one hour of audio is 112 500 windows = 0.16 TFLOP, well under a second on the box's cores.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include <new>

#include "../../include/fwamd.h"

// error reporting of the library (engine.hip); this file is plain host C++ built with g++ (no HIP headers)
namespace fw {
void set_error(const char* fmt, ...);
}
#define FW_CHECK_ARG(cond, ...)    \
  do {                             \
    if (!(cond)) {                 \
      fw::set_error(__VA_ARGS__);  \
      return FW_EINVAL;            \
    }                              \
  } while (0)

#include "vad_model.h"

namespace {

#define VAD_SIMD __attribute__((target_clones("avx512f", "avx2", "default")))

// e^x without a libm call so that the LSTM's 640 transcendentals per step vectorise: Cody-Waite reduction
// x = n ln2 + r, |r| <= ln2/2, degree-6 Taylor polynomial (truncation 1.2e-7 relative), 2^n through the exponent
// bits.  |result - expf| <= 2 ulp; the speech probabilities move by < 1e-6.
__attribute__((always_inline)) inline float fast_expf(float x) {
  x = x < -87.0f ? -87.0f : (x > 88.0f ? 88.0f : x);
  const float n = nearbyintf(x * 1.44269504088896341f);
  float r = x - n * 0.693145751953125f;
  r -= n * 1.42860682030941723212e-6f;
  float p = 1.0f / 720.0f;
  p = p * r + 1.0f / 120.0f;
  p = p * r + 1.0f / 24.0f;
  p = p * r + 1.0f / 6.0f;
  p = p * r + 0.5f;
  p = p * r + 1.0f;
  p = p * r + 1.0f;
  const int32_t bits = ((int32_t)n + 127) << 23;
  float scale;
  memcpy(&scale, &bits, sizeof(scale));
  return p * scale;
}
__attribute__((always_inline)) inline float sigmoidf(float x) { return 1.0f / (1.0f + fast_expf(-x)); }
__attribute__((always_inline)) inline float fast_tanhf(float x) { return 1.0f - 2.0f / (fast_expf(2.0f * x) + 1.0f); }

// y[n] += sum_k Wt[k*ldw + n] * x[k]  (k ascending for every n).  Outputs are processed in blocks of 64 that stay
// in vector registers across the whole k loop (4 zmm / 8 ymm), so the loop streams Wt once and touches y once.
__attribute__((always_inline)) inline void matvec_t(const float* Wt, size_t ldw, const float* x, float* y, int K,
                                                    int N) {
  int n0 = 0;
  for (; n0 + 64 <= N; n0 += 64) {
    float acc[64];
    for (int j = 0; j < 64; ++j) acc[j] = y[n0 + j];
    for (int k = 0; k < K; ++k) {
      const float xv = x[k];
      const float* wr = Wt + (size_t)k * ldw + n0;
      for (int j = 0; j < 64; ++j) acc[j] += wr[j] * xv;
    }
    for (int j = 0; j < 64; ++j) y[n0 + j] = acc[j];
  }
  if (n0 < N) {
    const int nt = N - n0;
    float acc[64];
    for (int j = 0; j < nt; ++j) acc[j] = y[n0 + j];
    for (int k = 0; k < K; ++k) {
      const float xv = x[k];
      const float* wr = Wt + (size_t)k * ldw + n0;
      for (int j = 0; j < nt; ++j) acc[j] += wr[j] * xv;
    }
    for (int j = 0; j < nt; ++j) y[n0 + j] = acc[j];
  }
}

// x [Tin][Cin] -> y [Tout][Cout] (time-major), kernel 3, zero padding 1, ReLU; w_t [Cin][3][Cout]
VAD_SIMD void conv_relu(const float* x, int cin, int tin, const float* w_t, const float* b, int cout, int stride,
                        int tout, float* y) {
  for (int t = 0; t < tout; ++t) {
    float* yt = y + (size_t)t * cout;
    for (int o = 0; o < cout; ++o) yt[o] = b[o];
    for (int k = 0; k < 3; ++k) {
      const int ti = t * stride + k - 1;
      if (ti < 0 || ti >= tin) continue;
      matvec_t(w_t + (size_t)k * cout, (size_t)3 * cout, x + (size_t)ti * cin, yt, cin, cout);
    }
    for (int o = 0; o < cout; ++o) yt[o] = yt[o] > 0.f ? yt[o] : 0.f;
  }
}

// one window: 576 samples -> the LSTM input projection gx[512] = W feat + Wb + Rb
VAD_SIMD void front_end(const Vad& v, const float* win, float* gx) {
  float xp[kPadded];
  memcpy(xp + kPad, win, kWin * sizeof(float));
  for (int j = 0; j < kPad; ++j) {
    xp[j] = win[kPad - j];                        // reflect (edge sample not repeated)
    xp[kPad + kWin + j] = win[kWin - 2 - j];
  }
  float mag[kFrames * kBins];                     // time-major [4][129]
  float spec[2 * kBins];
  for (int t = 0; t < kFrames; ++t) {
    const float* fr = xp + (t + 1) * kHop;        // frame 0 of the convolution is sliced away by the graph
    for (int k = 0; k < 2 * kBins; ++k) spec[k] = 0.f;
    matvec_t(v.basis_t.data(), 2 * kBins, fr, spec, kTaps, 2 * kBins);
    for (int k = 0; k < kBins; ++k) mag[t * kBins + k] = sqrtf(spec[k] * spec[k] + spec[k + kBins] * spec[k + kBins]);
  }
  float a[4 * 128], b2[4 * 128];
  conv_relu(mag, kC[0], kT[0], v.cw_t[0].data(), v.cb[0].data(), kC[1], kStride[0], kT[1], a);
  conv_relu(a, kC[1], kT[1], v.cw_t[1].data(), v.cb[1].data(), kC[2], kStride[1], kT[2], b2);
  conv_relu(b2, kC[2], kT[2], v.cw_t[2].data(), v.cb[2].data(), kC[3], kStride[2], kT[3], a);
  conv_relu(a, kC[3], kT[3], v.cw_t[3].data(), v.cb[3].data(), kC[4], kStride[3], kT[4], b2);   // feat = b2[128]
  for (int g = 0; g < kGates; ++g) gx[g] = v.lb[g];
  matvec_t(v.lw_t.data(), kGates, b2, gx, kHidden, kGates);
}

// one LSTM step + the output head; gates in ONNX order i, o, f, c
VAD_SIMD float lstm_step(const Vad& v, const float* gx, float* h, float* c) {
  float g[kGates];
  for (int r = 0; r < kGates; ++r) g[r] = gx[r];
  matvec_t(v.lr_t.data(), kGates, h, g, kHidden, kGates);
  float yk[kHidden];
  for (int k = 0; k < kHidden; ++k) {
    const float ig = sigmoidf(g[k]), og = sigmoidf(g[kHidden + k]), fg = sigmoidf(g[2 * kHidden + k]);
    const float cn = fg * c[k] + ig * fast_tanhf(g[3 * kHidden + k]);
    const float hn = og * fast_tanhf(cn);
    c[k] = cn;
    h[k] = hn;
    yk[k] = hn > 0.f ? v.dw[k] * hn : 0.f;
  }
  float y = v.db;
  for (int k = 0; k < kHidden; ++k) y += yk[k];   // sequential sum: the order of the definition
  return sigmoidf(y);
}

}  // namespace

extern "C" {

int32_t fw_vad_create(const fw_vad_weights* w, fw_vad** out) {
  FW_CHECK_ARG(w && out, "null argument");
  FW_CHECK_ARG(w->stft_basis && w->lstm_w && w->lstm_r && w->lstm_b && w->dec_w, "null weight pointer");
  for (int i = 0; i < 4; ++i) FW_CHECK_ARG(w->conv_w[i] && w->conv_b[i], "null convolution weight %d", i);
  fw_vad* v = new fw_vad();
  Vad& m = v->impl;
  m.basis_t.resize((size_t)kTaps * 2 * kBins);
  for (int k = 0; k < 2 * kBins; ++k)
    for (int i = 0; i < kTaps; ++i) m.basis_t[(size_t)i * 2 * kBins + k] = w->stft_basis[(size_t)k * kTaps + i];
  for (int l = 0; l < 4; ++l) {
    const int cin = kC[l], cout = kC[l + 1];
    m.cw_t[l].resize((size_t)cin * 3 * cout);
    for (int o = 0; o < cout; ++o)
      for (int c = 0; c < cin; ++c)
        for (int k = 0; k < 3; ++k)
          m.cw_t[l][((size_t)c * 3 + k) * cout + o] = w->conv_w[l][((size_t)o * cin + c) * 3 + k];
    m.cb[l].assign(w->conv_b[l], w->conv_b[l] + cout);
  }
  m.lw_t.resize((size_t)kHidden * kGates);
  m.lr_t.resize((size_t)kHidden * kGates);
  for (int g = 0; g < kGates; ++g)
    for (int k = 0; k < kHidden; ++k) {
      m.lw_t[(size_t)k * kGates + g] = w->lstm_w[(size_t)g * kHidden + k];
      m.lr_t[(size_t)k * kGates + g] = w->lstm_r[(size_t)g * kHidden + k];
    }
  m.lb.resize(kGates);
  for (int g = 0; g < kGates; ++g) m.lb[g] = w->lstm_b[g] + w->lstm_b[kGates + g];
  m.dw.assign(w->dec_w, w->dec_w + kHidden);
  m.db = w->dec_b;
  *out = v;
  return FW_OK;
}

void fw_vad_free(fw_vad* v) {
  if (v && v->dev) fw_vad_dev_release(v->dev);
  delete v;
}

int32_t fw_vad_forward(fw_vad* fv, const float* windows, int64_t n, int32_t n_threads, float* h, float* c,
                       float* probs) {
  FW_CHECK_ARG(fv && h && c && (n == 0 || (windows && probs)), "null argument");
  FW_CHECK_ARG(n >= 0, "negative window count");
  if (n == 0) return FW_OK;
  const Vad& v = fv->impl;
  std::vector<float> gx;
  try {
    gx.resize((size_t)n * kGates);
  } catch (const std::bad_alloc&) {
    fw::set_error("fw_vad_forward: out of host memory for %lld windows", (long long)n);
    return FW_ENOMEM;
  }
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(nt, 64), n / 64 + 1));
  auto work = [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) front_end(v, windows + i * kWin, &gx[(size_t)i * kGates]);
  };
  if (nt == 1) {
    work(0, n);
  } else {
    std::vector<std::thread> pool;
    const int64_t per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; ++t) {
      const int64_t lo = t * per, hi = std::min<int64_t>(n, lo + per);
      if (lo < hi) pool.emplace_back(work, lo, hi);
    }
    for (auto& th : pool) th.join();
  }
  // the recurrence: the N windows are the LSTM's sequence
  for (int64_t i = 0; i < n; ++i) probs[i] = lstm_step(v, &gx[(size_t)i * kGates], h, c);
  return FW_OK;
}

}  // extern "C"
