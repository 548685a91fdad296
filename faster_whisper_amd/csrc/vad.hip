// Silero VAD v6 on the GPU (MI355X-first form of csrc/vad_host.cpp).
//
// Validated on hardware against oracle/silero.py and the host path since round 4 (tests/test_gpu_vad.py, tests/test_gpu_c5.py);
// SileroVADModel(device="cuda") selects it.
//
// Two kernels (one hour of audio = 112 500 windows):
//   vad_front_kernel   one workgroup per 512-sample window: reflect pad, STFT as a [258 x 256] matrix product
//                      over 4 frames, magnitude, the four k=3 convolutions, the LSTM input projection
//                      gx = W feat + Wb + Rb.  Window-parallel: 112 500 workgroups, the 1.2 MB of weights stay
//                      in L2; all reduction loops run in the order of the definition (k outer, channel inner).
//   vad_lstm_kernel    ONE workgroup of 512 threads walks the windows sequentially: thread r keeps row r of the
//                      recurrence matrix R[512][128] in 128 VGPRs, h[128] lives in LDS (broadcast reads), two
//                      barriers per step; c stays in the registers of the first 128 threads.
#include <math.h>
#include <stdlib.h>

#include "common.h"
#include "engine.h"
#include "vad_model.h"

namespace {

struct VadDev {
  int device = 0;
  hipStream_t st = nullptr;
  float *basis_t = nullptr, *cw_t[4] = {nullptr, nullptr, nullptr, nullptr}, *cb[4] = {nullptr, nullptr, nullptr, nullptr};
  float *lw_t = nullptr, *lr_t = nullptr, *lb = nullptr, *dw = nullptr;
  float db = 0.f;
};

__device__ __forceinline__ float sigmoid_dev(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(256) void vad_front_kernel(const float* __restrict__ win, const float* __restrict__ basis_t,
                                                        const float* __restrict__ cw0, const float* __restrict__ cb0,
                                                        const float* __restrict__ cw1, const float* __restrict__ cb1,
                                                        const float* __restrict__ cw2, const float* __restrict__ cb2,
                                                        const float* __restrict__ cw3, const float* __restrict__ cb3,
                                                        const float* __restrict__ lw_t, const float* __restrict__ lb,
                                                        float* __restrict__ gx, const float* __restrict__ audio,
                                                        int64_t n_windows) {
  __shared__ float xp[kPadded];
  __shared__ float spec[kFrames][2 * kBins];
  __shared__ float mag[kFrames][kBins + 3];
  __shared__ float a0[4][128];
  __shared__ float a1[2][64];
  __shared__ float a2[64];
  __shared__ float feat[128];
  const int tid = threadIdx.x;
  // sample j of this window.  win: the caller framed [context 64 | window 512] rows.  audio (round 6): framed HERE, as the
  // reference's SileroVADModel.__call__ does on the host (vad.py:318-336): the context of window i is the tail of window
  // i - 1 (zeros for the first window) and the last 64 samples of the LAST window are zeroed (its in-place
  // `context[-1] = 0` on a view) — no [n][576] copy of the recording is built on the host or sent over PCIe
  const int64_t wi = blockIdx.x;
  const float* w = win ? win + (size_t)wi * kWin : nullptr;
  auto sample = [&](int j) -> float {
    if (w) return w[j];
    if (j < 64) return wi == 0 ? 0.f : audio[(wi - 1) * 512 + 448 + j];
    const int jj = j - 64;
    if (wi == n_windows - 1 && jj >= 448) return 0.f;
    return audio[wi * 512 + jj];
  };
  for (int i = tid; i < kWin; i += 256) xp[kPad + i] = sample(i);
  if (tid < kPad) {
    xp[tid] = sample(kPad - tid);                // reflect, edge sample not repeated
    xp[kPad + kWin + tid] = sample(kWin - 2 - tid);
  }
  __syncthreads();
  // STFT: frames 1..4 of the stride-128 convolution (frame 0 is sliced away by the graph)
  for (int k = tid; k < 2 * kBins; k += 256) {
    float acc[kFrames] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < kTaps; ++i) {
      const float b = basis_t[(size_t)i * 2 * kBins + k];
#pragma unroll
      for (int t = 0; t < kFrames; ++t) acc[t] += b * xp[(t + 1) * kHop + i];
    }
#pragma unroll
    for (int t = 0; t < kFrames; ++t) spec[t][k] = acc[t];
  }
  __syncthreads();
  for (int idx = tid; idx < kFrames * kBins; idx += 256) {
    const int t = idx / kBins, c = idx - t * kBins;
    const float re = spec[t][c], im = spec[t][c + kBins];
    mag[t][c] = sqrtf(re * re + im * im);
  }
  __syncthreads();
  {  // conv 129 -> 128, stride 1: thread = (output channel, frame pair)
    const int o = tid & 127, tp = tid >> 7;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = 2 * tp + tt;
      float acc = cb0[o];
      for (int k = 0; k < 3; ++k) {
        const int ti = t + k - 1;
        if (ti < 0 || ti >= 4) continue;
        for (int c = 0; c < kBins; ++c) acc += cw0[((size_t)c * 3 + k) * 128 + o] * mag[ti][c];
      }
      a0[t][o] = acc > 0.f ? acc : 0.f;
    }
  }
  __syncthreads();
  if (tid < 128) {  // conv 128 -> 64, stride 2: 2 output frames
    const int o = tid & 63, t = tid >> 6;
    float acc = cb1[o];
    for (int k = 0; k < 3; ++k) {
      const int ti = 2 * t + k - 1;
      if (ti < 0 || ti >= 4) continue;
      for (int c = 0; c < 128; ++c) acc += cw1[((size_t)c * 3 + k) * 64 + o] * a0[ti][c];
    }
    a1[t][o] = acc > 0.f ? acc : 0.f;
  }
  __syncthreads();
  if (tid < 64) {   // conv 64 -> 64, stride 2: 1 output frame
    float acc = cb2[tid];
    for (int k = 0; k < 3; ++k) {
      const int ti = k - 1;
      if (ti < 0 || ti >= 2) continue;
      for (int c = 0; c < 64; ++c) acc += cw2[((size_t)c * 3 + k) * 64 + tid] * a1[ti][c];
    }
    a2[tid] = acc > 0.f ? acc : 0.f;
  }
  __syncthreads();
  if (tid < 128) {  // conv 64 -> 128 on one frame: only the centre tap meets data
    float acc = cb3[tid];
    for (int c = 0; c < 64; ++c) acc += cw3[((size_t)c * 3 + 1) * 128 + tid] * a2[c];
    feat[tid] = acc > 0.f ? acc : 0.f;
  }
  __syncthreads();
  for (int g = tid; g < kGates; g += 256) {
    float acc = lb[g];
    for (int c = 0; c < kHidden; ++c) acc += lw_t[(size_t)c * kGates + g] * feat[c];
    gx[(size_t)blockIdx.x * kGates + g] = acc;
  }
}

__global__ __launch_bounds__(512) void vad_lstm_kernel(const float* __restrict__ gx, const float* __restrict__ lr_t,
                                                       const float* __restrict__ dw, float db, int64_t n,
                                                       float* __restrict__ h_io, float* __restrict__ c_io,
                                                       float* __restrict__ probs) {
  __shared__ float hs[kHidden];
  __shared__ float gs[kGates];
  __shared__ float part[2];
  const int r = threadIdx.x;
  float R[kHidden];
#pragma unroll
  for (int k = 0; k < kHidden; ++k) R[k] = lr_t[(size_t)k * kGates + r];   // row r of the recurrence matrix
  float cell = 0.f;
  if (r < kHidden) {
    hs[r] = h_io[r];
    cell = c_io[r];
  }
  const float dwr = r < kHidden ? dw[r] : 0.f;
  __syncthreads();
  float g_next = n > 0 ? gx[r] : 0.f;
  for (int64_t i = 0; i < n; ++i) {
    float g = g_next;
    if (i + 1 < n) g_next = gx[(size_t)(i + 1) * kGates + r];           // in flight during the dot product
    // four interleaved partial sums (k = 0, 4, 8, ... / 1, 5, ... / ...): the step is a chain of DEPENDENT fused
    // multiply-adds — 128 in a row were ~0.45 us of the ~1.1 us a step took, and an 8 h recording is 900 000 steps in front
    // of the first transcribed chunk (profiles/r06_vad_bench.json); the order of the additions differs from the host path's
    // k-ascending sum by fp32 round-off only (tests/test_gpu_vad.py: 1e-4 / 5e-5 bounds, measured ~1e-7)
    float g1 = 0.f, g2 = 0.f, g3 = 0.f;
#pragma unroll
    for (int k = 0; k < kHidden; k += 4) {                               // LDS broadcast reads
      g += R[k] * hs[k];
      g1 += R[k + 1] * hs[k + 1];
      g2 += R[k + 2] * hs[k + 2];
      g3 += R[k + 3] * hs[k + 3];
    }
    g = (g + g1) + (g2 + g3);
    gs[r] = g;
    __syncthreads();
    if (r < kHidden) {   // ONNX gate order i, o, f, c
      const float ig = sigmoid_dev(gs[r]), og = sigmoid_dev(gs[kHidden + r]), fg = sigmoid_dev(gs[2 * kHidden + r]);
      cell = fg * cell + ig * tanhf(gs[3 * kHidden + r]);
      const float hn = og * tanhf(cell);
      hs[r] = hn;
      float y = hn > 0.f ? dwr * hn : 0.f;
      y = wave_sum(y);
      if ((r & 63) == 0) part[r >> 6] = y;
    }
    __syncthreads();
    if (r == 0) probs[i] = sigmoid_dev(db + part[0] + part[1]);
  }
  if (r < kHidden) {
    h_io[r] = hs[r];
    c_io[r] = cell;
  }
}

// (Round 6 measured a second form — four waves, two gate rows per lane, the four gates of a unit meeting by one cross-lane
//  exchange, h double-buffered so that a step has ONE barrier — at 2.30 s for the 900 000 windows of an 8 h recording against
//  1.79 s for the form above: 256 registers of recurrence matrix per lane and half the waves to hide the LDS broadcast reads.
//  Removed; profiles/r06_vad_bench_call9_lstm_forms.json.)
template <typename T>
int upload(const std::vector<T>& src, T** dst) {
  FW_HIP(hipMalloc(reinterpret_cast<void**>(dst), src.size() * sizeof(T)));
  FW_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return FW_OK;
}

int ensure_dev(fw_vad* v, int device) {
  if (v->dev) {
    if (static_cast<VadDev*>(v->dev)->device == device) return FW_OK;
    fw_vad_dev_release(v->dev);
    v->dev = nullptr;
  }
  FW_HIP(hipSetDevice(device));
  VadDev* d = new VadDev();
  d->device = device;
  v->dev = d;   // released by fw_vad_free even if an upload below fails
  const Vad& m = v->impl;
  int rc;
  if ((rc = upload(m.basis_t, &d->basis_t))) return rc;
  for (int i = 0; i < 4; ++i) {
    if ((rc = upload(m.cw_t[i], &d->cw_t[i]))) return rc;
    if ((rc = upload(m.cb[i], &d->cb[i]))) return rc;
  }
  if ((rc = upload(m.lw_t, &d->lw_t))) return rc;
  if ((rc = upload(m.lr_t, &d->lr_t))) return rc;
  if ((rc = upload(m.lb, &d->lb))) return rc;
  if ((rc = upload(m.dw, &d->dw))) return rc;
  d->db = m.db;
  FW_HIP(hipStreamCreateWithFlags(&d->st, hipStreamNonBlocking));
  return FW_OK;
}

}  // namespace

void fw_vad_dev_release(void* dev) {
  VadDev* d = static_cast<VadDev*>(dev);
  if (!d) return;
  (void)hipSetDevice(d->device);
  float* ptrs[] = {d->basis_t, d->cw_t[0], d->cw_t[1], d->cw_t[2], d->cw_t[3], d->cb[0], d->cb[1], d->cb[2], d->cb[3],
                   d->lw_t,    d->lr_t,    d->lb,      d->dw};
  for (float* p : ptrs)
    if (p) (void)hipFree(p);
  if (d->st) (void)hipStreamDestroy(d->st);
  delete d;
}

// windows != null: n framed rows of 576 samples; audio != null: n * 512 samples, framed on the device
static int32_t vad_forward_dev_impl(fw_vad* v, int32_t device_index, const float* windows, const float* audio, int64_t n,
                                    float* h, float* c, float* probs) {
  FW_CHECK_ARG(v && h && c && (n == 0 || ((windows || audio) && probs)), "null argument");
  FW_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31), "window count out of range");
  if (n == 0) return FW_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_index < 0 || device_index >= ndev) {
    fw::set_error("fw_vad_forward_dev: no HIP device %d", device_index);
    return FW_ENODEV;
  }
  int rc = ensure_dev(v, device_index);
  if (rc) return rc;
  VadDev* d = static_cast<VadDev*>(v->dev);
  FW_HIP(hipSetDevice(d->device));
  float *dwin = nullptr, *dgx = nullptr, *dh = nullptr, *dc = nullptr, *dp = nullptr;
  auto cleanup = [&]() {
    for (float* p : {dwin, dgx, dh, dc, dp})
      if (p) (void)hipFree(p);
  };
#define VAD_TRY(call)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      fw::set_error("%s failed: %s", #call, hipGetErrorString(e_));                          \
      cleanup();                                                                             \
      return e_ == hipErrorOutOfMemory ? FW_ENOMEM : FW_ERUNTIME;                            \
    }                                                                                        \
  } while (0)
  const size_t in_floats = windows ? (size_t)n * kWin : (size_t)n * 512;
  VAD_TRY(hipMalloc(reinterpret_cast<void**>(&dwin), in_floats * sizeof(float)));
  VAD_TRY(hipMalloc(reinterpret_cast<void**>(&dgx), (size_t)n * kGates * sizeof(float)));
  VAD_TRY(hipMalloc(reinterpret_cast<void**>(&dh), kHidden * sizeof(float)));
  VAD_TRY(hipMalloc(reinterpret_cast<void**>(&dc), kHidden * sizeof(float)));
  VAD_TRY(hipMalloc(reinterpret_cast<void**>(&dp), (size_t)n * sizeof(float)));
  VAD_TRY(hipMemcpyAsync(dwin, windows ? windows : audio, in_floats * sizeof(float), hipMemcpyHostToDevice, d->st));
  VAD_TRY(hipMemcpyAsync(dh, h, kHidden * sizeof(float), hipMemcpyHostToDevice, d->st));
  VAD_TRY(hipMemcpyAsync(dc, c, kHidden * sizeof(float), hipMemcpyHostToDevice, d->st));
  vad_front_kernel<<<(unsigned)n, 256, 0, d->st>>>(windows ? dwin : nullptr, d->basis_t, d->cw_t[0], d->cb[0], d->cw_t[1],
                                                   d->cb[1], d->cw_t[2], d->cb[2], d->cw_t[3], d->cb[3], d->lw_t, d->lb, dgx,
                                                   windows ? nullptr : dwin, n);
  vad_lstm_kernel<<<1, 512, 0, d->st>>>(dgx, d->lr_t, d->dw, d->db, n, dh, dc, dp);
  VAD_TRY(hipGetLastError());
  VAD_TRY(hipMemcpyAsync(probs, dp, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, d->st));
  VAD_TRY(hipMemcpyAsync(h, dh, kHidden * sizeof(float), hipMemcpyDeviceToHost, d->st));
  VAD_TRY(hipMemcpyAsync(c, dc, kHidden * sizeof(float), hipMemcpyDeviceToHost, d->st));
  VAD_TRY(hipStreamSynchronize(d->st));
#undef VAD_TRY
  cleanup();
  return FW_OK;
}

extern "C" int32_t fw_vad_forward_dev(fw_vad* v, int32_t device_index, const float* windows, int64_t n, float* h,
                                      float* c, float* probs) {
  return vad_forward_dev_impl(v, device_index, windows, nullptr, n, h, c, probs);
}

extern "C" int32_t fw_vad_forward_audio_dev(fw_vad* v, int32_t device_index, const float* audio, int64_t n_samples,
                                            float* h, float* c, float* probs) {
  FW_CHECK_ARG(n_samples >= 0 && n_samples % 512 == 0, "the recording must be padded to a multiple of 512 samples");
  return vad_forward_dev_impl(v, device_index, nullptr, audio, n_samples / 512, h, c, probs);
}
