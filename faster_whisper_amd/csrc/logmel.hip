// K1 — log-mel front end on gfx950.
//
// Restates faster_whisper/feature_extractor.py:198-230 (FeatureExtractor.__call__)
// for a batch of ragged PCM chunks, entirely in HBM/LDS:
//   zero-pad 160 samples (:210-211) -> centre reflect-pad 200 (:117-121) ->
//   400-sample frames, hop 160 (:157-168) -> periodic Hann (:213,:170-171) ->
//   rDFT(400) (:189) -> drop last frame, |.|^2 (:222) -> mel filterbank matmul (:224) ->
//   log10(clip(.,1e-10)) (:226) -> max(x, globalmax-8) (:227) -> (x+4)/4 (:228)
// then the batched driver's [..., :-1] (transcribe.py:464) and pad_or_trim to
// 3000 frames with 0.0 (audio.py:111-123).
//
// MI355X mapping: one workgroup = 32 frames of one chunk. The windowed frames sit
// in LDS ([32][401] floats, odd stride => conflict-free column reads); the DFT and
// the mel projection are exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, an fmaf chain:
// same precision class as the reference's float32 pocketfft + sgemm). Twiddles come
// from a 400-entry cos table in LDS indexed by (freq*n) mod 400. PCM is read once
// (coalesced), the [n_mels, frames] result is written once; the per-chunk global
// max is one atomicMax per wave.
#include "common.h"
#include "kernels.h"

#define LM_FRAMES 32
#define LM_NFFT 400
#define LM_HOP 160
#define LM_FSTRIDE 401
#define LM_PSTRIDE 225
#define LM_NTILE_F 7 /* 7*32 = 224 >= 201 bins */

// consts: [0,400) cos(2*pi*j/400); [400,800) periodic Hann window (both computed in
// double on the host and rounded to float32, like numpy does).
__global__ __launch_bounds__(256) void logmel_power_kernel(
    const float* __restrict__ pcm, const int64_t* __restrict__ offsets,
    const float* __restrict__ consts, const float* __restrict__ filtT, int mel_pad,
    float* __restrict__ raw, int64_t raw_bstride, int raw_stride, int* __restrict__ chunk_max,
    int n_mels) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* frames = reinterpret_cast<float*>(smem_raw);               // [32][401], later power [32][225]
  float* ctab = frames + LM_FRAMES * LM_FSTRIDE;                    // [400]
  const int b = blockIdx.y;
  const int64_t off = offsets[b];
  const int N = (int)(offsets[b + 1] - off);
  const int L = N + LM_HOP;              // waveform + 160 zeros
  const int nf = L / LM_HOP;             // frames after the reference's stft[..., :-1]
  const int f0 = blockIdx.x * LM_FRAMES;
  if (f0 >= nf) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;

  for (int i = tid; i < LM_NFFT; i += 256) ctab[i] = consts[i];
  const int period = 2 * (L - 1);
  for (int idx = tid; idx < LM_FRAMES * LM_NFFT; idx += 256) {
    const int fr = idx / LM_NFFT, n = idx - fr * LM_NFFT;
    const int f = f0 + fr;
    float v = 0.f;
    if (f < nf) {
      int p = f * LM_HOP + n - LM_NFFT / 2;
      int q = p % period;
      if (q < 0) q += period;
      if (q >= L) q = period - q;
      const float x = (q < N) ? pcm[off + q] : 0.f;
      v = x * consts[LM_NFFT + n];
    }
    frames[fr * LM_FSTRIDE + n] = v;
  }
  __syncthreads();

  // ---- DFT: D[freq][frame] = sum_n tw[freq][n] * frame[frame][n] -------------------
  floatx16 re0 = {0}, im0 = {0}, re1 = {0}, im1 = {0};
  const int t0 = wave, t1 = wave + 4;  // freq tiles of this wave (t1 valid if < 7)
  const bool has1 = t1 < LM_NTILE_F;
  {
    const int fq0 = t0 * 32 + l31, fq1 = t1 * 32 + l31;
    int i0 = (fq0 * hi) % LM_NFFT, i1 = (fq1 * hi) % LM_NFFT;
    const int st0 = (2 * fq0) % LM_NFFT, st1 = (2 * fq1) % LM_NFFT;
    const float* frow = frames + l31 * LM_FSTRIDE + hi;
#pragma unroll 4
    for (int s = 0; s < LM_NFFT / 2; ++s) {
      const float x = frow[2 * s];
      const float c0 = ctab[i0];
      int j0 = i0 + 300; if (j0 >= LM_NFFT) j0 -= LM_NFFT;
      const float s0 = ctab[j0];
      re0 = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, x, re0, 0, 0, 0);
      im0 = __builtin_amdgcn_mfma_f32_32x32x2f32(s0, x, im0, 0, 0, 0);
      i0 += st0; if (i0 >= LM_NFFT) i0 -= LM_NFFT;
      if (has1) {
        const float c1 = ctab[i1];
        int j1 = i1 + 300; if (j1 >= LM_NFFT) j1 -= LM_NFFT;
        const float s1 = ctab[j1];
        re1 = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, x, re1, 0, 0, 0);
        im1 = __builtin_amdgcn_mfma_f32_32x32x2f32(s1, x, im1, 0, 0, 0);
        i1 += st1; if (i1 >= LM_NFFT) i1 -= LM_NFFT;
      }
    }
  }
  __syncthreads();  // every wave is done reading the frames: overlay the power tile
  float* pw = frames;  // [32 frames][225]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * hi;
    pw[l31 * LM_PSTRIDE + t0 * 32 + i] = re0[r] * re0[r] + im0[r] * im0[r];
    if (has1) pw[l31 * LM_PSTRIDE + t1 * 32 + i] = re1[r] * re1[r] + im1[r] * im1[r];
  }
  __syncthreads();

  // ---- mel: D[mel][frame] = sum_bin filt[mel][bin] * power[frame][bin] ---------------
  const int n_mtiles = (n_mels + 31) >> 5;
  float wmax = -3.0e38f;
  for (int mt = wave; mt < n_mtiles; mt += 4) {
    floatx16 acc = {0};
    const float* prow = pw + l31 * LM_PSTRIDE + hi;
    const float* frow = filtT + (size_t)hi * mel_pad + mt * 32 + l31;
#pragma unroll 4
    for (int s = 0; s < 101; ++s) {  // bins 0..201 (row 201 of filtT is zero)
      const float a = frow[(size_t)(2 * s) * mel_pad];
      const float p = prow[2 * s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, p, acc, 0, 0, 0);
    }
    const int f = f0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mel = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (mel < n_mels && f < nf) {
        const float v = log10f(fmaxf(acc[r], 1e-10f));
        raw[(size_t)b * raw_bstride + (size_t)mel * raw_stride + f] = v;
        wmax = fmaxf(wmax, v);
      }
    }
  }
  wmax = wave_max(wmax);
  if (lane == 0 && wmax > -1.0e38f) atomicMax(chunk_max + b, float_to_ordered(wmax));
}

// clamp to (max - 8), affine, drop the trailing frame(s), zero-pad to out_frames.
// Writes (optionally) the reference-layout float32 [B][n_mels][out_frames] and/or the
// encoder's fp16 channel-last image [B][out_frames + 2][c_pad] (row 0 and the last row
// are the conv zero padding; channels >= n_mels are zero).
__global__ __launch_bounds__(1024) void logmel_finish_kernel(
    const float* __restrict__ raw, int64_t raw_bstride, int raw_stride,
    const int* __restrict__ chunk_max, const int64_t* __restrict__ offsets, int drop_last,
    int n_mels, int out_frames, float* __restrict__ out_f32, half_t* __restrict__ out_cl,
    int c_pad, int* __restrict__ n_frames_out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int N = (int)(offsets[b + 1] - offsets[b]);
  const int nf = (N + LM_HOP) / LM_HOP;
  int keep = nf - drop_last;  // frames that carry data
  if (keep > out_frames) keep = out_frames;
  const float floorv = ordered_to_float(chunk_max[b]) - 8.0f;
  const int f0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  if (blockIdx.x == 0 && blockIdx.y == 0 && tx == 0 && ty == 0 && n_frames_out) n_frames_out[b] = keep;
  {
    const int mel = m0 + ty, f = f0 + tx;
    float v = 0.f;
    if (mel < n_mels && f < keep) {
      v = raw[(size_t)b * raw_bstride + (size_t)mel * raw_stride + f];
      v = (fmaxf(v, floorv) + 4.0f) * 0.25f;
    }
    if (out_f32 && mel < n_mels && f < out_frames)
      out_f32[((size_t)b * n_mels + mel) * out_frames + f] = v;
    tile[ty][tx] = v;
  }
  if (out_cl) {
    __syncthreads();
    const int f = f0 + ty, c = m0 + tx;
    if (f < out_frames && c < c_pad)
      out_cl[((size_t)b * (out_frames + 2) + 1 + f) * c_pad + c] = (half_t)tile[tx][ty];
  }
}

// zero the two conv-padding rows of the channel-last image and reset chunk maxima
__global__ void logmel_prep_kernel(half_t* out_cl, int out_frames, int c_pad, int* chunk_max, int B) {
  const int b = blockIdx.x;
  if (chunk_max && threadIdx.x == 0) chunk_max[b] = float_to_ordered(-3.0e38f);
  if (out_cl) {
    half_t* base = out_cl + (size_t)b * (out_frames + 2) * c_pad;
    for (int c = threadIdx.x; c < c_pad; c += blockDim.x) {
      base[c] = (half_t)0.f;
      base[(size_t)(out_frames + 1) * c_pad + c] = (half_t)0.f;
    }
  }
}

// features float32 [B][n_mels][3000] (host API path) -> fp16 channel-last image
__global__ __launch_bounds__(1024) void features_to_cl_kernel(const float* __restrict__ feats, int n_mels, int frames,
                                                               half_t* __restrict__ out_cl, int c_pad) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int f0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x, ty = threadIdx.y;
  {
    const int mel = m0 + ty, f = f0 + tx;
    tile[ty][tx] = (mel < n_mels && f < frames) ? feats[((size_t)b * n_mels + mel) * frames + f] : 0.f;
  }
  __syncthreads();
  const int f = f0 + ty, c = m0 + tx;
  if (f < frames && c < c_pad) out_cl[((size_t)b * (frames + 2) + 1 + f) * c_pad + c] = (half_t)tile[tx][ty];
}

namespace fwk {

size_t logmel_lds_bytes() { return (size_t)(LM_FRAMES * LM_FSTRIDE + LM_NFFT) * sizeof(float); }

void launch_logmel(hipStream_t st, const float* pcm_dev, const int64_t* offsets_dev, int B, int max_frames_total,
                   const float* consts, const float* filtT, int mel_pad, int n_mels, float* raw,
                   int64_t raw_bstride, int raw_stride, int* chunk_max, int drop_last, int out_frames,
                   float* out_f32, half_t* out_cl, int c_pad, int* n_frames_out) {
  logmel_prep_kernel<<<B, 128, 0, st>>>(out_cl, out_frames, c_pad, chunk_max, B);
  dim3 g1((max_frames_total + LM_FRAMES - 1) / LM_FRAMES, B);
  logmel_power_kernel<<<g1, 256, logmel_lds_bytes(), st>>>(pcm_dev, offsets_dev, consts, filtT, mel_pad, raw,
                                                           raw_bstride, raw_stride, chunk_max, n_mels);
  const int cdim = out_cl ? (c_pad > n_mels ? c_pad : n_mels) : n_mels;
  dim3 g2((out_frames + 31) / 32, (cdim + 31) / 32, B);
  logmel_finish_kernel<<<g2, dim3(32, 32), 0, st>>>(raw, raw_bstride, raw_stride, chunk_max, offsets_dev, drop_last,
                                                     n_mels, out_frames, out_f32, out_cl, c_pad, n_frames_out);
}

void launch_features_to_cl(hipStream_t st, const float* feats_dev, int B, int n_mels, int frames, half_t* out_cl,
                           int c_pad) {
  logmel_prep_kernel<<<B, 128, 0, st>>>(out_cl, frames, c_pad, (int*)nullptr, 0);
  dim3 g((frames + 31) / 32, (c_pad + 31) / 32, B);
  features_to_cl_kernel<<<g, dim3(32, 32), 0, st>>>(feats_dev, n_mels, frames, out_cl, c_pad);
}

}  // namespace fwk
