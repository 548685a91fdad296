/* Plain-C restatement of the reference log-mel front end — TEST INFRASTRUCTURE ONLY
 * (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it).
 *
 * Follows /root/reference/faster_whisper/feature_extractor.py:
 *   get_mel_filters :25-65 (float64 math, stored float32)
 *   __call__        :198-230  zero-pad 160 (:210-211), periodic Hann (:213), centre reflect pad 200
 *                             + 400-sample frames at hop 160 (:117-121,157-168), rDFT (:189),
 *                             drop last frame and |.|^2 (:222), mel matmul (:224),
 *                             log10(clip 1e-10) (:226), max(x, max-8) (:227), (x+4)/4 (:228)
 * The DFT is evaluated by definition in double precision (no FFT), so this file is also an
 * accuracy yardstick: the reference's own float32 pocketfft result differs from it only by
 * float32 round-off (tests/test_oracle_logmel.py pins both against the golden vectors).
 *
 *   int fw_oracle_logmel_full(const float* pcm, long n, int n_mels, float* out, long out_frames)
 *        out: [n_mels][n/160 + 1]  == FeatureExtractor.__call__(pcm)
 */
#define _USE_MATH_DEFINES
#include <math.h>
#include <stdlib.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#include <string.h>

#define N_FFT 400
#define HOP 160
#define N_BIN 201

static void mel_filters(int n_mels, float* filt /* [n_mels][201] */) {
  double fftfreqs[N_BIN];
  const double val = 1.0 / (N_FFT * (1.0 / 16000.0));
  for (int k = 0; k < N_BIN; ++k) fftfreqs[k] = k * val;
  const int n = n_mels + 2;
  double* freqs = (double*)malloc(sizeof(double) * n);
  const double max_mel = 45.245640471924965, f_sp = 200.0 / 3;
  const double min_log_mel = 1000.0 / f_sp, logstep = log(6.4) / 27.0;
  const double step = max_mel / (n - 1);
  for (int i = 0; i < n; ++i) {
    const double mel = (i == n - 1) ? max_mel : i * step;
    freqs[i] = (mel >= min_log_mel) ? 1000.0 * exp(logstep * (mel - min_log_mel)) : f_sp * mel;
  }
  for (int i = 0; i < n_mels; ++i) {
    const double lo = freqs[i], ce = freqs[i + 1], hi = freqs[i + 2];
    for (int k = 0; k < N_BIN; ++k) {
      const double rise = (fftfreqs[k] - lo) / (ce - lo), fall = (hi - fftfreqs[k]) / (hi - ce);
      double w = rise < fall ? rise : fall;
      if (w < 0.0) w = 0.0;
      filt[i * N_BIN + k] = (float)(w * (2.0 / (hi - lo)));
    }
  }
  free(freqs);
}

int fw_oracle_logmel_full(const float* pcm, long n, int n_mels, float* out, long out_frames) {
  const long L = n + HOP;          /* waveform + 160 zeros */
  const long nf = L / HOP;         /* frames kept by stft[..., :-1] */
  if (out_frames != nf || n_mels <= 0 || n_mels > 128) return -1;
  float* filt = (float*)malloc(sizeof(float) * n_mels * N_BIN);
  mel_filters(n_mels, filt);
  float window[N_FFT];
  double cs[N_FFT], sn[N_FFT];
  for (int i = 0; i < N_FFT; ++i) {
    window[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / N_FFT));
    cs[i] = cos(2.0 * M_PI * i / N_FFT);
    sn[i] = sin(2.0 * M_PI * i / N_FFT);
  }
  const long period = 2 * (L - 1);
  float frame[N_FFT];
  float power[N_BIN];
  float gmax = -INFINITY;
  for (long f = 0; f < nf; ++f) {
    for (int i = 0; i < N_FFT; ++i) {
      long p = f * HOP + i - N_FFT / 2;
      long q = p % period;
      if (q < 0) q += period;
      if (q >= L) q = period - q;
      const float x = (q < n) ? pcm[q] : 0.0f;
      frame[i] = x * window[i];                       /* float32 product, like numpy */
    }
    for (int k = 0; k < N_BIN; ++k) {
      double re = 0.0, im = 0.0;
      for (int i = 0; i < N_FFT; ++i) {
        const int idx = (int)(((long)k * i) % N_FFT);
        re += frame[i] * cs[idx];
        im -= frame[i] * sn[idx];
      }
      const float fre = (float)re, fim = (float)im;   /* the reference stores complex64 */
      const float mag = sqrtf(fre * fre + fim * fim); /* np.abs(.) ** 2 */
      power[k] = mag * mag;
    }
    for (int m = 0; m < n_mels; ++m) {
      float acc = 0.0f;
      for (int k = 0; k < N_BIN; ++k) acc += filt[m * N_BIN + k] * power[k];
      float v = log10f(acc < 1e-10f ? 1e-10f : acc);
      out[(long)m * nf + f] = v;
      if (v > gmax) gmax = v;
    }
  }
  const float floorv = gmax - 8.0f;
  for (long i = 0; i < (long)n_mels * nf; ++i) {
    float v = out[i] < floorv ? floorv : out[i];
    out[i] = (v + 4.0f) / 4.0f;
  }
  free(filt);
  return 0;
}
