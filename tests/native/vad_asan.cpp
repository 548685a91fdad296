// Sanitizer harness for the host Silero VAD network (csrc/vad_host.cpp): exact-size heap buffers for weights, windows, state
// and probabilities; n = 0, 1, 7, 333 windows; 1 and 3 threads must return the same bits.  Built and run by
// tests/test_flac_sanitizers.py under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build).
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fwamd.h"

namespace fw {
void set_error(const char* fmt, ...) { (void)fmt; }
}  // namespace fw
void fw_vad_dev_release(void*) {}   // (the device half lives in vad.hip; a host-only model holds none)

static uint64_t st = 0x2545F4914F6CDD1Dull;
static float rnd(float scale) {
  st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
  const uint32_t r = (uint32_t)((st * 2685821657736338717ull) >> 40);     // 24 bits
  return scale * ((float)r / 8388608.0f - 1.0f);
}
static float* filled(size_t n, float scale) {
  float* p = (float*)malloc(n * sizeof(float));
  for (size_t i = 0; i < n; ++i) p[i] = rnd(scale);
  return p;
}

int main() {
  fw_vad_weights w;
  memset(&w, 0, sizeof(w));
  const int co[4] = {128, 64, 64, 128}, ci[4] = {129, 128, 64, 64};
  float* own[16];
  int no = 0;
  w.stft_basis = own[no++] = filled(258 * 256, 0.05f);
  for (int i = 0; i < 4; ++i) {
    w.conv_w[i] = own[no++] = filled((size_t)co[i] * ci[i] * 3, 0.08f);
    w.conv_b[i] = own[no++] = filled(co[i], 0.05f);
  }
  w.lstm_w = own[no++] = filled(512 * 128, 0.08f);
  w.lstm_r = own[no++] = filled(512 * 128, 0.08f);
  w.lstm_b = own[no++] = filled(1024, 0.2f);
  w.dec_w = own[no++] = filled(128, 0.3f);
  w.dec_b = 0.1f;
  fw_vad* v = nullptr;
  if (fw_vad_create(&w, &v) != 0 || !v) { fprintf(stderr, "fw_vad_create failed\n"); return 1; }
  const int64_t sizes[4] = {0, 1, 7, 333};
  for (int64_t n : sizes) {
    float* win = filled((size_t)(n ? n : 1) * 576, 0.4f);
    float *p1 = (float*)malloc((size_t)(n ? n : 1) * 4), *p3 = (float*)malloc((size_t)(n ? n : 1) * 4);
    float *h1 = (float*)calloc(128, 4), *c1 = (float*)calloc(128, 4), *h3 = (float*)calloc(128, 4), *c3 = (float*)calloc(128, 4);
    if (fw_vad_forward(v, win, n, 1, h1, c1, p1) != 0 || fw_vad_forward(v, win, n, 3, h3, c3, p3) != 0) {
      fprintf(stderr, "fw_vad_forward failed at n = %lld\n", (long long)n);
      return 1;
    }
    if (memcmp(p1, p3, (size_t)n * 4) || memcmp(h1, h3, 512) || memcmp(c1, c3, 512)) {
      fprintf(stderr, "1 and 3 threads differ at n = %lld\n", (long long)n);
      return 1;
    }
    for (int64_t i = 0; i < n; ++i)
      if (!(p1[i] > 0.f && p1[i] < 1.f)) { fprintf(stderr, "probability %g\n", p1[i]); return 1; }
    free(win); free(p1); free(p3); free(h1); free(c1); free(h3); free(c3);
  }
  fw_vad_free(v);
  for (int i = 0; i < no; ++i) free(own[i]);
  printf("vad_asan: 0 / 1 / 7 / 333 windows, 1 and 3 threads: same bits; no sanitizer report\n");
  return 0;
}
