// Sanitizer harness for the native FLAC decoder (csrc/flac_host.cpp: the one parser of this repository that reads UNTRUSTED
// bytes — audio files handed to decode_audio).  Built by tests/test_flac_sanitizers.py with
//     g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all flac_fuzz.cpp ../../faster_whisper_amd/csrc/flac_host.cpp
// and run on the CPU (the task's rule: sanitizers on the CPU build only).  Every input lives in an exact-size heap block
// and every output buffer is exactly capacity x channels words, so a read or write one byte past either trips ASan.
//
//   flac_fuzz <file.flac> <iterations> <seed>
//     1. the file itself decodes, its MD5 signature matches (a whole file) or is absent (a cut one);
//     2. `iterations` mutants — byte flips, truncations, spliced headers, forged STREAMINFO fields (block sizes, channels,
//        bits per sample, total samples), forged frame headers — each go through fw_flac_info and fw_flac_decode:
//        any status is acceptable, a crash / sanitizer report / write past `capacity` is not.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <vector>

#include "../../include/fwamd.h"

namespace fw {
static char last_error[512];
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error, sizeof(last_error), fmt, ap);
  va_end(ap);
}
}  // namespace fw

static uint64_t rng_state;
static uint32_t rnd() {   // xorshift64*
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (uint32_t)((rng_state * 2685821657736338717ull) >> 32);
}

// decode `n` bytes held in an exact-size heap block; returns the status, checks the contract on success
static int decode_once(const uint8_t* src, size_t n, int64_t cap_limit, long* decoded_total) {
  uint8_t* data = (uint8_t*)malloc(n ? n : 1);
  memcpy(data, src, n);
  int32_t rate = 0, ch = 0, bps = 0, md5 = -2;
  int64_t total = -1, got = -1;
  int rc = fw_flac_info(data, (int64_t)n, &rate, &ch, &bps, &total);
  if (rc == 0) {
    if (ch < 1 || ch > 8 || bps < 4 || bps > 32 || rate < 0) { fprintf(stderr, "fw_flac_info accepted ch %d bps %d rate %d\n", ch, bps, rate); abort(); }
    // what a caller does: room for the announced length (bounded), or a guess when the length is unknown
    int64_t cap = total > 0 ? total : 65536;
    if (cap > cap_limit) cap = cap_limit;
    int32_t* out = (int32_t*)malloc((size_t)cap * ch * sizeof(int32_t));
    rc = fw_flac_decode(data, (int64_t)n, out, cap, &got, &md5);
    if (rc == 0) {
      if (got < 0 || got > cap) { fprintf(stderr, "decoded %lld samples into a buffer of %lld\n", (long long)got, (long long)cap); abort(); }
      if (md5 < -1 || md5 > 1) { fprintf(stderr, "md5 status %d\n", md5); abort(); }
      long long acc = 0;                       // touch every sample the decoder says it wrote (uninitialised tails would be ours)
      for (int64_t i = 0; i < got * ch; ++i) acc += out[i];
      if (decoded_total) *decoded_total += (long)got + (acc == 0x7fffffffffffffffLL);
    } else if (rc != FW_EINVAL && rc != FW_ENOSPC && rc != FW_ENOMEM) {
      fprintf(stderr, "unexpected status %d\n", rc); abort();
    }
    free(out);
  } else if (rc != FW_EINVAL) {
    fprintf(stderr, "unexpected fw_flac_info status %d\n", rc); abort();
  }
  free(data);
  return rc;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: flac_fuzz file iterations seed\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  std::vector<uint8_t> base;
  uint8_t buf[65536];
  size_t r;
  while ((r = fread(buf, 1, sizeof(buf), f)) > 0) base.insert(base.end(), buf, buf + r);
  fclose(f);
  const int iters = atoi(argv[2]);
  rng_state = 0x9E3779B97F4A7C15ull ^ (uint64_t)strtoull(argv[3], nullptr, 10);
  long decoded = 0;
  int rc = decode_once(base.data(), base.size(), 1 << 22, &decoded);
  if (rc != 0 || decoded <= 0) { fprintf(stderr, "the unmodified file does not decode: %d (%s)\n", rc, fw::last_error); return 1; }
  const long base_samples = decoded;
  int ok = 0, refused = 0;
  for (int it = 0; it < iters; ++it) {
    std::vector<uint8_t> m = base;
    const uint32_t kind = rnd() % 8;
    if (kind == 0) {                                   // truncate anywhere
      m.resize(rnd() % (m.size() + 1));
    } else if (kind == 1) {                            // forged STREAMINFO: min / max block size, frame sizes, rate | channels | bps | total
      for (int k = 0; k < 1 + (int)(rnd() % 6); ++k) m[8 + rnd() % 34] = (uint8_t)rnd();
    } else if (kind == 2) {                            // metadata block headers (type / last flag / 24-bit length) forged
      m[4 + rnd() % 4] = (uint8_t)rnd();
    } else if (kind == 3) {                            // burst of noise somewhere in the frames
      const size_t at = rnd() % m.size(), len = 1 + rnd() % 64;
      for (size_t k = at; k < at + len && k < m.size(); ++k) m[k] = (uint8_t)rnd();
    } else if (kind == 4) {                            // splice: the tail of the stream moved onto an earlier offset
      const size_t a = rnd() % m.size(), b = rnd() % m.size();
      const size_t len = (m.size() - (a > b ? a : b)) / 2;
      memmove(&m[a], &m[b], len);
    } else if (kind == 5) {                            // forge bytes right behind a frame sync code (block size / rate / channel / bps codes)
      for (size_t k = 42; k + 6 < m.size(); ++k)
        if (m[k] == 0xFF && (m[k + 1] & 0xFE) == 0xF8 && rnd() % 3 == 0) { m[k + 2 + rnd() % 3] = (uint8_t)rnd(); if (rnd() % 2) break; }
    } else if (kind == 6) {                            // the stream repeated (more samples than STREAMINFO announces)
      const size_t hdr = 42 < m.size() ? 42 : 0;
      m.insert(m.end(), base.begin() + hdr, base.end());
    } else {                                           // single bit flips
      for (int k = 0; k < 1 + (int)(rnd() % 4); ++k) m[rnd() % m.size()] ^= (uint8_t)(1u << (rnd() % 8));
    }
    // a small output buffer now and then: FW_ENOSPC must be reported without writing past it
    const int64_t cap_limit = (rnd() % 4 == 0) ? (int64_t)(1 + rnd() % (base_samples + 1)) : (1 << 22);
    long d = 0;
    rc = decode_once(m.data(), m.size(), cap_limit, &d);
    if (rc == 0) ++ok; else ++refused;
  }
  printf("flac_fuzz: %ld samples in the unmodified file; %d mutants: %d decoded (whole frames that survived), %d refused; no sanitizer report\n",
         base_samples, iters, ok, refused);
  return 0;
}
