// Native FLAC decoder, host C++ (the audio front of the pipelines).
//
// The reference decodes every container through PyAV / FFmpeg (faster_whisper/audio.py:19-76); its own test asset
// tests/data/jfk.flac is FLAC.  This is the native equivalent for that format behind the C ABI (fw_flac_info /
// fw_flac_decode): STREAMINFO, frame headers (CRC-8), CONSTANT / VERBATIM / FIXED / LPC subframes, Rice and Rice2
// residual partitions with escape codes, wasted bits, left-side / right-side / mid-side stereo, frame CRC-16, and the
// MD5 of the decoded PCM that the encoder stored in STREAMINFO — so a decode either reproduces the encoder's input
// bit for bit or says so.  Restated from the published FLAC format (xiph.org/flac/format.html, RFC 9639); no decoder
// source was consulted.  Plain host C++ built with g++ (no HIP): the GPU plays no part in entropy decoding.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/fwamd.h"

namespace fw {
void set_error(const char* fmt, ...);
}
#define FL_FAIL(...)             \
  do {                           \
    fw::set_error(__VA_ARGS__);  \
    return FW_EINVAL;            \
  } while (0)

namespace {

// ---- MD5 (RFC 1321), for the STREAMINFO signature ----
struct Md5 {
  uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
  uint64_t n = 0;
  uint8_t buf[64];
  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8,
        0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340,
        0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87,
        0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c,
        0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039,
        0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92,
        0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb,
        0xeb86d391};
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t m[16];
    for (int i = 0; i < 16; ++i)
      m[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
    uint32_t A = a, B = b, C = c, D = d;
    for (int i = 0; i < 64; ++i) {
      uint32_t f;
      int g;
      if (i < 16) { f = (B & C) | (~B & D); g = i; }
      else if (i < 32) { f = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
      else if (i < 48) { f = B ^ C ^ D; g = (3 * i + 5) & 15; }
      else { f = C ^ (B | ~D); g = (7 * i) & 15; }
      const uint32_t t = D;
      D = C; C = B;
      B = B + rol(A + f + K[i] + m[g], S[i]);
      A = t;
    }
    a += A; b += B; c += C; d += D;
  }
  void update(const uint8_t* p, size_t len) {
    size_t fill = (size_t)(n & 63);
    n += len;
    if (fill) {
      const size_t take = len < 64 - fill ? len : 64 - fill;
      memcpy(buf + fill, p, take);
      p += take; len -= take; fill += take;
      if (fill < 64) return;
      block(buf);
    }
    for (; len >= 64; p += 64, len -= 64) block(p);
    if (len) memcpy(buf, p, len);
  }
  void finish(uint8_t out[16]) {
    const uint64_t bits = n * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while ((n & 63) != 56) update(&zero, 1);
    uint8_t lenb[8];
    for (int i = 0; i < 8; ++i) lenb[i] = (uint8_t)(bits >> (8 * i));
    update(lenb, 8);
    const uint32_t r[4] = {a, b, c, d};
    for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(r[i >> 2] >> (8 * (i & 3)));
  }
};

// ---- CRC-8 (poly 0x07) over the frame header, CRC-16 (poly 0x8005) over the whole frame; both MSB first, init 0 ----
uint8_t crc8(const uint8_t* p, size_t n) {
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c ^= p[i];
    for (int k = 0; k < 8; ++k) c = (uint8_t)((c & 0x80) ? (c << 1) ^ 0x07 : c << 1);
  }
  return c;
}
struct Crc16Table {
  uint16_t t[256];
  Crc16Table() {
    for (int i = 0; i < 256; ++i) {
      uint16_t c = (uint16_t)(i << 8);
      for (int k = 0; k < 8; ++k) c = (uint16_t)((c & 0x8000) ? (c << 1) ^ 0x8005 : c << 1);
      t[i] = c;
    }
  }
};
uint16_t crc16(const uint8_t* p, size_t n) {
  static const Crc16Table tab;   // magic static: initialised once, thread-safe (decode workers may race to their first frame)
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ tab.t[(c >> 8) ^ p[i]]);
  return c;
}

// ---- MSB-first bit reader ----
struct Bits {
  const uint8_t* p;
  size_t n, pos = 0;      // pos in bits
  bool bad = false;
  Bits(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  uint32_t bit() {
    if (pos >= n * 8) { bad = true; return 0; }
    const uint32_t b = (p[pos >> 3] >> (7 - (pos & 7))) & 1u;
    ++pos;
    return b;
  }
  uint64_t u(int k) {      // k <= 57
    uint64_t v = 0;
    while (k > 0) {
      if (pos >= n * 8) { bad = true; return 0; }
      const int avail = 8 - (int)(pos & 7);
      const int take = k < avail ? k : avail;
      const uint32_t byte = p[pos >> 3];
      v = (v << take) | ((byte >> (avail - take)) & ((1u << take) - 1u));
      pos += take; k -= take;
    }
    return v;
  }
  int64_t s(int k) {       // two's complement, k bits
    if (k == 0) return 0;
    const uint64_t v = u(k);
    return (int64_t)(v << (64 - k)) >> (64 - k);
  }
  uint32_t unary() {       // number of 0 bits before the next 1
    uint32_t q = 0;
    while (!bad) {
      if ((pos & 7) == 0 && pos + 8 <= n * 8 && p[pos >> 3] == 0) { q += 8; pos += 8; continue; }
      if (bit()) break;
      ++q;
    }
    return q;
  }
  void align() { pos = (pos + 7) & ~(size_t)7; }
};

struct Info {
  int rate = 0, channels = 0, bps = 0, min_block = 0, max_block = 0;
  int64_t total = 0;
  uint8_t md5[16] = {0};
  size_t first_frame = 0;   // byte offset of the first audio frame
};

int parse_header(const uint8_t* d, size_t n, Info& inf) {
  size_t pos = 0;
  if (n >= 10 && d[0] == 'I' && d[1] == 'D' && d[2] == '3') {   // an ID3v2 tag in front of the stream: skip it
    const size_t sz = ((size_t)(d[6] & 0x7f) << 21) | ((size_t)(d[7] & 0x7f) << 14) | ((size_t)(d[8] & 0x7f) << 7) | (d[9] & 0x7f);
    pos = 10 + sz;
  }
  if (pos + 4 > n || memcmp(d + pos, "fLaC", 4) != 0) FL_FAIL("not a FLAC stream (no fLaC marker)");
  pos += 4;
  bool have_info = false;
  for (;;) {
    if (pos + 4 > n) FL_FAIL("FLAC: truncated metadata");
    const bool last = (d[pos] & 0x80) != 0;
    const int type = d[pos] & 0x7f;
    const size_t len = ((size_t)d[pos + 1] << 16) | ((size_t)d[pos + 2] << 8) | d[pos + 3];
    pos += 4;
    if (pos + len > n) FL_FAIL("FLAC: truncated metadata block");
    if (type == 0) {
      if (len < 34) FL_FAIL("FLAC: short STREAMINFO");
      const uint8_t* s = d + pos;
      inf.min_block = (s[0] << 8) | s[1];
      inf.max_block = (s[2] << 8) | s[3];
      inf.rate = (s[10] << 12) | (s[11] << 4) | (s[12] >> 4);
      inf.channels = ((s[12] >> 1) & 7) + 1;
      inf.bps = (((s[12] & 1) << 4) | (s[13] >> 4)) + 1;
      inf.total = ((int64_t)(s[13] & 0x0f) << 32) | ((int64_t)s[14] << 24) | ((int64_t)s[15] << 16) | ((int64_t)s[16] << 8) | s[17];
      memcpy(inf.md5, s + 18, 16);
      have_info = true;
    }
    pos += len;
    if (last) break;
  }
  if (!have_info) FL_FAIL("FLAC: no STREAMINFO block");
  if (inf.rate <= 0 || inf.bps < 4 || inf.bps > 32) FL_FAIL("FLAC: invalid STREAMINFO (rate %d, %d bits)", inf.rate, inf.bps);
  inf.first_frame = pos;
  return FW_OK;
}

// residual of one subframe into res[order .. block)
int read_residual(Bits& br, int block, int order, int32_t* res) {
  const int method = (int)br.u(2);
  if (method > 1) FL_FAIL("FLAC: reserved residual coding method");
  const int pbits = method == 0 ? 4 : 5, esc = method == 0 ? 15 : 31;
  const int porder = (int)br.u(4);
  const int parts = 1 << porder;
  if ((block >> porder) << porder != block && porder > 0) FL_FAIL("FLAC: block size %d not divisible by 2^%d partitions", block, porder);
  int i = order;
  for (int p = 0; p < parts; ++p) {
    int cnt = (block >> porder) - (p == 0 ? order : 0);
    if (cnt < 0) FL_FAIL("FLAC: partition smaller than the predictor order");
    const int k = (int)br.u(pbits);
    if (k == esc) {
      const int nb = (int)br.u(5);
      for (int j = 0; j < cnt; ++j) res[i++] = (int32_t)br.s(nb);
    } else {
      for (int j = 0; j < cnt; ++j) {
        const uint32_t q = br.unary();
        const uint32_t u = (q << k) | (uint32_t)br.u(k);
        res[i++] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
      }
    }
    if (br.bad) FL_FAIL("FLAC: residual runs past the end of the data");
  }
  return FW_OK;
}

int read_subframe(Bits& br, int block, int bps, int64_t* out) {
  if (br.bit()) FL_FAIL("FLAC: subframe padding bit set");
  const int type = (int)br.u(6);
  int wasted = 0;
  if (br.bit()) wasted = (int)br.unary() + 1;
  bps -= wasted;
  if (bps < 1) FL_FAIL("FLAC: wasted bits exceed the sample size");
  if (type == 0) {                                   // CONSTANT
    const int64_t v = br.s(bps);
    for (int i = 0; i < block; ++i) out[i] = v;
  } else if (type == 1) {                            // VERBATIM
    for (int i = 0; i < block; ++i) out[i] = br.s(bps);
  } else if (type >= 8 && type <= 12) {              // FIXED, order type - 8
    const int order = type - 8;
    if (order > block) FL_FAIL("FLAC: fixed order exceeds the block");
    std::vector<int32_t> res(block);
    for (int i = 0; i < order; ++i) out[i] = br.s(bps);
    int rc = read_residual(br, block, order, res.data());
    if (rc) return rc;
    // (the predictors run in WRAPPING 64-bit arithmetic: a valid stream never leaves its sample range, a corrupted one can
    //  make the recursion diverge — signed overflow would be undefined behaviour; the frame's CRC-16 rejects such a frame
    //  afterwards.  Found by tests/native/flac_fuzz.cpp under UBSan.)
    auto U = [](int64_t v) { return (uint64_t)v; };
    for (int i = order; i < block; ++i) {
      uint64_t p = 0;
      switch (order) {
        case 1: p = U(out[i - 1]); break;
        case 2: p = 2 * U(out[i - 1]) - U(out[i - 2]); break;
        case 3: p = 3 * U(out[i - 1]) - 3 * U(out[i - 2]) + U(out[i - 3]); break;
        case 4: p = 4 * U(out[i - 1]) - 6 * U(out[i - 2]) + 4 * U(out[i - 3]) - U(out[i - 4]); break;
        default: break;
      }
      out[i] = (int64_t)(p + U((int64_t)res[i]));
    }
  } else if (type >= 32) {                           // LPC, order type - 31
    const int order = type - 31;
    if (order > block) FL_FAIL("FLAC: LPC order exceeds the block");
    std::vector<int32_t> res(block);
    for (int i = 0; i < order; ++i) out[i] = br.s(bps);
    const int prec = (int)br.u(4) + 1;
    if (prec == 16) FL_FAIL("FLAC: invalid LPC precision");
    const int shift = (int)br.s(5);
    if (shift < 0) FL_FAIL("FLAC: negative LPC shift");
    int32_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = (int32_t)br.s(prec);
    int rc = read_residual(br, block, order, res.data());
    if (rc) return rc;
    for (int i = order; i < block; ++i) {
      uint64_t acc = 0;                              // wrapping, as above
      for (int j = 0; j < order; ++j) acc += (uint64_t)(int64_t)coef[j] * (uint64_t)out[i - 1 - j];
      out[i] = (int64_t)((uint64_t)((int64_t)acc >> shift) + (uint64_t)(int64_t)res[i]);
    }
  } else {
    FL_FAIL("FLAC: reserved subframe type %d", type);
  }
  if (br.bad) FL_FAIL("FLAC: subframe runs past the end of the data");
  if (wasted)
    for (int i = 0; i < block; ++i) out[i] = (int64_t)((uint64_t)out[i] << wasted);
  return FW_OK;
}

}  // namespace

extern "C" {

int32_t fw_flac_info(const uint8_t* data, int64_t n_bytes, int32_t* sample_rate, int32_t* channels, int32_t* bits_per_sample,
                     int64_t* total_samples) {
  if (!data || n_bytes < 8) FL_FAIL("null / short FLAC buffer");
  Info inf;
  int rc = parse_header(data, (size_t)n_bytes, inf);
  if (rc) return rc;
  if (sample_rate) *sample_rate = inf.rate;
  if (channels) *channels = inf.channels;
  if (bits_per_sample) *bits_per_sample = inf.bps;
  if (total_samples) *total_samples = inf.total;
  return FW_OK;
}

int32_t fw_flac_decode(const uint8_t* data, int64_t n_bytes, int32_t* out, int64_t capacity_samples, int64_t* n_decoded,
                       int32_t* md5_status) {
  if (!data || !out || !n_decoded) FL_FAIL("null argument");
  const size_t n = (size_t)n_bytes;
  Info inf;
  int rc = parse_header(data, n, inf);
  if (rc) return rc;
  const int C = inf.channels, bytes_ps = (inf.bps + 7) / 8;
  Md5 md5;
  int64_t done = 0;
  size_t pos = inf.first_frame;
  std::vector<int64_t> ch[8];
  std::vector<uint8_t> pcm;
  while (pos + 6 <= n) {
    if (inf.total > 0 && done >= inf.total) break;
    if (!(data[pos] == 0xff && (data[pos + 1] & 0xfe) == 0xf8)) {
      // between frames only padding / trailing tags are legal; a stream cut mid-frame ends here too
      if (done == 0) FL_FAIL("FLAC: no frame sync at the first frame");
      break;
    }
    Bits br(data + pos, n - pos);
    br.u(15);
    br.bit();                                         // blocking strategy (only tells how the frame is numbered)
    const int bs_code = (int)br.u(4), sr_code = (int)br.u(4), ch_code = (int)br.u(4), ss_code = (int)br.u(3);
    if (br.bit()) FL_FAIL("FLAC: reserved frame header bit set");
    {                                                 // UTF-8-style coded frame / sample number
      const uint32_t b0 = (uint32_t)br.u(8);
      int extra = 0;
      if (b0 >= 0xfe) extra = 6; else if (b0 >= 0xfc) extra = 5; else if (b0 >= 0xf8) extra = 4;
      else if (b0 >= 0xf0) extra = 3; else if (b0 >= 0xe0) extra = 2; else if (b0 >= 0xc0) extra = 1;
      else if (b0 >= 0x80) FL_FAIL("FLAC: bad coded frame number");
      for (int i = 0; i < extra; ++i) br.u(8);
    }
    int block;
    if (bs_code == 0) FL_FAIL("FLAC: reserved block size code");
    else if (bs_code == 1) block = 192;
    else if (bs_code <= 5) block = 576 << (bs_code - 2);
    else if (bs_code == 6) block = (int)br.u(8) + 1;
    else if (bs_code == 7) block = (int)br.u(16) + 1;
    else block = 256 << (bs_code - 8);
    if (sr_code == 12) br.u(8);
    else if (sr_code == 13 || sr_code == 14) br.u(16);
    else if (sr_code == 15) FL_FAIL("FLAC: invalid sample rate code");
    int bps = inf.bps;
    switch (ss_code) {
      case 0: break;
      case 1: bps = 8; break;
      case 2: bps = 12; break;
      case 4: bps = 16; break;
      case 5: bps = 20; break;
      case 6: bps = 24; break;
      case 7: bps = 32; break;
      default: FL_FAIL("FLAC: reserved sample size code");
    }
    if (bps != inf.bps) FL_FAIL("FLAC: sample size changes inside the stream (%d -> %d bits)", inf.bps, bps);
    int nch;
    if (ch_code <= 7) nch = ch_code + 1;
    else if (ch_code <= 10) nch = 2;
    else FL_FAIL("FLAC: reserved channel assignment");
    if (nch != C) FL_FAIL("FLAC: channel count changes inside the stream");
    if (br.bad || (br.pos & 7)) FL_FAIL("FLAC: truncated frame header");
    const size_t hdr_bytes = br.pos >> 3;
    if (pos + hdr_bytes + 1 > n) break;
    if (crc8(data + pos, hdr_bytes) != data[pos + hdr_bytes]) FL_FAIL("FLAC: frame header CRC-8 mismatch at byte %zu", pos);
    br.u(8);
    for (int c = 0; c < nch; ++c) {
      ch[c].resize(block);
      // the side channel of a decorrelated pair carries one more bit
      const int side = (ch_code == 8 && c == 1) || (ch_code == 9 && c == 0) || (ch_code == 10 && c == 1);
      if ((rc = read_subframe(br, block, bps + side, ch[c].data()))) {
        if (br.bad && done > 0) break;                     // the data ends inside this frame: keep the whole frames
        return rc;
      }
    }
    br.align();
    const size_t body = br.pos >> 3;
    if (br.bad || pos + body + 2 > n) {
      if (done > 0) break;                            // the stream ends inside this frame: keep what is whole
      FL_FAIL("FLAC: truncated first frame");
    }
    const uint16_t want = (uint16_t)((data[pos + body] << 8) | data[pos + body + 1]);
    if (crc16(data + pos, body) != want) FL_FAIL("FLAC: frame CRC-16 mismatch at byte %zu", pos);
    pos += body + 2;
    // undo the stereo decorrelation
    if (ch_code == 8) {
      for (int i = 0; i < block; ++i) ch[1][i] = ch[0][i] - ch[1][i];
    } else if (ch_code == 9) {
      for (int i = 0; i < block; ++i) ch[0][i] = ch[0][i] + ch[1][i];
    } else if (ch_code == 10) {
      for (int i = 0; i < block; ++i) {
        const int64_t side = ch[1][i];
        const int64_t mid = (int64_t)((uint64_t)ch[0][i] << 1) | (side & 1);
        ch[0][i] = (mid + side) >> 1;
        ch[1][i] = (mid - side) >> 1;
      }
    }
    int take = block;
    if (inf.total > 0 && done + take > inf.total) take = (int)(inf.total - done);
    if (done + take > capacity_samples) {   // its own status: the caller grows the buffer and retries (no message matching)
      fw::set_error("FLAC: output buffer of %lld samples per channel is too small", (long long)capacity_samples);
      return FW_ENOSPC;
    }
    pcm.resize((size_t)take * C * bytes_ps);
    size_t w = 0;
    for (int i = 0; i < take; ++i)
      for (int c = 0; c < C; ++c) {
        const int64_t v = ch[c][i];
        out[(done + i) * C + c] = (int32_t)v;
        for (int b = 0; b < bytes_ps; ++b) pcm[w++] = (uint8_t)((uint64_t)v >> (8 * b));
      }
    md5.update(pcm.data(), pcm.size());
    done += take;
  }
  *n_decoded = done;
  if (md5_status) {
    // 1: the decoded PCM has the MD5 the encoder stored; 0: it does not; -1: nothing to compare with (no signature in
    // the stream, or fewer samples than STREAMINFO announces — a truncated file)
    bool have = false;
    for (int i = 0; i < 16; ++i) have = have || inf.md5[i] != 0;
    if (!have || (inf.total > 0 && done != inf.total)) {
      *md5_status = -1;
    } else {
      uint8_t got[16];
      md5.finish(got);
      *md5_status = memcmp(got, inf.md5, 16) == 0 ? 1 : 0;
    }
  }
  return FW_OK;
}

}  // extern "C"
