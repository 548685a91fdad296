// Silero VAD v6: the model as both vad_host.cpp (g++, host forward) and vad.hip (device forward) see it.
// Plain C++ (no HIP types) so that the host file can be built without the HIP headers.
#pragma once
#include <vector>

constexpr int kWin = 576, kPad = 128, kPadded = kWin + 2 * kPad, kTaps = 256, kHop = 128;
constexpr int kBins = 129, kFrames = 4, kHidden = 128, kGates = 4 * kHidden;
constexpr int kC[5] = {kBins, 128, 64, 64, 128};      // channels through the four convolutions
constexpr int kT[5] = {kFrames, 4, 2, 1, 1};          // frames after each convolution
constexpr int kStride[4] = {1, 2, 2, 1};

// All matrices are stored TRANSPOSED (reduction index outermost, output index contiguous): every inner loop then
// runs over independent outputs, which the compiler vectorises without re-associating the sums (the per-output
// summation order stays the sequential one of the definition).
struct Vad {
  std::vector<float> basis_t;               // [256 taps][258]
  std::vector<float> cw_t[4], cb[4];        // [Cin][3][Cout], [Cout]
  std::vector<float> lw_t, lr_t, lb;        // [128][512], [128][512], [512] (Wb + Rb)
  std::vector<float> dw;                    // [128]
  float db = 0.f;
};

// device-side copy of the weights + scratch, owned by vad.hip (opaque here)
void fw_vad_dev_release(void* dev);

struct fw_vad {
  Vad impl;
  void* dev = nullptr;
};
