#!/bin/bash
# Builds libfwamd.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libfwamd.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p build
pids=()
for f in logmel gemm rowops attn_enc engine decoder dec_kernels vad; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ kernels.h -nt build/$f.o ] || [ engine.h -nt build/$f.o ] || [ dec_kernels.h -nt build/$f.o ] || [ ../../include/fwamd.h -nt build/$f.o ] || [ ../../include/fwamd_test.h -nt build/$f.o ] || [ vad_model.h -nt build/$f.o ] || [ build.sh -nt build/$f.o ]; then
    # attn_enc: MFMA accumulators in VGPRs (-amdgpu-mfma-vgpr-form): the softmax reads every score and rescales the output
    # accumulators, which with AGPR accumulators costs 191 v_accvgpr moves per 64-key tile of a vector-bound kernel (548 ->
    # 405 vector instructions per tile, 164 -> 138 registers)
    EXTRA=""
    if [ $f = attn_enc ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"; fi
    hipcc $FLAGS $EXTRA -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
# host-only C++ (Silero VAD network): g++, with AVX2 / AVX-512 clones selected at run time (target_clones)
if [ ! -f build/vad_host.o ] || [ vad_host.cpp -nt build/vad_host.o ] || [ ../../include/fwamd.h -nt build/vad_host.o ] || [ vad_model.h -nt build/vad_host.o ]; then
  g++ -O3 -std=c++17 -fPIC -Wall -c vad_host.cpp -o build/vad_host.o &
  pids+=($!)
fi
# host-only C++: the native FLAC decoder of the audio front (row f-4)
if [ ! -f build/flac_host.o ] || [ flac_host.cpp -nt build/flac_host.o ] || [ ../../include/fwamd.h -nt build/flac_host.o ]; then
  g++ -O2 -std=c++17 -fPIC -Wall -c flac_host.cpp -o build/flac_host.o &
  pids+=($!)
fi
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -pthread build/*.o -o $OUT
echo "built $(realpath $OUT)"
