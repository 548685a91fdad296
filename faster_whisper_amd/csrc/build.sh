#!/bin/bash
# Builds libfwamd.so for gfx950 (MI355X). hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libfwamd.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p build
pids=()
for f in logmel gemm rowops attn_enc engine decoder dec_kernels; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ kernels.h -nt build/$f.o ] || [ engine.h -nt build/$f.o ] || [ dec_kernels.h -nt build/$f.o ] || [ ../../include/fwamd.h -nt build/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o $OUT
echo "built $(realpath $OUT)"
