#!/bin/bash
# Round 6, GPU call 9: dec_gemm_big with the LDS reads issued before the DMA (and x-first / W-outer MFMA order): bit identity,
# the isolated table, the timeline; the second LSTM form of the device VAD: tests + timing.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vad.py tests/test_gpu_c5.py -q -k "dec_linear or vad or c5" > "$OUT/pytest_call9.log" 2>&1
echo "pytest rc=$?"; tail -4 "$OUT/pytest_call9.log"
DLB_VARIANTS=0,10,11,12,13,21 timeout 600 python profiles/dec_linear_bench.py 800 960 1280 1520 > "$OUT/dec_linear_bench_call9.txt" 2> "$OUT/dec_linear_bench_call9.err"
echo "dec_linear rc=$?"; cat "$OUT/dec_linear_bench_call9.txt"
cd "$R/profiles/ubench"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../faster_whisper_amd/csrc -I../../include dec_big_timeline.hip -o /tmp/dbt > "$OUT/dbt_build.log" 2>&1 && timeout 120 /tmp/dbt > "$OUT/dec_big_timeline_call9.txt" 2>&1
cat "$OUT/dec_big_timeline_call9.txt"
cd "$R"
for f in 1 0; do FWAMD_VAD_LSTM=$f timeout 600 python profiles/vad_bench.py 8 >> "$OUT/vad_bench_call9.json" 2>> "$OUT/vad_bench.err"; done
echo "vad (first line: first LSTM form, second: second form):"; cat "$OUT/vad_bench_call9.json"
