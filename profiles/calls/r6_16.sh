#!/bin/bash
# Round 6, GPU call 16: cfg 7 (128 x 128 with two k-steps per stage) — bit identity + table.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "dec_linear_big" > "$OUT/pytest_call16.log" 2>&1
echo "pytest rc=$?"; tail -4 "$OUT/pytest_call16.log"
DLB_VARIANTS=10,12,17 timeout 600 python profiles/dec_linear_bench.py 640 800 960 1280 1520 > "$OUT/dec_linear_bench_call16.txt" 2> "$OUT/dec_linear_bench_call16.err"
echo "dec_linear rc=$?"; cat "$OUT/dec_linear_bench_call16.txt"
