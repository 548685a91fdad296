#!/bin/bash
# Round 6, GPU call 10: LayerNorm statistics split between the two waves of a row block in dec_gemm_big_kernel: bit identity,
# the isolated table.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "dec_linear" > "$OUT/pytest_call10.log" 2>&1
echo "pytest rc=$?"; tail -4 "$OUT/pytest_call10.log"
DLB_VARIANTS=0,10,11,12,21 timeout 600 python profiles/dec_linear_bench.py 800 960 1280 1520 > "$OUT/dec_linear_bench_call10.txt" 2> "$OUT/dec_linear_bench_call10.err"
echo "dec_linear rc=$?"; cat "$OUT/dec_linear_bench_call10.txt"
