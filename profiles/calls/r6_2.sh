#!/bin/bash
# Round 6, GPU call 2: the staggered decoder-linear kernel (cfg 3 / 4): bit identity with the solo kernel, then the table.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "dec_linear_big" > "$OUT/pytest_call2.log" 2>&1
echo "pytest rc=$?"; tail -5 "$OUT/pytest_call2.log"
DLB_VARIANTS=0,10,12,13,14,21 timeout 600 python profiles/dec_linear_bench.py 800 960 1280 1520 1600 > "$OUT/dec_linear_bench_call2.txt" 2> "$OUT/dec_linear_bench_call2.err"
echo "dec_linear rc=$?"; cat "$OUT/dec_linear_bench_call2.txt"; tail -3 "$OUT/dec_linear_bench_call2.err"
