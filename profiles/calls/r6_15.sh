#!/bin/bash
# Round 6, GPU call 15: the 128 x 64 form with two k-steps per stage (cfg 5 / 6): bit identity + the isolated table.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "dec_linear_big" > "$OUT/pytest_call15.log" 2>&1
echo "pytest rc=$?"; tail -4 "$OUT/pytest_call15.log"
DLB_VARIANTS=0,11,15,16 timeout 600 python profiles/dec_linear_bench.py 832 960 1120 1280 1520 > "$OUT/dec_linear_bench_call15.txt" 2> "$OUT/dec_linear_bench_call15.err"
echo "dec_linear rc=$?"; cat "$OUT/dec_linear_bench_call15.txt"
