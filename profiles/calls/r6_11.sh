#!/bin/bash
# Round 6, GPU call 11: crossovers of the LDS-staged form after the LayerNorm-statistics split (rows 448 ... 1 120).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
DLB_VARIANTS=0,10,11,12 timeout 900 python profiles/dec_linear_bench.py 448 512 576 640 704 768 832 896 1024 1120 > "$OUT/dec_linear_bench_call11.txt" 2> "$OUT/dec_linear_bench_call11.err"
echo "dec_linear rc=$?"; cat "$OUT/dec_linear_bench_call11.txt"
