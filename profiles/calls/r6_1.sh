#!/bin/bash
# Round 6, GPU call 1: complementary CU masks (verdict item 1a) + the isolated decoder-linear table before any change.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 300 python profiles/dec_linear_bench.py 1280 1520 > "$OUT/dec_linear_bench_call1.txt" 2> "$OUT/dec_linear_bench_call1.err"
echo "dec_linear rc=$?"; tail -30 "$OUT/dec_linear_bench_call1.txt"
timeout 1500 python profiles/ab_r06_overlap.py --steps 96 > "$OUT/ab_overlap.jsonl" 2> "$OUT/ab_overlap.err"
echo "overlap rc=$?"; cut -c1-400 "$OUT/ab_overlap.jsonl"; tail -5 "$OUT/ab_overlap.err"
