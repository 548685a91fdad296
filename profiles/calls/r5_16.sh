#!/bin/bash
# Round 5, GPU call 16 (the last GPU minutes of the round): the driver's command on the FINAL build (layered cross-K/V launch,
# staged V^T epilogue) — whatever box it lands on.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
timeout 190 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd_final_build.json" 2> "$OUT/bench_driver_cmd_final_build.err"
echo "rc=$? $(cut -c1-220 "$OUT/bench_driver_cmd_final_build.json")"
