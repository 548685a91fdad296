#!/bin/bash
# Round 6, GPU call 8: cycle stamps inside dec_gemm_big_kernel's K loop (profiles/ubench/dec_big_timeline.hip).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R/profiles/ubench"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../faster_whisper_amd/csrc -I../../include dec_big_timeline.hip -o /tmp/dbt > "$OUT/dbt_build.log" 2>&1
echo "build rc=$?"
timeout 120 /tmp/dbt > "$OUT/dec_big_timeline.txt" 2>&1
echo "run rc=$?"; cat "$OUT/dec_big_timeline.txt"
