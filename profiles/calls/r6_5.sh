#!/bin/bash
# Round 6, GPU call 5: C5 + RCCL tests, where the VAD front of C5 spends its time, kernel trace of the distil pipeline with
# word timestamps (which kernels the align pass adds).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_c5.py tests/test_gpu_rccl_world1.py -q -s > "$OUT/pytest_c5_call5.log" 2>&1
echo "pytest rc=$?"; grep -E "C5|device VAD|passed|failed|Error|assert" "$OUT/pytest_c5_call5.log" | cut -c1-400 | tail -20
timeout 600 python profiles/vad_bench.py 8 > "$OUT/vad_bench_call5.json" 2> "$OUT/vad_bench.err"
echo "vad rc=$?"; cat "$OUT/vad_bench_call5.json"; tail -3 "$OUT/vad_bench.err"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
cd /tmp; export TMPDIR=/tmp
FWAMD_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_distil" -o kt -- \
    python "$R/bench.py" --model distil-large-v3 --word-timestamps --no-cpu-baseline --no-profile-pass --steps 8 --warmup 1 --pipeline-chunks 256 > "$OUT/prof_distil.log" 2>&1
f=$(find "$OUT/prof_distil" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_distil_pipeline.csv"
rm -rf "$OUT/prof_distil"
head -30 "$OUT/kernel_stats_distil_pipeline.csv" | cut -c1-200
