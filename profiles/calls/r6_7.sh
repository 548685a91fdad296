#!/bin/bash
# Round 6, GPU call 7: the whole -m gpu suite on the current tree, the VAD front after the LSTM change, the driver's command.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -s > "$OUT/pytest_gpu_call7.log" 2>&1
echo "pytest rc=$? $(( $(date +%s) - t0 ))s"; tail -5 "$OUT/pytest_gpu_call7.log" | cut -c1-300
for h in 1 8; do timeout 600 python profiles/vad_bench.py $h >> "$OUT/vad_bench_call7.json" 2>> "$OUT/vad_bench.err"; done
echo "vad:"; cat "$OUT/vad_bench_call7.json"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd_call7.json" 2> "$OUT/bench_driver_cmd_call7.err"
echo "driver-cmd rc=$?"; cut -c1-400 "$OUT/bench_driver_cmd_call7.json"; tail -3 "$OUT/bench_driver_cmd_call7.err"
