#!/bin/bash
# Round 6, GPU call 6: VAD tests (device framing), RCCL world-1, the VAD front's timing for 8 h, the distil C5 line again
# (VAD front without the host framing / per-window Python walk / chunk copies), C4 rank batch through the pipeline.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_vad.py tests/test_gpu_c5.py tests/test_gpu_rccl_world1.py -q -s > "$OUT/pytest_call6.log" 2>&1
echo "pytest rc=$?"; grep -E "C5|device VAD|passed|failed|Error|assert" "$OUT/pytest_call6.log" | cut -c1-400 | tail -20
timeout 600 python profiles/vad_bench.py 8 > "$OUT/vad_bench_call6.json" 2> "$OUT/vad_bench.err"
echo "vad rc=$?"; cat "$OUT/vad_bench_call6.json"; tail -3 "$OUT/vad_bench.err"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
timeout 900 python bench.py --gpus 1 --model distil-large-v3 --steps 20 --warmup 5 --word-timestamps --vad --no-cpu-baseline > "$OUT/bench_distil_large_v3_c5.json" 2> "$OUT/bench_distil_c5.err"
echo "distil rc=$?"; python -c "
import json; j=json.loads([l for l in open('$OUT/bench_distil_large_v3_c5.json') if l.startswith('{')][-1]); print(j['value'], json.dumps(j.get('steady')), json.dumps(j.get('pipeline')))"; tail -3 "$OUT/bench_distil_c5.err"
