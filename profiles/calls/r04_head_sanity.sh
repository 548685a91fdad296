#!/bin/bash
# HEAD's library (the round's kernels + the host-only FLAC object linked in): a last sanity pass on hardware
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_head
mkdir -p "$OUT"
cd "$R"
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_decode_group.py tests/test_gpu_logmel.py tests/test_gpu_pipeline.py -q -m gpu > "$OUT/pytest_part.log" 2>&1; echo "== pytest rc=$?"; tail -2 "$OUT/pytest_part.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile-pass > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "== bench rc=$? $(cut -c1-240 $OUT/bench.json)"
