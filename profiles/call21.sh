#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r02m; mkdir -p "$OUT"; cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
timeout 200 python -m pytest tests/test_gpu_logmel.py tests/test_gpu_decode_group.py -q -m gpu > "$OUT/pytest.log" 2>&1; echo "== pytest rc=$?"; tail -3 "$OUT/pytest.log"
for w in 12 16 6; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-profile-pass --steps 48 --workers $w > "$OUT/bench_w$w.json" 2> "$OUT/bench_w$w.err"
  echo "== workers=$w rc=$? $(python -c "
import json
j=json.load(open('$OUT/bench_w$w.json')); print(j['value'], j['ms_per_step'], j['config']['decode_group'])")"
done
