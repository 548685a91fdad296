#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r02n; mkdir -p "$OUT"; cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
for w in 20 24 32; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-profile-pass --steps 96 --workers $w > "$OUT/bench_w$w.json" 2> "$OUT/bench_w$w.err"
  echo "== workers=$w rc=$? $(python -c "
import json
j=json.load(open('$OUT/bench_w$w.json')); print(j['value'], j['ms_per_step'], j['config']['decode_group'])")"
  tail -1 "$OUT/bench_w$w.err" | cut -c1-200
done
timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-profile-pass --steps 96 --workers 16 --compute-type int8_float16 > "$OUT/bench_w16_i8.json" 2> "$OUT/bench_w16_i8.err"
echo "== int8 workers=16 $(python -c "
import json
j=json.load(open('$OUT/bench_w16_i8.json')); print(j['value'], j['ms_per_step'], j['config']['decode_group'])")"
