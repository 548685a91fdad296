#!/bin/bash
# what the explicit-LayerNorm evaluation order costs in the TIMED configuration (merged runs): one box, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_ln_bench
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
for i in 1 2; do
  for u in 0 2; do
    FWAMD_LN_UNFOLD=$u timeout 200 python bench.py --steps 96 --warmup 1 --no-secondary --no-profile-pass --no-cpu-baseline > "$OUT/unfold${u}_$i.json" 2> "$OUT/unfold${u}_$i.err"
    python - <<PY
import json
try:
    j=json.load(open("$OUT/unfold${u}_$i.json")); print("FWAMD_LN_UNFOLD=$u run $i:", j["value"], "x", j["ms_per_step"], "ms per step, verified", j["verified"])
except Exception as e: print("unreadable", e)
PY
  done
done
