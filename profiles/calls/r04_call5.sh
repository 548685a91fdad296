#!/bin/bash
# attention kernel built with MFMA accumulators in VGPRs: unit + model tests, micro-benchmark
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call5
mkdir -p "$OUT"
cd "$R"
t0=$SECONDS
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q -m gpu --maxfail=5 > "$OUT/pytest_part.log" 2>&1; echo "== pytest rc=$? $((SECONDS-t0))s"; tail -3 "$OUT/pytest_part.log" | cut -c1-300
timeout 100 python profiles/attn_bench.py 0 > "$OUT/attn_bench.txt" 2>&1; echo "== attn bench rc=$?"; cat "$OUT/attn_bench.txt"
B="python bench.py --steps 96 --warmup 1 --no-secondary --no-profile-pass --no-cpu-baseline"
for which in base new; do
  if [ $which = base ]; then export FWAMD_LIB=$R/faster_whisper_amd/libfwamd_base.so; else unset FWAMD_LIB; fi
  timeout 200 $B > "$OUT/${which}.json" 2> "$OUT/${which}.err"
  python - <<PY
import json
try:
    j=json.load(open("$OUT/${which}.json")); print("$which", j["value"], j["ms_per_step"], j["verified"])
except Exception as e: print("$which unreadable", e)
PY
done
echo "== total $((SECONDS-t0))s"
