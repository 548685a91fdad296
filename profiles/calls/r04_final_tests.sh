#!/bin/bash
# the whole GPU suite + smoke on the final build of the round (what the driver runs at round end)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_final
mkdir -p "$OUT"
cd "$R"
t0=$SECONDS
timeout 1500 python -m pytest tests/ -q -m gpu --maxfail=10 --durations=10 -s > "$OUT/pytest_gpu.log" 2>&1; echo "== pytest -m gpu rc=$? $((SECONDS-t0))s"
grep -E "MISMATCH|FAILED|Error" "$OUT/pytest_gpu.log" | head -20 | cut -c1-250
tail -16 "$OUT/pytest_gpu.log" | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "== smoke rc=$?"; tail -2 "$OUT/smoke.log"
echo "== total $((SECONDS-t0))s"
