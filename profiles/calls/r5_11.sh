#!/bin/bash
# Round 5, GPU call 11: the WHOLE -m gpu suite on the final build, with durations (the driver's step limit is 1 200 s), then smoke().
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
t0=$(date +%s)
timeout 1300 python -m pytest tests/ -x -q -m gpu -s --durations=30 > "$OUT/call11_pytest_full.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"
grep -E "passed|failed|Error|error" "$OUT/call11_pytest_full.log" | cut -c1-300 | tail -8
grep -A32 "slowest" "$OUT/call11_pytest_full.log" | cut -c1-160
t0=$(date +%s)
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/call11_smoke.log" 2>&1
echo "== smoke rc=$? $(( $(date +%s) - t0 ))s"; tail -3 "$OUT/call11_smoke.log" | cut -c1-300
