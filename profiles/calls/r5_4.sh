#!/bin/bash
# Round 5, GPU call 4: the WHOLE -m gpu suite on the current build, with durations (the driver's step limit is 1 200 s).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu -s --durations=25 > "$OUT/call4_pytest_full.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"
grep -E "passed|failed|Error|error" "$OUT/call4_pytest_full.log" | cut -c1-300 | tail -8
grep -A28 "slowest" "$OUT/call4_pytest_full.log" | cut -c1-160
