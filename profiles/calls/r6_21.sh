#!/bin/bash
# Round 6, GPU call 21: the whole -m gpu suite + smoke() on the final library, then the evidence set (collect_r06.sh).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -s > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$? $(( $(date +%s) - t0 ))s"; tail -4 "$OUT/pytest_gpu.log" | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?"; tail -1 "$OUT/smoke.log"
bash profiles/collect_r06.sh
