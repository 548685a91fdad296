#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
mkdir -p gpurun_out/r02
timeout 400 python -m pytest tests/ -q -m gpu --maxfail=8 > gpurun_out/r02/pytest_gpu.log 2>&1; echo "== pytest -m gpu rc=$?"; tail -4 gpurun_out/r02/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1000 bash profiles/collect.sh r02
