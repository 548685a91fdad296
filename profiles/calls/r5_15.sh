#!/bin/bash
# Round 5, GPU call 15: the pool / pipeline / int8 / sequential tests with the layered cross-K/V projection as the default
# (the whole suite ran on the build before it: profiles/r05_pytest_gpu.log; call 14 ran the model / full-size tests).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
t0=$(date +%s)
timeout 200 python -m pytest tests/test_gpu_decode_group.py tests/test_gpu_pipeline.py tests/test_gpu_sequential.py tests/test_gpu_int8.py tests/test_gpu_ct2_dir.py tests/test_gpu_vad.py -x -q > "$OUT/call15_pytest.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"; tail -2 "$OUT/call15_pytest.log" | cut -c1-200
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
