#!/bin/bash
# Round 5, GPU call 13: after the refactor of the idle-group rule into idle_lead_chunks (host-only unit test): the decode-group
# tests, the ABI test on the box, smoke, and the driver's burst once more.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu_decode_group.py tests/test_abi.py tests/test_host_logic.py tests/test_gpu_pipeline.py -x -q > "$OUT/call13_pytest.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"; tail -2 "$OUT/call13_pytest.log" | cut -c1-200
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-200
t0=$(date +%s)
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-profile-pass --no-cpu-baseline > "$OUT/call13_burst.json" 2> "$OUT/call13_burst.err"
echo "== burst rc=$? $(( $(date +%s) - t0 ))s $(cut -c1-180 "$OUT/call13_burst.json")"
