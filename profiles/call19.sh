#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r02k; mkdir -p "$OUT"; cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
timeout 120 python profiles/diag_int8.py > "$OUT/diag_int8.txt" 2>&1; echo "== diag rc=$?"; grep -v "^$" "$OUT/diag_int8.txt" | tail -30 | cut -c1-260
timeout 600 python -m pytest tests/test_gpu_int8.py tests/test_gpu_model.py tests/test_gpu_decode_group.py tests/test_gpu_pipeline.py -q -m gpu --maxfail=10 > "$OUT/pytest.log" 2>&1; echo "== pytest rc=$?"; tail -8 "$OUT/pytest.log" | cut -c1-200
timeout 200 python profiles/dec_linear_bench.py 80 320 640 > "$OUT/dec_linear_bench.txt" 2>&1; echo "== dec linear bench rc=$?"; cat "$OUT/dec_linear_bench.txt"
bash profiles/sweep_cus.sh r02k "0 96 128"
