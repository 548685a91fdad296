#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call7
mkdir -p "$OUT"
cd "$R"
t0=$SECONDS
timeout 600 python -m pytest "tests/test_gpu_full_size.py::test_distil_large_v3_float16" tests/test_gpu_int8.py tests/test_gpu_model.py tests/test_gpu_decode_group.py tests/test_gpu_pipeline.py tests/test_gpu_sequential.py -q -m gpu --maxfail=10 -s > "$OUT/pytest_part.log" 2>&1; echo "== pytest rc=$? $((SECONDS-t0))s"; tail -4 "$OUT/pytest_part.log" | cut -c1-300; grep -E "folded order|max prob diff|rule \(e\) tied|FAILED" "$OUT/pytest_part.log" | cut -c1-250
echo "== total $((SECONDS-t0))s"
