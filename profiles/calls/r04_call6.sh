#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call6
mkdir -p "$OUT"
cd "$R"
t0=$SECONDS
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_decode_group.py -q -m gpu --maxfail=5 -s > "$OUT/pytest_part.log" 2>&1; echo "== pytest rc=$? $((SECONDS-t0))s"; tail -3 "$OUT/pytest_part.log" | cut -c1-300; grep -E "^attention|resizes applied|encoder: max" "$OUT/pytest_part.log" | cut -c1-200
timeout 100 python profiles/attn_bench.py 0 > "$OUT/attn_bench.txt" 2>&1; echo "== attn bench rc=$?"; cat "$OUT/attn_bench.txt"
echo "== total $((SECONDS-t0))s"
