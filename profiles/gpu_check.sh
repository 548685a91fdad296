#!/bin/bash
# Validation call of a round: the GPU test suite in dependency order (single kernels first), then short bench runs.
#   gpurun --timeout 1500 -- 'bash profiles/gpu_check.sh r02b'
set -u
TAG=${1:-check}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t() { local name=$1; shift; local t0=$SECONDS; "$@" > "$OUT/$name.log" 2>&1; echo "== $name: rc=$? $((SECONDS-t0))s"; tail -${TAILN:-6} "$OUT/$name.log"; }
t kernels   timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_logmel.py -x -q
t group     timeout 300 python -m pytest tests/test_gpu_decode_group.py -x -q -s
t model     timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_int8.py tests/test_gpu_vad.py -q
t host      timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_sequential.py -q
TAILN=40 t fullsize timeout 700 python -m pytest tests/test_gpu_full_size.py -q -s
timeout 400 python bench.py --steps 16 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "== bench rc=$?"; cut -c1-1500 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
for w in 1 4; do
  timeout 200 python bench.py --workers $w --steps $((4*w)) --no-cpu-baseline --no-profile-pass --no-secondary > "$OUT/bench_w$w.json" 2> "$OUT/bench_w$w.err"
  echo "== bench workers=$w rc=$?"; cut -c1-400 "$OUT/bench_w$w.json"; tail -2 "$OUT/bench_w$w.err"
done
