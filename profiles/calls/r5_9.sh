#!/bin/bash
# Round 5, GPU call 9: the end-of-round evidence set (profiles/collect_r05.sh), the unmodified reference package on the shim
# (scratch copy next to the snapshot, untracked, removed afterwards), the C3 / C5 bench lines.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
bash profiles/collect_r05.sh
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$(date +%s)
( timeout 300 python profiles/run_reference_on_shim.py --ref _ref_scratch --backend engine > "$OUT/reference_on_shim.log" 2>&1; echo "rc=$?" >> "$OUT/reference_on_shim.log" )
echo "== shim $(( $(date +%s) - t0 ))s"; tail -6 "$OUT/reference_on_shim.log" | cut -c1-200
t0=$(date +%s)
timeout 300 python bench.py --compute-type int8_float16 --no-cpu-baseline > "$OUT/bench_int8_float16.json" 2> "$OUT/bench_int8_float16.err"
echo "== int8 rc=$? $(( $(date +%s) - t0 ))s"; cut -c1-200 "$OUT/bench_int8_float16.json"
t0=$(date +%s)
timeout 300 python bench.py --model distil-large-v3 --word-timestamps --no-cpu-baseline > "$OUT/bench_distil_large_v3.json" 2> "$OUT/bench_distil_large_v3.err"
echo "== distil rc=$? $(( $(date +%s) - t0 ))s"; cut -c1-200 "$OUT/bench_distil_large_v3.json"
