#!/bin/bash
# end of round 4: the C3 / C5 bench lines, then the whole GPU suite + smoke on HEAD
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_final2
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$SECONDS
B="python bench.py --no-cpu-baseline --steps 64"
timeout 300 $B --compute-type int8_float16 > "$OUT/bench_int8_float16.json" 2> "$OUT/bench_int8.err"; echo "== int8 rc=$? $(cut -c1-200 $OUT/bench_int8_float16.json)"
timeout 300 $B --model distil-large-v3 --word-timestamps > "$OUT/bench_distil_large_v3.json" 2> "$OUT/bench_distil.err"; echo "== distil rc=$? $(cut -c1-200 $OUT/bench_distil_large_v3.json)"
echo "== benches $((SECONDS-t0))s"
unset FWAMD_BLOB_CACHE
t1=$SECONDS
timeout 1300 python -m pytest tests/ -q -m gpu --maxfail=10 --durations=8 -s > "$OUT/pytest_gpu.log" 2>&1; echo "== pytest -m gpu rc=$? $((SECONDS-t1))s"
grep -E "MISMATCH|FAILED" "$OUT/pytest_gpu.log" | head -10 | cut -c1-250
tail -12 "$OUT/pytest_gpu.log" | cut -c1-200
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "== smoke rc=$?"; tail -1 "$OUT/smoke.log"
echo "== total $((SECONDS-t0))s"
