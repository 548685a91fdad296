#!/bin/bash
# Validation call of a round: the GPU test suite in dependency order (single kernels first), then bench runs,
# experiment knobs and a kernel trace of merged decode runs.
#   gpurun --timeout 1800 -- 'timeout 1700 bash profiles/gpu_check.sh r02c'
set -u
TAG=${1:-check}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t() { local name=$1; shift; local t0=$SECONDS; "$@" > "$OUT/$name.log" 2>&1; echo "== $name: rc=$? $((SECONDS-t0))s"; tail -${TAILN:-6} "$OUT/$name.log"; }
t kernels   timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_logmel.py -x -q -s
grep -E "gemm .*rel err|TFLOP" "$OUT/kernels.log" | head -20
t model     timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_int8.py tests/test_gpu_decode_group.py -q
TAILN=60 t fullsize timeout 700 python -m pytest tests/test_gpu_full_size.py -q -s
B="python bench.py --no-cpu-baseline"
timeout 400 $B --steps 32 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "== bench rc=$?"; cut -c1-700 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
Q="--no-profile-pass --no-secondary"
run() { local name=$1; shift; timeout 200 env "$@" $B $Q --steps 32 > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "== $name rc=$? $(python -c "import json,sys; j=json.load(open('$OUT/bench_$name.json')); print(j['value'], j['config']['decode_group'])" 2>&1 | tail -1)"; }
run fill100 FWAMD_GROUP_FILL=1.0
run fill25  FWAMD_GROUP_FILL=0.25
run serial  FWAMD_ENC_SERIAL=1
run serial_fill100 FWAMD_ENC_SERIAL=1 FWAMD_GROUP_FILL=1.0
timeout 200 $B $Q --steps 8 --workers 1 > "$OUT/bench_w1.json" 2> "$OUT/bench_w1.err"; echo "== w1 $(cut -c1-160 $OUT/bench_w1.json)"
timeout 300 $B $Q --steps 32 --compute-type int8_float16 > "$OUT/bench_int8.json" 2> "$OUT/bench_int8.err"; echo "== int8 $(cut -c1-160 $OUT/bench_int8.json)"
# kernel trace of merged decode runs (eager decode: rocprofv3 7.2 crashes on replayed graphs)
cd /tmp; export TMPDIR=/tmp
FWAMD_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_kt" -o kt -- \
    python "$R/bench.py" --no-cpu-baseline $Q --steps 8 --warmup 1 > "$OUT/prof_kt.log" 2>&1
f=$(find "$OUT/prof_kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/prof_kt"
head -25 "$OUT/kernel_stats.csv" | cut -c1-150
