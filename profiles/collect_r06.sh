#!/bin/bash
# Evidence set of round 6 (one box): collect.sh's passes (kernel traces w1 / w32, FETCH_SIZE w1 / w32, SQ w1, the default bench
# line, the driver's command) + the SQ pass of the TIMED configuration + the C3 (int8_float16) and C5 (distil-large-v3,
# native VAD + word timestamps inside the pipeline wall) lines.
#   gpurun --timeout 3000 -- 'bash profiles/collect_r06.sh'
set -u
TAG=r06
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
t0=$(date +%s)
timeout 1700 bash profiles/collect.sh $TAG > "$OUT/collect.log" 2>&1
echo "== collect.sh $(( $(date +%s) - t0 ))s"; tail -25 "$OUT/collect.log" | cut -c1-300
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
Q="--no-cpu-baseline --no-profile-pass --no-secondary --decode-lanes 1"
cd /tmp; export TMPDIR=/tmp
t0=$(date +%s)
FWAMD_NO_GRAPH=1 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/prof_sq_w32" -o pmc -- python "$R/bench.py" $Q --steps 32 --warmup 1 > "$OUT/prof_sq_w32.log" 2>&1
f=$(find "$OUT/prof_sq_w32" -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python "$R/profiles/parse_pmc.py" "$f" > "$OUT/pmc_sq_w32.json"
rm -rf "$OUT/prof_sq_w32"
echo "== sq w32 $(( $(date +%s) - t0 ))s"
cd "$R"
t0=$(date +%s)
timeout 600 python bench.py --compute-type int8_float16 --no-cpu-baseline > "$OUT/bench_int8_float16.json" 2> "$OUT/bench_int8.err"; echo "int8 rc=$? $(( $(date +%s) - t0 ))s"; cut -c1-250 "$OUT/bench_int8_float16.json"
t0=$(date +%s)
timeout 600 python bench.py --model distil-large-v3 --word-timestamps --vad --no-cpu-baseline > "$OUT/bench_distil_large_v3.json" 2> "$OUT/bench_distil.err"; echo "distil rc=$? $(( $(date +%s) - t0 ))s"; cut -c1-250 "$OUT/bench_distil_large_v3.json"
ls -la "$OUT" | tail -30
