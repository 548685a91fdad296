#!/bin/bash
# End-of-round evidence for round 5, trimmed to the GPU minutes that were left (collect.sh takes ~20): the kernel trace
# and the FETCH_SIZE pass of the TIMED configuration (32 workers, merged decode runs, one lane so that a kernel's duration is
# its own), then the default bench line (which reads this round's FETCH_SIZE pass for `roofline.traffic`) and the driver's
# command.  The one-batch-per-run trace / counter passes of round 4 and the SQ pass of call 1 (profiles/r05_pmc_sq_w32.json)
# are not repeated.
#   gpurun --timeout 900 -- 'bash profiles/collect_r05.sh'
set -u
TAG=r05
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
Q="--no-cpu-baseline --no-profile-pass --no-secondary --decode-lanes 1"
cd /tmp; export TMPDIR=/tmp
t0=$(date +%s)
FWAMD_NO_GRAPH=1 timeout 330 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_w32" -o kt -- \
    python "$R/bench.py" $Q --steps 64 --warmup 1 > "$OUT/prof_w32.log" 2>&1
f=$(find "$OUT/prof_w32" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_w32.csv" && cp "$f" "$R/profiles/${TAG}_kernel_stats_w32.csv"
rm -rf "$OUT/prof_w32"
echo "== trace w32 $(( $(date +%s) - t0 ))s"
t0=$(date +%s)
FWAMD_NO_GRAPH=1 timeout 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/prof_fetch_w32" -o pmc -- \
    python "$R/bench.py" $Q --steps 32 --warmup 1 > "$OUT/prof_fetch_w32.log" 2>&1
f=$(find "$OUT/prof_fetch_w32" -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python "$R/profiles/parse_pmc.py" "$f" > "$OUT/pmc_fetch_w32.json" && cp "$OUT/pmc_fetch_w32.json" "$R/profiles/${TAG}_pmc_fetch_w32.json"
rm -rf "$OUT/prof_fetch_w32"
echo "== fetch w32 $(( $(date +%s) - t0 ))s"
cd "$R"
t0=$(date +%s)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "driver-cmd bench rc=$? $(( $(date +%s) - t0 ))s"; cut -c1-300 "$OUT/bench_driver_cmd.json"
t0=$(date +%s)
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$? $(( $(date +%s) - t0 ))s"; cut -c1-400 "$OUT/bench.json"
head -8 "$OUT/kernel_stats_w32.csv" | cut -c1-150
ls -la "$OUT" | tail -12
