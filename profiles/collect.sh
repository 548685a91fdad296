#!/bin/bash
# One gpurun call that refreshes the evidence set of a round (run from the repo root on the GPU box):
#
#   gpurun --timeout 1200 -- 'timeout 1100 bash profiles/collect.sh r03'
#
# Writes under gpurun_out/<tag>/ and copies what is to be judged into profiles/<tag>_*:
#   bench.json                  python bench.py (the bench line: roofline + cpu_baseline + secondary measurements)
#   kernel_stats_w1.csv         rocprofv3 --kernel-trace --stats, ONE batch at a time (16-chunk decode runs), eager decode
#   kernel_stats_w32.csv        the same with the bench's 32 workers: merged decode runs next to the encoders (durations
#                               of concurrent kernels overlap in this one: read it for the decode stream)
#   pmc_fetch.json              FETCH_SIZE per kernel (x2 gfx950 correction applied by parse_pmc.py), one batch at a time
#   pmc_fetch_w32.json          the same for the configuration that is TIMED: 32 workers, merged decode runs (one lane);
#                               parse_pmc.py derives the chunks of the mean cross-attention launch from its grid
#   pmc_sq.json                 SQ wait / active / MFMA-busy / LDS-conflict counters per kernel, one batch at a time
# Counter passes are separate runs with --pmc only (gpurun refuses --pmc combined with the trace domains).
# The decode step runs eagerly under the profiler (FWAMD_NO_GRAPH=1): rocprofv3 7.2 crashes on replayed hipGraphs.
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
# kernel traces and counter passes: ONE decode run at a time (--decode-lanes 1), so that a kernel's duration is its own
Q="--no-cpu-baseline --no-profile-pass --no-secondary --decode-lanes 1"
cd "$R"
# (the bench line itself is taken AFTER the counter passes — bench.py reads this round's FETCH_SIZE pass for `traffic`)
cd /tmp; export TMPDIR=/tmp
trace() {   # name, bench args...
  local name=$1; shift
  FWAMD_NO_GRAPH=1 timeout 330 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$name" -o kt -- \
      python "$R/bench.py" $Q "$@" > "$OUT/prof_$name.log" 2>&1
  local f; f=$(find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_$name.csv" && cp "$f" "$R/profiles/${TAG}_kernel_stats_$name.csv"
  rm -rf "$OUT/prof_$name"
}
pmc() {     # name, bench args ... -- counters...
  local name=$1; shift
  local args=()
  while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
  FWAMD_NO_GRAPH=1 timeout 250 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/prof_$name" -o pmc -- \
      python "$R/bench.py" $Q "${args[@]}" > "$OUT/prof_$name.log" 2>&1
  local f; f=$(find "$OUT/prof_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$R/profiles/parse_pmc.py" "$f" > "$OUT/pmc_$name.json" && cp "$OUT/pmc_$name.json" "$R/profiles/${TAG}_pmc_$name.json"
  rm -rf "$OUT/prof_$name"
}
trace w1 --workers 1 --steps 2 --warmup 1
trace w32 --steps 64 --warmup 1
pmc fetch --workers 1 --steps 2 --warmup 1 -- FETCH_SIZE
pmc fetch_w32 --steps 32 --warmup 1 -- FETCH_SIZE
pmc sq --workers 1 --steps 1 --warmup 1 -- SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
cd "$R"
cp "$OUT/pmc_fetch.json" "$R/profiles/${TAG}_pmc_fetch.json" 2>/dev/null
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-400 "$OUT/bench.json"
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench_driver_cmd.err"; echo "driver-cmd bench rc=$?"; cut -c1-300 "$OUT/bench_driver_cmd.json"
head -12 "$OUT/kernel_stats_w1.csv" | cut -c1-150
head -8 "$OUT/kernel_stats_w32.csv" | cut -c1-150
ls -la "$OUT"
