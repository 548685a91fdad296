#!/bin/bash
# Round 5, GPU call 12: the one-batch-per-run passes of the evidence set on the final build (collect_r05.sh took the timed
# configuration only): kernel trace, FETCH_SIZE and SQ counters with --workers 1 (solo decode runs of 16 chunks).
set -u
TAG=r05
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
Q="--no-cpu-baseline --no-profile-pass --no-secondary --decode-lanes 1"
cd /tmp; export TMPDIR=/tmp
t0=$(date +%s)
FWAMD_NO_GRAPH=1 timeout 330 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_w1" -o kt -- \
    python "$R/bench.py" $Q --workers 1 --steps 2 --warmup 1 > "$OUT/prof_w1.log" 2>&1
f=$(find "$OUT/prof_w1" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_w1.csv"
rm -rf "$OUT/prof_w1"
echo "== trace w1 $(( $(date +%s) - t0 ))s"
pmc() {     # name, counters...
  local name=$1; shift
  local t0=$(date +%s)
  FWAMD_NO_GRAPH=1 timeout 250 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/prof_$name" -o pmc -- \
      python "$R/bench.py" $Q --workers 1 --steps 1 --warmup 1 > "$OUT/prof_$name.log" 2>&1
  local f; f=$(find "$OUT/prof_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$R/profiles/parse_pmc.py" "$f" > "$OUT/pmc_$name.json"
  rm -rf "$OUT/prof_$name"
  echo "== pmc $name $(( $(date +%s) - t0 ))s"
}
pmc fetch FETCH_SIZE
pmc sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
head -14 "$OUT/kernel_stats_w1.csv" | cut -c1-160
