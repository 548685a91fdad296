#!/bin/bash
# Round 5, GPU call 6: how an IDLE decode group leads its first run (FWAMD_IDLE_FILL_PCT), on the driver's 20-step burst and
# on the steady command; worker counts above 32.  One box, separate processes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
N="--no-secondary --no-profile-pass --no-cpu-baseline"
run() {   # tag, "bench args", env assignments...
  local tag=$1; local args=$2; shift; shift
  local t0=$(date +%s)
  env "$@" timeout 240 python bench.py $args $N > "$OUT/call6_$tag.json" 2> "$OUT/call6_$tag.err"
  echo "== $tag rc=$? $(( $(date +%s) - t0 ))s  $(python - "$OUT/call6_$tag.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g = j["config"]["decode_group"]
    print(j["value"], "x", j["ms_per_step"], "ms/step; cap", g.get("capacity_chunks"), "runs", g.get("decode_runs"), "chunks/run", g.get("chunks_per_run"), "largest", g.get("largest_run_chunks"), "verified", j.get("verified"))
except Exception as e:
    print("unreadable:", e)
PY
)"
}
B="--gpus 1 --steps 20 --warmup 5"
S="--steps 96 --warmup 1"
run burst_base "$B" FWAMD_NOP=1
run burst_idle50 "$B" FWAMD_IDLE_FILL_PCT=50
run burst_idle35 "$B" FWAMD_IDLE_FILL_PCT=35
run burst_idle65 "$B" FWAMD_IDLE_FILL_PCT=65
run burst_idle50_wait150 "$B --merge-wait-ms 150" FWAMD_IDLE_FILL_PCT=50
run burst_base2 "$B" FWAMD_NOP=1
run steady_base "$S" FWAMD_NOP=1
run steady_idle50 "$S" FWAMD_IDLE_FILL_PCT=50
run steady_w40 "--steps 120 --warmup 1 --workers 40" FWAMD_NOP=1
run steady_w48 "--steps 144 --warmup 1 --workers 48" FWAMD_NOP=1
