#!/bin/bash
# Round 5, GPU call 7: how an IDLE decode group leads its first run (FWAMD_IDLE_FILL_PCT), on the driver's 20-step burst and
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
  env "$@" timeout 240 python bench.py $args $N > "$OUT/call7_$tag.json" 2> "$OUT/call7_$tag.err"
  echo "== $tag rc=$? $(( $(date +%s) - t0 ))s  $(python - "$OUT/call7_$tag.json" <<'PY'
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
# (the idle group splits the known work evenly over two runs: FWAMD_IDLE_BALANCE; how long a leader waits after the last arrival)
run burst_balance "$B" FWAMD_IDLE_BALANCE=1
run burst_balance_wait150 "$B --merge-wait-ms 150" FWAMD_IDLE_BALANCE=1
run burst_balance_wait250 "$B --merge-wait-ms 250" FWAMD_IDLE_BALANCE=1
run steady_base "$S" FWAMD_NOP=1
run steady_balance_wait150 "$S --merge-wait-ms 150" FWAMD_IDLE_BALANCE=1
run steady_wait150 "$S --merge-wait-ms 150" FWAMD_NOP=1
run steady_wait250 "$S --merge-wait-ms 250" FWAMD_NOP=1
run burst_base "$B" FWAMD_NOP=1
