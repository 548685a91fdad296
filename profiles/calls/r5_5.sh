#!/bin/bash
# Round 5, GPU call 5: stream priorities, one box, separate processes (a stream's priority is fixed at creation).
# ROCm gives each priority level its own hardware queues: "high" decode streams no longer share a hardware queue with the
# 32 workers' encoder streams and take freed CUs first.  Settings alternate so that box drift shows (base first and last).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
Q="--steps 96 --warmup 1 --no-secondary --no-profile-pass --no-cpu-baseline"
run() {   # tag, env assignments...
  local tag=$1; shift
  local t0=$(date +%s)
  env "$@" timeout 200 python bench.py $Q > "$OUT/call5_$tag.json" 2> "$OUT/call5_$tag.err"
  echo "== $tag rc=$? $(( $(date +%s) - t0 ))s  $(python - "$OUT/call5_$tag.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    g = j["config"]["decode_group"]
    print(j["value"], "x", j["ms_per_step"], "ms/step; runs", g.get("decode_runs"), "chunks/run", g.get("chunks_per_run"), "verified", j.get("verified"))
except Exception as e:
    print("unreadable:", e)
PY
)"
}
run base1 FWAMD_NOP=1
run dec_high FWAMD_DEC_STREAM_PRIO=high
run dec_high_enc_low FWAMD_DEC_STREAM_PRIO=high FWAMD_ENC_STREAM_PRIO=low
run enc_low FWAMD_ENC_STREAM_PRIO=low
run base2 FWAMD_NOP=1
run dec_high2 FWAMD_DEC_STREAM_PRIO=high
# the driver's command (20-step burst) with and without
for t in base dec_high; do
  e=FWAMD_NOP=1; [ $t = dec_high ] && e=FWAMD_DEC_STREAM_PRIO=high
  env $e timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-profile-pass --no-cpu-baseline > "$OUT/call5_burst_$t.json" 2> "$OUT/call5_burst_$t.err"
  echo "== burst $t rc=$? $(cut -c1-160 "$OUT/call5_burst_$t.json")"
done
