#!/bin/bash
# Round 6, GPU call 4: C5 tests again, the RCCL world-1 test (device -> device blob broadcast), the burst with the idle
# group split into 3 runs (FWAMD_IDLE_MIN_RUNS), the C4 rank batch as two halves, the distil C5 bench line with --vad and
# the host-side sampling profile of the distil pipeline.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_c5.py tests/test_gpu_rccl_world1.py -x -q -s > "$OUT/pytest_c5_call4.log" 2>&1
echo "pytest rc=$?"; grep -E "C5|device VAD|passed|failed|Error|assert" "$OUT/pytest_c5_call4.log" | cut -c1-400 | tail -20
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
for runs in 2 3 2 3; do
  FWAMD_IDLE_MIN_RUNS=$runs timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-profile-pass --no-secondary > "$OUT/tmp_burst.json" 2>> "$OUT/ab_idle_runs.err"
  python - "$runs" "$OUT/tmp_burst.json" >> "$OUT/ab_idle_runs.jsonl" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(json.dumps({"setting": f"burst20_min_runs_{sys.argv[1]}", "value": j["value"], "ms_per_step": j["ms_per_step"],
                  "decode_group": j["config"]["decode_group"], "verified": j["verified"]}))
PY
done
for runs in 2 3; do
  FWAMD_IDLE_MIN_RUNS=$runs timeout 400 python bench.py --gpus 1 --steps 128 --warmup 1 --no-cpu-baseline --no-profile-pass --no-secondary > "$OUT/tmp_burst.json" 2>> "$OUT/ab_idle_runs.err"
  python - "$runs" "$OUT/tmp_burst.json" >> "$OUT/ab_idle_runs.jsonl" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(json.dumps({"setting": f"steady128_min_runs_{sys.argv[1]}", "value": j["value"], "ms_per_step": j["ms_per_step"],
                  "decode_group": j["config"]["decode_group"], "verified": j["verified"]}))
PY
done
echo "idle runs:"; cat "$OUT/ab_idle_runs.jsonl"
timeout 600 python bench.py --gpus 1 --steps 8 --warmup 1 --no-cpu-baseline --no-profile-pass --pipeline-chunks 64 > "$OUT/bench_c4_halves.json" 2> "$OUT/bench_c4_halves.err"
echo "c4 rc=$?"; python -c "
import json; j=json.loads([l for l in open('$OUT/bench_c4_halves.json') if l.startswith('{')][-1]); print(json.dumps(j['one_batch_at_a_time']))"
timeout 900 python bench.py --gpus 1 --model distil-large-v3 --steps 20 --warmup 5 --word-timestamps --vad --no-cpu-baseline > "$OUT/bench_distil_large_v3_c5.json" 2> "$OUT/bench_distil_c5.err"
echo "distil rc=$?"; python -c "
import json; j=json.loads([l for l in open('$OUT/bench_distil_large_v3_c5.json') if l.startswith('{')][-1]); print(j['value'], json.dumps(j.get('steady')), json.dumps(j.get('pipeline')))"; tail -3 "$OUT/bench_distil_c5.err"
timeout 900 python profiles/host_profile_pipeline.py --model distil-large-v3 --chunks 480 --word-timestamps > "$OUT/host_profile_distil.json" 2> "$OUT/host_profile_distil.err"
echo "hostprof rc=$?"; head -c 6000 "$OUT/host_profile_distil.json"; tail -3 "$OUT/host_profile_distil.err"
