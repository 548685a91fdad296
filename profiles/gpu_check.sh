#!/bin/bash
# Validation call of a round: the whole GPU test suite (as the driver runs it), then the bench configurations.
#   gpurun --timeout 1800 -- 'timeout 1700 bash profiles/gpu_check.sh r02g'
set -u
TAG=${1:-check}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$SECONDS
timeout 1200 python -m pytest tests/ -q -m gpu --maxfail=8 --durations=8 > "$OUT/pytest_gpu.log" 2>&1; echo "== pytest -m gpu rc=$? $((SECONDS-t0))s"; grep -E "MISMATCH|explicit-LayerNorm|teacher-forced cum|beam 5 chunk|max prob diff|align token" "$OUT/pytest_gpu.log" | cut -c1-220 | tail -60; tail -14 "$OUT/pytest_gpu.log"
summary() {
python - "$@" <<PY
import json, sys
for n in sys.argv[1:]:
    try:
        j=json.load(open("$OUT/%s.json"%n))
    except Exception as e:
        print(n, "unreadable", e); continue
    print(n, j["value"], "cap", j.get("cap_case",{}).get("value"), "pipeline", j.get("pipeline"), "single", j.get("single_utterance",{}).get("value"))
    print("   roofline", j.get("roofline"))
    print("   others", j.get("roofline_others"))
    print("   ms", j.get("families_ms_per_step"))
PY
}
B="python bench.py --no-cpu-baseline"
timeout 400 $B --steps 32 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "== bench rc=$?"; cut -c1-300 "$OUT/bench.json"; tail -2 "$OUT/bench.err"
if [ -n "${AB_ENV:-}" ]; then   # A/B of an experiment knob: AB_ENV="FWAMD_X=1"
  env $AB_ENV timeout 400 $B --steps 32 --no-secondary > "$OUT/bench_ab.json" 2> "$OUT/bench_ab.err"; echo "== bench with $AB_ENV rc=$? $(cut -c1-160 $OUT/bench_ab.json)"
fi
if [ -n "${QUICK:-}" ]; then summary bench bench_ab; exit 0; fi
timeout 400 $B --steps 32 --compute-type int8_float16 > "$OUT/bench_int8.json" 2> "$OUT/bench_int8.err"; echo "== int8 rc=$? $(cut -c1-160 $OUT/bench_int8.json)"
timeout 400 $B --steps 32 --model distil-large-v3 --word-timestamps > "$OUT/bench_distil.json" 2> "$OUT/bench_distil.err"; echo "== distil rc=$? $(cut -c1-160 $OUT/bench_distil.json)"
summary bench bench_int8 bench_distil
