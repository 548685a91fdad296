#!/bin/bash
# Round 6, GPU call 14: the vocabulary projection with its LayerNorm statistics split over the four waves of a workgroup —
# correctness (the logits / generate tests), then three builds on one box: the round-5 library (un-split), split, split +
# launch bounds for a third wave per SIMD: dec_logits ms per step from the bench's profiled round, and the steady state.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_logits_rules.py -q -k "logits or generate or greedy or beam" > "$OUT/pytest_call14.log" 2>&1
echo "pytest rc=$?"; tail -4 "$OUT/pytest_call14.log"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
rm -f "$OUT/ab_logits_split.jsonl"
for i in 1 2; do
  for which in r05 split_nobounds new; do
    if [ $which = new ]; then unset FWAMD_LIB; else export FWAMD_LIB=$R/faster_whisper_amd/libfwamd_$which.so; fi
    timeout 300 python bench.py --steps 64 --warmup 1 --no-secondary --no-cpu-baseline > "$OUT/tmp_ab.json" 2>> "$OUT/ab_logits.err"
    python - "$which" "$i" "$OUT/tmp_ab.json" >> "$OUT/ab_logits_split.jsonl" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    f = j.get("families_ms_per_step", {})
    print(json.dumps({"build": sys.argv[1], "round": int(sys.argv[2]), "value": j["value"], "ms_per_step": j["ms_per_step"],
                      "dec_logits_ms": f.get("dec_logits"), "dec_gemm_ms": round(sum(v for k, v in f.items() if k.startswith("dec_gemm")), 3),
                      "families_sum_ms": j.get("families_sum_ms"), "verified": j["verified"]}))
except Exception as e:
    print(json.dumps({"build": sys.argv[1], "round": int(sys.argv[2]), "error": str(e)}))
PY
  done
done
unset FWAMD_LIB
cat "$OUT/ab_logits_split.jsonl"
