#!/bin/bash
# Round 5, GPU call 14: the cross-K/V projections of all decoder layers as two launches (fw_test_knob 6): bit-identity tests,
# the large-v3 configuration against the oracle, and the A/B (one process, alternating) with the per-family profile.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$(date +%s)
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_full_size.py -x -q -s \
  -k "cross_kv or test_large_v3_float16 or single_utterance_large_v3_float16 or test_gpu_model and (align or detect or beam)" > "$OUT/call14_pytest.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"
grep -E "layered|passed|failed|Error" "$OUT/call14_pytest.log" | cut -c1-220 | tail -8
t0=$(date +%s)
timeout 400 python profiles/ab_r05.py --knob 6 --values 0,1 --rounds 2 --steps 64 --profile > "$OUT/call14_ab.jsonl" 2> "$OUT/call14_ab.err"
echo "== ab rc=$? $(( $(date +%s) - t0 ))s"
python - "$OUT/call14_ab.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try:
        j = json.loads(l)
    except Exception:
        continue
    f = j.get("families_ms_per_batch", {}); g = j.get("families_ms_solo_batch", {})
    print(j["round"], j["setting"], "merged", j.get("merged_rtf"), "one_batch", j.get("one_batch_ms"), "single", j.get("single_utterance_ms"),
          "cross_kv merged/solo", f.get("cross_kv_gemm"), g.get("cross_kv_gemm"), "enc_gemm", f.get("enc_gemm"), g.get("enc_gemm"), "same", j.get("same_results_as_first_setting"))
PY
