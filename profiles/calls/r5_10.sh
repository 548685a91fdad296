#!/bin/bash
# Round 5, GPU call 10: the staged V^T epilogue of the encoder GEMM (fw_test_knob 5): kernel / model tests (bit-identity with
# the direct stores), the large-v3 encoder against the oracle, the isolated A/B of the V^T projection, the pipeline A/B.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$(date +%s)
timeout 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_int8.py tests/test_gpu_full_size.py -x -q -s \
  -k "gemm or vt_epilogue or encode or attention or test_large_v3_float16 or test_distil" > "$OUT/call10_pytest.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"
grep -E "staged epilogue|V\^T epilogue|passed|failed|Error" "$OUT/call10_pytest.log" | cut -c1-220 | tail -12
t0=$(date +%s)
timeout 120 python profiles/vt_epilogue_ab.py --rounds 3 > "$OUT/call10_vt_ab.jsonl" 2> "$OUT/call10_vt_ab.err"
echo "== vt ab rc=$? $(( $(date +%s) - t0 ))s"; cat "$OUT/call10_vt_ab.jsonl"
t0=$(date +%s)
timeout 400 python profiles/ab_r05.py --knob 5 --values 0,1 --rounds 2 --steps 64 --profile > "$OUT/call10_ab_pipeline.jsonl" 2> "$OUT/call10_ab_pipeline.err"
echo "== pipeline ab rc=$? $(( $(date +%s) - t0 ))s"
python - "$OUT/call10_ab_pipeline.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try:
        j = json.loads(l)
    except Exception:
        continue
    f = j.get("families_ms_per_batch", {})
    print(j["round"], j["setting"], "merged", j.get("merged_rtf"), "one_batch", j.get("one_batch_ms"), "single", j.get("single_utterance_ms"),
          "enc_gemm", f.get("enc_gemm"), "enc_attn", f.get("enc_attn"), "same", j.get("same_results_as_first_setting"))
PY
