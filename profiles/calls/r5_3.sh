#!/bin/bash
# Round 5, GPU call 3: parity of the prefetch wave / hoisted residual loads / position blocks, then one-process A/Bs on the
# solo paths: prefetch wave (knob 3) and position blocks (knob 4).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$(date +%s)
timeout 700 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_full_size.py tests/test_gpu_sequential.py tests/test_gpu_pipeline.py -q -s -x --durations=8 \
  -k "test_gpu_model or dec_linear or single_utterance_large_v3_float16 or sequential or pipeline" > "$OUT/call3_pytest.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"
grep -E "passed|failed|Error|error|blocks|prefetch|assert|C2 single" "$OUT/call3_pytest.log" | cut -c1-300 | tail -25
t0=$(date +%s)
timeout 240 python profiles/ab_r05.py --knob 3 --values 0,1 --rounds 2 --no-merged > "$OUT/call3_ab_wprefetch.jsonl" 2> "$OUT/call3_ab_wprefetch.err"
echo "== ab wprefetch rc=$? $(( $(date +%s) - t0 ))s"; cat "$OUT/call3_ab_wprefetch.jsonl"; tail -3 "$OUT/call3_ab_wprefetch.err"
t0=$(date +%s)
timeout 240 python profiles/ab_r05.py --knob 4 --values 0,1 --rounds 2 --no-merged > "$OUT/call3_ab_blocks.jsonl" 2> "$OUT/call3_ab_blocks.err"
echo "== ab blocks rc=$? $(( $(date +%s) - t0 ))s"; cat "$OUT/call3_ab_blocks.jsonl"; tail -3 "$OUT/call3_ab_blocks.err"
