#!/bin/bash
# Round 5, GPU call 2: parity of the rewritten decode-step kernels (logits rules as scalar lane masks, self-attention
# second form, weight-prefetch branch), then one-process A/Bs: self-attention forms (knob 2: 1 = rounds 1-4, 0 = product)
# with per-family profiles, weight prefetch (knob 3) on the solo paths.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_logits_rules.py tests/test_gpu_model.py tests/test_gpu_decode_group.py -q -s -x --durations=8 \
  -k "logits_rules or forms or prefetch or generate or decode_group or merged or lanes or pool" > "$OUT/call2_pytest.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"
grep -E "passed|failed|Error|error|forms|prefetch|assert" "$OUT/call2_pytest.log" | cut -c1-300 | tail -25
t0=$(date +%s)
timeout 420 python profiles/ab_r05.py --knob 2 --values 1,0 --rounds 2 --profile > "$OUT/call2_ab_self_attn.jsonl" 2> "$OUT/call2_ab_self_attn.err"
echo "== ab self-attn rc=$? $(( $(date +%s) - t0 ))s"; cut -c1-1500 "$OUT/call2_ab_self_attn.jsonl"; tail -3 "$OUT/call2_ab_self_attn.err"
t0=$(date +%s)
timeout 240 python profiles/ab_r05.py --knob 3 --values 0,1 --rounds 2 --no-merged > "$OUT/call2_ab_wprefetch.jsonl" 2> "$OUT/call2_ab_wprefetch.err"
echo "== ab wprefetch rc=$? $(( $(date +%s) - t0 ))s"; cat "$OUT/call2_ab_wprefetch.jsonl"; tail -3 "$OUT/call2_ab_wprefetch.err"
