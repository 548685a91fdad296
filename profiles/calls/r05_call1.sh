#!/bin/bash
# Round 5, GPU call 1: the new parity tests (fp32-oracle legs, config C2 single utterance, attention first-tile hazard),
# the decoder-linear crossover sweep (register-streaming vs LDS-staged kernel, 640 .. 1520 rows), an SQ counter pass of
# the timed configuration (32 workers, merged runs: dec_gemm_big_kernel — round 4 had none), one steady bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r05
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_full_size.py -q -s --durations=12 \
  -k "test_gpu_model or attention or test_large_v3_float16 or single_utterance or peaked or distil" \
  > "$OUT/call1_pytest.log" 2>&1
echo "== pytest rc=$? $(( $(date +%s) - t0 ))s"
grep -E "vs fp32|C2 single|passed|failed|MISMATCH|Error|first tile" "$OUT/call1_pytest.log" | cut -c1-420 | tail -70
t0=$(date +%s)
DLB_VARIANTS=0,1,10,11,12 timeout 240 python profiles/dec_linear_bench.py 640 800 960 1120 1280 1520 > "$OUT/call1_dec_linear_bench.txt" 2>&1
echo "== dec_linear_bench rc=$? $(( $(date +%s) - t0 ))s"; cat "$OUT/call1_dec_linear_bench.txt" | cut -c1-150
t0=$(date +%s)
timeout 200 python bench.py --steps 96 --warmup 1 --no-secondary --no-profile-pass --no-cpu-baseline > "$OUT/call1_bench_steady.json" 2> "$OUT/call1_bench_steady.err"
echo "== bench rc=$? $(( $(date +%s) - t0 ))s"; cut -c1-300 "$OUT/call1_bench_steady.json"
t0=$(date +%s)
cd /tmp; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-profile-pass --no-secondary --decode-lanes 1"
FWAMD_NO_GRAPH=1 timeout 330 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
  --output-format csv -d "$OUT/prof_sq_w32" -o pmc -- python "$R/bench.py" $Q --steps 32 --warmup 1 > "$OUT/prof_sq_w32.log" 2>&1
f=$(find "$OUT/prof_sq_w32" -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python "$R/profiles/parse_pmc.py" "$f" > "$OUT/pmc_sq_w32.json" && cp "$OUT/pmc_sq_w32.json" "$R/profiles/r05_pmc_sq_w32.json"
rm -rf "$OUT/prof_sq_w32"
echo "== sq_w32 $(( $(date +%s) - t0 ))s"
python - <<PY
import json
try:
    j=json.load(open("$OUT/pmc_sq_w32.json"))
    for k,v in j.items():
        if any(s in k for s in ("gemm_big","self_attn","wave_kernel","logits_process","gemm_f16")): print(k[:70], {c: round(x["mean"]) for c,x in v.items() if isinstance(x,dict)})
except Exception as e: print("sq unreadable", e)
PY
