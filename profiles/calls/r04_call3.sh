#!/bin/bash
# Round-4 third GPU call: tests on the fixed permlane exchange, FETCH_SIZE per variant, the LayerNorm-order probe (clean), bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call3
mkdir -p "$OUT"
cd "$R"
t0=$SECONDS
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_decode_group.py tests/test_gpu_int8.py tests/test_gpu_pipeline.py "tests/test_gpu_full_size.py::test_large_v3_float16" -q -m gpu --maxfail=10 -s > "$OUT/pytest_part.log" 2>&1; echo "== pytest part rc=$? $((SECONDS-t0))s"; tail -4 "$OUT/pytest_part.log" | cut -c1-300; grep -E "cross-(K|V)|MISMATCH|max prob diff|align token|per token" "$OUT/pytest_part.log" | tail -12 | cut -c1-200
timeout 100 python profiles/attn_bench.py > "$OUT/attn_bench.txt" 2>&1; echo "== attn bench rc=$?"; cat "$OUT/attn_bench.txt"
timeout 200 python profiles/gemm_bench.py --iters 20 > "$OUT/gemm_bench.json" 2>&1; echo "== gemm_bench rc=$?"; python - <<PY
import json
try:
    j=json.load(open("$OUT/gemm_bench.json"))
    for k,v in j.items():
        if isinstance(v,dict) and "TFLOP/s" in v: print(k, v)
except Exception as e: print("unreadable", e)
PY
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_attn$v" -o pmc -- python "$R/profiles/attn_bench.py" $v > "$OUT/pmc_attn$v.log" 2>&1
  f=$(find "$OUT/pmc_attn$v" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$R/profiles/parse_pmc_dispatch.py" "$f" | grep -v rocclr > "$OUT/pmc_fetch_attn_variant$v.txt"
  rm -rf "$OUT/pmc_attn$v"; echo "== FETCH attention variant $v"; cat "$OUT/pmc_fetch_attn_variant$v.txt"
  timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_gemm$v" -o pmc -- python "$R/profiles/gemm_bench.py" --order $v --iters 5 > "$OUT/pmc_gemm$v.log" 2>&1
  f=$(find "$OUT/pmc_gemm$v" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$R/profiles/parse_pmc_dispatch.py" "$f" | grep -v rocclr > "$OUT/pmc_fetch_gemm_order$v.txt"
  rm -rf "$OUT/pmc_gemm$v"; echo "== FETCH gemm order $v"; cat "$OUT/pmc_fetch_gemm_order$v.txt"
done
cd "$R"
t1=$SECONDS
timeout 500 python profiles/ln_unfold_probe.py > "$OUT/ln_unfold_probe.txt" 2>&1; echo "== ln unfold probe rc=$? $((SECONDS-t1))s"; grep FWAMD_LN_UNFOLD "$OUT/ln_unfold_probe.txt"; tail -2 "$OUT/ln_unfold_probe.txt" | cut -c1-300
t1=$SECONDS
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench.err"; echo "== bench rc=$? $((SECONDS-t1))s"; python - <<PY
import json
try:
    j=json.load(open("$OUT/bench_driver_cmd.json"))
    print("value", j["value"], "steady", j.get("steady",{}).get("value"), "pipeline", j.get("pipeline",{}).get("value"), "cap", j.get("cap_case",{}).get("value"), "single", j.get("single_utterance",{}).get("latency_ms"), "one_batch", j.get("one_batch_at_a_time",{}).get("latency_ms_per_batch"))
    print(j.get("families_ms_per_step")); print(j.get("families_rate")); print(j["config"]["decode_group"], j.get("verified"))
    print(j.get("roofline_others",{}).get("dec_gemm"))
except Exception as e: print("unreadable", e)
PY
tail -3 "$OUT/bench.err"
echo "== total $((SECONDS-t0))s"
