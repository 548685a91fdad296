#!/bin/bash
# Round-4 second GPU call: kernel tests on the new epilogue / grid / reductions, A/B micro-benchmarks with FETCH_SIZE,
# the LayerNorm-order probe, the bench.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call2
mkdir -p "$OUT"
cd "$R"
t0=$SECONDS
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_decode_group.py tests/test_gpu_int8.py -q -m gpu --maxfail=10 -x > "$OUT/pytest_kernels.log" 2>&1; echo "== pytest kernels/model rc=$? $((SECONDS-t0))s"; tail -6 "$OUT/pytest_kernels.log" | cut -c1-300; grep -E "cross-(K|V)" "$OUT/pytest_kernels.log" | tail -4
timeout 120 python profiles/attn_bench.py > "$OUT/attn_bench.txt" 2>&1; echo "== attn bench rc=$?"; cat "$OUT/attn_bench.txt"
timeout 200 python profiles/gemm_bench.py --iters 20 > "$OUT/gemm_bench.json" 2>&1; echo "== gemm_bench rc=$?"; python - <<PY
import json
try:
    j=json.load(open("$OUT/gemm_bench.json"))
    for k,v in j.items():
        if isinstance(v,dict) and "TFLOP/s" in v: print(k, v)
except Exception as e: print("unreadable", e)
PY
cd /tmp; export TMPDIR=/tmp
for prog in attn_bench gemm_bench; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_$prog" -o pmc -- python "$R/profiles/$prog.py" > "$OUT/pmc_$prog.log" 2>&1
  f=$(find "$OUT/pmc_$prog" -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python "$R/profiles/parse_pmc_dispatch.py" "$f" > "$OUT/pmc_fetch_$prog.txt"; fi
  rm -rf "$OUT/pmc_$prog"
  echo "== FETCH $prog"; head -40 "$OUT/pmc_fetch_$prog.txt"
done
cd "$R"
t1=$SECONDS
timeout 500 python profiles/ln_unfold_probe.py > "$OUT/ln_unfold_probe.txt" 2>&1; echo "== ln unfold probe rc=$? $((SECONDS-t1))s"; grep FWAMD_LN_UNFOLD "$OUT/ln_unfold_probe.txt"; tail -3 "$OUT/ln_unfold_probe.txt" | cut -c1-300
t1=$SECONDS
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench.err"; echo "== bench rc=$? $((SECONDS-t1))s"; python - <<PY
import json
try:
    j=json.load(open("$OUT/bench_driver_cmd.json"))
    print("value", j["value"], "steady", j.get("steady",{}).get("value"), "pipeline", j.get("pipeline",{}).get("value"), "cap", j.get("cap_case",{}).get("value"), "single", j.get("single_utterance",{}).get("latency_ms"), "one_batch", j.get("one_batch_at_a_time",{}).get("latency_ms_per_batch"))
    print(j.get("families_ms_per_step")); print(j.get("families_rate")); print(j["config"]["decode_group"], j.get("verified"))
except Exception as e: print("unreadable", e)
PY
tail -3 "$OUT/bench.err"
echo "== total $((SECONDS-t0))s"
