#!/bin/bash
# Round 6, GPU call 18: the folded LayerNorm's statistics on the MATRIX pipe (ln_stat_step: ones x X^T and X x X^T) in the
# register-streaming, LDS-staged and vocabulary-projection kernels — every decoder-linear / logits / model test (fp64 references,
# bit identity between the kernel forms, oracle parity at micro / tiny.en), the isolated table, and the previous build against
# this one on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_decode_group.py tests/test_gpu_logits_rules.py -q > "$OUT/pytest_call18.log" 2>&1
echo "pytest rc=$?"; tail -6 "$OUT/pytest_call18.log" | cut -c1-300
DLB_VARIANTS=0,10,11,12,16 timeout 600 python profiles/dec_linear_bench.py 320 640 800 1280 > "$OUT/dec_linear_bench_call18.txt" 2> "$OUT/dec_linear_bench_call18.err"
echo "dec_linear rc=$?"; cat "$OUT/dec_linear_bench_call18.txt"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
rm -f "$OUT/ab_mfma_stats.jsonl"
for i in 1 2; do
  for which in prev new; do
    if [ $which = new ]; then unset FWAMD_LIB; else export FWAMD_LIB=$R/faster_whisper_amd/libfwamd_$which.so; fi
    timeout 300 python bench.py --steps 64 --warmup 1 --no-secondary --no-cpu-baseline > "$OUT/tmp_ab.json" 2>> "$OUT/ab_mfma.err"
    python - "$which" "$i" "$OUT/tmp_ab.json" >> "$OUT/ab_mfma_stats.jsonl" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    f = j.get("families_ms_per_step", {})
    print(json.dumps({"build": sys.argv[1], "round": int(sys.argv[2]), "value": j["value"], "ms_per_step": j["ms_per_step"],
                      "dec_logits_ms": f.get("dec_logits"), "dec_gemm": {k: v for k, v in f.items() if k.startswith("dec_gemm")},
                      "dec_gemm_ms": round(sum(v for k, v in f.items() if k.startswith("dec_gemm")), 3),
                      "families_sum_ms": j.get("families_sum_ms"), "verified": j["verified"]}))
except Exception as e:
    print(json.dumps({"build": sys.argv[1], "round": int(sys.argv[2]), "error": str(e)}))
PY
  done
done
unset FWAMD_LIB
cat "$OUT/ab_mfma_stats.jsonl"
