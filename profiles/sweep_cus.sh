#!/bin/bash
# Encoder CU partition sweep (fw_model_set_encoder_cus): bench line per setting.
#   gpurun --timeout 1200 -- 'timeout 1100 bash profiles/sweep_cus.sh r02j "0 96 112 128"'
set -u
TAG=${1:-cus}
LIST=${2:-"0 96 112 128"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
if [ -n "${TESTS:-}" ]; then
  timeout 600 python -m pytest $TESTS -q -m gpu --maxfail=10 > "$OUT/pytest.log" 2>&1; echo "== pytest rc=$?"; tail -25 "$OUT/pytest.log" | cut -c1-220
fi
for n in $LIST; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 32 --encoder-cus $n ${EXTRA:-} > "$OUT/bench_cus$n.json" 2> "$OUT/bench_cus$n.err"
  echo "== encoder_cus=$n rc=$? $(python - <<PY
import json
try:
    j=json.load(open("$OUT/bench_cus$n.json"))
    f=j.get("families_ms_per_step",{})
    print(j["value"], "ms/step", j["ms_per_step"], "enc_gemm", f.get("enc_gemm"), "cross", f.get("dec_cross_attn"), "dxd", f.get("dec_gemm_dxd"), "runs", j["config"]["decode_group"])
except Exception as e:
    print("unreadable", e)
PY
)"
  tail -2 "$OUT/bench_cus$n.err" | cut -c1-200
done
