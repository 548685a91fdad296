#!/bin/bash
# Round 6, GPU call 20: qkv / ffn1 of runs of >= 1 280 rows in the pipeline — the 256 x 128 form (144 KB of LDS, one workgroup per CU;
# the product) against the 128 x 128 form (64 KB, two per CU), equal in the isolated table: two builds alternating on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
rm -f "$OUT/ab_wide_cfg2.jsonl"
for i in 1 2 3; do
  for which in widecfg2 new; do
    if [ $which = new ]; then unset FWAMD_LIB; else export FWAMD_LIB=$R/faster_whisper_amd/libfwamd_$which.so; fi
    timeout 300 python bench.py --steps 64 --warmup 1 --no-secondary --no-cpu-baseline > "$OUT/tmp_ab.json" 2>> "$OUT/ab_wide.err"
    python - "$which" "$i" "$OUT/tmp_ab.json" >> "$OUT/ab_wide_cfg2.jsonl" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    f = j.get("families_ms_per_step", {})
    print(json.dumps({"build": sys.argv[1], "round": int(sys.argv[2]), "value": j["value"], "ms_per_step": j["ms_per_step"],
                      "dec_gemm": {k: v for k, v in f.items() if k.startswith("dec_gemm")},
                      "families_sum_ms": j.get("families_sum_ms"), "verified": j["verified"]}))
except Exception as e:
    print(json.dumps({"build": sys.argv[1], "round": int(sys.argv[2]), "error": str(e)}))
PY
  done
done
unset FWAMD_LIB
cat "$OUT/ab_wide_cfg2.jsonl"
