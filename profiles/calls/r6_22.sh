#!/bin/bash
# Round 6, GPU call 22: the ring depth of the 128 x 64 form (d x d, cross-q, ffn2 of merged runs) in the pipeline: 5 stages
# (60 KB of LDS, the product) against 4 (48 KB) and 3 (36 KB) — three builds alternating on one box.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
rm -f "$OUT/ab_cfg1_ring.jsonl"
for i in 1 2 3; do
  for which in cfg1nst4 cfg1nst3 new; do
    if [ $which = new ]; then unset FWAMD_LIB; else export FWAMD_LIB=$R/faster_whisper_amd/libfwamd_$which.so; fi
    timeout 300 python bench.py --steps 64 --warmup 1 --no-secondary --no-cpu-baseline > "$OUT/tmp_ab.json" 2>> "$OUT/ab_ring.err"
    python - "$which" "$i" "$OUT/tmp_ab.json" >> "$OUT/ab_cfg1_ring.jsonl" <<'PY'
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
cat "$OUT/ab_cfg1_ring.jsonl"
