#!/bin/bash
# Round 6, GPU call 12: A/B of two BUILDS on one box (boxes differ by 3-5 %): the round-5 library (git archive 11929e9
# faster_whisper_amd/csrc include | build.sh -> faster_whisper_amd/libfwamd_r05.so, loaded through FWAMD_LIB) against the
# current tree, alternating — the steady state (96 steps) and the driver's 20-step burst; then what the folded LayerNorm's
# statistics cost the decoder linears (the same shapes with the fold forced off).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
rm -f "$OUT/ab_builds.jsonl"
for i in 1 2; do
  for which in r05 new; do
    if [ $which = r05 ]; then export FWAMD_LIB=$R/faster_whisper_amd/libfwamd_r05.so; else unset FWAMD_LIB; fi
    for cmd in "--steps 96 --warmup 1" "--steps 20 --warmup 5"; do
      timeout 300 python bench.py $cmd --no-secondary --no-profile-pass --no-cpu-baseline > "$OUT/tmp_ab.json" 2>> "$OUT/ab_builds.err"
      python - "$which" "$i" "$cmd" "$OUT/tmp_ab.json" >> "$OUT/ab_builds.jsonl" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])
    print(json.dumps({"build": sys.argv[1], "round": int(sys.argv[2]), "cmd": sys.argv[3], "value": j["value"],
                      "ms_per_step": j["ms_per_step"], "decode_group": j["config"]["decode_group"], "verified": j["verified"]}))
except Exception as e:
    print(json.dumps({"build": sys.argv[1], "round": int(sys.argv[2]), "cmd": sys.argv[3], "error": str(e)}))
PY
    done
  done
done
unset FWAMD_LIB
echo "builds:"; cat "$OUT/ab_builds.jsonl"
for lnf in 1 0; do
  echo "DLB_LNF=$lnf" >> "$OUT/dec_linear_bench_call12_lnf.txt"
  DLB_LNF=$lnf DLB_VARIANTS=0,12 timeout 600 python profiles/dec_linear_bench.py 320 640 800 1280 >> "$OUT/dec_linear_bench_call12_lnf.txt" 2>> "$OUT/dec_linear_bench_call12.err"
done
cat "$OUT/dec_linear_bench_call12_lnf.txt"
