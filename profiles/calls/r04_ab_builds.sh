#!/bin/bash
# A/B of two builds of libfwamd.so on ONE GPU box (boxes differ by 3-5 %, more than most kernel changes): the build of
# the call-1 commit (shared cross-K/V pool, round-3 kernels) against the current tree, alternating, the bench's steady
# measurement (64 steps, 32 workers, two lanes), no secondary measurements.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_ab
mkdir -p "$OUT"
cd "$R"
B="python bench.py --steps 96 --warmup 1 --no-secondary --no-profile-pass --no-cpu-baseline"
for i in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export FWAMD_LIB=$R/faster_whisper_amd/libfwamd_base.so; else unset FWAMD_LIB; fi
    timeout 200 $B > "$OUT/${which}_$i.json" 2> "$OUT/${which}_$i.err"
    python - <<PY
import json
try:
    j=json.load(open("$OUT/${which}_$i.json")); print("$which $i", j["value"], j["ms_per_step"], j["config"]["decode_group"], j["verified"])
except Exception as e: print("$which $i unreadable", e)
PY
  done
done
unset FWAMD_LIB
cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_attn_sq" -o pmc -- python "$R/profiles/attn_bench.py" 0 > "$OUT/pmc_attn_sq.log" 2>&1
f=$(find "$OUT/pmc_attn_sq" -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python "$R/profiles/parse_pmc.py" "$f" > "$OUT/pmc_attn_sq.json"
rm -rf "$OUT/pmc_attn_sq"
python - <<PY
import json
try:
    j=json.load(open("$OUT/pmc_attn_sq.json"))
    for k,v in j.items():
        if "attn" in k: print(k, {c: round(x["mean"]) for c,x in v.items() if isinstance(x,dict)})
except Exception as e: print("sq unreadable", e)
PY
