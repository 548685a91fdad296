#!/bin/bash
# Short measurement call: standalone kernel durations (one batch at a time), SQ counters of the encoder GEMM, knobs.
set -u
TAG=${1:-probe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
B="python $R/bench.py --no-cpu-baseline --no-profile-pass --no-secondary"
run() { local name=$1; shift; timeout 200 env "$@" $B --steps 32 > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"; echo "== $name rc=$? $(python -c "import json,sys; j=json.load(open('$OUT/bench_$name.json')); print(j['value'], j['config']['decode_group'])" 2>&1 | tail -1)"; }
run default A=0
run prio FWAMD_DEC_PRIO=1
run prio_fill100 FWAMD_DEC_PRIO=1 FWAMD_GROUP_FILL=1.0
run noserial FWAMD_ENC_SERIAL=0
cd /tmp; export TMPDIR=/tmp
FWAMD_NO_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_kt" -o kt -- $B --workers 1 --steps 2 --warmup 1 > "$OUT/prof_kt.log" 2>&1
f=$(find "$OUT/prof_kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_w1.csv"; rm -rf "$OUT/prof_kt"
head -14 "$OUT/kernel_stats_w1.csv" | cut -c1-140
FWAMD_NO_GRAPH=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d "$OUT/prof_sq" -o pmc -- $B --workers 1 --steps 1 --warmup 1 > "$OUT/prof_sq.log" 2>&1
f=$(find "$OUT/prof_sq" -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python "$R/profiles/parse_pmc.py" "$f" > "$OUT/pmc_sq.json"; rm -rf "$OUT/prof_sq"
python - <<PY
import json
d=json.load(open("$OUT/pmc_sq.json"))
for k,v in d.items():
    if "gemm_f16" in k or "attn_enc" in k or "cross_attn" in k or "frag_kernel" in k:
        print(k[:60], {c: round(x["mean"]) for c,x in v.items() if isinstance(x, dict)})
PY
# merged decode runs with the encoders out of the way (the leader waits for all 8 requests)
FWAMD_NO_GRAPH=1 FWAMD_GROUP_FILL=1.0 timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_kt" -o kt -- $B --steps 8 --warmup 1 > "$OUT/prof_kt2.log" 2>&1
f=$(find "$OUT/prof_kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats_w8_fill100.csv"; rm -rf "$OUT/prof_kt"
head -14 "$OUT/kernel_stats_w8_fill100.csv" | cut -c1-140
