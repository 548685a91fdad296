#!/bin/bash
# One gpurun call that refreshes the whole evidence set of a round (run from the repo root on the GPU box):
#
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r02'
#
# Produces under gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   bench.json                       python bench.py (the bench line: roofline + cpu_baseline)
#   kernel_stats.csv                 rocprofv3 --kernel-trace --stats, single stream, eager decode
#   pmc_fetch.json                   FETCH_SIZE per kernel (x2 gfx950 correction applied by parse_pmc.py)
#   pmc_sq.json                      SQ wait / active / MFMA-busy / LDS-conflict counters per kernel (where the
#                                    cycles of the encoder GEMM and the skinny GEMM go)
#   sweep.jsonl                      single-stream / 8-in-flight RTF for the knob sets listed in SWEEPS below
# Counter passes are separate runs with --kernel-trace only (gpurun refuses --pmc combined with the other traces).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob_fp16.npy
BENCH_PROF="python $R/bench.py --steps 1 --warmup 1 --workers 1 --no-cpu-baseline --no-profile-pass"

cd "$R"
# code written after round 1's GPU budget was spent: run it under a hard timeout, separately from the regular suite
FWAMD_TEST_UNVALIDATED=1 timeout 240 python -m pytest tests/test_gpu_vad.py tests/test_gpu_full_size.py -q -s 2>&1 \
    | tail -15 > "$OUT/unvalidated_tests.log"
cat "$OUT/unvalidated_tests.log"
# int8 fragment-major decoder GEMM (opt-in at pack time): parity first, then speed against the LDS form
FWAMD_DEC_GEMM_I8=frag timeout 200 python -m pytest tests/test_gpu_int8.py -q 2>&1 | tail -3 > "$OUT/int8_frag_tests.log"
cat "$OUT/int8_frag_tests.log"
cd /tmp; export TMPDIR=/tmp
pmc_pass() {   # name, counters...
  local name=$1; shift
  FWAMD_NO_GRAPH=1 timeout 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/prof_$name" -o pmc -- $BENCH_PROF \
      > "$OUT/prof_$name.log" 2>&1
  local f; f=$(find "$OUT/prof_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$R/profiles/parse_pmc.py" "$f" > "$OUT/pmc_$name.json"
  rm -rf "$OUT/prof_$name"
}
pmc_pass fetch FETCH_SIZE
[ -s "$OUT/pmc_fetch.json" ] && cp "$OUT/pmc_fetch.json" "$R/profiles/${TAG}_pmc_fetch.json"
pmc_pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS

cd "$R"
timeout 300 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
# the opt-in end-to-end pipeline number (first hardware run: kept apart from the bench line above)
timeout 200 python bench.py --pipeline --no-cpu-baseline --no-profile-pass --steps 8 > "$OUT/bench_pipeline.json" 2> "$OUT/bench_pipeline.err"

cd /tmp
FWAMD_NO_GRAPH=1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_kt" -o kt -- \
    python "$R/bench.py" --steps 2 --warmup 1 --workers 1 --no-cpu-baseline --no-profile-pass > "$OUT/prof_kt.log" 2>&1
f=$(find "$OUT/prof_kt" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/prof_kt"

cd "$R"
SWEEPS=("A=0" "FWAMD_FRAG_ROWLOOP=1" "FWAMD_FRAG_LONGK_RT=1 FWAMD_FRAG_LONGK_NT=1")
for s in "${SWEEPS[@]}"; do
  timeout 120 env $s python profiles/sweep.py --workers 1,8 --tag "$s" >> "$OUT/sweep.jsonl" 2>> "$OUT/sweep.err"
done
cat "$OUT/sweep.jsonl"
cut -c1-600 "$OUT/bench.json"
for s in "A=0" "FWAMD_DEC_GEMM_I8=frag"; do
  timeout 150 env $s python profiles/sweep.py --compute-type int8_float16 --workers 1,8 --tag "int8 $s" \
      >> "$OUT/sweep_int8.jsonl" 2>> "$OUT/sweep.err"
done
cat "$OUT/sweep_int8.jsonl"
