#!/bin/bash
# Round 6, GPU call 3: the C5 composition tests (native VAD + batched word timestamps) and the register cap of the decoder
# cross-attention kernel (knob 7: does a thinner HBM-stream kernel let the other lane's kernels co-reside?).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gpu_c5.py -x -q -s > "$OUT/pytest_c5_call3.log" 2>&1
echo "pytest rc=$?"; grep -E "C5|device VAD|passed|failed|Error|assert" "$OUT/pytest_c5_call3.log" | cut -c1-400 | tail -20
timeout 1500 python profiles/ab_r05.py --knob 7 --values 0,1,2 --rounds 2 --steps 96 --profile > "$OUT/ab_cross_regs.jsonl" 2> "$OUT/ab_cross_regs.err"
echo "ab rc=$?"; cut -c1-330 "$OUT/ab_cross_regs.jsonl"; tail -3 "$OUT/ab_cross_regs.err"
