#!/bin/bash
# Round-4 first GPU call: the whole GPU suite on the shared cross-K/V pool + the new parity tests, the driver's bench
# command, the encoder GEMM at the guide's calibration size (4096^3) next to the encoder shapes, the grid-barrier ubench.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_call1
mkdir -p "$OUT"
cd "$R"
t0=$SECONDS
timeout 1500 python -m pytest tests/ -q -m gpu --maxfail=10 --durations=12 -s > "$OUT/pytest_gpu.log" 2>&1; echo "== pytest -m gpu rc=$? $((SECONDS-t0))s"
grep -E "MISMATCH|peaked|vocabulary projection|steps, engine score|224-step|decode run\(s\)|max prob diff|align token|teacher-forced cum" "$OUT/pytest_gpu.log" | cut -c1-260 | tail -50
tail -25 "$OUT/pytest_gpu.log" | cut -c1-300
t1=$SECONDS
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_cmd.json" 2> "$OUT/bench.err"; echo "== bench rc=$? $((SECONDS-t1))s"; cut -c1-1500 "$OUT/bench_driver_cmd.json"; tail -3 "$OUT/bench.err"
timeout 200 python profiles/gemm_bench.py --iters 20 > "$OUT/gemm_bench.json" 2>&1; echo "== gemm_bench rc=$?"; cat "$OUT/gemm_bench.json" | tr -d '\n' | cut -c1-1500; echo
timeout 200 python profiles/gemm_bench.py --square 4096,8192 --iters 20 > "$OUT/gemm_bench_square.json" 2>&1; echo "== gemm square rc=$?"; cat "$OUT/gemm_bench_square.json" | tr -d '\n' | cut -c1-800; echo
(cd profiles/ubench && timeout 120 hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o /tmp/gb && timeout 60 /tmp/gb) > "$OUT/grid_barrier.txt" 2>&1; echo "== grid barrier rc=$?"; tail -20 "$OUT/grid_barrier.txt"
echo "== total $((SECONDS-t0))s"
