#!/bin/bash
# bit-identity test of the GEMM-shaped decoder linear (dec_gemm_tile_kernel) + the bench with it wired for R >= 1024
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out/r02t; mkdir -p $OUT
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
timeout 150 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_decode_group.py -q -m gpu -k "tile_bit_identical or decode_group or merged or partition" > $OUT/pytest.log 2>&1; echo "== pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-160
timeout 200 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench.json 2> $OUT/bench.err; echo "== bench rc=$? $(python -c "
import json
j=json.load(open('$OUT/bench.json')); print(j['value'], j['ms_per_step'], j['config']['decode_group'], j['families_ms_per_step'])")"
