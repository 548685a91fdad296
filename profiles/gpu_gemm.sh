#!/bin/bash
# GEMM iteration call: correctness of the encoder GEMM (fp16 + int8 unit tests), then the micro-benchmark.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_int8.py -x -q -k "gemm" 2>&1 | tail -3
timeout 200 python profiles/gemm_bench.py --iters 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k.split(' ')[0]: v['TFLOP/s'] for k,v in d.items() if isinstance(v,dict) and 'TFLOP/s' in v})"
