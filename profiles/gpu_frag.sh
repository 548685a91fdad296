#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_int8.py -x -q -k "gemm" 2>&1 | tail -2
timeout 200 python profiles/gemm_bench.py --iters 10 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k.split(' ')[0]: v['TFLOP/s'] for k,v in d.items() if isinstance(v,dict) and 'TFLOP/s' in v})"
timeout 200 python profiles/gemm_bench.py --iters 10 --int8 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('int8', {k.split(' ')[0]: v['TFLOP/s'] for k,v in d.items() if isinstance(v,dict) and 'TFLOP/s' in v})"
B="python bench.py --no-cpu-baseline --no-profile-pass --no-secondary --steps 32"
for f in "0" "4,2,4" "2,2,8" "4,2,8" "2,4,4" "4,4,8"; do
  echo "FWAMD_FRAG=$f: $(FWAMD_FRAG=$f timeout 200 $B 2>/dev/null | python -c "import json,sys; j=json.load(sys.stdin); print(j['value'])")"
done
timeout 400 python -m pytest tests/test_gpu_full_size.py -q -s 2>&1 | grep -E "^\[|MISMATCH|passed|failed" | cut -c1-200
