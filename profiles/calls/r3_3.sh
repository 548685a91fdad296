set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_int8.py -m gpu -q -x -k "gemm or dec_linear" > gpurun_out/r03c/pytest_kernels.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03c/pytest_kernels.log
tail -5 gpurun_out/r03c/pytest_kernels.log
timeout 300 python profiles/gemm_bench.py --ab 4 > gpurun_out/r03c/gemm_ab.json 2> gpurun_out/r03c/gemm_ab.err
python - <<'P'
import json
j=json.load(open('gpurun_out/r03c/gemm_ab.json'))
for k,v in j.items():
    if 'pipe' in k: print(k, v)
P
DLB_VARIANTS=0,10,11,12,13,14,15 timeout 600 python profiles/dec_linear_bench.py 80 640 1280 1520 1680 > gpurun_out/r03c/dec_linear_bench.txt 2> gpurun_out/r03c/dec_linear_bench.err
cat gpurun_out/r03c/dec_linear_bench.txt; tail -3 gpurun_out/r03c/dec_linear_bench.err
