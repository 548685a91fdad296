set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_int8.py -m gpu -q -x -k "gemm or dec_linear" > gpurun_out/r03b/pytest_kernels.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03b/pytest_kernels.log
tail -5 gpurun_out/r03b/pytest_kernels.log
timeout 300 python profiles/gemm_bench.py --ab 5 > gpurun_out/r03b/gemm_ab.json 2> gpurun_out/r03b/gemm_ab.err
cat gpurun_out/r03b/gemm_ab.json | tr -d '\n' | head -c 3000; echo
DLB_VARIANTS=0,1,10,11,12,13,14 timeout 600 python profiles/dec_linear_bench.py 80 320 640 1280 1520 1680 > gpurun_out/r03b/dec_linear_bench.txt 2> gpurun_out/r03b/dec_linear_bench.err
cat gpurun_out/r03b/dec_linear_bench.txt
