cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "dec_linear" > gpurun_out/r03i/pytest_kernels.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03i/pytest_kernels.log
tail -4 gpurun_out/r03i/pytest_kernels.log
DLB_VARIANTS=21,22,10,12,40,41,42,43,44,11 timeout 600 python profiles/dec_linear_bench.py 80 1280 1520 > gpurun_out/r03i/dec_linear_bench.txt 2> gpurun_out/r03i/dec_linear_bench.err
cat gpurun_out/r03i/dec_linear_bench.txt; tail -3 gpurun_out/r03i/dec_linear_bench.err
