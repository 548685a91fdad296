cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "dec_linear" > gpurun_out/r03h/pytest_kernels.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03h/pytest_kernels.log
tail -4 gpurun_out/r03h/pytest_kernels.log
DLB_VARIANTS=0,1,21,22,10,11,12,13,14,15,16,17 timeout 600 python profiles/dec_linear_bench.py 80 1280 1520 1600 > gpurun_out/r03h/dec_linear_bench.txt 2> gpurun_out/r03h/dec_linear_bench.err
cat gpurun_out/r03h/dec_linear_bench.txt; tail -3 gpurun_out/r03h/dec_linear_bench.err
