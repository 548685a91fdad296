cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
DLB_VARIANTS=0,1,21,22,30,31,32,33,34,35,36,37 timeout 600 python profiles/dec_linear_bench.py 320 640 1280 1680 > gpurun_out/r03e/dec_linear_bench.txt 2> gpurun_out/r03e/dec_linear_bench.err
cat gpurun_out/r03e/dec_linear_bench.txt; tail -3 gpurun_out/r03e/dec_linear_bench.err
