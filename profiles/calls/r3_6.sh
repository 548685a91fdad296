cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
DLB_VARIANTS=12,16,17,11,18,19 timeout 600 python profiles/dec_linear_bench.py 80 1280 > gpurun_out/r03f/dec_linear_bench.txt 2> gpurun_out/r03f/dec_linear_bench.err
cat gpurun_out/r03f/dec_linear_bench.txt; tail -3 gpurun_out/r03f/dec_linear_bench.err
