cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
DLB_VARIANTS=0,1,23,24,25,26 timeout 600 python profiles/dec_linear_bench.py 5 16 80 > gpurun_out/r03l/dec_linear_small.txt 2> gpurun_out/r03l/dec_linear_small.err
cat gpurun_out/r03l/dec_linear_small.txt; tail -3 gpurun_out/r03l/dec_linear_small.err
