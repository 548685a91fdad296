cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
DLB_VARIANTS=0,1,2,4,5,6,7,9,20,21,22,23,24 timeout 600 python profiles/dec_linear_bench.py 640 1280 1680 > gpurun_out/r03d/dec_linear_bench.txt 2> gpurun_out/r03d/dec_linear_bench.err
cat gpurun_out/r03d/dec_linear_bench.txt; tail -3 gpurun_out/r03d/dec_linear_bench.err
