cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
timeout 100 python -m pytest tests/test_gpu_full_size.py -m gpu -q -x -k "test_large_v3_float16" --durations=3 > gpurun_out/r03w/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03w/pytest.log
tail -12 gpurun_out/r03w/pytest.log
