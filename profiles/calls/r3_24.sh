cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03t
timeout 400 python -m pytest tests/test_gpu_rccl_world1.py tests/test_gpu_pipeline.py -m gpu -q > gpurun_out/r03t/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03t/pytest.log
tail -30 gpurun_out/r03t/pytest.log
