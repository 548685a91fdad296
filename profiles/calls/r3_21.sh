cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
timeout 300 python -m pytest tests/test_gpu_decode_group.py tests/test_gpu_pipeline.py -m gpu -q > gpurun_out/r03q/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03q/pytest.log
tail -4 gpurun_out/r03q/pytest.log
