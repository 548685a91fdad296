cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03v
timeout 200 python -m pytest tests/test_gpu_decode_group.py -m gpu -q > gpurun_out/r03v/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03v/pytest.log
tail -15 gpurun_out/r03v/pytest.log
