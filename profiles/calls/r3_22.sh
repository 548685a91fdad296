cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03r
timeout 280 python -m pytest tests/test_gpu_rccl_world1.py -m gpu -q > gpurun_out/r03r/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03r/pytest.log
tail -25 gpurun_out/r03r/pytest.log
