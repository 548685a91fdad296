cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03u
timeout 200 python -m pytest tests/test_gpu_decode_group.py tests/test_gpu_pipeline.py -m gpu -q > gpurun_out/r03u/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03u/pytest.log
tail -3 gpurun_out/r03u/pytest.log
timeout 200 python profiles/burst_ab.py 20 8 40 > gpurun_out/r03u/burst_ab.txt 2>&1   # (script removed with the rule it measured: profiles/r03_burst_idle_lanes_ab.txt)
cat gpurun_out/r03u/burst_ab.txt | tail -14
