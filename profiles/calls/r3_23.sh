cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03s
timeout 280 python -m pytest tests/test_gpu_rccl_world1.py -m gpu -q > gpurun_out/r03s/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03s/pytest.log
tail -25 gpurun_out/r03s/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03s/smoke.log 2>&1; tail -3 gpurun_out/r03s/smoke.log
