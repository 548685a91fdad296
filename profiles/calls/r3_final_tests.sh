cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_final
timeout 1700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/r03_final/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_final/pytest_gpu.log
tail -22 gpurun_out/r03_final/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_final/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r03_final/smoke.log
