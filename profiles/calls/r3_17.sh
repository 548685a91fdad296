cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03o
timeout 600 python -m pytest tests/test_gpu_model.py "tests/test_gpu_full_size.py::test_large_v3_float16" -m gpu -q -s > gpurun_out/r03o/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03o/pytest.log
tail -3 gpurun_out/r03o/pytest.log; grep -c "every one of the" gpurun_out/r03o/pytest.log; grep "large-v3 float16\] chunk 13: detect" gpurun_out/r03o/pytest.log
timeout 600 python profiles/two_groups_probe.py 64 > gpurun_out/r03o/two_groups.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03o/two_groups.txt | tail -6
