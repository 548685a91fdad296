set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
( timeout 300 python profiles/run_reference_on_shim.py --ref _ref_scratch --backend engine > gpurun_out/r03a/reference_on_shim.log 2>&1; echo "rc=$?" >> gpurun_out/r03a/reference_on_shim.log )
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_vad.py tests/test_gpu_ct2_dir.py tests/test_gpu_decode_group.py tests/test_gpu_model.py tests/test_gpu_sequential.py -m gpu -q -s --durations=25 > gpurun_out/r03a/pytest_new.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03a/pytest_new.log
tail -5 gpurun_out/r03a/pytest_new.log; tail -8 gpurun_out/r03a/reference_on_shim.log
