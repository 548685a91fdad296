cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03m
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_int8.py tests/test_gpu_decode_group.py "tests/test_gpu_full_size.py::test_merged_run_large_v3_float16" "tests/test_gpu_full_size.py::test_merged_run_large_v3_int8_float16" -m gpu -q > gpurun_out/r03m/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03m/pytest.log
tail -6 gpurun_out/r03m/pytest.log
export FWAMD_BLOB_CACHE=/tmp/blob
for ct in float16 int8_float16; do
timeout 600 python bench.py --gpus 1 --steps 128 --warmup 2 --compute-type $ct --no-secondary --no-cpu-baseline > gpurun_out/r03m/bench_$ct.json 2> gpurun_out/r03m/bench_$ct.err
python - <<P
import json
j=json.loads(open('gpurun_out/r03m/bench_$ct.json').read().strip().splitlines()[-1])
print('$ct:', j['value'], j['ms_per_step'], j['verified'], j['config']['decode_group'], {k:round(v,1) for k,v in j.get('families_ms_per_step',{}).items()})
P
done
