cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03p
timeout 900 python -m pytest tests/test_gpu_decode_group.py tests/test_gpu_model.py tests/test_gpu_pipeline.py tests/test_gpu_sequential.py tests/test_gpu_int8.py tests/test_gpu_ct2_dir.py tests/test_gpu_logits_rules.py "tests/test_gpu_full_size.py::test_merged_run_large_v3_float16" "tests/test_gpu_full_size.py::test_merged_run_large_v3_int8_float16" -m gpu -q -x > gpurun_out/r03p/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03p/pytest.log
tail -5 gpurun_out/r03p/pytest.log; grep "merged\] 8 concurrent" gpurun_out/r03p/pytest.log
export FWAMD_BLOB_CACHE=/tmp/blob
timeout 600 python bench.py --gpus 1 --steps 128 --warmup 2 --no-secondary --no-cpu-baseline > gpurun_out/r03p/bench128.json 2> gpurun_out/r03p/bench128.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-profile-pass > gpurun_out/r03p/bench20.json 2> gpurun_out/r03p/bench20.err
for f in bench128 bench20; do python - <<P
import json
j=json.loads(open('gpurun_out/r03p/$f.json').read().strip().splitlines()[-1])
print('$f:', j['value'], j['ms_per_step'], j['verified'], j['config']['decode_group'], {k:round(v,1) for k,v in j.get('families_ms_per_step',{}).items()})
P
done
tail -3 gpurun_out/r03p/bench128.err
