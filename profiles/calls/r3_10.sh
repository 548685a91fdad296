cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_int8.py -m gpu -q -x -k "gemm or dec_linear" > gpurun_out/r03j/pytest_kernels.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03j/pytest_kernels.log
tail -4 gpurun_out/r03j/pytest_kernels.log
timeout 300 python profiles/gemm_bench.py > gpurun_out/r03j/gemm_bench.json 2> gpurun_out/r03j/gemm_bench.err
python -c "
import json; j=json.load(open('gpurun_out/r03j/gemm_bench.json'))
print({k:v for k,v in j.items() if 'pad' in k})"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03j/bench.json 2> gpurun_out/r03j/bench.err
echo "rc=$?"; tail -3 gpurun_out/r03j/bench.err
python - <<'P'
import json
j=json.loads(open('gpurun_out/r03j/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','verified','steady','cap_case','pipeline','single_utterance','one_batch_at_a_time','families_ms_per_step','families_rate'):
    print(k, j.get(k))
print(j['config']['decode_group']); print(j.get('roofline_others',{}).get('dec_gemm'))
P
