cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03g/bench.json 2> gpurun_out/r03g/bench.err
echo "rc=$?"; tail -3 gpurun_out/r03g/bench.err
python - <<'P'
import json
j=json.loads(open('gpurun_out/r03g/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','verified','steady','cap_case','pipeline','single_utterance','one_batch_at_a_time','families_ms_per_step','families_rate'):
    print(k, j.get(k))
print(j['config']); print(j['roofline']); print(j.get('roofline_others',{}).get('dec_gemm'))
P
