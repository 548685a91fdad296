cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03k
timeout 200 python profiles/attn_bench.py > gpurun_out/r03k/attn_bench.txt 2>&1; cat gpurun_out/r03k/attn_bench.txt
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -2
export FWAMD_BLOB_CACHE=/tmp/blob
for mf in 50 75 95; do
timeout 600 python bench.py --gpus 1 --steps 128 --warmup 2 --merge-fill $mf --no-secondary --no-cpu-baseline > gpurun_out/r03k/bench_mf$mf.json 2> gpurun_out/r03k/bench_mf$mf.err
python - <<P
import json
j=json.loads(open('gpurun_out/r03k/bench_mf$mf.json').read().strip().splitlines()[-1])
print('merge fill $mf:', j['value'], j['ms_per_step'], j['config']['decode_group'], {k:round(v,1) for k,v in j.get('families_ms_per_step',{}).items() if k.startswith('dec_gemm') or k.startswith('enc_')})
P
done
