cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03n
export FWAMD_BLOB_CACHE=/tmp/blob
for rep in 1 2; do
for mf in 50 70 90; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --merge-fill $mf --no-secondary --no-cpu-baseline --no-profile-pass > gpurun_out/r03n/bench20_mf${mf}_$rep.json 2> gpurun_out/r03n/bench20_mf${mf}_$rep.err
python - <<P
import json
j=json.loads(open('gpurun_out/r03n/bench20_mf${mf}_$rep.json').read().strip().splitlines()[-1])
print('20 steps, merge fill $mf rep $rep:', j['value'], j['ms_per_step'], j['config']['decode_group'])
P
done
done
