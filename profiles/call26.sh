#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out/r02q; mkdir -p $OUT
t0=$SECONDS
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "== driver command rc=$? $((SECONDS-t0))s $(cut -c1-120 $OUT/bench_driver.json)"
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
for w in 16 24 20; do
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 2 --workers $w --no-secondary --no-cpu-baseline --no-profile-pass > $OUT/bench_k20_w$w.json 2> $OUT/bench_k20_w$w.err
  echo "== K=20 workers=$w $(python3 -c "
import json
j=json.load(open('$OUT/bench_k20_w$w.json')); print(j['value'], j['ms_per_step'], j['config']['decode_group'])")"
done
python3 -c "
import json
j=json.load(open('$OUT/bench_driver.json')); print('driver cmd:', j['value'], j['ms_per_step'], j['config']['decode_group'], 'cap', j['cap_case'], 'pipeline', j['pipeline']['value'])"
