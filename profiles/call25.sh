#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out/r02p; mkdir -p $OUT
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_int8.py tests/test_gpu_decode_group.py -q -m gpu --maxfail=5 > $OUT/pytest.log 2>&1; echo "== pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 150 python profiles/dec_linear_bench.py 80 320 1280 > $OUT/dec_linear_bench.txt 2>&1; grep -A3 "^R =" $OUT/dec_linear_bench.txt
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench.json 2> $OUT/bench.err; echo "== bench rc=$? $(python -c "
import json
j=json.load(open('$OUT/bench.json')); print(j['value'], j['ms_per_step'], j['families_ms_per_step'])")"
cd /tmp; export TMPDIR=/tmp
FWAMD_NO_GRAPH=1 timeout 250 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o pmc -- python $R/bench.py --no-cpu-baseline --no-profile-pass --no-secondary --workers 1 --steps 2 --warmup 1 > $OUT/prof_fetch.log 2>&1
f=$(find $OUT/prof_fetch -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $R/profiles/parse_pmc.py "$f" > $OUT/pmc_fetch.json; rm -rf $OUT/prof_fetch
python - <<PY
import json
j=json.load(open("$OUT/pmc_fetch.json"))
for k,v in j.items():
    if "dec_gemm" in k: print(k[:75], v["FETCH_SIZE"]["dispatches"], round(v["hbm_read_bytes_per_launch_corrected"]/1e6,2), "MB")
PY
