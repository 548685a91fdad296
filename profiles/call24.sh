#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=gpurun_out/r02z; mkdir -p $OUT
export FWAMD_BLOB_CACHE=/tmp/fwamd_blob
timeout 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "== bench rc=$? $(cut -c1-140 $OUT/bench.json)"
timeout 400 python bench.py --no-cpu-baseline --compute-type int8_float16 > $OUT/bench_int8.json 2> $OUT/bench_int8.err; echo "== int8 rc=$? $(cut -c1-140 $OUT/bench_int8.json)"
timeout 400 python bench.py --no-cpu-baseline --model distil-large-v3 --word-timestamps > $OUT/bench_distil.json 2> $OUT/bench_distil.err; echo "== distil rc=$? $(cut -c1-140 $OUT/bench_distil.json)"
python - <<PY
import json
for n in ("bench","bench_int8","bench_distil"):
    try:
        j=json.load(open("$OUT/%s.json"%n))
    except Exception as e:
        print(n, "unreadable", e); continue
    print(n, j["value"], "cap", j.get("cap_case",{}).get("value"), "pipeline", j.get("pipeline",{}).get("value"), "single", j.get("single_utterance",{}).get("value"), "roofline", j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["traffic"]["source"] if j["roofline"].get("traffic") else None)
PY
