#!/bin/bash
set -u
O=gpurun_out/r02i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "fullsize or train_cli" > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -5 $O/gpu_tests.txt | cut -c1-300
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events"
run() { # name, env, args
  env $2 timeout 300 python bench.py $B $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], d["config"].get("launch"), d["config"]["loss"])
except Exception as e: print("$1 failed", e); print(open("$O/$1.err").read()[-1500:])
PY
}
run graph_base "A=1" ""
run graph_side "OFQ_DW_SIDE_STREAM=1" ""
run eager_base "A=1" "--no-graph"
run eager_side "OFQ_DW_SIDE_STREAM=1" "--no-graph"
run graph_base2 "A=1" ""
run graph_side2 "OFQ_DW_SIDE_STREAM=1" ""
