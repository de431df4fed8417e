#!/bin/bash
set -u
O=gpurun_out/r02l; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q -k "teacher" > $O/tests.txt 2>&1; echo "teacher test rc=$?"
tail -12 $O/tests.txt | cut -c1-300
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events"
run() { env $2 timeout 300 python bench.py $B $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], d["config"].get("launch"), d["config"]["workload"][-60:])
except Exception as e: print("$1 failed", e); print(open("$O/$1.err").read()[-1500:])
PY
}
run base "A=1" ""
run teacher_hip "A=1" "--with-teacher"
run teacher_stock "A=1" "--with-teacher --stock-teacher"
run teacher_hip2 "A=1" "--with-teacher"
