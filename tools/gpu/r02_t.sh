#!/bin/bash
set -u
O=gpurun_out/r02t; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q -k "teacher or plane_gemm" > $O/tests.txt 2>&1; echo "tests rc=$?"
tail -12 $O/tests.txt | cut -c1-300
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events --with-teacher"
run() { timeout 300 python bench.py $B $2 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], d["config"]["workload"][-70:])
except Exception as e: print("$1 failed", e); print(open("$O/$1.err").read()[-1500:])
PY
}
run t_x9 "--teacher-gemm bf16x9"
run t_x6 "--teacher-gemm bf16x6"
run t_f32 "--teacher-gemm f32"
run t_stock "--stock-teacher"
