#!/bin/bash
# round 2, call E: optimised recompute kernel: parity + per-site A/B (same box)
set -u
O=gpurun_out/r02e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "recompute or i8" > $O/kernel_tests.txt 2>&1; echo "kernel tests rc=$?"
tail -12 $O/kernel_tests.txt | cut -c1-300
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -12 $O/gpu_tests.txt | cut -c1-300
B="--steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events"
run() { # name, env, args
  env $2 timeout 300 python bench.py $B $3 > $O/$1.json 2> $O/$1.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], d["ms_per_step"], d["config"].get("launch"))
except Exception as e: print("$1 failed", e); print(open("$O/$1.err").read()[-1500:])
PY
}
run none "OFQ_RECOMPUTE=" ""
run qkx "OFQ_RECOMPUTE=qkx" ""
run qkx_v "OFQ_RECOMPUTE=qkx,v" ""
run all "OFQ_RECOMPUTE=qkx,v,fc1" ""
run none2 "OFQ_RECOMPUTE=" ""
run qkx2 "OFQ_RECOMPUTE=qkx" ""
cd /tmp
OFQ_RECOMPUTE=qkx,v,fc1 timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $db 45 > $O/kernel_stats_all.txt; tail -1 $O/kernel_stats_all.txt
python tools/rocpd_stats.py $db 200 grid | grep -E "i8_nt_kernel<0>|i8_lsqbwd|lsq_kernel<2, true" > $O/kernel_stats_grid.txt
find $O/prof -name "*.db" -delete
