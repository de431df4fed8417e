#!/bin/bash
# round 2, call A: captured-step tests, full GPU suite, graph vs eager bench on C3 / C2 / force-dp
set -u
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q > $O/graph_tests.txt 2>&1; echo "graph tests rc=$?"
tail -15 $O/graph_tests.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_graph_gpu.py > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -8 $O/gpu_tests.txt
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline > $O/bench_eager.json 2> $O/bench_eager.err; echo "bench eager rc=$?"
for f in bench_graph bench_eager; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["config"].get("launch"))
except Exception as e: print("$f failed", e); print(open("$O/$f.err").read()[-2000:])
PY
done
C2="--model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --no-cpu-baseline --no-roofline-events"
timeout 300 python bench.py --steps 20 --warmup 5 $C2 > $O/bench_c2_graph.json 2> $O/bench_c2_graph.err; echo "c2 graph rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 $C2 --no-graph > $O/bench_c2_eager.json 2> $O/bench_c2_eager.err; echo "c2 eager rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --force-dp --no-cpu-baseline --no-roofline-events > $O/bench_dp_graph.json 2> $O/bench_dp_graph.err; echo "dp graph rc=$?"
for f in bench_c2_graph bench_c2_eager bench_dp_graph; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["config"].get("launch"))
except Exception as e: print("$f failed", e); print(open("$O/$f.err").read()[-2000:])
PY
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_graph -o graph -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events > $GRAFT_REPO_ROOT/$O/prof_graph.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
ls -R $O/prof_graph | head -20
