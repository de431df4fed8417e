#!/bin/bash
# round 2, call B: captured-step tests after the stream fix, graph bench on C3 / force-dp, rocprof kernel stats (graph + eager)
set -u
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q > $O/graph_tests.txt 2>&1; echo "graph tests rc=$?"
tail -15 $O/graph_tests.txt | cut -c1-400
timeout 300 python -m pytest tests/test_modules_gpu.py -q -k "rowsum or full_step" > $O/mod_tests.txt 2>&1; echo "module tests rc=$?"
tail -5 $O/mod_tests.txt | cut -c1-400
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_graph.json 2> $O/bench_graph.err; echo "bench graph rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --force-dp --no-cpu-baseline --no-roofline-events > $O/bench_dp_graph.json 2> $O/bench_dp_graph.err; echo "dp graph rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 --force-dp --no-graph --no-cpu-baseline --no-roofline-events > $O/bench_dp_eager.json 2> $O/bench_dp_eager.err; echo "dp eager rc=$?"
for f in bench_graph bench_dp_graph bench_dp_eager; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", d["value"], d["ms_per_step"], d["config"].get("launch"))
except Exception as e: print("$f failed", e); print(open("$O/$f.err").read()[-1500:])
PY
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_graph -o graph -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events > $GRAFT_REPO_ROOT/$O/prof_graph.log 2>&1; echo "rocprof graph rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_eager -o eager -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-roofline-events > $GRAFT_REPO_ROOT/$O/prof_eager.log 2>&1; echo "rocprof eager rc=$?"
cd $GRAFT_REPO_ROOT
for k in graph eager; do
  db=$(find $O/prof_$k -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py $db 60 > $O/kernel_stats_$k.txt
  tail -1 $O/kernel_stats_$k.txt
  tail -2 $O/prof_$k.log | cut -c1-600
  find $O/prof_$k -name "*.db" -delete
done
