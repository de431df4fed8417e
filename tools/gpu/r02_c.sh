#!/bin/bash
# round 2, call C: full GPU suite (new goldens, graph tests after the aliasing fix), idle-gap analysis graph vs eager
set -u
O=gpurun_out/r02c; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"
tail -12 $O/gpu_tests.txt | cut -c1-300
cd /tmp
for k in graph eager; do
  extra=""; [ $k = eager ] && extra="--no-graph"
  timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof_$k -o $k -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events $extra > $GRAFT_REPO_ROOT/$O/prof_$k.log 2>&1; echo "rocprof $k rc=$?"
done
cd $GRAFT_REPO_ROOT
for k in graph eager; do
  db=$(find $O/prof_$k -name "*.db" | head -1)
  echo "== $k"; python tools/rocpd_gaps.py $db 0.6 | tee $O/gaps_$k.txt
  grep -o '"ms_per_step": [0-9.]*' $O/prof_$k.log
  find $O/prof_$k -name "*.db" -delete
done
