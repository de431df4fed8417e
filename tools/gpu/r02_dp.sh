#!/bin/bash
# DataParallel (one rank over RCCL) against the plain step, both with eager launches and both replayed: what does it add?
set -u
O=gpurun_out/r02_dp; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-roofline-events --steps 20 --warmup 5"
run() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
timeout 300 python bench.py $F 2>/dev/null | run "graph      "
timeout 300 python bench.py $F --force-dp 2>/dev/null | run "graph + dp "
timeout 300 python bench.py $F --no-graph 2>/dev/null | run "eager      "
timeout 300 python bench.py $F --no-graph --force-dp 2>/dev/null | run "eager + dp "
done
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/dp -o dp -- python3 $R/bench.py --no-cpu-baseline --no-roofline-events --steps 10 --warmup 3 --force-dp > $R/$O/dp.log 2>&1; echo "rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/gr -o gr -- python3 $R/bench.py --no-cpu-baseline --no-roofline-events --steps 10 --warmup 3 > $R/$O/gr.log 2>&1; echo "rc=$?"
cd $R
for t in dp gr; do
  db=$(find $O/$t -name "*.db" | head -1)
  python tools/rocpd_stats.py $db 400 > $O/stats_$t.txt
  python tools/rocpd_gaps.py $db 0.6 > $O/gaps_$t.txt
done
find $O -name "*.db" -delete
