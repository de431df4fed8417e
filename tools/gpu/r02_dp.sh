#!/bin/bash
# DataParallel (one rank over RCCL, eager launches) against the plain eager step: which kernels / gaps does it add?
set -u
O=gpurun_out/r02_dp; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
F="--no-cpu-baseline --no-roofline-events --steps 10 --warmup 3"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/dp -o dp -- python3 $R/bench.py $F --force-dp > $R/$O/dp.log 2>&1; echo "rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/eg -o eg -- python3 $R/bench.py $F --no-graph > $R/$O/eg.log 2>&1; echo "rc=$?"
cd $R
for t in dp eg; do
  db=$(find $O/$t -name "*.db" | head -1)
  python tools/rocpd_stats.py $db 400 > $O/stats_$t.txt
  python tools/rocpd_gaps.py $db 0.6 > $O/gaps_$t.txt
  grep '"metric"' $O/$t.log | cut -c1-160
done
find $O -name "*.db" -delete
