#!/bin/bash
set -u
O=gpurun_out/r02_grid; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events > $R/$O/kt.log 2>&1; echo "rc=$?"
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 400 grid > $O/kernel_stats_grid.txt
find $O -name "*.db" -delete
