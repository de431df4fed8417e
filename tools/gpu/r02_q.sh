#!/bin/bash
set -u
O=gpurun_out/r02q; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --force-dp --no-graph --no-cpu-baseline --no-roofline-events > $GRAFT_REPO_ROOT/$O/prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
db=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $db 200 > $O/kernel_stats_dp.txt
grep -iE "ccl|copy|foreach|multi_tensor|Memcpy|div|fill" $O/kernel_stats_dp.txt | cut -c1-170 | head -20
tail -1 $O/kernel_stats_dp.txt
find $O/prof -name "*.db" -delete
