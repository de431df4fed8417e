#!/bin/bash
# kernel-trace stats of the default bench -> gpurun_out/$1/kernel_stats.txt
set -u
TAG=${1:-kt}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-events ${BENCH_ARGS:-} > $R/$O/kt.log 2>&1; echo "kernel-trace rc=$?"
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 70 > $O/kernel_stats.txt; tail -1 $O/kernel_stats.txt
grep -h '"metric"' $O/kt.log | tail -1 | cut -c1-300
find $O -name "*.db" -delete
head -45 $O/kernel_stats.txt | cut -c1-150
