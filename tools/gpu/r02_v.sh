#!/bin/bash
# Swin-T W3A3 kernel breakdown (128 images)
set -u
O=gpurun_out/r02_v; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/bench.py --model swin_t --wbits 3 --abits 3 --batch-per-gpu 128 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline-events > $R/$O/kt.log 2>&1; echo "rc=$?"
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 60 > $O/kernel_stats_swin.txt
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 120 grid > $O/kernel_stats_swin_grid.txt
find $O -name "*.db" -delete
grep '"metric"' $O/kt.log | cut -c1-200
