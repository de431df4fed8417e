#!/bin/bash
# C2 kernel breakdown (DeiT-T W4A4, plain attention, 256 images)
set -u
O=gpurun_out/r02_u; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/bench.py --model deit_tiny_distilled_patch16_224 --wbits 4 --abits 4 --no-qkr --batch-per-gpu 256 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events > $R/$O/kt.log 2>&1; echo "rc=$?"
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 60 > $O/kernel_stats_c2.txt; python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 200 grid > $O/kernel_stats_c2_grid.txt
find $O -name "*.db" -delete
tail -3 $O/kt.log | cut -c1-300
