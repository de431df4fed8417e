#!/bin/bash
set -u
O=gpurun_out/r04_kt_swin_grid; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-events --model swin_t --wbits 3 --abits 3 > $R/$O/kt.log 2>&1
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 400 grid > $O/kernel_stats_grid.txt
find $O -name "*.db" -delete
grep "qgemm_bf16s_tn_kernel<false>\|qgemm_bf16s_nt_kernel<3, false, 1\|lsq_kernel<2, true, false, false>" $O/kernel_stats_grid.txt | head -40 | cut -c1-150
