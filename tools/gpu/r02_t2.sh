#!/bin/bash
set -u
O=gpurun_out/r02_t2; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python3 $R/tools/teacher_profile.py > $R/$O/kt.log 2>&1; echo "rc=$?"
cd $R
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) 40 > $O/kernel_stats_teacher.txt
find $O -name "*.db" -delete
grep -i "forward" $O/kt.log
